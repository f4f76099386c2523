// Calibration probe (not part of the library): what the host link and the host's memory system
// give the host-pointer entry points -- pageable vs pinned copies, duplex, the price of pinning
// in place, and threaded memcpy into pinned staging.
//   hipcc --offload-arch=gfx950 -O2 -o tools/pcie_probe tools/pcie_probe.hip -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));                    \
            return 1;                                                              \
        }                                                                          \
    } while (0)

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void par_copy(char* dst, const char* src, size_t bytes, int nt) {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([=] {
            const size_t lo = bytes * t / nt, hi = bytes * (t + 1) / nt;
            std::memcpy(dst + lo, src + lo, hi - lo);
        });
    for (auto& t : th) t.join();
}

int main() {
    const size_t B = size_t(1) << 30;
    char *pageable = (char*)std::malloc(B), *pageable2 = (char*)std::malloc(B), *pin_in, *pin_out, *d_a, *d_b;
    std::memset(pageable, 1, B);
    std::memset(pageable2, 2, B);
    CK(hipHostMalloc((void**)&pin_in, B, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&pin_out, B, hipHostMallocDefault));
    std::memset(pin_in, 3, B);
    std::memset(pin_out, 4, B);
    CK(hipMalloc((void**)&d_a, B));
    CK(hipMalloc((void**)&d_b, B));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    const double gib = B / 1e9;
    for (int rep = 0; rep < 2; ++rep) {
        double t = now();
        CK(hipMemcpy(d_a, pageable, B, hipMemcpyHostToDevice));
        std::printf("pageable H2D        %6.1f GB/s\n", gib / (now() - t));
        t = now();
        CK(hipMemcpy(pageable2, d_a, B, hipMemcpyDeviceToHost));
        std::printf("pageable D2H        %6.1f GB/s\n", gib / (now() - t));
        t = now();
        CK(hipMemcpyAsync(d_a, pin_in, B, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        std::printf("pinned H2D          %6.1f GB/s\n", gib / (now() - t));
        t = now();
        CK(hipMemcpyAsync(pin_out, d_b, B, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s1));
        std::printf("pinned D2H          %6.1f GB/s\n", gib / (now() - t));
        t = now();
        CK(hipMemcpyAsync(d_a, pin_in, B, hipMemcpyHostToDevice, s0));
        CK(hipMemcpyAsync(pin_out, d_b, B, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s0));
        CK(hipStreamSynchronize(s1));
        std::printf("pinned duplex       %6.1f GB/s each way\n", gib / (now() - t));
        {   // pinning in place: not permitted for an ordinary user on every machine
            t = now();
            const hipError_t e = hipHostRegister(pageable, B, hipHostRegisterDefault);
            const double treg = now() - t;
            if (e == hipSuccess) {
                t = now();
                CK(hipMemcpyAsync(d_a, pageable, B, hipMemcpyHostToDevice, s0));
                CK(hipStreamSynchronize(s0));
                const double tcp = now() - t;
                t = now();
                CK(hipHostUnregister(pageable));
                std::printf("register 1 GiB      %6.1f ms, H2D from it %6.1f GB/s, unregister %6.1f ms\n", treg * 1e3,
                            gib / tcp, (now() - t) * 1e3);
            } else {
                (void)hipGetLastError();
                std::printf("hipHostRegister(1 GiB): %s\n", hipGetErrorString(e));
            }
        }
        for (int nt : {1, 2, 4, 8, 16}) {
            t = now();
            par_copy(pin_in, pageable, B, nt);
            const double a = now() - t;
            t = now();
            par_copy(pageable2, pin_out, B, nt);
            std::printf("memcpy %2d threads   to pinned %6.1f GB/s, from pinned %6.1f GB/s\n", nt, gib / a, gib / (now() - t));
        }
    }
    // fresh pageable destination (first touch) as a caller's new output vector would be
    char* fresh = (char*)std::malloc(B);
    double t = now();
    par_copy(fresh, pin_out, B, 4);
    std::printf("memcpy 4 threads into untouched memory %6.1f GB/s\n", gib / (now() - t));
    return 0;
}
