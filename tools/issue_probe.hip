// How fast does a wave issue ordinary instructions while the other waves of its SIMD issue FP64 matrix instructions
// back to back?  One workgroup of 1024 threads per CU (16 waves, four per SIMD).  Waves 0..11 (three per SIMD) run a
// register-only v_mfma_f64_4x4x4_4b_f64 loop until a flag in LDS is raised; waves 12..15 (one per SIMD) time chains of
// vector, scalar and LDS instructions with s_memtime, then raise the flag.  Run with `mfma` = 0 for the same chains
// on an otherwise idle SIMD, and prio = 0..3 for the timed wave's s_setprio.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/issue_probe && tools/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kChain = 256;

template <int PRIO>
__global__ __launch_bounds__(1024) void probe(unsigned long long* out, int mfma_waves, double seed) {
    __shared__ volatile int stop;
    __shared__ double buf[2048];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) stop = 0;
    buf[tid] = seed * tid, buf[tid + 1024] = seed + tid;
    __syncthreads();
    if (wave < 12) {
        if (wave >= mfma_waves) return;
        double a = seed * lane + 1.0, b = seed - lane;
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
        for (int it = 0; it < 20000 && !stop; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
                c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
                c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
                c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
            }
        }
        if (c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 == 12345.678) out[4096 + tid] = 1;
        return;
    }
    __builtin_amdgcn_s_setprio(PRIO);
    // let the matrix waves get going
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(1);
    unsigned long long t[6];
    unsigned v = lane, sacc = __builtin_amdgcn_readfirstlane(wave);
    t[0] = __builtin_readcyclecounter();
    // (1) dependent 32-bit vector instructions
#pragma unroll
    for (int i = 0; i < kChain; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v) : "v"(lane));
    t[1] = __builtin_readcyclecounter();
    // (2) dependent scalar instructions
#pragma unroll
    for (int i = 0; i < kChain; ++i) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));
    t[2] = __builtin_readcyclecounter();
    // (3) LDS reads, each waited for (latency) ...
    double acc = 0;
    int idx = lane;
#pragma unroll 1
    for (int i = 0; i < kChain; ++i) {
        const double d = buf[idx & 2047];
        acc += d;
        idx += (int)d & 1;
    }
    t[3] = __builtin_readcyclecounter();
    // (4) ... and 16 in flight at a time (throughput)
#pragma unroll 1
    for (int i = 0; i < kChain / 16; ++i) {
        double d[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] = buf[(lane + 64 * j + i) & 2047];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += d[j];
    }
    t[4] = __builtin_readcyclecounter();
    // (5) independent 64-bit vector instructions
    double e0 = acc, e1 = seed, e2 = seed + 1, e3 = seed + 2;
#pragma unroll
    for (int i = 0; i < kChain / 4; ++i) {
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(e0) : "v"(seed));
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(e1) : "v"(seed));
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(e2) : "v"(seed));
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(e3) : "v"(seed));
    }
    t[5] = __builtin_readcyclecounter();
    if (lane == 0) {
        unsigned long long* o = out + (blockIdx.x * 4 + (wave - 12)) * 8;
        for (int i = 0; i < 5; ++i) o[i] = t[i + 1] - t[i];
        o[5] = v + sacc + (unsigned long long)(e0 + e1 + e2 + e3);
    }
    stop = 1;
}

template <int PRIO> int run(unsigned long long* d_out, int mfma_waves, const char* what) {
    hipLaunchKernelGGL(probe<PRIO>, dim3(256), dim3(1024), 0, 0, d_out, mfma_waves, 1.000001);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256 * 4 * 8);
    CHECK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    double sum[5] = {};
    for (int w = 0; w < 1024; ++w)
        for (int i = 0; i < 5; ++i) sum[i] += (double)h[w * 8 + i];
    printf("%-34s prio %d: v_xor chain %6.1f  s_add chain %6.1f  LDS read latency %6.1f  LDS read x16 %6.1f  v_add_f64 x4 %6.1f  (s_memtime ticks per instruction)\n",
           what, PRIO, sum[0] / 1024 / kChain, sum[1] / 1024 / kChain, sum[2] / 1024 / kChain, sum[3] / 1024 / kChain, sum[4] / 1024 / kChain);
    return 0;
}

int main() {
    unsigned long long* d_out = nullptr;
    CHECK(hipMalloc(&d_out, (4096 + 1024) * 8 * 8));
    for (int rep = 0; rep < 2; ++rep) {
        if (run<0>(d_out, 0, "SIMD otherwise idle")) return 1;
        if (run<0>(d_out, 4, "one matrix wave on the SIMD")) return 1;
        if (run<0>(d_out, 12, "three matrix waves on the SIMD")) return 1;
        if (run<3>(d_out, 12, "three matrix waves on the SIMD")) return 1;
    }
    CHECK(hipFree(d_out));
    return 0;
}
