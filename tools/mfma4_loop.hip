// Probe (not part of the library): the product loop of fir_f64_mfma4_kernel in isolation -- cycles per MFMA for one
// wave per SIMD, with and without its LDS operand reads.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o tools/mfma4_loop tools/mfma4_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int RB = 16, kShift = 6, kPadStep = 66, kHalo = 320, kWaveOut = 1024, kSeg = 64;

template <int VARIANT>
__global__ __launch_bounds__(256) void loop(double* out, int reps, int steps, unsigned long long* clk) {
    __shared__ double imgd[(kHalo + 4096) * 66 / 64];
    __shared__ double tapz[328];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (kHalo + 4096) * 66 / 64; i += 256) imgd[i] = 1.0 / (i + 1);
    for (int i = tid; i < 328; i += 256) tapz[i] = 0.5 / (i + 1);
    __syncthreads();
    const int b_base = (kHalo + wave * kWaveOut) / kSeg * kPadStep + kPadStep * (lane & 15) + (lane >> 4);
    const int a_base = 3 + (lane & 3) - (lane >> 4);
    double acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        auto V = [&](int sidx) {
            const int off = -4 * sidx;
            if constexpr (VARIANT == 2) return imgd[b_base + off];          // no pad arithmetic
            return imgd[b_base + off + 2 * (off >> kShift)];
        };
        double w[RB];
#pragma unroll
        for (int j = 1; j < RB; ++j) w[j] = V(j - RB);
        w[0] = V(0);
        double a_now = tapz[a_base];
        for (int el = 0; el + RB <= steps; el += RB) {
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const double a = a_now;
                if constexpr (VARIANT != 1) a_now = tapz[a_base + 4 * (el + u + 1)];
#pragma unroll
                for (int rb = RB - 1; rb >= 0; --rb) {
                    acc[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, w[(u - rb + RB) % RB], acc[rb], 0, 0, 0);
                    if constexpr (VARIANT != 1)
                        if (rb == RB - 1) w[(u + 1) % RB] = V(el + u + 1);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) s += acc[rb];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int V> void run(const char* name) {
    unsigned long long* clk;
    double* out;
    const int blocks = 256, reps = 2000, steps = 64;
    hipMalloc(&clk, 8 * blocks);
    hipMalloc(&out, 8 * 256 * blocks);
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(loop<V>, dim3(blocks), dim3(256), 0, 0, out, reps, steps, clk);
    hipDeviceSynchronize();
    static unsigned long long h[256];
    hipMemcpy(h, clk, 8 * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < blocks; ++i) avg += (double)h[i];
    printf("%-44s %.2f cycles per MFMA\n", name, avg / blocks / ((double)reps * steps * RB));
    hipFree(clk);
    hipFree(out);
}

int main() {
    run<0>("loop as in the kernel");
    run<1>("operands never reloaded");
    run<2>("LDS reads without the pad arithmetic");
    return 0;
}
