// Probe (not part of the library): does the issue rate of v_mfma_f64_4x4x4_4b_f64 depend on which VGPR banks its
// operands sit in?  Hand-numbered registers, 16 accumulators, one A operand per 16 instructions, 16 B operands.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma4_banks tools/mfma4_banks.hip
#include <hip/hip_runtime.h>
#include <cstdio>

// accumulators v[4:35]; B operands start at v[BBASE]; A at v[ABASE]
#define STR2(x) #x
#define STR(x) STR2(x)
#define MF(c, a, b) "v_mfma_f64_4x4x4_4b_f64 v[" STR(c) ":" STR(c) "+1], v[" STR(a) ":" STR(a) "+1], v[" STR(b) ":" STR(b) "+1], v[" STR(c) ":" STR(c) "+1]\n"
#define STEP(A, B0) \
    MF(4, A, B0) MF(6, A, B0 + 2) MF(8, A, B0 + 4) MF(10, A, B0 + 6) MF(12, A, B0 + 8) MF(14, A, B0 + 10) MF(16, A, B0 + 12) MF(18, A, B0 + 14) \
    MF(20, A, B0 + 16) MF(22, A, B0 + 18) MF(24, A, B0 + 20) MF(26, A, B0 + 22) MF(28, A, B0 + 24) MF(30, A, B0 + 26) MF(32, A, B0 + 28) MF(34, A, B0 + 30)

template <int VARIANT>
__global__ __launch_bounds__(256) void loop(double* out, int iters, unsigned long long* clk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (VARIANT == 0)       // B pair in the same banks as its C pair (B0 = 40: 40 % 4 == 4 % 4), A at 104 (banks 0,1)
            asm volatile(STEP(104, 40) STEP(104, 40) STEP(104, 40) STEP(104, 40) ::: "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");
        else if constexpr (VARIANT == 1)  // B pair in the other banks (B0 = 42), A at 104
            asm volatile(STEP(104, 42) STEP(104, 42) STEP(104, 42) STEP(104, 42) ::: "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");
        else if constexpr (VARIANT == 2)  // all B the same register pair (42), A at 104
            asm volatile(STEP(104, 42) ::: "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    if (iters < 0) out[threadIdx.x] = 1.0;
}

template <int V> void run(const char* name, int per_iter, int waves_per_simd) {
    unsigned long long* clk;
    double* out;
    const int blocks = 256 * waves_per_simd, iters = 20000;
    hipMalloc(&clk, 8 * blocks);
    hipMalloc(&out, 8 * 256);
    hipLaunchKernelGGL(loop<V>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    hipLaunchKernelGGL(loop<V>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    hipDeviceSynchronize();
    static unsigned long long h[4096];
    hipMemcpy(h, clk, 8 * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < blocks; ++i) avg += (double)h[i];
    avg /= blocks;
    printf("%-40s waves/SIMD=%d  %.2f cycles per MFMA per wave (x%d waves sharing the SIMD)\n", name, waves_per_simd,
           avg / ((double)iters * per_iter), waves_per_simd);
    hipFree(clk);
    hipFree(out);
}

int main() {
    run<0>("B and C pairs in the same banks", 64, 1);
    run<1>("B and C pairs in different banks", 64, 1);
    run<2>("one B pair, different banks", 16, 1);
    run<0>("B and C pairs in the same banks", 64, 2);
    run<1>("B and C pairs in different banks", 64, 2);
    return 0;
}
