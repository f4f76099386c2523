// Calibration microbenchmark (not part of the library): sustained FP64 rate of the chip with
// operands in registers -- v_fma_f64 (vector) and v_mfma_f64_16x16x4_f64 (matrix) -- and the
// clock the chip holds while doing it, over runs long enough for DVFS to settle.
//   hipcc --offload-arch=gfx950 -O3 -o tools/fp64_peak tools/fp64_peak.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

static std::string g_json;  // one record per loop, written to the file named on the command line

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

template <int ACC>
__global__ __launch_bounds__(256) void fma_loop(double* out, int iters, double a, double b, unsigned long long* clk) {
    double acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

// Same loop with per-lane pseudo-random multiplicands and addends (random mantissa bits
// toggle far more logic than loop-invariant operands do, which lowers the clock the
// chip can hold).
template <int ACC>
__global__ __launch_bounds__(256) void fma_loop_random(double* out, int iters, double a, double b, unsigned long long* clk) {
    double acc[ACC], xm[ACC], ya[ACC];
    unsigned long long z = 0x9E3779B97F4A7C15ull * (blockIdx.x * 256 + threadIdx.x + 1);
#pragma unroll
    for (int i = 0; i < ACC; ++i) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        xm[i] = (double)(z >> 11) * 0x1.0p-53 * a;
        z += 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        ya[i] = (double)(z >> 11) * 0x1.0p-53 - 0.5 + b;
        acc[i] = ya[i];
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_fma(acc[i], xm[(i + u) % ACC], ya[(i + 3 * u) % ACC]);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int ACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters, double a, double b, unsigned long long* clk) {
    d4 acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    // per-lane pseudo-random A and B operands, four of each, cycled through the unrolled body
    double av4[4], bv4[4];
    unsigned long long z = 0x9E3779B97F4A7C15ull * (blockIdx.x * 256 + threadIdx.x + 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        av4[i] = ((double)(z >> 11) * 0x1.0p-52 - 1.0) * a;
        z += 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        bv4[i] = ((double)(z >> 11) * 0x1.0p-52 - 1.0) + b;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < ACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av4[(u + i) & 3], bv4[u], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

// The four-block 4x4x4 form (v_mfma_f64_4x4x4_4b_f64): one accumulator value per lane per chain.
template <int ACC>
__global__ __launch_bounds__(256) void mfma4_loop(double* out, int iters, double a, double b, unsigned long long* clk) {
    double acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = 0.0;
    double av4[4], bv4[4];
    unsigned long long z = 0x9E3779B97F4A7C15ull * (blockIdx.x * 256 + threadIdx.x + 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        av4[i] = ((double)(z >> 11) * 0x1.0p-52 - 1.0) * a;
        z += 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        bv4[i] = ((double)(z >> 11) * 0x1.0p-52 - 1.0) + b;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < ACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av4[(u + i) & 3], bv4[u], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

// Both pipes at once: every wave interleaves independent v_fma_f64 chains (VF of them per MFMA)
// with v_mfma_f64_16x16x4_f64 chains, all on per-lane random operands.
template <int MACC, int VF>
__global__ __launch_bounds__(256) void mixed_loop(double* out, int iters, double a, double b, unsigned long long* clk) {
    d4 macc[MACC];
    double vacc[8], xm[8], ya[8], av4[4], bv4[4];
    unsigned long long z = 0x9E3779B97F4A7C15ull * (blockIdx.x * 256 + threadIdx.x + 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        xm[i] = (double)(z >> 11) * 0x1.0p-53 * a;
        z += 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        ya[i] = (double)(z >> 11) * 0x1.0p-53 - 0.5 + b;
        vacc[i] = ya[i];
        if (i < 4) av4[i] = xm[i] - 0.3, bv4[i] = ya[i];
    }
#pragma unroll
    for (int i = 0; i < MACC; ++i) macc[i] = d4{0.0, 0.0, 0.0, 0.0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < MACC; ++i) {
                macc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av4[(u + i) & 3], bv4[u], macc[i], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < VF; ++v) {
                    const int q = (i * VF + v + u) & 7;
                    vacc[q] = __builtin_fma(vacc[q], xm[(q + u) & 7], ya[(q + 3 * u) & 7]);
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < MACC; ++i) s += macc[i][0] + macc[i][1] + macc[i][2] + macc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += vacc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <typename K> void run(const char* name, K kernel, double flops_per_thread_iter, int blocks_per_cu, int iters, double seconds) {
    const int blocks = 256 * blocks_per_cu;
    double* out;
    unsigned long long* clk;
    CK(hipMalloc(&out, sizeof(double) * blocks * 256));
    CK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999999, 1e-7, clk);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999999, 1e-7, clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms1;
    CK(hipEventElapsedTime(&ms1, e0, e1));
    const int reps = (int)(seconds * 1e3 / ms1) + 1;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999999, 1e-7, clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = flops_per_thread_iter * iters * 256.0 * blocks * reps;
    std::vector<unsigned long long> h(2 * blocks);
    CK(hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
    double ghz = 0;
    for (int i = 0; i < blocks; ++i) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
    printf("%-10s waves/SIMD=%d  first launch %.2f TFLOP/s, sustained %.2f TFLOP/s over %.2f s, clock at end %.3f GHz\n", name,
           blocks_per_cu, flops / reps / (ms1 * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12, ms * 1e-3, ghz / blocks);
    char rec[512];
    snprintf(rec, sizeof rec,
             "%s\n  {\"loop\": \"%s\", \"waves_per_simd\": %d, \"first_launch_tflops\": %.2f, \"sustained_tflops\": %.2f, "
             "\"seconds\": %.2f, \"clock_ghz\": %.3f}",
             g_json.empty() ? "" : ",", name, blocks_per_cu, flops / reps / (ms1 * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12,
             ms * 1e-3, ghz / blocks);
    g_json += rec;
    CK(hipFree(out));
    CK(hipFree(clk));
}

int main(int argc, char** argv) {
    run("fma invariant-operands acc8 (50 ms)", fma_loop<8>, 2.0 * 8 * 8, 4, 10000, 0.05);
    run("fma invariant-operands acc8", fma_loop<8>, 2.0 * 8 * 8, 4, 10000, 2.0);
    run("fma invariant-operands acc8", fma_loop<8>, 2.0 * 8 * 8, 2, 10000, 2.0);
    run("fma rand", fma_loop_random<8>, 2.0 * 8 * 8, 4, 10000, 2.0);
    run("fma rand", fma_loop_random<8>, 2.0 * 8 * 8, 2, 10000, 2.0);
    // one MFMA = 16*16*4*2 flops per wave = 32 flops per lane
    run("mfma acc4", mfma_loop<4>, 32.0 * 4 * 4, 1, 10000, 2.0);
    run("mfma acc4", mfma_loop<4>, 32.0 * 4 * 4, 2, 10000, 2.0);
    run("mfma acc8", mfma_loop<8>, 32.0 * 8 * 4, 1, 5000, 2.0);
    run("mfma acc2", mfma_loop<2>, 32.0 * 2 * 4, 4, 10000, 2.0);
    // mixed: per MFMA (32 flops/lane) VF vector FMAs (2 flops/lane each)
    run("mix 4:4", mixed_loop<4, 4>, (32.0 + 2.0 * 4) * 4 * 4, 2, 5000, 2.0);
    run("mix 4:8", mixed_loop<4, 8>, (32.0 + 2.0 * 8) * 4 * 4, 2, 5000, 2.0);
    run("mix 4:12", mixed_loop<4, 12>, (32.0 + 2.0 * 12) * 4 * 4, 2, 5000, 2.0);
    run("mix 4:16", mixed_loop<4, 16>, (32.0 + 2.0 * 16) * 4 * 4, 2, 5000, 2.0);
    run("mix 4:16", mixed_loop<4, 16>, (32.0 + 2.0 * 16) * 4 * 4, 4, 5000, 2.0);
    // one 4x4x4 four-block MFMA = 4 * 4*4*4*2 flops per wave = 8 flops per lane
    run("mfma4x4 acc8", mfma4_loop<8>, 8.0 * 8 * 4, 1, 10000, 2.0);
    run("mfma4x4 acc8", mfma4_loop<8>, 8.0 * 8 * 4, 2, 10000, 2.0);
    run("mfma4x4 acc16", mfma4_loop<16>, 8.0 * 16 * 4, 1, 10000, 2.0);
    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        if (f) {
            fprintf(f, "{\"tool\": \"tools/fp64_peak.hip\", \"what\": \"sustained FP64 rate with operands in registers, per-lane random "
                       "operands unless the loop name says otherwise; clock = s_memtime / s_memrealtime at the end of the run\",\n"
                       " \"loops\": [%s\n ]}\n", g_json.c_str());
            fclose(f);
        }
    }
    return 0;
}
