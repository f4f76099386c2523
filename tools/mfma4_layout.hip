// Probe (not part of the library): operand layout of v_mfma_f64_4x4x4_4b_f64 on this chip.  For every lane la, A is
// one-hot at la and B holds lane + 1: every nonzero D lane then names the B lane it was multiplied with.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma4_layout tools/mfma4_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(double* out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la) {
        const double a = lane == la ? 1.0 : 0.0;
        const double b = lane + 1.0;
        out[la * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    }
}

int main() {
    double* d;
    static double h[64 * 64];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d ->", la);
        for (int l = 0; l < 64; ++l)
            if (h[la * 64 + l] != 0.0) printf(" D%d=B%d", l, (int)h[la * 64 + l] - 1);
        printf("\n");
    }
    return 0;
}
