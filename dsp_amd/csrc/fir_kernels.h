// Internal launch interface between the C ABI (dspx_capi.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace dspx {

struct FirShape {
    unsigned long long x_stride;     // elements between channels of x
    unsigned long long hist_stride;  // elements between channels of the history buffer
    unsigned long long y_stride;     // elements between channels of y
    long long n;                     // samples per channel
    unsigned long long total_tiles;  // nchan * tiles_per_chan  (< 2^31)
    unsigned tiles_per_chan;         // ceil(n / fir_tile())
    int ntaps;
};

int fir_tile();  // outputs per workgroup

hipError_t launch_fir_f64(const double* taps, const double* x, const double* hist, double* y, const FirShape& shape,
                          bool exact, bool vec2, hipStream_t stream);
hipError_t launch_fill_splitmix(double* x, unsigned long long n, unsigned long long seed, unsigned long long offset,
                                hipStream_t stream);
hipError_t launch_checksum(const double* y, unsigned long long n, unsigned long long offset, unsigned long long* out,
                           hipStream_t stream);

}  // namespace dspx
