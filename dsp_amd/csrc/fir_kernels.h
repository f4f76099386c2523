// Internal launch interface between the C ABI (dspx_capi.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace dspx {

struct FirShape {
    unsigned long long x_stride;     // elements between channels of x
    unsigned long long hist_stride;  // elements between channels of the history buffer
    unsigned long long y_stride;     // elements between channels of y
    long long n;                     // samples per channel
    unsigned long long div_magic;    // tile_id / tiles_per_chan == (tile_id * div_magic) >> div_shift
    unsigned div_shift;
    unsigned total_tiles;            // nchan * tiles_per_chan  (vector kernel: the grid, < 2^24 as grid * 256 threads must
                                     // stay below 2^32; matrix kernel: what its persistent workgroups walk)
    unsigned tiles_per_chan;         // ceil(n / fir_tile())
    int ntaps;
};

// Fills div_magic / div_shift so that x / d == (x * div_magic) >> div_shift for x < 2^31.
inline void set_divider(FirShape& s, unsigned d) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    s.div_shift = 31 + l;
    s.div_magic = ((1ull << s.div_shift) + d - 1) / d;
}
inline void set_tile_divider(FirShape& s) { set_divider(s, s.tiles_per_chan); }

// One row of a filter bank (bank mode of the kernel): its taps, where its results go, the factor
// applied to the first difference of its convolution, and how far its output is advanced.
struct BankRow {
    unsigned long long taps_off;  // offset of the row's taps in the packed tap buffer
    unsigned long long out_off;   // row of the output buffer that receives the results (times FirShape::y_stride)
    double scale;
    int ntaps;
    int advance;
    int win_steps;  // fir_mfma_window_steps(ntaps): the steps of four tap offsets one window of the matrix-pipe kernels serves
};

int fir_tile();         // outputs per workgroup
int fir_mfma_tile();         // outputs per workgroup and tile of the matrix-pipe kernel
int fir_mfma_bank_stride();  // ... in bank mode
int fir_bank_stride();  // outputs per workgroup in bank mode

hipError_t launch_fir_f64(const double* taps, const double* x, const double* hist, double* y, const FirShape& shape,
                          bool exact, bool vec2, hipStream_t stream);
// Matrix-pipe (Toeplitz-block MFMA) variant of launch_fir_f64; tiles are fir_mfma_tile() outputs.  Over 16-byte aligned
// buffers the wave-autonomous kernels run (one window per chunk up to 569 taps, several beyond) unless `tiled` asks
// for the workgroup-tiled one (same results bit for bit).
hipError_t launch_fir_mfma_f64(const double* taps, const double* x, const double* hist, double* y,
                               const FirShape& shape, bool vec2, bool tiled, hipStream_t stream);
// Bank mode on the matrix pipe: the wave-autonomous kernel (1022-output chunks of a row; aligned or not) unless `tiled`
// asks for the workgroup-tiled one (2046-output tiles).  The two chunkings group a row's taps differently, so their
// results differ in rounding; a row's bits depend on neither the other rows nor the alignment of the buffers.
hipError_t launch_fir_mfma_bank_f64(const double* taps, const BankRow* bank, const double* x, double* y,
                                    const FirShape& shape, bool vec2, bool tiled, hipStream_t stream);
int fir_mfma_window_steps(int ntaps);  // for BankRow::win_steps
// The matrix-pipe kernels keep a small set of device counters per stream they have been launched on (launches of one
// stream hand their work out through them).  A caller that destroys a stream tells the launcher first, with the
// stream idle and its device current, so that the set can serve another stream.
void forget_stream(hipStream_t stream);
// Bank mode: shape.n = length of the single input and of every output row, shape.y_stride = elements
// between output rows, shape.tiles_per_chan = ceil(n / fir_bank_stride()), shape.total_tiles =
// rows * 8 * ceil(tiles_per_chan / 8) workgroups (each row is cut into 8 runs of tiles, one per XCD; the
// divider divides by the run length); the other strides and ntaps are unused.
hipError_t launch_fir_bank_f64(const double* taps, const BankRow* bank, const double* x, double* y,
                               const FirShape& shape, bool exact, bool vec2, hipStream_t stream);
// lfilter<float>: float sums of products formed in float (containers of float) or in double (containers of double),
// bit-identical to the reference loop; tiles are fir_tile() outputs, one workgroup each.
hipError_t launch_fir_f32(const float* taps, const float* x, const float* hist, float* y, const FirShape& shape,
                          hipStream_t stream);
hipError_t launch_fir_f32_from_f64(const double* taps, const double* x, const double* hist, float* y,
                                   const FirShape& shape, hipStream_t stream);
// lfilter<double> over containers of float: float products, widened and added to a double sum.
hipError_t launch_fir_f64_from_f32(const float* taps, const float* x, const float* hist, double* y,
                                   const FirShape& shape, hipStream_t stream);
hipError_t launch_fill_splitmix(double* x, unsigned long long n, unsigned long long seed, unsigned long long offset,
                                hipStream_t stream);
hipError_t launch_checksum(const double* y, unsigned long long n, unsigned long long offset, unsigned long long* out,
                           hipStream_t stream);

}  // namespace dspx
