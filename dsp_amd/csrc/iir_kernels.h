// Internal launch interface of the recursive tail of dsp::lfilter (dsp.hpp:366-368):
//   y[i] = w[i] - sum_{j=1..P} a[j] * y[i-j],   P = na - 1,
// applied in place to a buffer that holds w (the output of the FIR part).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace dspx {

constexpr int kIirMaxOrder = 8;   // P of the kernels below (na <= 9)
constexpr int kIirLevels = 8;     // log2 of the 256 lanes (or tiles) one scan covers
constexpr int kIirTile = 4096;    // samples per workgroup: 256 lanes x 16 consecutive samples
constexpr int kIirTileRun = 16;   // tiles a lane of the tile-state scan walks by itself
constexpr int kIirTileLevels = kIirLevels + 4;  // powers of A^4096 kept: 2^0 .. 2^11 (2^4 = one lane's run)

// Coefficients by value (kernel argument): a[j-1] = a_j.
struct IirCoef {
    double a[kIirMaxOrder];
};

// Powers of the companion matrix A of the recurrence (state = the P most recent outputs, newest first):
// lane[d] = (A^16)^(2^d) carries a state across 2^d lanes, tile[d] = (A^4096)^(2^d) across 2^d tiles
// (tile[4 + d] across 2^d lanes of the tile-state scan, whose lanes walk 16 tiles each).
// Each entry is an unevaluated sum hi + lo of two doubles: a state of a narrow-band filter is a smooth
// sequence, and a matrix power applied to it cancels terms thousands of times larger than the result, so
// the entries and the dot products carry about twice the precision of a double.
struct IirMatrix {
    double hi[kIirMaxOrder][kIirMaxOrder];
    double lo[kIirMaxOrder][kIirMaxOrder];
};
struct IirTables {
    IirMatrix lane[kIirLevels];
    IirMatrix tile[kIirTileLevels];
};

// The FIR part, when the tail kernels compute it on the fly from x instead of reading it from y (short
// numerators: x is then read twice and y written once, 24 bytes per sample instead of 40).  nb == 0: not fused.
constexpr int kIirMaxFusedTaps = 16;
struct IirFront {
    const double* x;
    const double* xhist;              // per channel the nb-1 samples before x[0], oldest first (nullable)
    unsigned long long x_stride, xhist_stride;
    const double* b;                  // nb taps in device memory
    int nb;
    int vec2;                         // x is 16-byte aligned and x_stride is even
};

struct IirShape {
    unsigned long long y_stride;      // elements between channels of y
    unsigned long long state_stride;  // elements between channels of the incoming state
    long long n;                      // samples per channel
    unsigned tiles_per_chan;          // ceil(n / kIirTile)
    unsigned nchan;
    int vec2;                         // y is 16-byte aligned and y_stride is even: 16-byte global accesses
    unsigned lane_plain, tile_plain;  // bit d: level d of that scan may use plain products (IirPlan)
};

// Bytes of device workspace one launch of the scan path needs for `nchan` channels of `n` samples: the
// tables, plus one state per tile unless the launch has enough channels for one workgroup to walk each.
size_t iir_workspace_bytes(size_t nchan, size_t n, int order);

// What the host learns from the matrix powers (computed in double, as the device computes them in
// double-double).  usable: every entry stays far below overflow, i.e. the scan path can be used; otherwise
// (a recurrence that grows without bound, non-finite coefficients) only the sequential kernel is meaningful.
// lane_plain / tile_plain, bit d: the power used at level d of the lane / tile scan has absolute row sums of
// at most 1 / (16 (P + 1)), so applying it with plain fused multiply-adds perturbs a state by less than
// eps / 16 of its own size per level -- far below the rounding of the recurrence itself -- and the compensated
// products are kept for the levels whose powers are not yet small (all of them for poles next to the unit circle).
struct IirPlan {
    bool usable;
    unsigned lane_plain, tile_plain;
};
IirPlan iir_plan(const double* a, int order);

// Block-parallel path (fused multiply-adds; rounding differs from the sequential loop).  `work` holds
// iir_workspace_bytes(); `state_in` (nullable) holds per channel the P outputs that precede y[0], oldest first.
// `front` (nullable): compute the FIR part from x on the fly (one fused multiply-add per tap, taps ascending,
// as the vector-pipe FIR kernel does) instead of expecting it in y.
hipError_t launch_iir_scan(const IirCoef& coef, int order, double* y, const double* state_in, const IirShape& shape,
                           const IirFront* front, void* work, hipStream_t stream);

// One lane per channel, samples in order.  exact: product and subtraction rounded separately, j ascending,
// bit-identical to dsp.hpp:366-368; otherwise one fused multiply-add per coefficient.
hipError_t launch_iir_sequential(const IirCoef& coef, int order, double* y, const double* state_in,
                                 const IirShape& shape, bool exact, hipStream_t stream);
// The tail of lfilter<float> (dsp.hpp:366-368 with T = float) in place on float outputs; `in_double`: products and
// differences in double (double denominators) instead of float.
hipError_t launch_iir_narrow_tail(const IirCoef& coef, int order, bool in_double, float* y, unsigned long long y_stride,
                                  unsigned long long nchan, long long n, hipStream_t stream);

}  // namespace dspx
