// gfx950 kernels for the recursive tail of dsp::lfilter (dsp.hpp:366-368), in place on a buffer
// that holds the FIR part w:   y[i] = w[i] - sum_{j=1..P} a[j] * y[i-j].
//
// Block-parallel path.  The recurrence is linear, so the state (the P most recent outputs) after a run
// of L samples is  A^L * (state before) + (state the run reaches from zero).  A lane owns 16 consecutive
// samples, a workgroup a tile of 4096:
//   1. iir_tile_state_kernel: every lane runs its 16 samples from zero state; a Kogge-Stone scan over the
//      256 lanes with the constant matrices (A^16)^(2^d) gives the state the tile reaches from zero.
//   2. iir_tile_scan_kernel: one workgroup per channel scans those tile states with (A^4096)^(2^d), 4096 tiles
//      per pass, and leaves in their place the state every tile starts from.
//   3. iir_apply_kernel: as 1., with the tile's true starting state folded into lane 0; every lane then knows
//      the state it starts from and runs its 16 samples again from it, which yields the outputs.
// HBM traffic: w is read twice and y written once (24 bytes per sample, on top of the FIR pass's 16).  For a
// numerator of at most 16 taps the kernels of 1. and 3. compute w from x on the fly instead (FUSED): x is read
// twice, y written once, and no FIR pass runs.  With many channels (iir_walk_kernel) a workgroup walks a whole
// channel and carries the state from tile to tile itself: one read and one write per sample.
//
// Sequential path (iir_seq_kernel): one lane per channel, samples in order, staged through LDS in 64x64
// blocks so that global accesses stay coalesced; with EXACT it is bit-identical to the reference loop.
#include <hip/hip_runtime.h>

#include <cmath>

#include "iir_kernels.h"

namespace dspx {
namespace {

constexpr int kBlock = 256;
constexpr int kR = kIirTile / kBlock;  // samples per lane
static_assert(kR == 16 && (1 << kIirLevels) == kBlock, "scan geometry");
constexpr int kLaneStride = kR + 2;                  // tile image: sample q at q + 2*(q/16); lanes 144 bytes apart,
constexpr int kImg = (kBlock + 1) * kLaneStride;     // which keeps 16-byte LDS reads aligned and conflict-free;
                                                     // one more slot in front holds the 16 samples before the tile

template <bool EXACT> __device__ __forceinline__ double tail_step(double a, double yp, double acc) {
    if constexpr (EXACT)
        return __dsub_rn(acc, __dmul_rn(a, yp));
    else
        return __builtin_fma(-a, yp, acc);
}

// One step of the recurrence for a lane: st[0] is the most recent output.
template <int P> __device__ __forceinline__ double advance(double (&st)[P], const IirCoef& coef, double w) {
    double acc = w;
#pragma unroll
    for (int j = 0; j < P; ++j) acc = __builtin_fma(-coef.a[j], st[j], acc);
#pragma unroll
    for (int k = P - 1; k > 0; --k) st[k] = st[k - 1];
    st[0] = acc;
    return acc;
}

// Double-double helpers (the library is built with -ffp-contract=off and without fast-math, so every
// operation below is the IEEE operation written).
struct dd {
    double hi, lo;
};
__device__ __forceinline__ dd two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd quick_two_sum(double a, double b) {  // |a| >= |b|
    const double s = a + b;
    return {s, b - (s - a)};
}
__device__ __forceinline__ dd dd_add(dd a, dd b) {
    dd s = two_sum(a.hi, b.hi);
    s.lo += a.lo + b.lo;
    return quick_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ dd dd_mul(dd a, dd b) {
    const double p = a.hi * b.hi;
    double e = __builtin_fma(a.hi, b.hi, -p);
    e = __builtin_fma(a.hi, b.lo, e);
    e = __builtin_fma(a.lo, b.hi, e);
    return quick_two_sum(p, e);
}

// s <- s + M * p as a compensated dot product (products and running sum split exactly into a leading part
// and an error term; the error terms are added up apart), rounded to double once per component
template <int P>
__device__ __forceinline__ void add_matvec(double (&s)[P], const IirMatrix& M, const double (&p)[P]) {
#pragma unroll
    for (int r = 0; r < P; ++r) {
        double lead = s[r], rest = 0.0;
#pragma unroll
        for (int c = 0; c < P; ++c) {
            const double prod = M.hi[r][c] * p[c];
            double e = __builtin_fma(M.hi[r][c], p[c], -prod);
            e = __builtin_fma(M.lo[r][c], p[c], e);
            const dd t = two_sum(lead, prod);
            lead = t.hi;
            rest += t.lo + e;
        }
        s[r] = lead + rest;
    }
}

// The same with plain fused multiply-adds, for powers whose entries are small (see IirPlan).
template <int P>
__device__ __forceinline__ void add_matvec_plain(double (&s)[P], const IirMatrix& M, const double (&p)[P]) {
#pragma unroll
    for (int r = 0; r < P; ++r) {
        double acc = s[r];
#pragma unroll
        for (int c = 0; c < P; ++c) acc = __builtin_fma(M.hi[r][c], p[c], acc);
        s[r] = acc;
    }
}

// Inclusive scan over the 256 lanes of a workgroup: s_k <- sum_{i <= k} M0^(k-i) s_i, with M[d] = M0^(2^d).
// buf: 2 * P * kBlock doubles of LDS (component-major, so lanes hit consecutive banks).
template <int P>
__device__ __forceinline__ void scan_states(double (&s)[P], const IirMatrix* M, unsigned plain, double* buf, int tid) {
    int cur = 0;
    for (int d = 0; d < kIirLevels; ++d) {
        double* b = buf + cur * P * kBlock;
#pragma unroll
        for (int k = 0; k < P; ++k) b[k * kBlock + tid] = s[k];
        __syncthreads();
        const int from = tid - (1 << d);
        if (from >= 0) {
            double p[P];
#pragma unroll
            for (int k = 0; k < P; ++k) p[k] = b[k * kBlock + from];
            if ((plain >> d) & 1u)  // uniform
                add_matvec_plain<P>(s, M[d], p);
            else
                add_matvec<P>(s, M[d], p);
        }
        cur ^= 1;  // the next level writes the other half: nobody can still be reading it (one barrier between)
    }
}

// Loads the tile into the LDS image (zeros past the end of the channel): sample q of the tile at
// kLaneStride + q + 2*(q/16), i.e. lane t's samples in slot t + 1.
__device__ __forceinline__ void stage_tile(const double* __restrict__ src, long long left, bool vec2, double* img,
                                           int tid) {
    double* body = img + kLaneStride;
    if (vec2 && left >= kIirTile) {  // whole tile, 16-byte loads (uniform branch)
#pragma unroll
        for (int i = 0; i < kR / 2; ++i) {
            const int q = 2 * (tid + i * kBlock);
            *reinterpret_cast<double2*>(body + q + 2 * (q >> 4)) = reinterpret_cast<const double2*>(src)[tid + i * kBlock];
        }
    } else {
#pragma unroll
        for (int i = 0; i < kR; ++i) {
            const int q = tid + i * kBlock;
            body[q + 2 * (q >> 4)] = q < left ? src[q] : 0.0;
        }
    }
}

// The lane's 16 values of the FIR part: read from the staged tile, or (FUSED) computed from the staged input
// and the 16 samples before the lane's own, taps ascending, one fused multiply-add per tap.
template <bool FUSED>
__device__ __forceinline__ void lane_fir_part(const double* img, int tid, const IirFront& f, double (&w)[kR]) {
    const double2* mine = reinterpret_cast<const double2*>(img + kLaneStride * (tid + 1));
    if constexpr (!FUSED) {
#pragma unroll
        for (int i = 0; i < kR / 2; ++i) {
            const double2 v = mine[i];
            w[2 * i] = v.x, w[2 * i + 1] = v.y;
        }
    } else {
        double X[2 * kR];
        const double2* prev = reinterpret_cast<const double2*>(img + kLaneStride * tid);
#pragma unroll
        for (int i = 0; i < kR / 2; ++i) {
            const double2 p = prev[i], v = mine[i];
            X[2 * i] = p.x, X[2 * i + 1] = p.y;
            X[kR + 2 * i] = v.x, X[kR + 2 * i + 1] = v.y;
        }
#pragma unroll
        for (int t = 0; t < kR; ++t) w[t] = 0.0;
#pragma unroll
        for (int j = 0; j < kIirMaxFusedTaps; ++j) {
            if (j < f.nb) {  // uniform
                const double bj = f.b[j];
#pragma unroll
                for (int t = 0; t < kR; ++t) w[t] = __builtin_fma(bj, X[kR + t - j], w[t]);
            }
        }
    }
}

// Stages what a tile needs and returns the lane's FIR part.  FUSED: the input tile and the 16 samples before
// it (from x, from the history in front of x, or zeros).
template <bool FUSED>
__device__ __forceinline__ void load_fir_part(const double* __restrict__ y, const IirFront& f, const IirShape& s,
                                              unsigned chan, long long start, double* img, int tid, double (&w)[kR]) {
    static_assert(kIirMaxFusedTaps - 1 <= kR, "the slot in front of the tile holds the samples the taps reach");
    if constexpr (FUSED) {
        const double* __restrict__ xc = f.x + (unsigned long long)chan * f.x_stride;
        stage_tile(xc + start, s.n - start, f.vec2 != 0, img, tid);
        if (tid < kR) {
            const long long g = start - kR + tid;
            double v = 0.0;
            if (g >= 0)
                v = xc[g];
            else if (f.xhist != nullptr && g >= -(long long)(f.nb - 1))
                v = f.xhist[(unsigned long long)chan * f.xhist_stride + (f.nb - 1) + g];
            img[tid] = v;
        }
    } else {
        stage_tile(y + (unsigned long long)chan * s.y_stride + start, s.n - start, s.vec2 != 0, img, tid);
    }
    __syncthreads();
    lane_fir_part<FUSED>(img, tid, f, w);
}

__device__ __forceinline__ void locate(const IirShape& s, unsigned& chan, unsigned& tile) {
    chan = blockIdx.x / s.tiles_per_chan;
    tile = blockIdx.x - chan * s.tiles_per_chan;
}

// Computes (A^16)^(2^d) and (A^4096)^(2^d) by repeated squaring in double-double; one workgroup of 64
// threads, thread = element.
__global__ __launch_bounds__(64) void iir_tables_kernel(IirCoef coef, int order, IirTables* out) {
    __shared__ double mh[kIirMaxOrder][kIirMaxOrder], ml[kIirMaxOrder][kIirMaxOrder];
    const int r = threadIdx.x / kIirMaxOrder, c = threadIdx.x % kIirMaxOrder;
    // companion matrix, padded with zeros beyond `order`
    double v = 0.0;
    if (r < order && c < order) v = r == 0 ? -coef.a[c] : (c == r - 1 ? 1.0 : 0.0);
    mh[r][c] = v, ml[r][c] = 0.0;
    __syncthreads();
    auto square = [&]() {
        dd acc{0.0, 0.0};
        for (int k = 0; k < kIirMaxOrder; ++k) acc = dd_add(acc, dd_mul(dd{mh[r][k], ml[r][k]}, dd{mh[k][c], ml[k][c]}));
        __syncthreads();
        mh[r][c] = acc.hi, ml[r][c] = acc.lo;
        __syncthreads();
        return acc;
    };
    dd e{v, 0.0};
    for (int i = 0; i < 4; ++i) e = square();  // A^16
    for (int d = 0; d < kIirLevels; ++d) {
        out->lane[d].hi[r][c] = e.hi, out->lane[d].lo[r][c] = e.lo;
        e = square();
    }
    // e = (A^16)^256 = A^4096
    for (int d = 0; d < kIirTileLevels; ++d) {
        out->tile[d].hi[r][c] = e.hi, out->tile[d].lo[r][c] = e.lo;
        if (d + 1 < kIirTileLevels) e = square();
    }
}

template <int P, bool FUSED>
__global__ __launch_bounds__(kBlock) void iir_tile_state_kernel(IirCoef coef, IirFront front,
                                                                const double* __restrict__ y,
                                                                const IirTables* __restrict__ tables,
                                                                double* __restrict__ states, IirShape s) {
    __shared__ __align__(16) double img[kImg];
    __shared__ double buf[2 * P * kBlock];
    const int tid = threadIdx.x;
    unsigned chan, tile;
    locate(s, chan, tile);
    const long long start = (long long)tile * kIirTile;
    double w[kR];
    load_fir_part<FUSED>(y, front, s, chan, start, img, tid, w);
    double st[P];
#pragma unroll
    for (int k = 0; k < P; ++k) st[k] = 0.0;
#pragma unroll
    for (int t = 0; t < kR; ++t) advance<P>(st, coef, w[t]);
    scan_states<P>(st, tables->lane, s.lane_plain, buf, tid);
    if (tid == kBlock - 1) {
        double* dst = states + ((unsigned long long)chan * s.tiles_per_chan + tile) * P;
#pragma unroll
        for (int k = 0; k < P; ++k) dst[k] = st[k];
    }
}

// One workgroup per channel.  On entry states[chan][t] is the state tile t reaches from zero; on exit it is
// the state tile t starts from.  A lane walks 16 consecutive tiles by itself (state <- A^4096 * state + tile
// state), the 256 lane results are joined by the same scan as the lanes of a tile, and the lane walks its
// tiles again from the state it really starts from: 4096 tiles per pass.
template <int P>
__global__ __launch_bounds__(kBlock) void iir_tile_scan_kernel(const IirTables* __restrict__ tables,
                                                               double* __restrict__ states,
                                                               const double* __restrict__ state_in, IirShape s) {
    static_assert((1 << 4) == kIirTileRun, "tile[4] carries a state across one lane's run");
    __shared__ double buf[2 * P * kBlock];
    __shared__ double last[P];
    const int tid = threadIdx.x;
    const unsigned chan = blockIdx.x;
    double* mine = states + (unsigned long long)chan * s.tiles_per_chan * P;
    const IirMatrix& step = tables->tile[0];
    double carry[P];
#pragma unroll
    for (int k = 0; k < P; ++k)  // state_in is oldest first
        carry[k] = state_in ? state_in[(unsigned long long)chan * s.state_stride + (P - 1 - k)] : 0.0;
    for (unsigned base = 0; base < s.tiles_per_chan; base += kBlock * kIirTileRun) {
        const unsigned t0 = base + tid * kIirTileRun;
        // tiles past the end count as tiles that reach the zero state from zero: nothing after them is used
        double st[P], z[P];
#pragma unroll
        for (int k = 0; k < P; ++k) st[k] = 0.0;
        for (int j = 0; j < kIirTileRun; ++j) {
            const unsigned t = t0 + j;
#pragma unroll
            for (int k = 0; k < P; ++k) z[k] = t < s.tiles_per_chan ? mine[(unsigned long long)t * P + k] : 0.0;
            if (s.tile_plain & 1u)  // z + A^4096 * st
                add_matvec_plain<P>(z, step, st);
            else
                add_matvec<P>(z, step, st);
#pragma unroll
            for (int k = 0; k < P; ++k) st[k] = z[k];
        }
        if (tid == 0) add_matvec<P>(st, tables->tile[4], carry);  // what the first lane of this pass ends in
        scan_states<P>(st, tables->tile + 4, s.tile_plain >> 4, buf, tid);
        // a lane starts from what the lane before it ends in
        __syncthreads();
#pragma unroll
        for (int k = 0; k < P; ++k) buf[k * kBlock + tid] = st[k];
        if (tid == kBlock - 1) {
#pragma unroll
            for (int k = 0; k < P; ++k) last[k] = st[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < P; ++k) st[k] = tid == 0 ? carry[k] : buf[k * kBlock + tid - 1];
        for (int j = 0; j < kIirTileRun; ++j) {
            const unsigned t = t0 + j;
            if (t >= s.tiles_per_chan) break;
#pragma unroll
            for (int k = 0; k < P; ++k) {
                z[k] = mine[(unsigned long long)t * P + k];
                mine[(unsigned long long)t * P + k] = st[k];  // the state tile t starts from
            }
            if (s.tile_plain & 1u)
                add_matvec_plain<P>(z, step, st);
            else
                add_matvec<P>(z, step, st);
#pragma unroll
            for (int k = 0; k < P; ++k) st[k] = z[k];
        }
#pragma unroll
        for (int k = 0; k < P; ++k) carry[k] = last[k];
        __syncthreads();
    }
}

// One tile, start to finish: stage, run every lane from zero, join the lane states with the tile's starting
// state `in` folded into lane 0, run every lane again from the state it really starts from, store.  On return
// `in` holds the state the tile ends in (every thread has it).
template <int P, bool FUSED>
__device__ __forceinline__ void apply_tile(const IirCoef& coef, const IirFront& front, double* __restrict__ y,
                                           const IirTables* __restrict__ tables, const IirShape& s, unsigned chan,
                                           unsigned tile, double (&in)[P], double* img, double* buf, double* last,
                                           int tid) {
    const long long start = (long long)tile * kIirTile;
    const long long left = s.n - start;
    double* __restrict__ yt = y + (unsigned long long)chan * s.y_stride + start;
    double w[kR];
    load_fir_part<FUSED>(y, front, s, chan, start, img, tid, w);
    double st[P];
#pragma unroll
    for (int k = 0; k < P; ++k) st[k] = 0.0;
#pragma unroll
    for (int t = 0; t < kR; ++t) advance<P>(st, coef, w[t]);
    if (tid == 0) add_matvec<P>(st, tables->lane[0], in);
    scan_states<P>(st, tables->lane, s.lane_plain, buf, tid);
    // a lane starts from what the lane before it ends in; the last lane ends in the tile's final state
    __syncthreads();
#pragma unroll
    for (int k = 0; k < P; ++k) buf[k * kBlock + tid] = st[k];
    if (tid == kBlock - 1) {
#pragma unroll
        for (int k = 0; k < P; ++k) last[k] = st[k];
    }
    __syncthreads();
    if (tid > 0) {
#pragma unroll
        for (int k = 0; k < P; ++k) in[k] = buf[k * kBlock + tid - 1];
    }
    double2* mine = reinterpret_cast<double2*>(img + kLaneStride * (tid + 1));
#pragma unroll
    for (int i = 0; i < kR / 2; ++i) {
        const double a = advance<P>(in, coef, w[2 * i]);
        const double b = advance<P>(in, coef, w[2 * i + 1]);
        mine[i] = make_double2(a, b);
    }
    __syncthreads();
    if (s.vec2 && left >= kIirTile) {
#pragma unroll
        for (int i = 0; i < kR / 2; ++i) {
            const int q = 2 * (tid + i * kBlock);
            reinterpret_cast<double2*>(yt)[tid + i * kBlock] =
                *reinterpret_cast<const double2*>(img + kLaneStride + q + 2 * (q >> 4));
        }
    } else {
#pragma unroll
        for (int i = 0; i < kR; ++i) {
            const int q = tid + i * kBlock;
            if (q < left) yt[q] = img[kLaneStride + q + 2 * (q >> 4)];
        }
    }
#pragma unroll
    for (int k = 0; k < P; ++k) in[k] = last[k];
}

template <int P, bool FUSED>
__global__ __launch_bounds__(kBlock) void iir_apply_kernel(IirCoef coef, IirFront front, double* __restrict__ y,
                                                           const IirTables* __restrict__ tables,
                                                           const double* __restrict__ states, IirShape s) {
    __shared__ __align__(16) double img[kImg];
    __shared__ double buf[2 * P * kBlock];
    __shared__ double last[P];
    const int tid = threadIdx.x;
    unsigned chan, tile;
    locate(s, chan, tile);
    const double* src = states + ((unsigned long long)chan * s.tiles_per_chan + tile) * P;
    double in[P];
#pragma unroll
    for (int k = 0; k < P; ++k) in[k] = src[k];  // the state the tile starts from
    apply_tile<P, FUSED>(coef, front, y, tables, s, chan, tile, in, img, buf, last, tid);
}

// Many channels: one workgroup walks a whole channel tile by tile and carries the state itself, so every
// sample is read once and written once and one scan per tile suffices (no tile states, no second pass).
template <int P, bool FUSED>
__global__ __launch_bounds__(kBlock) void iir_walk_kernel(IirCoef coef, IirFront front, double* __restrict__ y,
                                                          const IirTables* __restrict__ tables,
                                                          const double* __restrict__ state_in, IirShape s) {
    __shared__ __align__(16) double img[kImg];
    __shared__ double buf[2 * P * kBlock];
    __shared__ double last[P];
    const int tid = threadIdx.x;
    const unsigned chan = blockIdx.x;
    double in[P];
#pragma unroll
    for (int k = 0; k < P; ++k)  // state_in is oldest first
        in[k] = state_in ? state_in[(unsigned long long)chan * s.state_stride + (P - 1 - k)] : 0.0;
    for (unsigned tile = 0; tile < s.tiles_per_chan; ++tile) {
        apply_tile<P, FUSED>(coef, front, y, tables, s, chan, tile, in, img, buf, last, tid);
        __syncthreads();  // the image and `last` are free for the next tile
    }
}

// The recursive tail of lfilter<float> (reference dsp.hpp:366-368 with T = float), in place on the FIR part: one lane
// per channel, samples in order.  `y_i -= denominators[j] * y[i-j]` is evaluated in PT, the type denominators[j] * float
// promotes to (float for float denominators, double for double ones): the product and the difference are rounded
// separately in PT and the difference is converted back to float, j ascending -- bit-identical to the reference loop.
// Functional, not fast: the recurrence is sequential along a channel.
template <typename PT>
__global__ __launch_bounds__(64) void iir_narrow_tail_kernel(IirCoef coef, int order, float* __restrict__ y,
                                                             unsigned long long y_stride, unsigned long long nchan,
                                                             long long n) {
    const unsigned long long c = (unsigned long long)blockIdx.x * 64 + threadIdx.x;
    if (c >= nchan) return;
    float* __restrict__ yc = y + c * y_stride;
    PT a[kIirMaxOrder];
#pragma unroll
    for (int j = 0; j < kIirMaxOrder; ++j) a[j] = static_cast<PT>(coef.a[j]);
    float st[kIirMaxOrder];  // st[j] = y[i - 1 - j]
#pragma unroll
    for (int j = 0; j < kIirMaxOrder; ++j) st[j] = 0.0f;
    for (long long i = 0; i < n; ++i) {
        float v = yc[i];
#pragma unroll
        for (int j = 0; j < kIirMaxOrder; ++j) {
            if (j < order && j < i) {
#pragma clang fp contract(off)
                const PT prod = a[j] * static_cast<PT>(st[j]);
                const PT diff = static_cast<PT>(v) - prod;
                v = static_cast<float>(diff);
            }
        }
#pragma unroll
        for (int k = kIirMaxOrder - 1; k > 0; --k) st[k] = st[k - 1];
        st[0] = v;
        yc[i] = v;
    }
}

// 64 channels per workgroup (one wave), one lane per channel, 64 samples per pass through LDS.
constexpr int kSeqChans = 64, kSeqSamples = 64;
template <int P, bool EXACT>
__global__ __launch_bounds__(kSeqChans) void iir_seq_kernel(IirCoef coef, double* __restrict__ y,
                                                            const double* __restrict__ state_in, IirShape s) {
    __shared__ double blk[kSeqChans][kSeqSamples + 1];
    const int lane = threadIdx.x;
    const unsigned long long c0 = (unsigned long long)blockIdx.x * kSeqChans;
    const int rows = (int)min((unsigned long long)kSeqChans, s.nchan - c0);
    const bool have = lane < rows;
    double st[P];
    int known = 0;  // how many outputs before the current sample exist, up to P (the reference skips the others)
#pragma unroll
    for (int k = 0; k < P; ++k) st[k] = 0.0;
    if (state_in && have) {
#pragma unroll
        for (int k = 0; k < P; ++k) st[k] = state_in[(c0 + lane) * s.state_stride + (P - 1 - k)];
        known = P;
    }
    for (long long i0 = 0; i0 < s.n; i0 += kSeqSamples) {
        const int cols = (int)min((long long)kSeqSamples, s.n - i0);
        for (int r = 0; r < rows; ++r)
            if (lane < cols) blk[r][lane] = y[(c0 + r) * s.y_stride + i0 + lane];
        __syncthreads();
        if (have) {
            int t = 0;
            // until P outputs exist the reference subtracts only the terms it has (dsp.hpp:359)
            for (; t < cols && known < P; ++t) {
                double acc = blk[lane][t];
#pragma unroll
                for (int j = 0; j < P; ++j)
                    if (j < known) acc = tail_step<EXACT>(coef.a[j], st[j], acc);
#pragma unroll
                for (int k = P - 1; k > 0; --k) st[k] = st[k - 1];
                st[0] = acc;
                ++known;
                blk[lane][t] = acc;
            }
            // steady state, eight samples at a time: their LDS reads are issued together, off the dependent chain
            for (; t + 8 <= cols; t += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = blk[lane][t + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    double acc = v[u];
#pragma unroll
                    for (int j = 0; j < P; ++j) acc = tail_step<EXACT>(coef.a[j], st[j], acc);
#pragma unroll
                    for (int k = P - 1; k > 0; --k) st[k] = st[k - 1];
                    st[0] = acc;
                    v[u] = acc;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) blk[lane][t + u] = v[u];
            }
            for (; t < cols; ++t) {
                double acc = blk[lane][t];
#pragma unroll
                for (int j = 0; j < P; ++j) acc = tail_step<EXACT>(coef.a[j], st[j], acc);
#pragma unroll
                for (int k = P - 1; k > 0; --k) st[k] = st[k - 1];
                st[0] = acc;
                blk[lane][t] = acc;
            }
        }
        __syncthreads();
        for (int r = 0; r < rows; ++r)
            if (lane < cols) y[(c0 + r) * s.y_stride + i0 + lane] = blk[r][lane];
        __syncthreads();
    }
}

// From this many channels on a launch has enough workgroups when each walks one channel.
constexpr unsigned kWalkMinChans = 1024;

template <int P>
hipError_t scan_path(const IirCoef& coef, double* y, const double* state_in, const IirShape& s, const IirFront* front,
                     IirTables* tables, double* states, hipStream_t stream) {
    const dim3 tiles(s.nchan * s.tiles_per_chan), block(kBlock);
    const IirFront f = front ? *front : IirFront{};
    hipLaunchKernelGGL(iir_tables_kernel, dim3(1), dim3(64), 0, stream, coef, P, tables);
    if (s.nchan >= kWalkMinChans) {
        if (front)
            hipLaunchKernelGGL((iir_walk_kernel<P, true>), dim3(s.nchan), block, 0, stream, coef, f, y, tables, state_in, s);
        else
            hipLaunchKernelGGL((iir_walk_kernel<P, false>), dim3(s.nchan), block, 0, stream, coef, f, y, tables, state_in, s);
        return hipGetLastError();
    }
    if (front)
        hipLaunchKernelGGL((iir_tile_state_kernel<P, true>), tiles, block, 0, stream, coef, f, y, tables, states, s);
    else
        hipLaunchKernelGGL((iir_tile_state_kernel<P, false>), tiles, block, 0, stream, coef, f, y, tables, states, s);
    hipLaunchKernelGGL(iir_tile_scan_kernel<P>, dim3(s.nchan), block, 0, stream, tables, states, state_in, s);
    if (front)
        hipLaunchKernelGGL((iir_apply_kernel<P, true>), tiles, block, 0, stream, coef, f, y, tables, states, s);
    else
        hipLaunchKernelGGL((iir_apply_kernel<P, false>), tiles, block, 0, stream, coef, f, y, tables, states, s);
    return hipGetLastError();
}

template <int P>
hipError_t seq_path(const IirCoef& coef, double* y, const double* state_in, const IirShape& s, bool exact,
                    hipStream_t stream) {
    const unsigned groups = (s.nchan + kSeqChans - 1) / kSeqChans;
    if (exact)
        hipLaunchKernelGGL((iir_seq_kernel<P, true>), dim3(groups), dim3(kSeqChans), 0, stream, coef, y, state_in, s);
    else
        hipLaunchKernelGGL((iir_seq_kernel<P, false>), dim3(groups), dim3(kSeqChans), 0, stream, coef, y, state_in, s);
    return hipGetLastError();
}

constexpr size_t kTablesBytes = (sizeof(IirTables) + 255) & ~size_t(255);

}  // namespace

size_t iir_workspace_bytes(size_t nchan, size_t n, int order) {
    if (nchan >= kWalkMinChans) return kTablesBytes;  // iir_walk_kernel carries the state itself: no tile states
    const size_t tiles = (n + kIirTile - 1) / kIirTile;
    return kTablesBytes + nchan * tiles * (size_t)order * sizeof(double);
}

IirPlan iir_plan(const double* a, int order) {
    // the same squarings as iir_tables_kernel: after k of them m = A^(2^k); lane[d] is k = 4 + d, tile[d] is k = 12 + d
    IirPlan plan{true, 0u, 0u};
    const double small = 1.0 / (16.0 * (order + 1));
    double m[kIirMaxOrder][kIirMaxOrder] = {}, t[kIirMaxOrder][kIirMaxOrder];
    for (int r = 0; r < order; ++r)
        for (int c = 0; c < order; ++c) m[r][c] = r == 0 ? -a[c] : (c == r - 1 ? 1.0 : 0.0);
    for (int k = 1; k <= 4 + kIirLevels + kIirTileLevels - 1; ++k) {
        double worst_row = 0.0;
        for (int r = 0; r < order; ++r) {
            double row = 0.0;
            for (int c = 0; c < order; ++c) {
                double acc = 0.0;
                for (int j = 0; j < order; ++j) acc += m[r][j] * m[j][c];
                if (!(std::fabs(acc) < 1e150)) return IirPlan{false, 0u, 0u};  // also for NaN
                t[r][c] = acc;
                row += std::fabs(acc);
            }
            worst_row = std::fmax(worst_row, row);
        }
        for (int r = 0; r < order; ++r)
            for (int c = 0; c < order; ++c) m[r][c] = t[r][c];
        if (worst_row <= small) {
            if (k >= 4 && k < 4 + kIirLevels) plan.lane_plain |= 1u << (k - 4);
            if (k >= 4 + kIirLevels) plan.tile_plain |= 1u << (k - 4 - kIirLevels);
        }
    }
    return plan;
}

#define DSPX_IIR_DISPATCH(call)       \
    switch (order) {                  \
    case 1: return call(1);           \
    case 2: return call(2);           \
    case 3: return call(3);           \
    case 4: return call(4);           \
    case 5: return call(5);           \
    case 6: return call(6);           \
    case 7: return call(7);           \
    case 8: return call(8);           \
    default: return hipErrorInvalidValue; \
    }

hipError_t launch_iir_scan(const IirCoef& coef, int order, double* y, const double* state_in, const IirShape& shape,
                           const IirFront* front, void* work, hipStream_t stream) {
    if (shape.nchan == 0 || shape.n == 0) return hipSuccess;
    // a launch of 2^32 or more threads is dropped by the runtime without an error: refuse it here
    if ((unsigned long long)shape.nchan * shape.tiles_per_chan * kBlock >= (1ull << 32)) return hipErrorInvalidConfiguration;
    IirTables* tables = static_cast<IirTables*>(work);
    double* states = reinterpret_cast<double*>(static_cast<char*>(work) + kTablesBytes);
#define DSPX_IIR_SCAN(P) scan_path<P>(coef, y, state_in, shape, front, tables, states, stream)
    DSPX_IIR_DISPATCH(DSPX_IIR_SCAN)
#undef DSPX_IIR_SCAN
}

hipError_t launch_iir_sequential(const IirCoef& coef, int order, double* y, const double* state_in,
                                 const IirShape& shape, bool exact, hipStream_t stream) {
    if (shape.nchan == 0 || shape.n == 0) return hipSuccess;
#define DSPX_IIR_SEQ(P) seq_path<P>(coef, y, state_in, shape, exact, stream)
    DSPX_IIR_DISPATCH(DSPX_IIR_SEQ)
#undef DSPX_IIR_SEQ
}

hipError_t launch_iir_narrow_tail(const IirCoef& coef, int order, bool in_double, float* y, unsigned long long y_stride,
                                  unsigned long long nchan, long long n, hipStream_t stream) {
    if (nchan == 0 || n == 0 || order == 0) return hipSuccess;
    const dim3 grid((unsigned)((nchan + 63) / 64)), block(64);
    if (in_double)
        hipLaunchKernelGGL(iir_narrow_tail_kernel<double>, grid, block, 0, stream, coef, order, y, y_stride, nchan, n);
    else
        hipLaunchKernelGGL(iir_narrow_tail_kernel<float>, grid, block, 0, stream, coef, order, y, y_stride, nchan, n);
    return hipGetLastError();
}

}  // namespace dspx
