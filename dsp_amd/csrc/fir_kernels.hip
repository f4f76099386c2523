// FP64 direct-form FIR kernels for gfx950 (MI355X).
//
// Computes y[i] = sum_j taps[j] * x[i-j] with taps ascending per output, i.e. the
// inner loop of the reference lfilter (dsp.hpp:356-365) for a denominator of size 1,
// with the samples before x[0] taken as zero (or from an explicit history buffer).
// Two kernels: fir_f64_kernel on the vector pipe (described here) and fir_f64_mfma_kernel on
// the matrix pipe (described where it starts, further down).
//
// Work decomposition (vector pipe)
//   * one workgroup (256 threads = 4 waves) produces a tile of 2048 consecutive
//     outputs of one channel; thread t owns outputs 8t .. 8t+7 of the tile;
//   * taps are consumed in stages of at most 256; per stage the workgroup copies the
//     input window [tile_start - j0 - 256, tile_start - j0 + 2048) from HBM/L2 into LDS
//     with coalesced 16-byte loads;
//   * inside a stage a thread handles 16 taps at a time: it reads the 24 samples those
//     taps touch from LDS (12 x ds_read_b128) and issues 128 v_fmac_f64 whose tap operand
//     is an SGPR pair (scalar loads), so the VALU issues nothing but FMAs in the steady
//     state; latency of the LDS and scalar loads is covered by the other resident waves;
//   * accumulators go back through LDS so that the tile is stored with coalesced
//     16-byte stores.
//
// LDS image: slot q of the window is stored at q + 2*(q/8) (two pad slots per 8 samples).
// Thread t reads slots 8t + const, which the padding turns into a stride of 10 doubles
// (80 bytes): 16-byte aligned for ds_read_b128 and conflict-free, because the 16 lanes
// of a ds_read_b128 group start at 16 distinct multiples of 4 banks.
#include <hip/hip_runtime.h>

#include "fir_kernels.h"

namespace dspx {

namespace {

constexpr int kBlock = 256;               // threads per workgroup
// A launch of 2^32 or more threads is dropped by the runtime without an error: refuse it here.
constexpr bool grid_fits(unsigned workgroups) { return (unsigned long long)workgroups * kBlock < (1ull << 32); }
constexpr int kR = 8;                     // outputs per thread
constexpr int kGroup = 16;                // taps per unrolled group
constexpr int kWin = kGroup + kR;         // samples a group reads per thread (23 needed, 24 loaded)
constexpr int kTile = kBlock * kR;        // outputs per workgroup
constexpr int kStageTaps = 256;           // taps per LDS stage
constexpr int kHalo = kStageTaps;         // window slots in front of the tile
constexpr int kSlots = kHalo + kTile;     // window slots per stage
constexpr int kLdsPairs = (kSlots + kSlots / 4) / 2;             // 16-byte pairs in the padded image
constexpr int kStageIters = (kSlots / 2 + kBlock - 1) / kBlock;  // 16-byte loads per thread per stage
constexpr int kBankStride = kTile - kR;   // outputs per tile in bank mode (the tile's last thread only supplies c[i+1])

static_assert(kR == 8 && kGroup == 16, "the window indexing below assumes 8 outputs per thread, 16-tap groups");
static_assert(kHalo % 16 == 0 && kTile % 8 == 0 && kStageTaps % kGroup == 0, "padding groups must align");
static_assert(kStageIters == 5 && (kStageIters - 1) * 2 * kBlock <= kTile, "only the last staging pass is partial");

// The image stores slot q at double index q + 2*(q/8); an even slot q and its neighbour
// q+1 form the 16-byte pair with this index.
__host__ __device__ constexpr int pair_of(int q) { return (q >> 1) + (q >> 3); }

template <bool EXACT> __device__ __forceinline__ double mac(double b, double w, double acc) {
    if constexpr (EXACT) {
#pragma clang fp contract(off)
        const double prod = b * w;
        return acc + prod;
    } else {
        return __builtin_fma(b, w, acc);
    }
}

// One input sample of a channel at signed position g: history (or zero) before the
// channel start, zero past its end.
__device__ __forceinline__ double sample_at(const double* __restrict__ xc, const double* __restrict__ hc,
                                            long long g, long long n, int ntaps) {
    if (g >= 0) return g < n ? xc[g] : 0.0;
    if (hc != nullptr && g >= -(long long)(ntaps - 1)) return hc[(ntaps - 1) + g];
    return 0.0;
}

// Loads the 24 samples that taps [16g, 16g+16) of a stage touch for one thread.
// W[i] is the sample that tap 16g + t multiplies into output r when i = t - r + 7, i.e.
// image slot kHalo - 16g + 8*tid + 7 - i.  Slots (i+1, i) for even i form the aligned pair
// pair_of(kHalo + 8*tid + 6) - 10g - 5*(i/8) - (i%8)/2; `top` points at the pair of i = 0, g = 0.
__device__ __forceinline__ void load_window(const double2* top, int g, double (&W)[kWin]) {
    const double2* p = top - 10 * g;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const double2 v = p[-5 * m - h];
            W[8 * m + 2 * h] = v.y;
            W[8 * m + 2 * h + 1] = v.x;
        }
    }
    // W[23] is never multiplied; keeping it "used" stops the compiler from splitting the
    // bottom row of 16-byte reads into 8-byte ones (ds_read2_b64 is slower and conflicts).
    asm volatile("" ::"v"(W[kWin - 1]));
}

// BANK = false: nchan independent channels, one tap vector (the lfilter path).
// BANK = true: one input, a bank of tap vectors of different lengths (the wavelet path, reference
// dsp.hpp:1136-1217): row r computes c[t] = sum_k taps_r[k] * x[t + advance_r - k] with zeros outside
// [0, n), and stores (c[i+1] - c[i]) * scale_r for i < n.  A tile still computes 2048 values of c but
// advances by kBankStride = 2040 outputs, so that the c[i+1] (or c[i+2], see `odd`) of its last
// output is its own.
template <bool EXACT, bool VEC2, bool BANK>
__global__ __launch_bounds__(kBlock) void fir_f64_kernel(const double* __restrict__ taps,
                                                          const double* __restrict__ x,
                                                          const double* __restrict__ hist, double* __restrict__ y,
                                                          const BankRow* __restrict__ bank, FirShape s) {
    __shared__ double2 lds[kLdsPairs];
    const int tid = threadIdx.x;

    // Blocks b and b+8 share an XCD (and its L2).  Tiles that are neighbours in time share a halo,
    // so every XCD gets contiguous runs of tiles and finds the shared samples in its own L2.
    const unsigned b = blockIdx.x;
    const unsigned grp = b % 8, k = b / 8;
    unsigned chan, tile;
    if constexpr (BANK) {
        // Rows of a bank differ in cost (tap count), so each row is cut into 8 runs, one per
        // XCD, and all XCDs work through the rows together (s.div_* divide by the run length).
        const unsigned run = (s.tiles_per_chan + 7) / 8;
        chan = (unsigned)(((unsigned long long)k * s.div_magic) >> s.div_shift);
        tile = grp * run + (k - chan * run);
        if (tile >= s.tiles_per_chan) return;  // the whole workgroup leaves together
    } else {
        // Equal-cost tiles: XCD g takes the g-th eighth of all tiles (s.div_* divide by tiles_per_chan).
        const unsigned q8 = s.total_tiles / 8, r8 = s.total_tiles % 8;
        const unsigned tile_id = (grp < r8 ? grp * (q8 + 1) : r8 * (q8 + 1) + (grp - r8) * q8) + k;
        chan = (unsigned)(((unsigned long long)tile_id * s.div_magic) >> s.div_shift);
        tile = tile_id - chan * s.tiles_per_chan;
    }
    const long long n = s.n;
    long long tile_start, pos0;  // first output of the tile; input position its tap 0 multiplies
    int ntaps;
    int odd = 0;  // bank mode: 1 when the row's advance is odd (the tile then computes c one sample early)
    double post = 1.0;
    const double* __restrict__ xc;
    const double* __restrict__ hc;
    const double* __restrict__ tapsc;
    double* __restrict__ yrow;
    if constexpr (BANK) {
        const BankRow row = bank[chan];
        tile_start = (long long)tile * kBankStride;
        // Keep the staged window 16-byte aligned in memory: for an odd advance compute
        // c'[i] = c[i + advance - 1] instead and shift the difference by one in the epilogue.
        odd = row.advance & 1;
        pos0 = tile_start + row.advance - odd;
        ntaps = row.ntaps;
        post = row.scale;
        xc = x, hc = nullptr;
        tapsc = taps + row.taps_off;
        yrow = y + row.out_off * s.y_stride;
    } else {
        tile_start = (long long)tile * kTile;
        pos0 = tile_start;
        ntaps = s.ntaps;
        xc = x + (unsigned long long)chan * s.x_stride;
        hc = hist ? hist + (unsigned long long)chan * s.hist_stride : nullptr;
        tapsc = taps;
        yrow = y + (unsigned long long)chan * s.y_stride;
    }
    const int stage_pair = tid + (tid >> 2);  // pair_of(2*tid): where this thread's staged pairs go

    double acc[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) acc[r] = 0.0;

    for (int j0 = 0; j0 < ntaps; j0 += kStageTaps) {
        const int S = min(kStageTaps, ntaps - j0);
        const int ngroups = S / kGroup;
        const int rem = S % kGroup;

        // ---- stage the window: slot q <- sample pos0 - j0 - kHalo + q, for q >= qlo ----
        const int qlo = kHalo - kGroup * (ngroups + (rem ? 1 : 0));  // lowest slot the window loads touch
        const int npairs = (kSlots - qlo) / 2;
        const long long g_lo = pos0 - j0 - kHalo + qlo;              // sample held by slot qlo
        if (j0 > 0) __syncthreads();                                  // readers of the previous stage are done
        double2* stage = lds + pair_of(qlo) + stage_pair;
        if (g_lo >= 0 && g_lo + (kSlots - qlo) <= n) {  // interior tile (workgroup-uniform): no edge tests
            const double* __restrict__ src = xc + g_lo;
            double2 v[kStageIters];
#pragma unroll
            for (int i = 0; i < kStageIters; ++i) {
                const int pi = tid + i * kBlock;
                if (i < kStageIters - 1 || pi < npairs) {
                    if constexpr (VEC2)
                        v[i] = reinterpret_cast<const double2*>(src)[pi];
                    else
                        v[i] = make_double2(src[2 * pi], src[2 * pi + 1]);
                }
            }
#pragma unroll
            for (int i = 0; i < kStageIters; ++i) {
                if (i < kStageIters - 1 || tid + i * kBlock < npairs) stage[i * pair_of(2 * kBlock)] = v[i];
            }
        } else {  // first / last tiles of a channel: every sample is range-checked (kept small, not fast)
#pragma unroll 1
            for (int i = 0; i < kStageIters; ++i) {
                const int pi = tid + i * kBlock;
                if (pi < npairs)
                    stage[i * pair_of(2 * kBlock)] = make_double2(sample_at(xc, hc, g_lo + 2 * pi, n, ntaps),
                                                                  sample_at(xc, hc, g_lo + 2 * pi + 1, n, ntaps));
            }
        }
        __syncthreads();

        // ---- accumulate the stage's taps, 16 per group ----
        const double* __restrict__ tp = tapsc + j0;
        const double2* top = lds + pair_of(kHalo + 6) + 5 * tid;
        for (int g = 0; g < ngroups; ++g) {
            double W[kWin], bt[kGroup];
            load_window(top, g, W);
#pragma unroll
            for (int t = 0; t < kGroup; ++t) bt[t] = tp[g * kGroup + t];
#pragma unroll
            for (int t = 0; t < kGroup; ++t) {
#pragma unroll
                for (int r = 0; r < kR; ++r) acc[r] = mac<EXACT>(bt[t], W[t - r + 7], acc[r]);
            }
        }
        if (rem) {  // last, partial group: taps predicated (wave-uniform branches)
            double W[kWin], bt[kGroup - 1];
            load_window(top, ngroups, W);
#pragma unroll
            for (int t = 0; t < kGroup - 1; ++t) bt[t] = tp[min(ngroups * kGroup + t, S - 1)];
            // keep all scalar tap loads here, ahead of the branches (one wait instead of one per tap)
#pragma unroll
            for (int t = 0; t < kGroup - 1; ++t) asm volatile("" : "+s"(bt[t]));
#pragma unroll
            for (int t = 0; t < kGroup - 1; ++t) {
                if (t < rem) {
#pragma unroll
                    for (int r = 0; r < kR; ++r) acc[r] = mac<EXACT>(bt[t], W[t - r + 7], acc[r]);
                }
            }
        }
    }

    // ---- store the tile: accumulators -> LDS (stride 5 pairs) -> coalesced 16-byte stores ----
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kR; r += 2) lds[5 * tid + r / 2] = make_double2(acc[r], acc[r + 1]);
    __syncthreads();
    double* __restrict__ yt = yrow + tile_start;
    const double2* back = lds + stage_pair;
    const long long left = n - tile_start;  // outputs of this row from tile_start on
    if constexpr (BANK) {
#pragma unroll
        for (int i = 0; i < kTile / 2 / kBlock; ++i) {
            const int pi = tid + i * kBlock, pn = pi + 1;
            const double2 c01 = back[i * pair_of(2 * kBlock)];
            const double2 c23 = lds[pn + (pn >> 2)];  // pair_of(2 * pn)
            // first difference times the row's factor, as two separately rounded operations
            // (reference dsp.hpp:1186-1191)
            const double2 o = odd ? make_double2((c23.x - c01.y) * post, (c23.y - c23.x) * post)
                                  : make_double2((c01.y - c01.x) * post, (c23.x - c01.y) * post);
            if (2 * pi < kBankStride) {
                if (2 * pi + 1 < left) {
                    if constexpr (VEC2)
                        reinterpret_cast<double2*>(yt)[pi] = o;
                    else
                        yt[2 * pi] = o.x, yt[2 * pi + 1] = o.y;
                } else if (2 * pi < left) {
                    yt[2 * pi] = o.x;
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < kTile / 2 / kBlock; ++i) {
            const int pi = tid + i * kBlock;
            const double2 o = back[i * pair_of(2 * kBlock)];
            if (left >= kTile) {  // full tile (workgroup-uniform)
                if constexpr (VEC2)
                    reinterpret_cast<double2*>(yt)[pi] = o;
                else
                    yt[2 * pi] = o.x, yt[2 * pi + 1] = o.y;
            } else if (2 * pi + 1 < left) {
                if constexpr (VEC2)
                    reinterpret_cast<double2*>(yt)[pi] = o;
                else
                    yt[2 * pi] = o.x, yt[2 * pi + 1] = o.y;
            } else if (2 * pi < left) {
                yt[2 * pi] = o.x;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Toeplitz-block variant on the FP64 matrix pipe (v_mfma_f64_16x16x4_f64).
//
// One wave computes 16 segments (matrix columns n) of 16*OB consecutive outputs each, as OB
// 16-row blocks per segment (OB = 4 or 2).  For a tap offset d (a multiple of 4) the product
//     D[m][n] += sum_k A[m][k] * B[k][n],   A[m][k] = taps[d + m - k],   B[k][n] = x[seg(n) + i0 - d + k]
// adds, to output i0 + m of every segment, the four taps d+m-3 .. d+m.  A is the same 16x4
// Toeplitz block for every segment and every i0 (one LDS read per step from a zero-padded copy
// of the stage's taps); B is four consecutive input samples of each segment (one LDS read per
// product, or per step where the operand ring of the four-block tiling reuses it across row blocks).  A filter of N taps takes floor((N+2)/4) + 4 steps: the band is N + 15 columns
// wide instead of N, so 6 % of the multiplies are by zero at 255 taps and 0.4 % at 4095.
//
// Why it exists: on random data the chip holds ~2.39 GHz while issuing these products but only
// ~1.9 GHz while issuing v_fma_f64 back to back, so the matrix pipe delivers more FP64 work
// per second even after the zero multiplies (measured: tools/fp64_peak.hip).
//
// Differences from the vector kernel: within a step the four taps of an output are added in
// descending order (steps ascend), so rounding differs slightly (same error bound).  The zero
// taps at the band edges multiply real samples, and 0 * Inf is NaN; so a workgroup whose
// staged window holds a non-finite sample computes its tile with a plain per-output FMA loop
// instead, which adds exactly the products the reference adds.
constexpr int kMHalo = 256;                     // staged samples in front of the tile
constexpr int kMStageSteps = (kMHalo + 16) / 4; // steps of four tap offsets one staged window serves (68)
constexpr int kMTapSlots = 4 * kMStageSteps + 24;
// Geometry for OB row blocks of 16 outputs per segment (OB = 4: 4096-output tiles, fewest LDS
// reads per product; OB = 2: 2048-output tiles, twice the workgroups per CU for short filters).
template <int OB> struct MGeo {
    static_assert(OB == 2 || OB == 4, "row blocks per segment");
    static constexpr int kSeg = 16 * OB;                    // outputs per segment = pad group of the image
    static constexpr int kPadStep = kSeg + 2;               // doubles between segments in the image
    static constexpr int kShift = OB == 4 ? 6 : 5;          // log2(kSeg)
    static constexpr int kWaveOut = 16 * kSeg;              // outputs per wave
    static constexpr int kTile = (kBlock / 64) * kWaveOut;  // outputs per workgroup
    static constexpr int kBankStride = kTile - 2;           // bank mode: two spare values of c per tile
    static constexpr int kSlots = kMHalo + kTile;
    static constexpr int kPairs = (kSlots + 2 * (kSlots / kSeg)) / 2;  // image: 2 pad doubles per segment length
    static constexpr int kStageIters = (kSlots / 2 + kBlock - 1) / kBlock;
};
typedef double mfma_d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool not_finite(double v) {
    return (__double_as_longlong(v) & 0x7ff0000000000000ll) == 0x7ff0000000000000ll;
}

template <int OB, bool VEC2, bool BANK>
__global__ __launch_bounds__(kBlock) void fir_f64_mfma_kernel(const double* __restrict__ taps,
                                                               const double* __restrict__ x,
                                                               const double* __restrict__ hist,
                                                               double* __restrict__ y,
                                                               const BankRow* __restrict__ bank, FirShape s) {
    using G = MGeo<OB>;
    __shared__ double2 img[G::kPairs];
    __shared__ double tapz[kMTapSlots];
    __shared__ int bad_flag;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    // tile assignment: as in fir_f64_kernel
    const unsigned b = blockIdx.x;
    const unsigned grp = b % 8, k8 = b / 8;
    unsigned chan, tile;
    if constexpr (BANK) {
        const unsigned run = (s.tiles_per_chan + 7) / 8;
        chan = (unsigned)(((unsigned long long)k8 * s.div_magic) >> s.div_shift);
        tile = grp * run + (k8 - chan * run);
        if (tile >= s.tiles_per_chan) return;
    } else {
        const unsigned q8 = s.total_tiles / 8, r8 = s.total_tiles % 8;
        const unsigned tile_id = (grp < r8 ? grp * (q8 + 1) : r8 * (q8 + 1) + (grp - r8) * q8) + k8;
        chan = (unsigned)(((unsigned long long)tile_id * s.div_magic) >> s.div_shift);
        tile = tile_id - chan * s.tiles_per_chan;
    }
    const long long n = s.n;
    long long tile_start, pos0;
    int ntaps, odd = 0;
    double post = 1.0;
    const double* __restrict__ xc;
    const double* __restrict__ hc;
    const double* __restrict__ tapsc;
    double* __restrict__ yrow;
    if constexpr (BANK) {
        const BankRow row = bank[chan];
        tile_start = (long long)tile * G::kBankStride;
        odd = row.advance & 1;
        pos0 = tile_start + row.advance - odd;
        ntaps = row.ntaps;
        post = row.scale;
        xc = x, hc = nullptr;
        tapsc = taps + row.taps_off;
        yrow = y + row.out_off * s.y_stride;
    } else {
        tile_start = (long long)tile * G::kTile;
        pos0 = tile_start;
        ntaps = s.ntaps;
        xc = x + (unsigned long long)chan * s.x_stride;
        hc = hist ? hist + (unsigned long long)chan * s.hist_stride : nullptr;
        tapsc = taps;
        yrow = y + (unsigned long long)chan * s.y_stride;
    }

    mfma_d4 acc[OB];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) acc[ob] = mfma_d4{0.0, 0.0, 0.0, 0.0};
    if (tid == 0) bad_flag = 0;

    const double* imgd = reinterpret_cast<const double*>(img);
    // lane (k = lane >> 4, n = lane & 15): image address of sample seg(n) + k, and tap slot of m - k at d = -12
    const int b_base = (kMHalo + wave * G::kWaveOut) / G::kSeg * G::kPadStep + G::kPadStep * (lane & 15) + (lane >> 4);
    const int a_base = 3 + (lane & 15) - (lane >> 4);
    bool bad = false;

    // Steps of four tap offsets (d = -12, -8, ...) that cover the whole band.  A stage is up to 68
    // consecutive steps (what the staged halo reaches), an even number of them except in the last stage:
    // it re-stages the samples those offsets reach and the taps d-3 .. d+15 they multiply, so stage
    // boundaries cost no extra zero columns.
    const int total_steps = ntaps ? (ntaps + 2) / 4 + 4 : 0;
    const int nstages = (total_steps + kMStageSteps - 1) / kMStageSteps;
    const int per_stage = nstages ? (((total_steps + nstages - 1) / nstages + 1) & ~1) : 0;  // even split, even count
    for (int e0 = 0; e0 < total_steps; e0 += per_stage) {
        const int steps = min(per_stage, total_steps - e0);
        const int j0 = 4 * e0;  // the stage's first offset is d = j0 - 12
        __syncthreads();  // previous stage fully consumed (and bad_flag initialised)
        // tapz[i] = tap j0 - 15 + i (zero outside the filter)
        for (int i = tid; i < kMTapSlots; i += kBlock) {
            const int j = j0 - 15 + i;
            tapz[i] = (j >= 0 && j < ntaps) ? tapsc[j] : 0.0;
        }
        // samples: slot q <- sample pos0 - j0 - kMHalo + q, from the lowest pad group the products read
        const int qlo = (kMHalo - (4 * steps - 16)) & ~(G::kSeg - 1);
        const int npairs = (G::kSlots - qlo) / 2;
        const long long g_lo = pos0 - j0 - kMHalo + qlo;
        double2* stage = img + (qlo >> 1) + (qlo >> G::kShift);
        bool mine_bad = false;
        if (g_lo >= 0 && g_lo + (G::kSlots - qlo) <= n) {  // interior tile: no edge tests
            const double* __restrict__ src = xc + g_lo;
#pragma unroll
            for (int i = 0; i < G::kStageIters; ++i) {
                const int pi = tid + i * kBlock;
                if (i < G::kStageIters - 1 || pi < npairs) {
                    double2 v;
                    if constexpr (VEC2)
                        v = reinterpret_cast<const double2*>(src)[pi];
                    else
                        v = make_double2(src[2 * pi], src[2 * pi + 1]);
                    mine_bad = mine_bad || not_finite(v.x) || not_finite(v.y);
                    stage[pi + (pi >> (G::kShift - 1))] = v;
                }
            }
        } else {
#pragma unroll 1
            for (int pi = tid; pi < npairs; pi += kBlock) {
                const double2 v = make_double2(sample_at(xc, hc, g_lo + 2 * pi, n, ntaps),
                                               sample_at(xc, hc, g_lo + 2 * pi + 1, n, ntaps));
                mine_bad = mine_bad || not_finite(v.x) || not_finite(v.y);
                stage[pi + (pi >> (G::kShift - 1))] = v;
            }
        }
        if (mine_bad) bad_flag = 1;
        __syncthreads();
        bad = bad_flag != 0;  // workgroup-uniform
        if (bad) break;

        // A sample vector read for row block 0 at step e is the operand of row block 1 at step e + 4, of block 2
        // at e + 8, ...: iterations of 4*(OB-1) steps keep those in registers, so a step reads one new sample
        // vector and one tap vector from LDS instead of OB + 1 (less LDS traffic: the chip is power-bound here,
        // and the saving shows up as clock).
        // Worth it with four row blocks (+1..3 %); with two it saves one read in three and costs ten registers
        // (-0.5 %), so that tiling reads every operand from LDS in the loop further down.
        constexpr int kRing = 4 * (OB - 1);
        constexpr bool kUseRing = OB == 4;
        auto sample_vec = [&](int off) { return imgd[b_base + off + 2 * (off >> G::kShift)]; };
        double ring[kRing];  // ring[j] = the block-0 operand of step e - kRing + j
        if constexpr (kUseRing) {
#pragma unroll
            for (int j = 0; j < kRing; ++j) ring[j] = sample_vec(12 + 4 * (kRing - j));
        }
        int e_first = 0;
        for (; kUseRing && e_first + kRing <= steps; e_first += kRing) {
            double f[kRing], a[kRing];
#pragma unroll
            for (int u = 0; u < kRing; ++u) {
                f[u] = sample_vec(12 - 4 * (e_first + u));
                a[u] = tapz[a_base + 4 * (e_first + u)];
            }
#pragma unroll
            for (int u = 0; u < kRing; ++u) {
#pragma unroll
                for (int ob = 0; ob < OB; ++ob) {
                    const double bv = u >= 4 * ob ? f[u - 4 * ob] : ring[kRing - 4 * ob + u];
                    acc[ob] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], bv, acc[ob], 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < kRing; ++j) ring[j] = f[j];
        }
        // the remaining steps (all of them without the ring), two per iteration with every operand read from LDS
        for (int e = e_first; e + 1 < steps; e += 2) {
            const double a0 = tapz[a_base + 4 * e];
            const double a1 = tapz[a_base + 4 * e + 4];
            double b0[OB], b1[OB];
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) {
                const int off0 = 16 * ob + 12 - 4 * e;   // i0 - d, a multiple of 4, may be negative
                const int off1 = off0 - 4;
                b0[ob] = imgd[b_base + off0 + 2 * (off0 >> G::kShift)];
                b1[ob] = imgd[b_base + off1 + 2 * (off1 >> G::kShift)];
            }
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
                acc[ob] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0[ob], acc[ob], 0, 0, 0);
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
                acc[ob] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1[ob], acc[ob], 0, 0, 0);
        }
        if ((steps - e_first) & 1) {  // the band's last step
            const int e = steps - 1;
            const double a0 = tapz[a_base + 4 * e];
            double b0[OB];
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) {
                const int off0 = 16 * ob + 12 - 4 * e;
                b0[ob] = imgd[b_base + off0 + 2 * (off0 >> G::kShift)];
            }
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
                acc[ob] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0[ob], acc[ob], 0, 0, 0);
        }
    }

    double* __restrict__ yt = yrow + tile_start;
    const long long left = n - tile_start;
    constexpr int kOutputs = BANK ? G::kBankStride : G::kTile;

    if (bad) {
        // A non-finite sample is in reach of this tile: add exactly the products the reference adds,
        // one output at a time (every sample comes from memory again; rare and slow by design).
        const int extra = BANK ? 1 + odd : 0;  // bank mode needs c one or two samples past the tile
        __syncthreads();
        double* cbuf = reinterpret_cast<double*>(img);
        for (int o = tid; o < kOutputs + extra + 1; o += kBlock) {
            double sum = 0.0;
            for (int j = 0; j < ntaps; ++j) {
                const long long g = pos0 + o - j;
                if (g >= 0 && g < n)
                    sum = __builtin_fma(tapsc[j], xc[g], sum);
                else if (g < 0 && hc != nullptr && g >= -(long long)(ntaps - 1))
                    sum = __builtin_fma(tapsc[j], hc[(ntaps - 1) + g], sum);
            }
            cbuf[o] = sum;
        }
        __syncthreads();
        for (int o = tid; o < kOutputs; o += kBlock) {
            if (o < left) yt[o] = BANK ? (cbuf[o + odd + 1] - cbuf[o + odd]) * post : cbuf[o];
        }
        return;
    }

    // ---- accumulators -> LDS -> coalesced stores.  acc[ob][r] is output 16*ob + (lane >> 4) + 4r of segment lane & 15 ----
    __syncthreads();
    double* outd = reinterpret_cast<double*>(img);
    const int o_base = wave * (G::kWaveOut / G::kSeg * G::kPadStep) + G::kPadStep * (lane & 15) + (lane >> 4);
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) {
#pragma unroll
        for (int r = 0; r < 4; ++r) outd[o_base + 16 * ob + 4 * r] = acc[ob][r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < G::kTile / 2 / kBlock; ++i) {
        const int pi = tid + i * kBlock;
        double2 o = img[pi + (pi >> (G::kShift - 1))];
        if constexpr (BANK) {
            // (c[i+1] - c[i]) * factor, shifted by one when the advance was rounded down
            const int pn = min(pi + 1, G::kTile / 2 - 1);
            const double2 c23 = img[pn + (pn >> (G::kShift - 1))];
            o = odd ? make_double2((c23.x - o.y) * post, (c23.y - c23.x) * post)
                    : make_double2((o.y - o.x) * post, (c23.x - o.y) * post);
            if (2 * pi >= G::kBankStride) continue;
        }
        if (2 * pi + 1 < left) {
            if constexpr (VEC2)
                reinterpret_cast<double2*>(yt)[pi] = o;
            else
                yt[2 * pi] = o.x, yt[2 * pi + 1] = o.y;
        } else if (2 * pi < left) {
            yt[2 * pi] = o.x;
        }
    }
}

// SplitMix64 finaliser.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
constexpr unsigned long long kGamma = 0x9E3779B97F4A7C15ull;

__global__ __launch_bounds__(256) void fill_splitmix_kernel(double* __restrict__ x, unsigned long long n,
                                                            unsigned long long seed, unsigned long long offset) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long bits = mix64(seed + (offset + i + 1) * kGamma) >> 11;
        x[i] = (double)bits * 0x1.0p-52 - 1.0;
    }
}

__global__ __launch_bounds__(256) void checksum_kernel(const double* __restrict__ yv, unsigned long long n,
                                                       unsigned long long offset, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long part[4];
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        acc += mix64((unsigned long long)__double_as_longlong(yv[i]) + (offset + i + 1) * kGamma);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace

int fir_tile() { return kTile; }
int fir_mfma_tile(int row_blocks) { return row_blocks == 2 ? MGeo<2>::kTile : MGeo<4>::kTile; }
int fir_mfma_bank_stride(int row_blocks) { return row_blocks == 2 ? MGeo<2>::kBankStride : MGeo<4>::kBankStride; }

template <int OB, bool BANK>
static void launch_mfma_variant(const double* taps, const double* x, const double* hist, double* y, const BankRow* bank,
                                const FirShape& shape, bool vec2, hipStream_t stream) {
    const dim3 grid(shape.total_tiles), block(kBlock);
    if (vec2)
        hipLaunchKernelGGL((fir_f64_mfma_kernel<OB, true, BANK>), grid, block, 0, stream, taps, x, hist, y, bank, shape);
    else
        hipLaunchKernelGGL((fir_f64_mfma_kernel<OB, false, BANK>), grid, block, 0, stream, taps, x, hist, y, bank, shape);
}

hipError_t launch_fir_mfma_f64(const double* taps, const double* x, const double* hist, double* y,
                               const FirShape& shape, int row_blocks, bool vec2, hipStream_t stream) {
    if (shape.total_tiles == 0) return hipSuccess;
    if (!grid_fits(shape.total_tiles)) return hipErrorInvalidConfiguration;
    if (row_blocks == 2)
        launch_mfma_variant<2, false>(taps, x, hist, y, nullptr, shape, vec2, stream);
    else
        launch_mfma_variant<4, false>(taps, x, hist, y, nullptr, shape, vec2, stream);
    return hipGetLastError();
}

hipError_t launch_fir_mfma_bank_f64(const double* taps, const BankRow* bank, const double* x, double* y,
                                    const FirShape& shape, int row_blocks, bool vec2, hipStream_t stream) {
    if (shape.total_tiles == 0) return hipSuccess;
    if (!grid_fits(shape.total_tiles)) return hipErrorInvalidConfiguration;
    if (row_blocks == 2)
        launch_mfma_variant<2, true>(taps, x, nullptr, y, bank, shape, vec2, stream);
    else
        launch_mfma_variant<4, true>(taps, x, nullptr, y, bank, shape, vec2, stream);
    return hipGetLastError();
}

int fir_bank_stride() { return kBankStride; }

hipError_t launch_fir_f64(const double* taps, const double* x, const double* hist, double* y, const FirShape& shape,
                          bool exact, bool vec2, hipStream_t stream) {
    if (shape.total_tiles == 0) return hipSuccess;
    if (!grid_fits(shape.total_tiles)) return hipErrorInvalidConfiguration;
    const dim3 grid(shape.total_tiles), block(kBlock);
    const BankRow* none = nullptr;
    if (exact) {
        if (vec2)
            hipLaunchKernelGGL((fir_f64_kernel<true, true, false>), grid, block, 0, stream, taps, x, hist, y, none, shape);
        else
            hipLaunchKernelGGL((fir_f64_kernel<true, false, false>), grid, block, 0, stream, taps, x, hist, y, none, shape);
    } else {
        if (vec2)
            hipLaunchKernelGGL((fir_f64_kernel<false, true, false>), grid, block, 0, stream, taps, x, hist, y, none, shape);
        else
            hipLaunchKernelGGL((fir_f64_kernel<false, false, false>), grid, block, 0, stream, taps, x, hist, y, none, shape);
    }
    return hipGetLastError();
}

hipError_t launch_fir_bank_f64(const double* taps, const BankRow* bank, const double* x, double* y,
                               const FirShape& shape, bool exact, bool vec2, hipStream_t stream) {
    if (shape.total_tiles == 0) return hipSuccess;
    if (!grid_fits(shape.total_tiles)) return hipErrorInvalidConfiguration;
    const dim3 grid(shape.total_tiles), block(kBlock);  // rows * 8 * ceil(tiles_per_chan / 8)
    const double* none = nullptr;
    if (exact) {
        if (vec2)
            hipLaunchKernelGGL((fir_f64_kernel<true, true, true>), grid, block, 0, stream, taps, x, none, y, bank, shape);
        else
            hipLaunchKernelGGL((fir_f64_kernel<true, false, true>), grid, block, 0, stream, taps, x, none, y, bank, shape);
    } else {
        if (vec2)
            hipLaunchKernelGGL((fir_f64_kernel<false, true, true>), grid, block, 0, stream, taps, x, none, y, bank, shape);
        else
            hipLaunchKernelGGL((fir_f64_kernel<false, false, true>), grid, block, 0, stream, taps, x, none, y, bank, shape);
    }
    return hipGetLastError();
}

hipError_t launch_fill_splitmix(double* x, unsigned long long n, unsigned long long seed, unsigned long long offset,
                                hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned long long want = (n + 255) / 256;
    const unsigned grid = (unsigned)(want < 65536ull ? want : 65536ull);
    hipLaunchKernelGGL(fill_splitmix_kernel, dim3(grid), dim3(256), 0, stream, x, n, seed, offset);
    return hipGetLastError();
}

hipError_t launch_checksum(const double* yv, unsigned long long n, unsigned long long offset, unsigned long long* out,
                           hipStream_t stream) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess || n == 0) return e;
    const unsigned long long want = (n + 255) / 256;
    const unsigned grid = (unsigned)(want < 8192ull ? want : 8192ull);
    hipLaunchKernelGGL(checksum_kernel, dim3(grid), dim3(256), 0, stream, yv, n, offset, out);
    return hipGetLastError();
}

}  // namespace dspx
