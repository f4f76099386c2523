// FP64 direct-form FIR kernels for gfx950 (MI355X).
//
// Computes y[i] = sum_j taps[j] * x[i-j] with taps ascending per output, i.e. the
// inner loop of the reference lfilter (dsp.hpp:356-365) for a denominator of size 1,
// with the samples before x[0] taken as zero (or from an explicit history buffer).
// Two kernels: fir_f64_kernel on the vector pipe (described here) and fir_f64_mfma4_kernel on
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

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>

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

typedef double dvec2 __attribute__((ext_vector_type(2)));

// 16-byte accesses to data that is read or written once per launch carry the non-temporal hint (measured on
// MI355X: a 2-tap filter, i.e. a scaled copy, moves 6.23 TB/s with it and 6.00 without).
__device__ __forceinline__ dvec2 stream_load(const dvec2* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stream_store(dvec2* p, dvec2 v) { __builtin_nontemporal_store(v, p); }
// A wave-uniform pointer the compiler must keep in scalar registers: a memory instruction then takes it as its scalar
// base with one shared per-lane offset, instead of a 64-bit per-lane address held in two vector registers.
// The thread index as a value the compiler cannot compute ahead: what a phase derives from it (offsets, range checks)
// is then formed where it is used instead of being carried in registers through the whole tile loop.
__device__ __forceinline__ int fresh(int v) {
    asm volatile("" : "+v"(v));
    return v & (kBlock - 1);  // (tells the compiler the range again)
}
template <typename T> __device__ __forceinline__ auto scalar_base(T* p) {
    typedef T __attribute__((address_space(1))) * global_ptr;  // (a pointer made from an integer would be a flat one)
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (global_ptr)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ dvec2 stream_load(const dvec2 __attribute__((address_space(1))) * p) { return __builtin_nontemporal_load(p); }

__device__ __forceinline__ bool not_finite(double v) {
    return (__double_as_longlong(v) & 0x7ff0000000000000ll) == 0x7ff0000000000000ll;
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
    bool edge = false;  // some window of this tile reached outside the channel (workgroup-uniform)

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
            dvec2 v[kStageIters];
#pragma unroll
            for (int i = 0; i < kStageIters; ++i) {
                const int pi = tid + i * kBlock;
                if (i < kStageIters - 1 || pi < npairs) {
                    if constexpr (VEC2)  // samples that this launch reads once (one stage, one row) are streamed
                        v[i] = (!BANK && ntaps <= kStageTaps) ? stream_load(reinterpret_cast<const dvec2*>(src) + pi)
                                                              : reinterpret_cast<const dvec2*>(src)[pi];
                    else
                        v[i] = dvec2{src[2 * pi], src[2 * pi + 1]};
                }
            }
#pragma unroll
            for (int i = 0; i < kStageIters; ++i) {
                if (i < kStageIters - 1 || tid + i * kBlock < npairs) reinterpret_cast<dvec2*>(stage)[i * pair_of(2 * kBlock)] = v[i];
            }
        } else {  // first / last tiles of a channel: every sample is range-checked (kept small, not fast)
            edge = true;
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

    // The zeros staged for samples outside the channel were multiplied by real taps, and 0 * Inf (or NaN) is NaN
    // where the reference, which never multiplies a missing sample (dsp.hpp:359-364), adds nothing.  Only a
    // non-finite tap does that, and it leaves a non-finite accumulator: such outputs are summed again over the
    // samples that exist, in the same order.
    if (edge) {
        bool redo = false;
#pragma unroll
        for (int r = 0; r < kR; ++r) redo = redo || not_finite(acc[r]);
        if (redo) {
#pragma unroll 1
            for (int r = 0; r < kR; ++r) {
                const long long p = pos0 + kR * tid + r;
                double sum = 0.0;
                for (int j = 0; j < ntaps; ++j) {
                    const long long g = p - j;
                    if (g >= 0 && g < n)
                        sum = mac<EXACT>(tapsc[j], xc[g], sum);
                    else if (g < 0 && hc != nullptr && g >= -(long long)(ntaps - 1))
                        sum = mac<EXACT>(tapsc[j], hc[(ntaps - 1) + g], sum);
                }
                acc[r] = sum;
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
                    stream_store(reinterpret_cast<dvec2*>(yt) + pi, dvec2{o.x, o.y});
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
// Toeplitz-block variant on the FP64 matrix pipe, four-block form of the instruction (v_mfma_f64_4x4x4_4b_f64).
//
// Why it exists: on random data the chip holds its clock while issuing these products but drops to ~1.9 GHz while
// issuing v_fma_f64 back to back, so the matrix pipe delivers more FP64 work per second (tools/fp64_peak.hip).
//
// One instruction multiplies four independent 4x4 blocks.  Operand layout (measured on gfx950 with one-hot
// operands): A[i][k] of block b sits in lane i + 4b + 16k, B[k][j] in lane j + 4b + 16k, D[i][j] in lane
// j + 4b + 16i, one double per lane each.  Here the 16 (block, j) pairs are the 16 segments of a wave
// (n = lane & 15), i is one of four consecutive outputs and k one of four consecutive samples:
//     D[i][n] += sum_k A[i][k] * B[k][n],   A[i][k] = taps[d + i - k],   B[k][n] = x[seg(n) + i0 - d + k]
// adds the taps d+i-3 .. d+i to output i0 + i of every segment.  A is the same for all blocks and all i0 (one LDS
// read per step of four tap offsets); the B operand of outputs i0 at offset d is the B operand of outputs i0 + 4
// at offset d + 4, so a wave that owns RB row blocks of four outputs per segment reads ONE new sample vector per
// step and keeps the RB - 1 before it in registers.  A filter of N taps takes floor((N + 2) / 4) + 1 steps: the
// band is N + 3 columns wide (the 16x16x4 form of the instruction needs N + 15), and this form issues at its full
// rate (one per 16 cycles and SIMD on random data, 76 TFLOP/s on the chip, against 66 for the 16x16x4 form).
//
// Differences from the vector kernel: within a step the four taps of an output are added in descending order
// (steps ascend), so rounding differs slightly (same error bound), and which four taps share a step depends on an
// output's distance from the start of its call modulo 4.
//
// Workgroups are persistent: the grid holds a few workgroups per CU and each walks its share of the tiles.  While the
// products of one window run, the global loads of the next window (the next stage of a long filter, or the next
// tile) are already in flight into registers; they are written to LDS when the products are done.  Results pass
// through a second LDS buffer into registers and are stored one pass later, so that nothing ever waits for a store
// to be acknowledged.  Per window a workgroup meets at two barriers.
template <int OB> struct M4Geo {
    static constexpr int kSeg = 16 * OB;                    // outputs per segment (4 * OB row blocks of four) = pad group
    static constexpr int kPadStep = kSeg + 2;               // doubles between segments in the image
    static constexpr int kShift = OB == 4 ? 6 : 5;          // log2(kSeg)
    static constexpr int kWaveOut = 16 * kSeg;              // outputs per wave
    static constexpr int kTile = (kBlock / 64) * kWaveOut;  // outputs per workgroup and tile
    static constexpr int kBankStride = kTile - 2;           // bank mode: two spare values of c per tile
    static constexpr int kHalo = OB == 4 ? 320 : 288;       // image slots in front of the tile (a multiple of kSeg; sized so
                                                            // that 2 resp. 4 workgroups fit the 160 KB of a CU)
    static constexpr int kStageSteps = kHalo / 4 - 1;       // steps one window serves (79 / 71): the loop reads one step ahead
    static constexpr int kTapSlots = 4 * kStageSteps + 8;   // likewise
    static constexpr int kTapIters = (kTapSlots + kBlock - 1) / kBlock;
    static constexpr int kSlots = kHalo + kTile;
    static constexpr int kPairs = (kSlots + 2 * (kSlots / kSeg)) / 2;
    static constexpr int kOutStep = kSeg + 1;               // doubles between segments in the output buffer
    static constexpr int kOutDoubles = (kTile / kSeg * kOutStep + 3) & ~1;  // (two more: bank mode reads c[q + 2])
    static constexpr int kStageIters = (kSlots / 2 + kBlock - 1) / kBlock;
    static constexpr int kResident = OB == 4 ? 2 : 4;       // workgroups per CU the grid is sized for
    static_assert(kHalo % kSeg == 0, "the image's pad groups start at the window's first slot");
    static_assert(kResident * (kPairs * 16 + kOutDoubles * 8 + kTapSlots * 8 + 16) <= 160 * 1024, "LDS of the resident workgroups");
};

// What a workgroup needs to know about one tile (all wave-uniform).
struct M4Tile {
    const double* xc;
    const double* hc;
    const double* tapsc;
    double* yrow;
    long long tile_start, pos0;
    double post;
    int ntaps, odd;
};

// SINGLE: lfilter mode with a filter that one window serves (the common case, up to 4 * kStageSteps - 3 taps): every
// tile is one pass of the loop and the tap table stays in LDS for the whole launch, which the compiler is told at
// compile time so that the per-tile code is a straight line.
template <int OB, bool VEC2, bool BANK, bool SINGLE>
__global__ __launch_bounds__(kBlock, OB == 4 ? 2 : 4) void fir_f64_mfma4_kernel(const double* __restrict__ taps,
                                                                                 const double* __restrict__ x,
                                                                                 const double* __restrict__ hist,
                                                                                 double* __restrict__ y,
                                                                                 const BankRow* __restrict__ bank,
                                                                                 FirShape s) {
    using G = M4Geo<OB>;
    constexpr int RB = 4 * OB;  // row blocks of four outputs per segment
    // One LDS block, the tap table first: LDS instructions reach 64 KB from their address register, and the product
    // loop reads the table at constant distances from one pointer.
    constexpr int kTapDoubles = (G::kTapSlots + 1) & ~1;
    __shared__ double2 lds[kTapDoubles / 2 + G::kPairs + G::kOutDoubles / 2];
    double* const tapz = reinterpret_cast<double*>(lds);
    double2* const img = lds + kTapDoubles / 2;
    double* const outd = reinterpret_cast<double*>(img + G::kPairs);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const long long n = s.n;
    // Sign of a 64-bit wave-uniform value by its high word: the scalar unit has no 64-bit ordered compare, and a
    // compare on the vector unit costs this wave far more than its one instruction (see below).
    auto negative = [](long long v) __attribute__((always_inline)) {
        int hi = (int)((unsigned long long)v >> 32);
        asm("" : "+s"(hi));  // (in a scalar register, and not to be read as v < 0 again)
        return hi < 0;
    };

    // Blocks b and b + 8 share an XCD (and its L2): XCD g owns a contiguous run of tile indices, and its workgroups
    // walk that run side by side (neighbouring tiles share a halo, which the XCD's L2 then serves).
    const unsigned grp = blockIdx.x % 8, k8 = blockIdx.x / 8, per_xcd = gridDim.x / 8;
    unsigned first, count;  // this XCD's tile indices: first .. first + count - 1 (bank mode: positions k of its runs)
    if constexpr (BANK) {
        first = 0, count = s.total_tiles / 8;  // rows * run positions; those past a row's last tile are skipped
    } else {
        const unsigned q8 = s.total_tiles / 8, r8 = s.total_tiles % 8;
        first = grp < r8 ? grp * (q8 + 1) : r8 * (q8 + 1) + (grp - r8) * q8;
        count = q8 + (grp < r8 ? 1u : 0u);
    }
    // Fills `t` for position k of this XCD; false when there is no tile there.
    auto locate = [&](unsigned k, M4Tile& t) __attribute__((always_inline)) -> bool {
        if (k >= count) return false;
        if constexpr (BANK) {
            const unsigned run = (s.tiles_per_chan + 7) / 8;
            const unsigned chan = (unsigned)(((unsigned long long)k * s.div_magic) >> s.div_shift);
            const unsigned tile = grp * run + (k - chan * run);
            if (tile >= s.tiles_per_chan) return false;
            const BankRow row = bank[chan];
            t.tile_start = (long long)tile * G::kBankStride;
            t.odd = row.advance & 1;
            t.pos0 = t.tile_start + row.advance - t.odd;
            t.ntaps = row.ntaps;
            t.post = row.scale;
            t.xc = x, t.hc = nullptr;
            t.tapsc = taps + row.taps_off;
            t.yrow = y + row.out_off * s.y_stride;
        } else {
            const unsigned tile_id = first + k;
            const unsigned chan = (unsigned)(((unsigned long long)tile_id * s.div_magic) >> s.div_shift);
            const unsigned tile = tile_id - chan * s.tiles_per_chan;
            t.tile_start = (long long)tile * G::kTile;
            t.pos0 = t.tile_start;
            t.ntaps = s.ntaps;
            t.odd = 0, t.post = 1.0;
            t.xc = x + (unsigned long long)chan * s.x_stride;
            t.hc = hist ? hist + (unsigned long long)chan * s.hist_stride : nullptr;
            t.tapsc = taps;
            t.yrow = y + (unsigned long long)chan * s.y_stride;
        }
        return true;
    };
    // Steps of four tap offsets d = 0, 4, ... cover the band; a window serves up to kStageSteps of them, longer
    // filters take several windows (stages), split evenly.
    auto stage_steps = [](int ntaps, int& total, int& per) __attribute__((always_inline)) {
        total = ntaps ? (ntaps + 2) / 4 + 1 : 0;
        const int nstages = (total + G::kStageSteps - 1) / G::kStageSteps;
        per = nstages > 1 ? (total + nstages - 1) / nstages : max(total, 1);
    };
    // In lfilter mode a filter that fits one window keeps its tap table in LDS for the whole launch.
    static_assert(!(BANK && SINGLE), "a bank's rows differ in length");
    int total_steps, per_stage;
    stage_steps(BANK ? 0 : s.ntaps, total_steps, per_stage);
    constexpr bool taps_stay = SINGLE;

    // A wave's ordinary vector instructions are issued very slowly while the other waves of its SIMD issue matrix
    // instructions back to back (measured with tools/issue_probe.hip: a dependent v_xor takes ~50 times as long next
    // to three such waves; scalar instructions are not slowed).  Everything a tile does outside the product loop is
    // therefore written to need as few vector-unit instructions as possible: per-lane addresses are formed once, below,
    // and kept; memory instructions take a scalar base plus one of those offsets; LDS instructions take constant
    // distances from them; range checks that do not hold for a whole tile are in branches of their own, whose per-lane
    // arithmetic starts from fresh(tid) so that none of it is computed ahead and carried through the loop.

    // ---- the window in flight: samples (and taps) of (tile at position fk, first step fe0) on their way into
    // registers.  Image slot q <- sample pos0 - 4*e0 - kHalo + q, from the lowest pad group the products read; tap
    // slot i <- tap 4*e0 - 3 + i (zero outside the filter).  A window that reaches outside [0, n) is not fetched
    // ahead: commit() loads it with range checks (first and last windows of a channel; kept small, not fast).
    dvec2 R[G::kStageIters];
    double T[G::kTapIters];
    unsigned fk = 0;
    int fe0 = 0, fsteps = 1;
    bool finterior = false;
    auto window_lo = [](int steps) __attribute__((always_inline)) { return (G::kHalo - 4 * (steps - 1)) & ~(G::kSeg - 1); };
    auto fetch_taps = [&](const M4Tile& t, int e0) __attribute__((always_inline)) {
        const int ft = fresh(tid);
#pragma unroll
        for (int r = 0; r < G::kTapIters; ++r) {
            const int j = 4 * e0 - 3 + ft + r * kBlock;
            T[r] = (j >= 0 && j < t.ntaps) ? t.tapsc[j] : 0.0;
        }
    };
    auto fetch = [&](const M4Tile& t, unsigned kpos, int e0, int steps) __attribute__((always_inline)) {
        fk = kpos, fe0 = e0, fsteps = steps;
        if (!taps_stay) fetch_taps(t, e0);
        const int qlo = window_lo(steps);
        const long long g_lo = t.pos0 - 4 * e0 - G::kHalo + qlo;
        finterior = !negative(g_lo) && !negative(n - (g_lo + (G::kSlots - qlo)));
        if (!finterior) return;
        const int npairs = (G::kSlots - qlo) / 2;
        const double* src = t.xc + g_lo;  // wave-uniform
#pragma unroll
        for (int i = 0; i < G::kStageIters; ++i) {
            if (i < G::kStageIters - 1 || tid + i * kBlock < npairs) {
                if constexpr (VEC2) {  // a long filter reads its samples once per window: those loads stay cacheable
                    const auto p = scalar_base(reinterpret_cast<const dvec2*>(src) + i * kBlock);
                    R[i] = taps_stay ? stream_load(p + (unsigned)tid) : p[(unsigned)tid];
                } else {
                    const double* p = src + 2 * i * kBlock;
                    R[i] = dvec2{p[2u * (unsigned)tid], p[2u * (unsigned)tid + 1u]};
                }
            }
        }
    };
    const int stage_lane = tid + (tid >> (G::kShift - 1));  // where a thread's first pair goes, relative to the window's start
    constexpr int kPairStep = kBlock + (kBlock >> (G::kShift - 1));  // image distance of pairs kBlock apart
    auto put_taps = [&]() __attribute__((always_inline)) {
        const int ft = fresh(tid);
#pragma unroll
        for (int r = 0; r < G::kTapIters; ++r) {
            const int i = ft + r * kBlock;
            if (r < G::kTapIters - 1 || i < G::kTapSlots) tapz[i] = T[r];
        }
    };
    // registers -> LDS for the window in flight
    auto commit = [&]() __attribute__((always_inline)) {
        if (!taps_stay) put_taps();
        const int qlo = window_lo(fsteps);
        const int npairs = (G::kSlots - qlo) / 2;
        if (finterior) {
            dvec2* stage = reinterpret_cast<dvec2*>(img) + (qlo >> 1) + (qlo >> G::kShift) + stage_lane;
#pragma unroll
            for (int i = 0; i < G::kStageIters; ++i)  // pair tid + i*kBlock: a constant distance per i
                if (i < G::kStageIters - 1 || tid + i * kBlock < npairs) stage[i * kPairStep] = R[i];
        } else {
            M4Tile t{};
            (void)locate(fk, t);
            const long long g_lo = t.pos0 - 4 * fe0 - G::kHalo + qlo;
            const int ft = fresh(tid);
            dvec2* stage = reinterpret_cast<dvec2*>(img) + (qlo >> 1) + (qlo >> G::kShift) + ft + (ft >> (G::kShift - 1));
#pragma unroll 1
            for (int pi = ft; pi < npairs; pi += kBlock, stage += kPairStep)
                *stage = dvec2{sample_at(t.xc, t.hc, g_lo + 2 * pi, n, t.ntaps), sample_at(t.xc, t.hc, g_lo + 2 * pi + 1, n, t.ntaps)};
        }
    };

    // lane (k = lane >> 4, n = lane & 15): image address of sample seg(n) + k, and tap slot of i - k at d = 0
    const double* imgd = reinterpret_cast<const double*>(img);
    const double* w_home = imgd + (G::kHalo + wave * G::kWaveOut) / G::kSeg * G::kPadStep + G::kPadStep * (lane & 15) + (lane >> 4);
    const double* a_home = tapz + 3 + (lane & 3) - (lane >> 4);
    // The output buffer holds tile output q at q + q / kSeg (one spare double per segment): acc[rb] is output
    // 4*rb + (lane >> 4) of segment lane & 15, and with an odd distance between segments the 16 lanes of a ds_write_b64
    // group fall into 16 different bank pairs.  It is read back 8 bytes per lane, lanes in output order.
    double* o_home = outd + wave * (G::kWaveOut / G::kSeg * G::kOutStep) + G::kOutStep * (lane & 15) + (lane >> 4);
    constexpr int kStores = G::kTile / kBlock;                                  // outputs per thread: tid + i*kBlock
    constexpr int kOutRowStep = kBlock + kBlock / G::kSeg;                      // buffer distance of outputs kBlock apart
    const double* g_home = outd + tid + (tid >> G::kShift);                     // output tid
    const double* g_home1 = outd + (tid + 1) + ((tid + 1) >> G::kShift);        // bank mode: outputs tid + 1, tid + 2
    const double* g_home2 = outd + (tid + 2) + ((tid + 2) >> G::kShift);

    // What the epilogue needs of a tile; kept from the moment the tile's first window is fetched.
    struct Out {
        double* yt;
        long long left;
        double post;
        int odd;
    };
    auto out_of = [&](const M4Tile& t) __attribute__((always_inline)) { return Out{t.yrow + t.tile_start, n - t.tile_start, t.post, t.odd}; };

    // the first tile of this workgroup (positions k8, k8 + per_xcd, ... of the XCD's run)
    unsigned k = k8;
    Out cur_out{}, nxt_out{};
    {
        M4Tile first_tile{};
        bool have = false;
        for (; k < count && !(have = locate(k, first_tile)); k += per_xcd) {}
        if (!have) return;  // the whole workgroup leaves together
        if constexpr (BANK) stage_steps(first_tile.ntaps, total_steps, per_stage);
        if (taps_stay) {
            fetch_taps(first_tile, 0);
            put_taps();
        }
        cur_out = out_of(first_tile);
        fetch(first_tile, k, 0, min(per_stage, max(total_steps, 1)));
        commit();
    }
    // A finished tile's results wait in O[] and are stored after the next window's loads have been issued: the step that
    // takes a window from registers to LDS waits for every memory instruction issued before it, and what was issued a
    // whole window earlier has long been done, while stores issued just before it would be waited for until memory
    // has acknowledged them.
    constexpr int kOutputs = BANK ? G::kBankStride : G::kTile;
    double O[kStores];
    double* pend_yt = nullptr;
    int pend_left = 0;  // outputs of the tile that exist, at most kOutputs
    bool stores_due = false;
    auto store_tile = [&]() __attribute__((always_inline)) {
        if (pend_left >= kOutputs) {  // all of the tile
#pragma unroll
            for (int i = 0; i < kStores; ++i) {
                const auto p = scalar_base(pend_yt + i * kBlock);
                if (!BANK || i < kStores - 1 || tid < kBlock - (G::kTile - G::kBankStride)) __builtin_nontemporal_store(O[i], p + (unsigned)tid);
            }
        } else {
            const int ft = fresh(tid);
#pragma unroll
            for (int i = 0; i < kStores; ++i)
                if (ft + i * kBlock < pend_left) pend_yt[ft + i * kBlock] = O[i];
        }
    };

    double acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = 0.0;
    int e0 = 0;
    // One pass per window: a tile has one unless its filter is longer than a window reaches.
    for (bool more = true; more;) {
        const int steps = SINGLE ? total_steps : fsteps;  // the window in the image
        // start the loads of the window after this one: the next stage of this tile, or the next tile
        const int e_next = e0 + steps;
        const bool tile_done = SINGLE || e_next >= total_steps;
        unsigned kn = k;
        int next_total = total_steps, next_per = per_stage;
        {
            M4Tile t{};
            bool have = false;
            if (!tile_done) {
                have = locate(k, t);
            } else {
                for (kn = k + per_xcd; kn < count && !(have = locate(kn, t)); kn += per_xcd) {}
                if (have) {
                    if constexpr (BANK) stage_steps(t.ntaps, next_total, next_per);
                    nxt_out = out_of(t);
                }
            }
            more = have;
            if (have) {
                const int ne0 = tile_done ? 0 : e_next;
                fetch(t, kn, ne0, min(next_per, max(next_total, 1) - ne0));
            }
        }
        if (stores_due) store_tile();
        stores_due = false;
        __syncthreads();

        if (total_steps > 0) {
            // V(s) = the four-sample vector at offset -4s from the lane's segment start: w_home[-4s], two doubles
            // lower for every pad group crossed.  Row block rb uses V(el - rb) at step el.  w[] is a circular window
            // over them, V(s) in slot s mod RB: a step runs its row blocks from the last to the first, so the slot the
            // oldest vector leaves is refilled (for the next step's row block 0) right after its last use, a whole
            // step ahead of its first.  Steps run in groups of RB: within a group every refill V(el + u + 1) lies in
            // one pad group, so a group reads at constant distances from one pointer that moves by kPadStep per group.
            double w[RB];
#pragma unroll
            for (int j = 1; j < RB; ++j) w[j] = w_home[4 * (RB - j)];
            w[0] = w_home[0];
            const double* wp = w_home - G::kPadStep;  // V(el + u + 1) = wp[kSeg - 4*(u + 1)] for the group at el
            const double* ap = a_home;                // tap vector of step el + u = ap[4*u]
            double a_now = ap[0];
            int el = 0;
            for (; el + RB <= steps; el += RB, wp -= G::kPadStep, ap += 4 * RB) {
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const double a = a_now;
                    a_now = ap[4 * (u + 1)];  // the table has a spare step past the last
#pragma unroll
                    for (int rb = RB - 1; rb >= 0; --rb) {
                        acc[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, w[(u - rb + RB) % RB], acc[rb], 0, 0, 0);
                        if (rb == RB - 1) w[(u + 1) % RB] = wp[G::kSeg - 4 * (u + 1)];  // at most one step past the window: in the image
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep every step's two LDS reads in their step (a step ahead of their use)
                }
            }
            if (el < steps) {  // the window's last steps, fewer than RB of them (wave-uniform branches)
                const int rem = steps - el;
#pragma unroll
                for (int u = 0; u < RB - 1; ++u) {
                    if (u < rem) {
                        const double a = a_now;
                        a_now = ap[4 * (u + 1)];
#pragma unroll
                        for (int rb = RB - 1; rb >= 0; --rb) {
                            acc[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, w[(u - rb + RB) % RB], acc[rb], 0, 0, 0);
                            if (rb == RB - 1) w[(u + 1) % RB] = wp[G::kSeg - 4 * (u + 1)];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        if (!tile_done) {
            e0 = e_next;
            __syncthreads();  // the image is rewritten for the next window
            commit();         // (a tile's last window is never the launch's last: `more` holds)
            continue;
        }

        // ---- accumulators -> output buffer ----
        // The zero taps at the band's edges multiply real samples and the zero samples in front of a channel multiply
        // real taps; 0 * Inf (or NaN) is NaN where the reference adds nothing.  Any such product leaves a non-finite
        // accumulator, and then so is the sum of the lane's accumulators.  A wave that finds one computes its own
        // 16 * kSeg values again with a plain loop that adds exactly the reference's products, and puts them into
        // the output buffer in place of the accumulators: no other wave needs to know.  (A sum of finite values
        // that overflows only sends a wave there needlessly.)
        double probe = acc[0];
#pragma unroll
        for (int rb = 1; rb < RB; ++rb) probe += acc[rb];
        if (__builtin_amdgcn_ballot_w64(not_finite(probe)) != 0) {
            // one value at a time, every sample from memory again (rare and slow by design)
            M4Tile t{};
            (void)locate(k, t);
            const int ft = fresh(tid);
#pragma unroll 1
            for (int q = (ft >> 6) * G::kWaveOut + (ft & 63); q < ((ft >> 6) + 1) * G::kWaveOut; q += 64) {
                double sum = 0.0;
                for (int j = 0; j < t.ntaps; ++j) {
                    const long long g = t.pos0 + q - j;
                    if (g >= 0 && g < n)
                        sum = __builtin_fma(t.tapsc[j], t.xc[g], sum);
                    else if (g < 0 && t.hc != nullptr && g >= -(long long)(t.ntaps - 1))
                        sum = __builtin_fma(t.tapsc[j], t.hc[(t.ntaps - 1) + g], sum);
                }
                outd[q + (q >> G::kShift)] = sum;
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) o_home[4 * rb] = acc[rb];
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = 0.0;
        __syncthreads();  // the tile's values are in the output buffer, and everybody is past its products

        // the results wait in registers; stored while the next window's products run
        double* __restrict__ yt = cur_out.yt;
        const int left = negative(cur_out.left - kOutputs) ? (int)cur_out.left : kOutputs;
#pragma unroll
        for (int i = 0; i < kStores; ++i) {
            if constexpr (BANK) {
                // (c[q+1] - c[q]) * factor, shifted by one when the advance was rounded down (wave-uniform)
                const double c0 = g_home[i * kOutRowStep], c1 = g_home1[i * kOutRowStep];
                if (cur_out.odd) {
                    const double c2 = (i < kStores - 1 || tid < kBlock - 2) ? g_home2[i * kOutRowStep] : c1;  // (never stored)
                    O[i] = (c2 - c1) * cur_out.post;
                } else {
                    O[i] = (c1 - c0) * cur_out.post;
                }
            } else {
                O[i] = g_home[i * kOutRowStep];
            }
        }
        pend_yt = yt, pend_left = left, stores_due = true;
        if (more) commit();  // everybody is past this tile's products
        k = kn, e0 = 0;
        cur_out = nxt_out;
        total_steps = next_total, per_stage = next_per;
    }
    if (stores_due) store_tile();
}

// ---------------------------------------------------------------------------------------------
// Wave-autonomous form of the matrix-pipe kernel for lfilter mode with a filter that one window serves (up to 569 taps)
// over 16-byte aligned buffers.  Same products, same order, same bits as fir_f64_mfma4_kernel; what differs is who
// shares what:
//   * every WAVE has an LDS image of its own (halo + one chunk of samples: 512 outputs and 288 halo slots in 6.8 KB,
//     or 1024 outputs and 320 / 576 halo slots in 11.1 / 13.2 KB) and walks a run of consecutive chunks of one
//     channel.  Nothing but the read-only tap table is shared between the waves of a workgroup, so there is no
//     workgroup barrier after the table has been written: the LDS instructions of one wave execute in order, which is
//     all the hand-over between a wave's own phases needs;
//   * the halo of chunk k + 1 is the tail of chunk k's image: it is read into registers while the products of chunk k
//     start and written to its new place when they are done, so every sample of a run is fetched from memory exactly
//     once, 64 (128) bytes per lane and chunk (measured: HBM read traffic within 0.1 % of the algorithmic 8 bytes per
//     output, where the workgroup-tiled form re-reads 11 %);
//   * results are transposed through the body's place in the wave's own image (dead once the products are done),
//     read back in output order, 512 contiguous bytes per store instruction, and stored one chunk late, after the
//     next chunk's loads and first LDS reads have been issued;
//   * a change of run costs what a change of chunk costs: a run's last chunk fetches the whole first window of the
//     wave's next run while its products run.  On large inputs that next run is drawn from a device counter of the
//     launch's stream (FirRuns below), on small ones and under stream capture every wave has a fixed share.
// Per chunk a wave issues, besides its matrix instructions, 4 (8) loads, 8 (16) stores, about 30 (45) LDS instructions
// and about two dozen vector instructions, and waits for nobody but itself: between the last matrix instruction of a
// chunk and the first of the next lies one LDS round trip.  The 1024-output chunks halve the LDS reads per matrix
// instruction and the hand-overs per output at the price of two or three resident workgroups per CU instead of four;
// they win from 78 taps on.
// Everything a wave loads ahead of its use (the next chunk's samples, the halo's tail, a next run's first window) is
// loaded by all of its lanes -- a lane past the end of the range takes the range's last element -- and only the stores
// that follow stand under a lane condition: registers that a load defines on some lanes' paths only are copied around
// by the compiler, chunk after chunk, on the vector unit, whose cycles are the matrix instructions' own (measured: 69
// vector instructions per chunk before, 28 after; 128 -> 89 registers per lane for the 512-output chunks).
//
// What bounds it (timestamps inside the kernel, tools/m4w_timing.py; builds with the loads or the stores left out):
// the product loop's marginal cost is exactly its matrix-pipe time, the pipe is occupied 87 % (101 taps) to 95 %
// (255 taps) of the time, and the clock the chip holds falls with the HBM traffic the kernel drives -- 2.0 GHz at
// 101 taps (4.1 TB/s), 2.2 GHz at 255 taps (2.1 TB/s), 2.26 / 2.35 GHz for the same kernels without their loads.
#ifndef M4W_ABLATE
// Development builds only (tools/build_variant.sh): 2 leaves out the result stores, 16 the sample loads of the steady
// state, to time the kernel without that memory traffic.  The results of such a build are wrong.
#define M4W_ABLATE 0
#endif
constexpr unsigned kMaxRunDecks = 8, kTicketStride = 64;   // (64 unsigned = 256 bytes between the counters of two groups)
struct FirRuns {
    unsigned long long div_magic;  // run / runs_per_chan == (run * div_magic) >> div_shift
    unsigned div_shift;
    unsigned runs_per_chan;        // a channel's chunks [0, lead_chunks) are cut into this many runs ...
    unsigned run_chunks;           // ... of this many chunks (the last one may be shorter)
    unsigned chunks_per_chan;      // ceil(n / outputs per chunk)
    unsigned total_runs;           // lead_runs + nchan * tail_runs_per_chan
    // Runs handed out in order by a counter (tickets != nullptr; otherwise wave w takes runs w, w + waves, ...).  The
    // waves of a chip do not progress equally fast -- its eight XCDs hold clocks 5-7 % apart under this kernel, and a
    // SIMD serves its waves oldest first --, so equal shares end with the slowest.  With a counter every wave draws
    // its next run while it works on the current one; the runs that are drawn last are short, so that the waves end
    // within a few chunks of each other.
    // The workgroups of a launch form `decks` groups (1, 2, 4 or 8); group g is dealt the runs g, g + decks, ... from a
    // counter of its own (kTicketStride apart).  One counter serves some 40 draws per microsecond: a launch of several
    // milliseconds stays far below that with a single counter, a launch of one millisecond with its few dozen draws
    // per wave needs eight.  A group is made of workgroups of every XCD (blockIdx.x / 8 % decks: blocks b and b + 8
    // share an XCD), because the XCDs of a chip hold different clocks: within each group the waves of the fast XCDs
    // then draw more runs than those of the slow ones, and all groups end together.
    unsigned* tickets;             // per group: [0] the group's next draw minus its waves, [1] its workgroups that have ended
    unsigned decks;
    unsigned lead_runs;            // nchan * runs_per_chan: runs [0, lead_runs) are the long ones
    unsigned lead_chunks;          // chunks of a channel that the long runs cover (chunks_per_chan without a tail part)
    unsigned tail_chan0, tail_c0;  // the short runs start at this channel and, within a channel, at this chunk
    unsigned tail_runs_per_chan;   // a channel's chunks [tail_c0, chunks_per_chan) are cut into this many runs ...
    unsigned tail_run_chunks;      // ... of this many chunks
    unsigned tail_div_shift;
    unsigned long long tail_div_magic;
};

// Run number -> the chunks [c0, c1) of channel `chan` it covers.
struct RunSpan {
    unsigned chan, c0, c1;
};
__host__ __device__ __forceinline__ RunSpan run_span(const FirRuns& rn, unsigned run) {
    RunSpan r;
    if (run < rn.lead_runs) {
        r.chan = (unsigned)(((unsigned long long)run * rn.div_magic) >> rn.div_shift);
        r.c0 = (run - r.chan * rn.runs_per_chan) * rn.run_chunks;
        const unsigned end = r.c0 + rn.run_chunks;
        r.c1 = end < rn.lead_chunks ? end : rn.lead_chunks;
    } else {
        const unsigned u = run - rn.lead_runs;
        const unsigned tc = (unsigned)(((unsigned long long)u * rn.tail_div_magic) >> rn.tail_div_shift);
        r.chan = rn.tail_chan0 + tc;
        r.c0 = rn.tail_c0 + (u - tc * rn.tail_runs_per_chan) * rn.tail_run_chunks;
        const unsigned end = r.c0 + rn.tail_run_chunks;
        r.c1 = end < rn.chunks_per_chan ? end : rn.chunks_per_chan;
    }
    return r;
}

__device__ __forceinline__ void wave_fence() {
    // Orders this wave's LDS accesses as written (other lanes of the wave read what this lane wrote); the hardware
    // executes a wave's LDS instructions in order, so nothing is emitted for it.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

#ifdef DSPX_M4W_TIMING
// Development build only: per-wave cycle counts of the phases of a chunk (tools/m4w_timing.py reads them).
__device__ unsigned long long g_m4w_timing[16 * 8192];
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#endif

// OB: a segment is 16 * OB consecutive outputs (4 * OB row blocks, one accumulator each), a chunk 16 segments.
// HG: pad groups (of one segment's length) in front of the samples of a chunk.  <HG 9, OB 2>: 512-output chunks, 288
// halo slots, four workgroups per CU (89 VGPRs; up to 77 taps); <5, 4>, <7, 4> and <9, 4>: 1024-output chunks, 320 /
// 448 / 576 halo slots (up to 313 / 441 / 569 taps), three / three / two workgroups per CU (148 / 154 / 158 VGPRs; LDS
// holds no third with 576 slots).
template <int RESIDENT, int HG, int OB>
__global__ __launch_bounds__(kBlock, RESIDENT) void fir_f64_mfma4w_kernel(const double* __restrict__ taps,
                                                                           const double* __restrict__ x,
                                                                           const double* __restrict__ hist,
                                                                           double* __restrict__ y, FirShape s, FirRuns rn) {
    using G = M4Geo<OB>;
    constexpr int RB = 4 * OB;                                   // row blocks of four outputs per segment
    constexpr int kChunk = G::kWaveOut;                          // outputs per wave and chunk (512)
    constexpr int kHaloGroups = HG;
    constexpr int kHalo = HG * G::kSeg;                          // image slots in front of the chunk (288 / 576)
    constexpr int kStageSteps = kHalo / 4 - 1;                   // steps the image serves (71 / 143)
    constexpr int kTapSlots = 4 * kStageSteps + 8;
    constexpr int kTapIters = (kTapSlots + kBlock - 1) / kBlock;
    constexpr int kImgGroups = (kHalo + kChunk) / G::kSeg;       // 25 / 34 pad groups of 32 slots (34 doubles) per image
    constexpr int kGroupPairs = G::kPadStep / 2;                 // 17 (33 for OB = 4)
    constexpr int kPG = G::kShift - 1;                           // log2 of the sample pairs per group
    constexpr int kImgPairs = kImgGroups * kGroupPairs;          // 425 / 578 pairs = 6800 / 9248 bytes per wave
    constexpr int kTapDoubles = (kTapSlots + 1) & ~1;
    constexpr int kBodyIters = kChunk / 2 / 64;                  // 16-byte loads per lane and chunk (4)
    constexpr int kHaloIters = (kHaloGroups * kGroupPairs + 63) / 64;          // 3 / 5
    constexpr int kFirstIters = ((kHalo + kChunk) / 2 + 63) / 64;              // 7 / 9
    static_assert(RESIDENT * (kTapDoubles * 8 + (kBlock / 64) * kImgPairs * 16) <= 160 * 1024, "LDS of the resident workgroups");
    constexpr int kStores = kChunk / 64;                         // 8
    static_assert(kChunk / G::kSeg * G::kOutStep <= (kImgGroups - kHaloGroups) * G::kPadStep,
                  "the transposed results fit in the body's place");
    __shared__ double2 lds[kTapDoubles / 2 + (kBlock / 64) * kImgPairs];
    __shared__ unsigned waves_ended;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* const tapz = reinterpret_cast<double*>(lds);
    dvec2* const imgw = reinterpret_cast<dvec2*>(lds + kTapDoubles / 2 + wave * kImgPairs);
    double* const imgd = reinterpret_cast<double*>(imgw);
    const long long n = s.n;
    const int ntaps = s.ntaps;
    auto negative = [](long long v) __attribute__((always_inline)) {
        int hi = (int)((unsigned long long)v >> 32);
        asm("" : "+s"(hi));
        return hi < 0;
    };

    // tap slot i <- tap i - 3 (zero outside the filter), once per launch
#pragma unroll
    for (int r = 0; r < kTapIters; ++r) {
        const int i = tid + r * kBlock, j = i - 3;
        if (i < kTapSlots) tapz[i] = (j >= 0 && j < ntaps) ? taps[j] : 0.0;
    }
    if (tid == 0) waves_ended = 0;
    __syncthreads();

    const int steps = (ntaps + 2) / 4 + 1;                                        // <= kStageSteps (the launcher picks HG so)
    const int qlo = (kHalo - 4 * (steps - 1)) & ~(G::kSeg - 1);                // lowest image slot the products read
    const int hgroups = (kHalo - qlo) / G::kSeg;                               // pad groups of halo in use (1..9)
    const int hpairs = hgroups * kGroupPairs;

    // per-lane LDS addresses, formed once
    const double* w_home = imgd + kHaloGroups * G::kPadStep + G::kPadStep * (lane & 15) + (lane >> 4);
    const double* a_home = tapz + 3 + (lane & 3) - (lane >> 4);
    // results pass through the body's place in the image, output q at q + q / 32 (one spare double per segment)
    double* o_home = imgd + kHaloGroups * G::kPadStep + G::kOutStep * (lane & 15) + (lane >> 4);   // acc[rb] -> o_home[4 * rb]
    const double* g_home = imgd + kHaloGroups * G::kPadStep + lane + (lane >> G::kShift);           // output lane + 64 j at g_home[66 j]
    constexpr int kBackStep = 64 + 64 / G::kSeg;
    // halo carry: pair lane + 64 i of the image's tail goes to the same pair of the halo's place.  A lane past the tail's
    // end reads the tail's last pair instead and writes nothing.  A load that every lane executes defines its registers
    // on every path: the compiler then needs no copies of them around it (loads under a lane condition cost some forty
    // moves per chunk, on the unit the products use); the condition is on the stores, which define nothing.
    int t_off[kHaloIters];
#pragma unroll
    for (int i = 0; i < kHaloIters; ++i) t_off[i] = lane + 64 * i < hpairs ? lane + 64 * i : hpairs - 1;
    dvec2* const t_dst = imgw + (kHaloGroups - hgroups) * kGroupPairs + lane;
    const dvec2* const t_base = imgw + (kImgGroups - hgroups) * kGroupPairs;
    dvec2* b_dst = imgw + kHaloGroups * kGroupPairs + lane + (lane >> kPG);       // body pair lane + 64 i at b_dst[i * kPairStep]
    dvec2* f_dst = imgw + (qlo / G::kSeg) * kGroupPairs + lane + (lane >> kPG);   // first window: pair lane + 64 i from qlo
    constexpr int kPairStep = 64 + 128 / G::kSeg;   // image distance of pairs 64 apart (one pad pair per group)

    double O[kStores];
    double* pend_yt = nullptr;
    int pend_left = 0;
    bool stores_due = false;
    auto store_chunk = [&]() __attribute__((always_inline)) {
        if (pend_left >= kChunk) {
#pragma unroll
            for (int j = 0; j < kStores; ++j) __builtin_nontemporal_store(O[j], scalar_base(pend_yt + 64 * j) + (unsigned)lane);
        } else {
            const int fl = fresh(lane) & 63;
#pragma unroll
            for (int j = 0; j < kStores; ++j)
                if (fl + 64 * j < pend_left) pend_yt[fl + 64 * j] = O[j];
        }
    };

    double acc[RB];

    const unsigned nwaves = gridDim.x * (kBlock / 64);
#ifdef DSPX_M4W_TIMING
    unsigned long long t_prod = 0, t_probe = 0, t_rest = 0, t_last = 0, n_chunks = 0, t_64 = 0;
    const unsigned long long t_begin = stamp();
    const unsigned long long r_begin = __builtin_amdgcn_s_memrealtime();
#endif
    // Drawn runs: this workgroup's group, its counters, its waves, and this wave's place among them.
    const unsigned decks = rn.decks;                       // (a power of two; 8 * decks divides the grid)
    const unsigned deck = (blockIdx.x >> 3) & (decks - 1);
    unsigned* const tickets = rn.tickets != nullptr ? rn.tickets + deck * kTicketStride : nullptr;
    const unsigned deck_blocks = gridDim.x / decks;
    const unsigned deck_waves = deck_blocks * (kBlock / 64);
    const unsigned deck_place = ((blockIdx.x >> 3) / decks) * 8 + (blockIdx.x & 7);   // this workgroup among its group's
    const int npairs = (kHalo + kChunk - qlo) / 2;   // sample pairs of a run's first window
    static_assert(kBodyIters + kHaloIters >= kFirstIters, "a first window fits in the registers of a body and a halo");
    bool window_ready = false;   // the image holds the run's first window already (fetched during the run before)
    // The pair of a first window that this lane fetches in its i-th load: lane + 64 i, the window's last pair for the
    // lanes past its end (as for the halo carry).  Formed where it is used, not carried through the products.
    auto window_pair = [&](int i) __attribute__((always_inline)) {
        const int p = fresh(lane) + 64 * i;
        return p < npairs ? p : npairs - 1;
    };
    for (unsigned run = tickets != nullptr ? decks * (deck_place * (kBlock / 64) + wave) + deck
                                           : blockIdx.x * (kBlock / 64) + wave;
         run < rn.total_runs;) {
        const RunSpan span = run_span(rn, run);
        const unsigned chan = span.chan, c0 = span.c0, c1 = span.c1;
        const double* __restrict__ xc = x + (unsigned long long)chan * s.x_stride;
        const double* __restrict__ hc = hist ? hist + (unsigned long long)chan * s.hist_stride : nullptr;
        double* __restrict__ yc = y + (unsigned long long)chan * s.y_stride;

        // ---- the run's first window: image slot q <- sample 512*c0 - kHalo + q for q >= qlo ----
        if (!window_ready) {
            const long long start = (long long)c0 * kChunk;
            const long long g_lo = start - kHalo + qlo;
            if (!negative(g_lo) && !negative(n - (start + kChunk))) {
                const dvec2* src = reinterpret_cast<const dvec2*>(xc + g_lo);
                dvec2 F[kFirstIters];
#pragma unroll
                for (int i = 0; i < kFirstIters; ++i) F[i] = scalar_base(src)[(unsigned)window_pair(i)];
#pragma unroll
                for (int i = 0; i < kFirstIters; ++i)
                    if (lane + 64 * i < npairs) f_dst[i * kPairStep] = F[i];
            } else {  // the window reaches outside the channel: every sample range-checked (kept small, not fast)
#pragma unroll 1
                for (int p = fresh(lane) & 63; p < npairs; p += 64)
                    imgw[(qlo / G::kSeg) * kGroupPairs + p + (p >> kPG)] =
                        dvec2{sample_at(xc, hc, g_lo + 2 * p, n, ntaps), sample_at(xc, hc, g_lo + 2 * p + 1, n, ntaps)};
            }
            wave_fence();
        }
        // The next run is drawn now (the first window's loads have arrived, so nothing waits for this one): its number
        // is back long before this run's last chunk, which fetches that run's first window.
        unsigned drawn = 0;
        if (tickets != nullptr && lane == 0)
            drawn = __hip_atomic_fetch_add(tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned next_run = rn.total_runs;
        window_ready = false;

        const double* xn = xc + (unsigned long long)(c0 + 1) * kChunk;  // the body of the chunk after the current one
        double* yk = yc + (unsigned long long)c0 * kChunk;
        long long left = n - (long long)c0 * kChunk;                    // outputs of the channel from this chunk on
        // One chunk of the run.  `last` (compile time): the run's last chunk, whose hand-over installs not the next
        // chunk's body and halo but the whole first window of the wave's next run, fetched while the products run --
        // a change of run then costs what a change of chunk costs.  (Two copies of the code, so that the registers of
        // the two cases never meet.)
        auto do_chunk = [&](auto last, const unsigned k) __attribute__((always_inline)) {
            constexpr bool have_next = !decltype(last)::value;
            const bool next_full = have_next && !negative(left - 2 * kChunk) && !(M4W_ABLATE & 16);  // the next body lies inside the channel
            dvec2 R[have_next ? kBodyIters : 1];
            dvec2 F[have_next ? 1 : kFirstIters];
            bool switch_run = false;
            if constexpr (have_next) {
                if (next_full) {
#pragma unroll
                    for (int i = 0; i < kBodyIters; ++i)
                        R[i] = stream_load(scalar_base(reinterpret_cast<const dvec2*>(xn) + 64 * i) + (unsigned)lane);
                } else {
                    // (never read on this path; "defined" without an instruction, or the compiler clears the sixteen
                    // registers at the top of every chunk -- on the vector unit, which the matrix pipe then waits for)
#pragma unroll
                    for (int i = 0; i < kBodyIters; ++i) asm volatile("" : "=v"(R[i]));
                }
            } else {
                asm volatile("" ::"v"(drawn));   // (the drawn number is waited for here on every path, as the samples below)
                next_run = tickets != nullptr ? decks * (deck_waves + (unsigned)__builtin_amdgcn_readfirstlane((int)drawn)) + deck
                                              : run + nwaves;
                if (next_run < rn.total_runs) {
                    const RunSpan ns = run_span(rn, next_run);
                    const long long nstart = (long long)ns.c0 * kChunk, ng_lo = nstart - kHalo + qlo;
                    if (!negative(ng_lo) && !negative(n - (nstart + kChunk))) {   // (else: range-checked, at the top of that run)
                        switch_run = true;
                        const dvec2* src = reinterpret_cast<const dvec2*>(x + (unsigned long long)ns.chan * s.x_stride + ng_lo);
#pragma unroll
                        for (int i = 0; i < kFirstIters; ++i) F[i] = scalar_base(src)[(unsigned)window_pair(i)];
                    }
                }
            }
            // The first sample vectors and the image's tail (the next chunk's halo) are asked for before the finished
            // chunk's stores wait for its values: everything below up to the first matrix instruction is then one
            // LDS round trip.
            double w[RB];
#pragma unroll
            for (int j = 1; j < RB; ++j) w[j] = w_home[4 * (RB - j)];
            w[0] = w_home[0];
            const double* wp = w_home - G::kPadStep;
            const double* ap = a_home;
            double a_now = ap[0];
            dvec2 Tl[have_next ? kHaloIters : 1];
            if constexpr (have_next) {
#pragma unroll
                for (int i = 0; i < kHaloIters; ++i) Tl[i] = t_base[t_off[i]];
            }
            if (stores_due && !(M4W_ABLATE & 2)) store_chunk();
            if ((M4W_ABLATE & 2) && stores_due) {
#pragma unroll
                for (int j = 0; j < kStores; ++j) asm volatile("" ::"v"(O[j]));
            }
            stores_due = false;

            // ---- products (the loop of fir_f64_mfma4_kernel, on this wave's image) ----
#ifdef DSPX_M4W_TIMING
            const unsigned long long ta = stamp();
            if (t_last) t_rest += ta - t_last;
#endif
            {
                int el = 0;
                // `first`: the chunk's first step starts its sums from zero (no separate clearing of the accumulators)
                auto group = [&](auto first) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const double a = a_now;
                        a_now = ap[4 * (u + 1)];
#pragma unroll
                        for (int rb = RB - 1; rb >= 0; --rb) {
                            acc[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, w[(u - rb + RB) % RB], (first.value && u == 0) ? 0.0 : acc[rb], 0, 0, 0);
                            if (rb == RB - 1) w[(u + 1) % RB] = wp[G::kSeg - 4 * (u + 1)];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    el += RB, wp -= G::kPadStep, ap += 4 * RB;
                };
                if (steps >= RB) {
                    group(std::true_type{});
                } else {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc[rb] = 0.0;
                }
                while (el + RB <= steps) group(std::false_type{});
                if (el < steps) {
                    const int rem = steps - el;
#pragma unroll
                    for (int u = 0; u < RB - 1; ++u) {
                        if (u < rem) {
                            const double a = a_now;
                            a_now = ap[4 * (u + 1)];
#pragma unroll
                            for (int rb = RB - 1; rb >= 0; --rb) {
                                acc[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, w[(u - rb + RB) % RB], acc[rb], 0, 0, 0);
                                if (rb == RB - 1) w[(u + 1) % RB] = wp[G::kSeg - 4 * (u + 1)];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
#ifdef DSPX_M4W_TIMING
            const unsigned long long tb = stamp();
            t_prod += tb - ta;
#endif

            // ---- the exactness guard of fir_f64_mfma4_kernel: a wave with a non-finite accumulator sums its outputs
            // again over the products the reference forms.  Eight independent compares joined on the scalar unit.
            unsigned long long any_bad = 0;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) any_bad |= __builtin_amdgcn_ballot_w64(not_finite(acc[rb]));
#ifdef DSPX_M4W_TIMING
            {
                unsigned long long tc;
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc) : "s"(any_bad) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                t_probe += tc - tb;
                t_last = tc;
                if (++n_chunks == 64) t_64 = tc;
            }
#endif
            wave_fence();
            // The whole image is dead now (its tail is in Tl): the halo of the next chunk goes to its place, the
            // results pass through the body's place, and the next body follows them there.  A wave's LDS instructions
            // execute in order, so none of these waits for another.
            if constexpr (have_next) {
#pragma unroll
                for (int i = 0; i < kHaloIters; ++i)
                    if (lane + 64 * i < hpairs) t_dst[64 * i] = Tl[i];
            }
            if (any_bad != 0) {
                const long long pos = (long long)k * kChunk;
#pragma unroll 1
                for (int q = fresh(lane) & 63; q < kChunk; q += 64) {
                    double sum = 0.0;
                    for (int j = 0; j < ntaps; ++j) {
                        const long long g = pos + q - j;
                        if (g >= 0 && g < n)
                            sum = __builtin_fma(taps[j], xc[g], sum);
                        else if (g < 0 && hc != nullptr && g >= -(long long)(ntaps - 1))
                            sum = __builtin_fma(taps[j], hc[(ntaps - 1) + g], sum);
                    }
                    imgd[kHaloGroups * G::kPadStep + q + (q >> G::kShift)] = sum;
                }
            } else {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) o_home[4 * rb] = acc[rb];
            }
            wave_fence();
#pragma unroll
            for (int j = 0; j < kStores; ++j) O[j] = g_home[j * kBackStep];
            wave_fence();
            // The prefetched samples are waited for here on every path, also on those that will not use them: the
            // compiler cannot tell that a path which skipped the loads skips their use too, and would otherwise wait
            // for "loads still in flight" where the registers are written next -- at the top of a chunk, behind that
            // chunk's own prefetch.
            if constexpr (have_next) {
#pragma unroll
                for (int i = 0; i < kBodyIters; ++i) asm volatile("" ::"v"(R[i]));
            } else {
#pragma unroll
                for (int i = 0; i < kFirstIters; ++i) asm volatile("" ::"v"(F[i]));
            }
            if constexpr (have_next) {
                if (next_full) {
#pragma unroll
                    for (int i = 0; i < kBodyIters; ++i) b_dst[i * kPairStep] = R[i];
                } else if (!(M4W_ABLATE & 16)) {  // the channel ends inside the next chunk: zeros past its end
                    const long long pos = (long long)(k + 1) * kChunk;
#pragma unroll 1
                    for (int p = fresh(lane) & 63; p < kChunk / 2; p += 64)
                        imgw[kHaloGroups * kGroupPairs + p + (p >> kPG)] =
                            dvec2{sample_at(xc, hc, pos + 2 * p, n, ntaps), sample_at(xc, hc, pos + 2 * p + 1, n, ntaps)};
                }
            } else if (switch_run) {
#pragma unroll
                for (int i = 0; i < kFirstIters; ++i)
                    if (lane + 64 * i < npairs) f_dst[i * kPairStep] = F[i];
                window_ready = true;
            }
            wave_fence();
            pend_yt = yk;
            pend_left = negative(left - kChunk) ? (int)left : kChunk;
            stores_due = true;
        };
        unsigned k = c0;
        for (; k + 1 < c1; ++k, xn += kChunk, yk += kChunk, left -= kChunk) do_chunk(std::false_type{}, k);
        do_chunk(std::true_type{}, k);
        run = next_run;
    }
    if (stores_due) store_chunk();
    if (tickets != nullptr && lane == 0) {
        // The last workgroup of the group to end leaves its counters at zero for the next launch on this stream.
        // (Every wave has drawn its last, out-of-range number before it counts itself here; a workgroup's waves count
        // themselves in LDS, so that the counter in memory sees one addition per workgroup.)
        if (__hip_atomic_fetch_add(&waves_ended, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == kBlock / 64 - 1 &&
            __hip_atomic_fetch_add(tickets + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == deck_blocks - 1) {
            __hip_atomic_store(tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(tickets + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#ifdef DSPX_M4W_TIMING
    {
        const unsigned long long t_end = stamp();
        const unsigned gw = blockIdx.x * (kBlock / 64) + wave;
        if (lane == 0 && gw < 8192) {
            unsigned long long* o = g_m4w_timing + 16 * gw;
            o[0] = t_prod, o[1] = t_probe, o[2] = t_rest, o[3] = n_chunks, o[4] = t_begin, o[5] = t_end;
            o[6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4)                     // HW_ID
                   | ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15) << 32);  // XCC_ID
            o[7] = __builtin_amdgcn_s_memrealtime() - r_begin;
            o[8] = t_64;
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// The wave-autonomous form for filters that need several windows (more than 569 taps; lfilter mode, 16-byte aligned
// buffers) and for wavelet banks.  As in fir_f64_mfma4w_kernel every wave is a pipeline of its own -- an LDS image
// and, here, a tap window of its own, no workgroup barrier -- but a chunk's sums are built from several windows:
// window w multiplies the tap offsets [4*e0, 4*(e0 + steps)) with the samples that lie 4*e0 further back, so every
// window brings its own image and its own tap slots.  What runs is <HG 9, OB 4>: chunks of 1024 values, 576 halo
// slots, windows of up to 143 steps (572 tap offsets), 17.8 KB of LDS per wave, two workgroups per CU (chunks of 512
// with windows of 71 steps, four workgroups per CU, measured 1-2 % slower from 570 to 4095 taps).  There is no halo
// to carry; a sample is read by every window that slides over it, and what keeps those reads out of HBM is the
// order in which the chunks are taken: the waves of an XCD work on adjacent chunks at the same time, so all but the
// first read of a sample are hits in that XCD's L2 (at 4095 taps over one stream the HBM reads stay within a few per
// cent of the algorithmic 8 bytes per output; 12.8 x of them when every wave walked a run of consecutive chunks of
// its own).  While the products of one window run, the samples and taps of the next -- the next chunk's first
// window after a chunk's last -- are in flight into registers; between two windows lies one LDS round trip.  Same
// products in the same order as fir_f64_mfma4_kernel: same bits.
// Large inputs with filters of up to 3072 taps, and wavelet banks, keep that order for the first seven eighths of
// every channel (row) only: the XCDs of a chip hold clocks a few per cent apart, so with equal shares the launch ends
// with the slowest.  The last eighth of every channel is a pool from which the waves draw chunk by chunk through the
// counters of the launch's stream (FirRuns::tickets), a whole chunk before they need it; the XCDs that are through
// with their stretches first take more of it.  The pool's chunks land on any XCD and their overlapping reads miss its
// L2 (measured as L2 fetches from the fabric: 1023 taps 1.12 x the algorithmic read bytes instead of 1.01 x, for
// +1.4 % of throughput; with every chunk drawn 2.0 x for +1.2 %).  Longer filters run at the chip's full clock on
// every XCD and keep the fixed order throughout.
//
// BANK (the wavelet path, reference dsp.hpp:1136-1217): one input, rows of taps of different lengths.  A unit of work is
// a chunk of one row: 1024 values of c[t] = sum_k taps_r[k] * x[t + advance_r - k] (zeros outside [0, n)), of which the
// first 1022 leave as (c[i+1] - c[i]) * scale_r -- the first difference, the scaling and the crop fused into the
// read-back, as in the workgroup-tiled kernel; a chunk advances by 1022 outputs so that the c[i+1] (c[i+2] when the
// row's advance is odd and was rounded down to keep the windows 16-byte aligned) of its last output is its own.  Rows
// come longest first; the waves of the chip work through them together, row after row, so they all meet chunks of
// the same cost at the same time.  VEC2 = false reads the windows with 8-byte loads (a signal that is only 8-byte
// aligned); everything else is the same, so a row's bits do not depend on the alignment of the buffers.
template <int RESIDENT, int HG, int OB, bool BANK, bool VEC2>
__global__ __launch_bounds__(kBlock, RESIDENT) void fir_f64_mfma4wm_kernel(const double* __restrict__ taps,
                                                                            const double* __restrict__ x,
                                                                            const double* __restrict__ hist,
                                                                            double* __restrict__ y,
                                                                            const BankRow* __restrict__ bank, FirShape s,
                                                                            FirRuns rn) {
    using G = M4Geo<OB>;                                         // <HG 9, OB 2>: 512-output chunks, 288 halo slots; <9, 4>: 1024, 576
    constexpr int RB = 4 * OB;
    constexpr int kHalo = HG * G::kSeg;                          // image slots in front of the chunk
    constexpr int kStageSteps = kHalo / 4 - 1;                   // steps of four tap offsets one window serves (71 / 143)
    constexpr int kTapSlots = 4 * kStageSteps + 8;               // 292 / 580
    constexpr int kChunk = G::kWaveOut;                          // values per chunk: 512 / 1024
    constexpr int kOutputs = BANK ? kChunk - 2 : kChunk;         // outputs a chunk stores (and advances by)
    constexpr int kHaloGroups = HG;
    constexpr int kImgGroups = (kHalo + kChunk) / G::kSeg;    // 25 / 25
    constexpr int kGroupPairs = G::kPadStep / 2;                 // 17 / 33
    constexpr int kPG = G::kShift - 1;                           // log2 of the sample pairs per group
    constexpr int kImgPairs = kImgGroups * kGroupPairs;          // 425 / 825
    constexpr int kTapPairs = (kTapSlots + 3) / 4 * 2;        // pairs that hold the tap slots of a window
    constexpr int kWinIters = ((kHalo + kChunk) / 2 + 63) / 64;   // 16-byte loads per lane and window (7 / 13)
    constexpr int kTapLoads = (kTapSlots + 63) / 64;              // 8-byte loads per lane and window (5 / 10)
    constexpr int kStores = kChunk / 64;
    constexpr int kPairStep = 64 + 128 / G::kSeg;                // image distance of pairs 64 apart
    constexpr int kBackStep = 64 + 64 / G::kSeg;
    static_assert(RESIDENT * (kBlock / 64) * (kTapPairs + kImgPairs) * 16 <= 160 * 1024, "LDS of the resident workgroups");
    static_assert(!BANK || (kChunk + 1) + ((kChunk + 1) >> G::kShift) < (kImgGroups - kHaloGroups) * G::kPadStep,
                  "bank mode reads c[q + 2] of a chunk's last values inside the image");
    __shared__ double2 lds[(kBlock / 64) * (kTapPairs + kImgPairs)];
    __shared__ unsigned waves_ended;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (rn.tickets != nullptr) {   // (drawn chunks: the workgroup counts its waves as they end)
        if (tid == 0) waves_ended = 0;
        __syncthreads();
    }
    double* const tapw = reinterpret_cast<double*>(lds + wave * (kTapPairs + kImgPairs));
    dvec2* const imgw = reinterpret_cast<dvec2*>(lds + wave * (kTapPairs + kImgPairs) + kTapPairs);
    double* const imgd = reinterpret_cast<double*>(imgw);
    const long long n = s.n;
    auto negative = [](long long v) __attribute__((always_inline)) {
        int hi = (int)((unsigned long long)v >> 32);
        asm("" : "+s"(hi));
        return hi < 0;
    };
    // The filter of a chunk.  Steps of four tap offsets cover its band; a window serves up to kStageSteps of them, the
    // windows split them evenly (bank mode: the host has done that division per row).
    struct Filter {
        const double* tc;  // its taps
        double post;       // bank mode: the factor on the first difference
        int ntaps, odd;    // bank mode: whether the row's advance was rounded down
        int total_steps, per_stage;
    };
    Filter cur{};
    cur.tc = taps, cur.post = 1.0, cur.ntaps = s.ntaps, cur.odd = 0;
    cur.total_steps = (s.ntaps + 2) / 4 + 1;
    {
        const int nstages = (cur.total_steps + kStageSteps - 1) / kStageSteps;
        cur.per_stage = (cur.total_steps + nstages - 1) / nstages;
    }
    Filter nxt = cur;  // (lfilter mode: one filter per launch, never written again)

    const double* w_home = imgd + kHaloGroups * G::kPadStep + G::kPadStep * (lane & 15) + (lane >> 4);
    const double* a_home = tapw + 3 + (lane & 3) - (lane >> 4);
    double* o_home = imgd + kHaloGroups * G::kPadStep + G::kOutStep * (lane & 15) + (lane >> 4);
    const double* g_home = imgd + kHaloGroups * G::kPadStep + lane + (lane >> G::kShift);
    const double* g_home1 = imgd + kHaloGroups * G::kPadStep + (lane + 1) + ((lane + 1) >> G::kShift);   // bank mode: values lane + 1, + 2
    const double* g_home2 = imgd + kHaloGroups * G::kPadStep + (lane + 2) + ((lane + 2) >> G::kShift);
    dvec2* const c_home = imgw + lane + (lane >> kPG);   // window pair lane + 64 i, counted from slot 0, at c_home[i * kPairStep]

    double O[kStores];
    double* pend_yt = nullptr;
    int pend_left = 0;
    bool stores_due = false;
    auto store_chunk = [&]() __attribute__((always_inline)) {
        if (pend_left >= kOutputs) {
#pragma unroll
            for (int j = 0; j < kStores; ++j)
                if (!BANK || j < kStores - 1 || lane < 64 - (kChunk - kOutputs))
                    __builtin_nontemporal_store(O[j], scalar_base(pend_yt + 64 * j) + (unsigned)lane);
        } else {
            const int fl = fresh(lane) & 63;
#pragma unroll
            for (int j = 0; j < kStores; ++j)
                if (fl + 64 * j < pend_left) pend_yt[fl + 64 * j] = O[j];
        }
    };

    // ---- the window in flight: image slot q <- sample pos - 4*e0 - kHalo + q (the whole image, whatever part of it
    // the window's steps read), tap slot i <- tap 4*e0 - 3 + i (zero outside the filter).  A window that reaches
    // outside the channel, or a tap window that reaches outside the filter, is not fetched ahead: commit() fills it
    // with range checks.
    constexpr int kWinPairs = (kHalo + kChunk) / 2;   // 400 = 6 * 64 + 16 (800 = 12 * 64 + 32)
    dvec2 R[kWinIters];
    double T[kTapLoads];
    bool f_interior = false, f_taps_inside = false;
    auto fetch = [&](const Filter& f, const double* xc, long long pos, int e0) __attribute__((always_inline)) {
        const long long g_lo = pos - 4 * e0 - kHalo;
        f_interior = !negative(g_lo) && !negative(n - (g_lo + kHalo + kChunk));
        if (f_interior) {
#pragma unroll
            for (int i = 0; i < kWinIters; ++i) {
                if (i < kWinIters - 1 || lane < kWinPairs - 64 * (kWinIters - 1)) {
                    if constexpr (VEC2) {
                        R[i] = scalar_base(reinterpret_cast<const dvec2*>(xc + g_lo) + 64 * i)[(unsigned)lane];
                    } else {
                        const auto p = scalar_base(xc + g_lo + 128 * i);
                        R[i] = dvec2{p[2u * (unsigned)lane], p[2u * (unsigned)lane + 1u]};
                    }
                }
            }
        }
        const int j0 = 4 * e0 - 3;
        f_taps_inside = j0 >= 0 && j0 + kTapSlots <= f.ntaps;
        if (f_taps_inside) {
#pragma unroll
            for (int r = 0; r < kTapLoads; ++r)
                if (r < kTapLoads - 1 || lane < kTapSlots - 64 * (kTapLoads - 1)) T[r] = scalar_base(f.tc + j0 + 64 * r)[(unsigned)lane];
        } else {  // the filter's first and last windows: the nearest tap is fetched, commit() puts zeros where none exists
            const int fl = fresh(lane) & 63;
#pragma unroll
            for (int r = 0; r < kTapLoads; ++r) T[r] = f.tc[min(max(j0 + fl + 64 * r, 0), f.ntaps - 1)];
        }
    };
    auto commit = [&](const Filter& f, const double* xc, const double* hc, long long pos, int e0) __attribute__((always_inline)) {
        if (f_interior) {
#pragma unroll
            for (int i = 0; i < kWinIters; ++i)
                if (i < kWinIters - 1 || lane < kWinPairs - 64 * (kWinIters - 1)) c_home[i * kPairStep] = R[i];
        } else {
            const long long g_lo = pos - 4 * e0 - kHalo;
#pragma unroll 1
            for (int p = fresh(lane) & 63; p < kWinPairs; p += 64)
                imgw[p + (p >> kPG)] = dvec2{sample_at(xc, hc, g_lo + 2 * p, n, f.ntaps), sample_at(xc, hc, g_lo + 2 * p + 1, n, f.ntaps)};
        }
        if (f_taps_inside) {
#pragma unroll
            for (int r = 0; r < kTapLoads; ++r)
                if (r < kTapLoads - 1 || lane < kTapSlots - 64 * (kTapLoads - 1)) tapw[lane + 64 * r] = T[r];
        } else {
            const int fl = fresh(lane) & 63;
#pragma unroll
            for (int r = 0; r < kTapLoads; ++r) {
                const int i = fl + 64 * r, j = 4 * e0 - 3 + i;
                if (i < kTapSlots) tapw[i] = (unsigned)j < (unsigned)f.ntaps ? T[r] : 0.0;
            }
        }
        wave_fence();
    };

    double acc[RB];
    // Blocks b and b + 8 share an XCD and its L2.  A chunk's windows slide back over ntaps + chunk samples, which the
    // chunks next to it read as well: the waves of an XCD therefore take ADJACENT runs at the same time (the grid is a
    // multiple of 8 workgroups; the launcher makes a run one chunk), so that what one wave fetched is found in that
    // XCD's L2 by its neighbours, and the XCD moves on to its next stretch of the input as a whole.
    const unsigned xcd = blockIdx.x % 8, in_xcd = (blockIdx.x / 8) * (kBlock / 64) + wave, per_xcd = gridDim.x / 8 * (kBlock / 64);
    // A wave's place among its XCD's waves moves on by one with every stretch: the chunks at the start of a channel
    // cost more (their windows are filled from the history, synchronously), and with channels a power of two long a
    // fixed place would hand all of them to the same few waves, which the rest of the chip then waits for.
    // The last stretch is usually a partial one: its runs are dealt round-robin over the XCDs instead, so that what is
    // left (all there is, for a small input) is spread over the whole chip and not over the first XCDs only.
    const unsigned per_stretch = 8 * per_xcd;
    const unsigned lead = rn.tickets != nullptr ? rn.lead_runs : rn.total_runs;   // the runs dealt in stretches (all, without a pool)
    auto run_at = [&](unsigned first, unsigned pl) __attribute__((always_inline)) {
        return first + per_stretch > lead ? first + in_xcd * 8 + xcd : first + xcd * per_xcd + pl;
    };
    unsigned stretch = 0, place = in_xcd;  // `stretch`: the first run of the stretch the wave is in
    unsigned run = run_at(stretch, place);   // (with a pool: every wave has one, the launcher sees to that)
    // Drawn chunks (rn.tickets; counter groups made of workgroups of every XCD, as in fir_f64_mfma4w_kernel).  The first
    // seven eighths of every channel (row) -- runs [0, lead), in the run order of run_span() -- are dealt in stretches,
    // XCD by XCD, as above; the last eighth of every channel is a pool: a wave's first two pool chunks are fixed, every
    // later one is drawn from its group's counter a whole chunk before it is needed (the chunk after the current one
    // must be known when the current one's last window starts).  The XCDs that are through with their stretches first
    // take more of the pool, so the launch does not end with the slowest XCD, and only the pool's chunks lose the
    // XCD-by-XCD order that keeps overlapping reads in an XCD's L2.
    const unsigned decks = rn.decks;
    const unsigned deck = (blockIdx.x >> 3) & (decks - 1);
    unsigned* const tickets = rn.tickets != nullptr ? rn.tickets + deck * kTicketStride : nullptr;
    const unsigned deck_blocks = gridDim.x / decks, deck_waves = deck_blocks * (kBlock / 64);
    const unsigned deck_wave = (((blockIdx.x >> 3) / decks) * 8 + (blockIdx.x & 7)) * (kBlock / 64) + wave;
    const unsigned pool_first = lead + decks * deck_wave + deck, pool_second = lead + decks * (deck_waves + deck_wave) + deck;
    bool pooled = false;   // the chunk in hand comes from the pool
    unsigned known_next = 0, drawn = 0;
    if (run < rn.total_runs) {
        // The chunk in hand: its channel (row), `pos` = the sample position of its first value, `yk` / `opos` = where its
        // first output goes and which output of the channel that is.
        const double *xc, *hc;
        double* yk;
        long long pos, opos;
        unsigned left_in_run;   // chunks of the run after the current one
        auto locate = [&](unsigned r, Filter& f, const double*& cx, const double*& ch, double*& cy, long long& cpos,
                          long long& copos, unsigned& more) __attribute__((always_inline)) {
            const RunSpan span = run_span(rn, r);
            const unsigned chan = span.chan, c0 = span.c0;
            more = span.c1 - c0 - 1;
            copos = (long long)c0 * kOutputs;
            if constexpr (BANK) {
                const BankRow row = bank[chan];
                f.tc = taps + row.taps_off, f.post = row.scale, f.ntaps = row.ntaps, f.odd = row.advance & 1;
                f.total_steps = (row.ntaps + 2) / 4 + 1, f.per_stage = row.win_steps;
                cpos = copos + row.advance - f.odd;
                cx = x, ch = nullptr;
                cy = y + row.out_off * s.y_stride + copos;
            } else {
                cpos = copos;
                cx = x + (unsigned long long)chan * s.x_stride;
                ch = hist ? hist + (unsigned long long)chan * s.hist_stride : nullptr;
                cy = y + (unsigned long long)chan * s.y_stride + copos;
            }
        };
        locate(run, cur, xc, hc, yk, pos, opos, left_in_run);
        // One pass of the loop per window; the window about to be multiplied is (the chunk at `pos`, first step e0).
        int e0 = 0;
        fetch(cur, xc, pos, 0);  // the wave's first window: waited for on the spot
        for (;;) {
            commit(cur, xc, hc, pos, e0);
            const int steps = min(cur.per_stage, cur.total_steps - e0);
            const bool last_window = e0 + steps >= cur.total_steps;
            // the window after this one: the chunk's next, the first of the run's next chunk, or of the wave's next run
            const unsigned next_place = place + 1 == per_xcd ? 0 : place + 1;
            const unsigned stretch_next = run_at(stretch + per_stretch, next_place);   // (past `lead`: none)
            const unsigned next_run = tickets == nullptr ? stretch_next : pooled ? known_next : stretch_next < lead ? stretch_next : pool_first;
            const bool have_next = !last_window || left_in_run > 0 || next_run < rn.total_runs;
            const double *n_xc = xc, *n_hc = hc;
            double* n_yk = yk;
            long long n_pos = pos, n_opos = opos;
            unsigned n_left = left_in_run;
            int n_e0 = e0 + steps;
            if constexpr (BANK) nxt = cur;
            if (last_window && have_next) {
                n_e0 = 0;
                if (left_in_run > 0)
                    n_pos = pos + kOutputs, n_opos = opos + kOutputs, n_yk = yk + kOutputs, n_left = left_in_run - 1;
                else
                    locate(next_run, nxt, n_xc, n_hc, n_yk, n_pos, n_opos, n_left);
            }
            if (have_next) fetch(BANK ? nxt : cur, n_xc, n_pos, n_e0);
            double w[RB];
#pragma unroll
            for (int j = 1; j < RB; ++j) w[j] = w_home[4 * (RB - j)];
            w[0] = w_home[0];
            const double* wp = w_home - G::kPadStep;
            const double* ap = a_home;
            double a_now = ap[0];
            if (stores_due) store_chunk();
            stores_due = false;
            {
                int el = 0;
                auto group = [&](auto first) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const double a = a_now;
                        a_now = ap[4 * (u + 1)];
#pragma unroll
                        for (int rb = RB - 1; rb >= 0; --rb) {
                            acc[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, w[(u - rb + RB) % RB], (first.value && u == 0) ? 0.0 : acc[rb], 0, 0, 0);
                            if (rb == RB - 1) w[(u + 1) % RB] = wp[G::kSeg - 4 * (u + 1)];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    el += RB, wp -= G::kPadStep, ap += 4 * RB;
                };
                if (e0 == 0) {  // a chunk's first step starts its sums from zero
                    if (steps >= RB) {
                        group(std::true_type{});
                    } else {
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) acc[rb] = 0.0;
                    }
                }
                while (el + RB <= steps) group(std::false_type{});
                if (el < steps) {
                    const int rem = steps - el;
#pragma unroll
                    for (int u = 0; u < RB - 1; ++u) {
                        if (u < rem) {
                            const double a = a_now;
                            a_now = ap[4 * (u + 1)];
#pragma unroll
                            for (int rb = RB - 1; rb >= 0; --rb) {
                                acc[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, w[(u - rb + RB) % RB], acc[rb], 0, 0, 0);
                                if (rb == RB - 1) w[(u + 1) % RB] = wp[G::kSeg - 4 * (u + 1)];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
            wave_fence();
            if (last_window) {
                // ---- the chunk is complete: the exactness guard, then the results through the body's place ----
                unsigned long long any_bad = 0;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) any_bad |= __builtin_amdgcn_ballot_w64(not_finite(acc[rb]));
                if (any_bad != 0) {
#pragma unroll 1
                    for (int q = fresh(lane) & 63; q < kChunk; q += 64) {
                        double sum = 0.0;
                        for (int j = 0; j < cur.ntaps; ++j) {
                            const long long g = pos + q - j;
                            if (g >= 0 && g < n)
                                sum = __builtin_fma(cur.tc[j], xc[g], sum);
                            else if (g < 0 && hc != nullptr && g >= -(long long)(cur.ntaps - 1))
                                sum = __builtin_fma(cur.tc[j], hc[(cur.ntaps - 1) + g], sum);
                        }
                        imgd[kHaloGroups * G::kPadStep + q + (q >> G::kShift)] = sum;
                    }
                } else {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) o_home[4 * rb] = acc[rb];
                }
                wave_fence();
                if constexpr (BANK) {
                    // (c[q+1] - c[q]) * factor, shifted by one when the advance was rounded down (wave-uniform); the last
                    // two lanes of the last read form values that are never stored
                    if (cur.odd) {
#pragma unroll
                        for (int j = 0; j < kStores; ++j) O[j] = (g_home2[j * kBackStep] - g_home1[j * kBackStep]) * cur.post;
                    } else {
#pragma unroll
                        for (int j = 0; j < kStores; ++j) O[j] = (g_home1[j * kBackStep] - g_home[j * kBackStep]) * cur.post;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < kStores; ++j) O[j] = g_home[j * kBackStep];
                }
                wave_fence();
                const long long left = n - opos;
                pend_yt = yk;
                pend_left = negative(left - kOutputs) ? (int)left : kOutputs;
                stores_due = true;
                if (left_in_run == 0) {
                    run = next_run;
                    if (tickets == nullptr || (!pooled && stretch_next < lead)) {
                        stretch += per_stretch, place = next_place;
                    } else if (have_next) {
                        // into the pool, or on within it: the chunk after the next becomes known, the one after that is drawn
                        // (a draw is only made while the known chunk lies inside the input, and only read when one was made)
                        known_next = !pooled ? pool_second
                                             : lead + decks * (2 * deck_waves + (unsigned)__builtin_amdgcn_readfirstlane((int)drawn)) + deck;
                        pooled = true;
                        if (known_next < rn.total_runs && lane == 0)
                            drawn = __hip_atomic_fetch_add(tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (!have_next) break;
            xc = n_xc, hc = n_hc, yk = n_yk, pos = n_pos, opos = n_opos, left_in_run = n_left, e0 = n_e0;
            if constexpr (BANK) cur = nxt;
        }
    }
    if (stores_due) store_chunk();
    if (tickets != nullptr) asm volatile("" ::"v"(drawn));   // (a draw that nobody read has arrived before the wave counts itself)
    if (tickets != nullptr && lane == 0) {
        // the last workgroup of the group to end leaves its counters at zero for the next launch on this stream
        if (__hip_atomic_fetch_add(&waves_ended, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == kBlock / 64 - 1 &&
            __hip_atomic_fetch_add(tickets + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == deck_blocks - 1) {
            __hip_atomic_store(tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(tickets + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// lfilter over narrow element types (reference dsp.hpp:361-364): the sum has the result type TOUT, and every product
// is formed in TIN, the type the containers' elements promote to -- float for containers of float, double for
// containers of double -- converted to TOUT and added, taps ascending.  Separately rounded operations in exactly the
// reference's order, so the results are bit-identical to the reference loop; a product with a sample that does not
// exist (before the start of a channel without history) is never formed.  <float, float> and <double, float> are the
// reference's lfilter<float>, <float, double> is lfilter<double> over containers of float (float products, double
// sum).  Functional paths, not tuned ones: one workgroup per tile of 2048 outputs, thread t owns outputs
// t, t + 256, ..., taps and samples through LDS in chunks of 256 taps.
constexpr int kNarrowTaps = 256;
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(kBlock) void fir_narrow_kernel(const TIN* __restrict__ taps, const TIN* __restrict__ x,
                                                             const TIN* __restrict__ hist, TOUT* __restrict__ y,
                                                             FirShape s) {
    __shared__ TIN xs[kNarrowTaps - 1 + kTile];
    __shared__ TIN tapl[kNarrowTaps];
    const int tid = threadIdx.x;
    const unsigned chan = blockIdx.x / s.tiles_per_chan, tile = blockIdx.x - chan * s.tiles_per_chan;
    const long long n = s.n, tile_start = (long long)tile * kTile;
    const int ntaps = s.ntaps;
    const TIN* __restrict__ xc = x + (unsigned long long)chan * s.x_stride;
    const TIN* __restrict__ hc = hist ? hist + (unsigned long long)chan * s.hist_stride : nullptr;
    TOUT acc[kR];
    long long jmax[kR];  // output r adds taps j < jmax[r]: all of them, or as many as there are samples behind it
#pragma unroll
    for (int r = 0; r < kR; ++r) {
        acc[r] = 0;
        const long long i = tile_start + tid + r * kBlock;
        jmax[r] = hc ? (long long)ntaps : min((long long)ntaps, i + 1);
    }
    for (int j0 = 0; j0 < ntaps; j0 += kNarrowTaps) {
        const int S = min(kNarrowTaps, ntaps - j0);
        if (j0 > 0) __syncthreads();
        if (tid < S) tapl[tid] = taps[j0 + tid];
        // slot q <- sample tile_start - j0 - (kNarrowTaps - 1) + q; tap j0 + jj of output o reads slot kNarrowTaps - 1 + o - jj
        for (int q = tid; q < kNarrowTaps - 1 + kTile; q += kBlock) {
            const long long g = tile_start - j0 - (kNarrowTaps - 1) + q;
            TIN v = 0;
            if (g >= 0) {
                if (g < n) v = xc[g];
            } else if (hc != nullptr && g >= -(long long)(ntaps - 1)) {
                v = hc[(ntaps - 1) + g];
            }
            xs[q] = v;
        }
        __syncthreads();
        for (int jj = 0; jj < S; ++jj) {
            const TIN b = tapl[jj];
#pragma unroll
            for (int r = 0; r < kR; ++r) {
#pragma clang fp contract(off)
                const TIN prod = b * xs[kNarrowTaps - 1 + tid + r * kBlock - jj];
                const TOUT term = static_cast<TOUT>(prod);
                if (j0 + jj < jmax[r]) acc[r] = acc[r] + term;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kR; ++r) {
        const long long i = tile_start + tid + r * kBlock;
        if (i < n) y[(unsigned long long)chan * s.y_stride + i] = acc[r];
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

// Row blocks of 16 outputs per segment of the matrix-pipe kernel.  2 (2048-output tiles, four workgroups per CU)
// measured faster than 4 (4096-output tiles, two per CU) at every tap count from 63 to 4095.
constexpr int kMatrixOB = 2;

int fir_tile() { return kTile; }
int fir_mfma_tile() { return M4Geo<kMatrixOB>::kTile; }
int fir_mfma_bank_stride() { return M4Geo<kMatrixOB>::kBankStride; }

// Compute units of the current device (256 on MI355X); queried once per device.
static unsigned compute_units() {
    static unsigned cached[64] = {};
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev]) return cached[dev];
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cached[dev] = (unsigned)cus;
}

static std::atomic<int> g_share_mode{0};   // test hook (dspx_test_run_shares): 1 fixed shares, 2 drawn runs for any input size

// Cuts `nchan` channels of `n` samples into runs of consecutive chunks of `kChunk` outputs for the wave-autonomous kernels: a
// run never crosses a channel; with a large input there are about kRunsPerWave runs per resident wave, so that the
// waves end together, and never fewer than kMinRun chunks per run unless the input is too small to occupy every
// wave that way (a run's first window is fetched with its halo, the only samples read twice).
//
// With `tickets` (the counter set of the launch's stream) and an input of kDrawnFrom chunks per wave or more, the runs
// are handed out in order instead (see FirRuns): long runs over the first three quarters of the input -- of every
// channel when there are few channels, the first three quarters of the channels otherwise --, about kLeadPerWave per
// wave, then runs an eighth as long over the rest.
// (Measured: at 64 chunks per wave -- launches of half a millisecond to a millisecond -- drawn runs are 1-4 % slower than
// fixed shares, because the long runs a wave commits to when it starts are most of its work; at 100 and more they are
// equal or faster.)
constexpr unsigned long long kDrawnFrom = 100, kDrawnFromBig = 80;   // (chunks of 512 / of 1024 outputs)
// The several-windows kernel draws single chunks from a wave's third chunk on, so it adapts from the start: it draws
// from kDrawnChunksFrom chunks per wave.
constexpr unsigned long long kDrawnChunksFrom = 32;
static bool large_enough_to_draw(unsigned nchan, long long n, unsigned max_waves, unsigned long long kChunk,
                                 unsigned long long from = kDrawnFrom) {
    const unsigned long long cpc = ((unsigned long long)n + kChunk - 1) / kChunk;
    return cpc <= 0x7fffffffull && (cpc * nchan / max_waves >= from || g_share_mode.load(std::memory_order_relaxed) == 2);
}
static bool plan_runs(unsigned nchan, long long n, unsigned max_waves, FirRuns& rn, unsigned long long kChunk,
                      unsigned long long fixed_run = 0, unsigned* tickets = nullptr) {
    constexpr unsigned long long kRunsPerWave = 8, kMinRun = 32, kLeadPerWave = 6, kTailDiv = 8, kFewChannels = 16;
    const unsigned long long cpc = ((unsigned long long)n + kChunk - 1) / kChunk, total_chunks = cpc * nchan;
    unsigned long long want = (total_chunks + kRunsPerWave * max_waves - 1) / (kRunsPerWave * max_waves);
    if (want < kMinRun) want = std::min(kMinRun, std::max<unsigned long long>(1, total_chunks / max_waves));
    if (fixed_run) want = fixed_run;   // (a kernel that carries nothing from chunk to chunk)
    const unsigned long long per_wave = total_chunks / max_waves;
    const bool drawn = tickets != nullptr && !fixed_run && large_enough_to_draw(nchan, n, max_waves, kChunk);
    unsigned long long lead_nchan = nchan, lead_chunks = cpc, tail_run = 0;
    if (drawn) {
        want = std::max(kMinRun, per_wave * 3 / 4 / kLeadPerWave);
        tail_run = std::max<unsigned long long>(4, std::min(want, cpc) / kTailDiv);
        if (nchan < kFewChannels) {
            lead_chunks = cpc * 3 / 4 / std::min(want, cpc) * std::min(want, cpc);   // whole long runs
        } else {
            lead_nchan = (unsigned long long)nchan * 3 / 4;
        }
    }
    want = std::min(want, std::max<unsigned long long>(lead_chunks, 1));
    rn = FirRuns{};
    rn.decks = 1;
    rn.chunks_per_chan = (unsigned)cpc;
    rn.lead_chunks = (unsigned)lead_chunks;
    rn.runs_per_chan = (unsigned)((lead_chunks + want - 1) / want);
    rn.run_chunks = rn.runs_per_chan ? (unsigned)((lead_chunks + rn.runs_per_chan - 1) / rn.runs_per_chan) : 1;
    rn.runs_per_chan = (unsigned)((lead_chunks + rn.run_chunks - 1) / rn.run_chunks);
    const unsigned long long lead_runs = lead_nchan * rn.runs_per_chan;
    unsigned long long tail_runs = 0;
    FirShape d{};
    if (drawn) {
        rn.tail_chan0 = nchan < kFewChannels ? 0u : (unsigned)lead_nchan;
        rn.tail_c0 = nchan < kFewChannels ? (unsigned)lead_chunks : 0u;
        const unsigned long long tail_chunks = cpc - rn.tail_c0;
        rn.tail_run_chunks = (unsigned)std::min(tail_run, std::max<unsigned long long>(tail_chunks, 1));
        rn.tail_runs_per_chan = (unsigned)((tail_chunks + rn.tail_run_chunks - 1) / rn.tail_run_chunks);
        tail_runs = (unsigned long long)(nchan - rn.tail_chan0) * rn.tail_runs_per_chan;
        set_divider(d, std::max(rn.tail_runs_per_chan, 1u));
        rn.tail_div_magic = d.div_magic, rn.tail_div_shift = d.div_shift;
        rn.tickets = tickets;
    }
    const unsigned long long total_runs = lead_runs + tail_runs;
    if (cpc > 0x7fffffffull || total_runs >= (1ull << 31)) return false;  // (the dividers cover 31 bits)
    rn.lead_runs = (unsigned)lead_runs;
    rn.total_runs = (unsigned)total_runs;
    set_divider(d, std::max(rn.runs_per_chan, 1u));
    rn.div_magic = d.div_magic, rn.div_shift = d.div_shift;
    return true;
}

// Counters for launches whose runs are drawn (FirRuns::tickets): one set per stream, so that two launches that may
// run at the same time never share one, while the launches of one stream, which run one after the other, each find
// the zeros the launch before them left.  A device has kTicketSlots sets.  A stream keeps its set until
// forget_stream() gives it back (the library does that for the streams it creates itself, before it destroys them);
// when all sets are taken, further streams get none and their launches use fixed shares.  Launches that are being
// captured into a graph always do: a graph may be replayed on any stream.
constexpr int kTicketSlots = 64;
__device__ unsigned g_run_tickets[kTicketSlots][kMaxRunDecks * kTicketStride];

struct TicketTable {
    unsigned* base = nullptr;
    bool looked_up = false;
    bool taken[kTicketSlots] = {};
    hipStream_t streams[kTicketSlots] = {};
};
static std::mutex g_ticket_mutex;
static TicketTable g_ticket_tables[64];   // per device

static unsigned* ticket_set(hipStream_t stream) {
    if (g_share_mode.load(std::memory_order_relaxed) == 1) return nullptr;
    if (stream == hipStreamPerThread || stream == hipStreamLegacy) return nullptr;   // (handles that name several streams)
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capture) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (capture != hipStreamCaptureStatusNone) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_ticket_mutex);
    TicketTable& d = g_ticket_tables[dev];
    if (!d.looked_up) {
        d.looked_up = true;
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_run_tickets)) == hipSuccess)
            d.base = static_cast<unsigned*>(p);
        else
            (void)hipGetLastError();
    }
    if (d.base == nullptr) return nullptr;
    int slot = -1, vacant = -1;
    for (int i = 0; i < kTicketSlots && slot < 0; ++i) {
        if (d.taken[i] && d.streams[i] == stream) slot = i;
        if (!d.taken[i] && vacant < 0) vacant = i;
    }
    if (slot < 0) slot = vacant;
    if (slot < 0) return nullptr;
    d.taken[slot] = true, d.streams[slot] = stream;
    return d.base + (size_t)slot * kMaxRunDecks * kTicketStride;
}

void forget_stream(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    std::lock_guard<std::mutex> lock(g_ticket_mutex);
    TicketTable& d = g_ticket_tables[dev];
    for (int i = 0; i < kTicketSlots; ++i)
        if (d.taken[i] && d.streams[i] == stream) d.taken[i] = false;
}

// Drawn chunks for the several-windows kernel (runs of one chunk each): takes the stream's counters when the input is
// large enough, picks the number of counter groups from the expected draw rate and makes the grid a multiple of 8 per
// group.  `ntaps`: the (mean) filter length, for the estimate of the launch's duration.
static void draw_chunks(FirRuns& rn, unsigned& wgs, unsigned nchan, long long n, double ntaps, unsigned* tickets) {
    constexpr unsigned kWaves = kBlock / 64;
    if (tickets == nullptr || wgs < 8 || rn.run_chunks != 1) return;
    // The last eighth of every channel's chunks is the pool the waves draw from; what comes before is dealt in
    // stretches of one chunk per wave, XCD by XCD.  Worth it from four pool chunks per wave (two of them are fixed)
    // and one stretch before them.
    const unsigned cpc = rn.chunks_per_chan, lead_chunks = (unsigned)((unsigned long long)cpc * 7 / 8), tail_chunks = cpc - lead_chunks;
    const double outputs = (double)n * nchan;
    const double micros = std::max(outputs * 2.0 * ntaps / 60e6, outputs * 16.0 / 5e6);
    const unsigned long long pool = (unsigned long long)nchan * tail_chunks, lead = (unsigned long long)nchan * lead_chunks;
    unsigned decks = 1;
    while (decks < kMaxRunDecks && pool > 10.0 * (micros / 8) * decks) decks *= 2;   // (the pool's draws fall into the launch's last eighth)
    while (decks > 1 && wgs < 8 * decks) decks /= 2;
    const unsigned grid = std::max(wgs / (8 * decks) * (8 * decks), 8u), waves = grid * kWaves;
    if (lead_chunks == 0 || tail_chunks == 0 || lead < waves || pool < 4ull * waves) return;
    wgs = grid;
    rn.decks = decks;
    rn.tickets = tickets;
    rn.runs_per_chan = lead_chunks, rn.lead_chunks = lead_chunks, rn.lead_runs = (unsigned)lead;
    rn.tail_chan0 = 0, rn.tail_c0 = lead_chunks, rn.tail_runs_per_chan = tail_chunks, rn.tail_run_chunks = 1;
    FirShape d{};
    set_divider(d, lead_chunks);
    rn.div_magic = d.div_magic, rn.div_shift = d.div_shift;
    set_divider(d, tail_chunks);
    rn.tail_div_magic = d.div_magic, rn.tail_div_shift = d.div_shift;
}

// The several-windows kernel as the wavelet bank runs it: 1024-value chunks, 576 halo slots (143 steps per window).
constexpr int kWaveBankHG = 9, kWaveBankOB = 4;

template <bool BANK>
static void launch_mfma_variant(const double* taps, const double* x, const double* hist, double* y, const BankRow* bank,
                                const FirShape& shape, bool vec2, bool tiled, hipStream_t stream) {
    // persistent workgroups: a few per CU, in multiples of 8 (one share per XCD), never more than there are tiles
    using G = M4Geo<kMatrixOB>;
    const unsigned per_xcd = std::min((shape.total_tiles + 7) / 8, compute_units() * G::kResident / 8);
    const dim3 grid(8 * std::max(per_xcd, 1u)), block(kBlock);
    const bool single = !BANK && shape.ntaps > 0 && (shape.ntaps + 2) / 4 + 1 <= G::kStageSteps;  // one window per tile
    if constexpr (BANK) {
        // The wave-autonomous form: a unit of work is a 1022-output chunk of one row; the positions of the tiled form's
        // walk (rows * 8 runs of tiles) give the number of rows back.
        const unsigned tile_run = (shape.tiles_per_chan + 7) / 8;
        const unsigned rows = tile_run ? shape.total_tiles / (8 * tile_run) : 0;
        constexpr unsigned kWaves = kBlock / 64;
        const unsigned max_waves = compute_units() * 2u * kWaves;
        FirRuns rn;
        if (!tiled && rows > 0 && plan_runs(rows, shape.n, max_waves, rn, M4Geo<kWaveBankOB>::kWaveOut - 2, 1)) {
            unsigned wgs = (std::min((rn.total_runs + kWaves - 1) / kWaves, max_waves / kWaves) + 7) / 8 * 8;
            if (large_enough_to_draw(rows, shape.n, max_waves, M4Geo<kWaveBankOB>::kWaveOut - 2, kDrawnChunksFrom))
                draw_chunks(rn, wgs, rows, shape.n, 550.0, ticket_set(stream));
            if (vec2)
                hipLaunchKernelGGL((fir_f64_mfma4wm_kernel<2, kWaveBankHG, kWaveBankOB, true, true>), dim3(wgs), block, 0, stream, taps, x, hist, y, bank, shape, rn);
            else
                hipLaunchKernelGGL((fir_f64_mfma4wm_kernel<2, kWaveBankHG, kWaveBankOB, true, false>), dim3(wgs), block, 0, stream, taps, x, hist, y, bank, shape, rn);
            return;
        }
    }
    if constexpr (!BANK) {
        if (vec2 && !tiled && shape.ntaps > 0) {
            // The wave-autonomous forms (measured on one box, tools/tap_sweep.py and same-box A/B): up to 77 taps 512-output
            // chunks with a 288-slot halo, four workgroups per CU; up to 569 taps 1024-output chunks (half the LDS reads
            // per matrix instruction, half the chunk hand-overs) with a 320-, 448- or 576-slot halo, three, three or two
            // workgroups per CU; beyond that several windows per 1024-output chunk, each with a 576-slot halo (143 steps; 512-output
            // chunks with windows of 71 steps measured 1-2 % slower from 570 to 4095 taps).
            constexpr unsigned kWaves = kBlock / 64;
            // Steps of four tap offsets from which the 1024-output chunks win: 78 taps, with three workgroups per CU
            // (+0.4...1.5 % from 80 to 176 taps, +1 % on a single stream of 2^28 samples at 101 taps; -0.5 % at 63 taps,
            // -5 % there on a single stream.  With two workgroups per CU they won from 190 taps only).
            constexpr int kBigFrom = 21;
            const int steps = (shape.ntaps + 2) / 4 + 1;
            const bool big = steps >= kBigFrom && steps <= 2 * M4Geo<2>::kStageSteps + 1;
            const bool several = steps > 2 * M4Geo<2>::kStageSteps + 1;   // several windows per chunk: runs of one chunk
            // Workgroups per CU: the 512-output chunks four (a fifth: -1 % at 63 taps); the 1024-output chunks three
            // where the halo has 320 or 448 slots (up to 313 / 441 taps: +1 % over two), else two, as many as a CU's LDS
            // holds.
            constexpr int kMidSteps = 7 * M4Geo<4>::kSeg / 4 - 1;   // steps a 448-slot halo serves (111)
            const bool mid = big && steps <= kMidSteps;
            const unsigned max_waves = compute_units() * (mid ? 3u : (big || several) ? 2u : 4u) * kWaves;
            FirRuns rn;
            const unsigned nchan = shape.total_tiles / shape.tiles_per_chan, chunk = (big || several) ? 1024 : 512;
            // The several-windows kernel has a pool of drawn chunks only for filters of up to kDrawnChunkTaps taps: beyond,
            // the chip sits at its full clock on every XCD and there is nothing to balance.
            constexpr int kDrawnChunkTaps = 3072;
            const bool may_draw = large_enough_to_draw(nchan, shape.n, max_waves, chunk, several ? kDrawnChunksFrom : big ? kDrawnFromBig : kDrawnFrom) &&
                                  (!several || shape.ntaps <= kDrawnChunkTaps);
            unsigned* const tickets = may_draw ? ticket_set(stream) : nullptr;
            if (plan_runs(nchan, shape.n, max_waves, rn, chunk, several ? 1 : 0, several ? nullptr : tickets)) {
                unsigned wgs = std::min((rn.total_runs + kWaves - 1) / kWaves, max_waves / kWaves);
                if (several) wgs = (wgs + 7) / 8 * 8;   // (its waves are addressed by XCD)
                if (several && tickets != nullptr) draw_chunks(rn, wgs, nchan, shape.n, (double)shape.ntaps, tickets);
                if (rn.tickets != nullptr && !several) {
                    // how long the launch will take, roughly (60 TFLOP/s, 5 TB/s), and the draws per microsecond that makes
                    const double outputs = (double)shape.n * nchan;
                    const double micros = std::max(outputs * 2.0 * shape.ntaps / 60e6, outputs * 16.0 / 5e6);
                    constexpr double kDrawsPerMicro = 10.0;   // per counter
                    rn.decks = 1;
                    while (rn.decks < kMaxRunDecks && rn.total_runs > kDrawsPerMicro * micros * rn.decks) rn.decks *= 2;
                    while (rn.decks > 1 && wgs < 8 * rn.decks) rn.decks /= 2;   // (a group has workgroups of every XCD)
                    wgs = std::max(wgs / (8 * rn.decks) * (8 * rn.decks), 8u);   // (groups of equal size)
                }
                const dim3 wgrid(wgs);
                if (steps < kBigFrom)
                    hipLaunchKernelGGL((fir_f64_mfma4w_kernel<4, 9, 2>), wgrid, block, 0, stream, taps, x, hist, y, shape, rn);
                else if (steps <= M4Geo<4>::kStageSteps)
                    hipLaunchKernelGGL((fir_f64_mfma4w_kernel<3, 5, 4>), wgrid, block, 0, stream, taps, x, hist, y, shape, rn);
                else if (steps <= kMidSteps)
                    hipLaunchKernelGGL((fir_f64_mfma4w_kernel<3, 7, 4>), wgrid, block, 0, stream, taps, x, hist, y, shape, rn);
                else if (big)
                    hipLaunchKernelGGL((fir_f64_mfma4w_kernel<2, 9, 4>), wgrid, block, 0, stream, taps, x, hist, y, shape, rn);
                else
                    hipLaunchKernelGGL((fir_f64_mfma4wm_kernel<2, kWaveBankHG, kWaveBankOB, false, true>), wgrid, block, 0, stream, taps, x, hist, y, bank, shape, rn);
                return;
            }
        }
        if (single) {
            if (vec2)
                hipLaunchKernelGGL((fir_f64_mfma4_kernel<kMatrixOB, true, false, true>), grid, block, 0, stream, taps, x, hist, y, bank, shape);
            else
                hipLaunchKernelGGL((fir_f64_mfma4_kernel<kMatrixOB, false, false, true>), grid, block, 0, stream, taps, x, hist, y, bank, shape);
            return;
        }
    }
    if (vec2)
        hipLaunchKernelGGL((fir_f64_mfma4_kernel<kMatrixOB, true, BANK, false>), grid, block, 0, stream, taps, x, hist, y, bank, shape);
    else
        hipLaunchKernelGGL((fir_f64_mfma4_kernel<kMatrixOB, false, BANK, false>), grid, block, 0, stream, taps, x, hist, y, bank, shape);
}

hipError_t launch_fir_mfma_f64(const double* taps, const double* x, const double* hist, double* y,
                               const FirShape& shape, bool vec2, bool tiled, hipStream_t stream) {
    if (shape.total_tiles == 0) return hipSuccess;
    launch_mfma_variant<false>(taps, x, hist, y, nullptr, shape, vec2, tiled, stream);
    return hipGetLastError();
}

hipError_t launch_fir_mfma_bank_f64(const double* taps, const BankRow* bank, const double* x, double* y,
                                    const FirShape& shape, bool vec2, bool tiled, hipStream_t stream) {
    if (shape.total_tiles == 0) return hipSuccess;
    launch_mfma_variant<true>(taps, x, nullptr, y, bank, shape, vec2, tiled, stream);
    return hipGetLastError();
}

int fir_mfma_window_steps(int ntaps) {
    constexpr int kStageSteps = kWaveBankHG * M4Geo<kWaveBankOB>::kSeg / 4 - 1;   // what a window of the bank's kernel serves
    const int total = (ntaps + 2) / 4 + 1;
    const int nstages = (total + kStageSteps - 1) / kStageSteps;
    return (total + nstages - 1) / nstages;
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

template <typename TIN, typename TOUT>
static hipError_t launch_fir_narrow(const TIN* taps, const TIN* x, const TIN* hist, TOUT* y, const FirShape& shape,
                                    hipStream_t stream) {
    if (shape.total_tiles == 0) return hipSuccess;
    if (!grid_fits(shape.total_tiles)) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL((fir_narrow_kernel<TIN, TOUT>), dim3(shape.total_tiles), dim3(kBlock), 0, stream, taps, x, hist, y, shape);
    return hipGetLastError();
}
hipError_t launch_fir_f32(const float* taps, const float* x, const float* hist, float* y, const FirShape& shape,
                          hipStream_t stream) {
    return launch_fir_narrow<float, float>(taps, x, hist, y, shape, stream);
}
hipError_t launch_fir_f32_from_f64(const double* taps, const double* x, const double* hist, float* y,
                                   const FirShape& shape, hipStream_t stream) {
    return launch_fir_narrow<double, float>(taps, x, hist, y, shape, stream);
}
hipError_t launch_fir_f64_from_f32(const float* taps, const float* x, const float* hist, double* y,
                                   const FirShape& shape, hipStream_t stream) {
    return launch_fir_narrow<float, double>(taps, x, hist, y, shape, stream);
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

// Test hooks (not part of the C ABI, include/dspx.h does not declare them; results do not depend on them).
// dspx_test_run_shares: 1 makes every launch of the wave-autonomous kernels use fixed shares of runs per wave, as
// launches under stream capture do; 2 makes them draw their runs from the stream's counter whatever the input's
// size (by default only inputs of 128 chunks per wave or more do); 0 restores the default.  Returns the previous
// setting.  dspx_test_forget_stream: gives an idle stream's counters back (what the library does for its own
// streams before it destroys them).  dspx_test_run_plan: the runs plan_runs() cuts an input into, decoded as the kernels decode them
// (chan, first chunk, end chunk per run; `drawn` 0: fixed shares, 1: drawn runs, 2: the several-windows kernel's
// stretches and pool; *lead_runs = the runs before the short runs / before the pool); returns the number of runs, or
// -1 when the plan does not fit `cap`.
extern "C" __attribute__((visibility("default"))) int dspx_test_run_shares(int mode) {
    return dspx::g_share_mode.exchange(mode);
}
extern "C" __attribute__((visibility("default"))) void dspx_test_forget_stream(void* stream) {
    dspx::forget_stream(static_cast<hipStream_t>(stream));
}
extern "C" __attribute__((visibility("default"))) long long dspx_test_run_plan(unsigned nchan, long long n, unsigned max_waves,
                                                                            unsigned chunk, int drawn, unsigned* spans,
                                                                            size_t cap, unsigned* lead_runs) {
    dspx::FirRuns rn;
    unsigned dummy[1];
    if (drawn == 2) {   // the several-windows kernel's plan: runs of one chunk, the last eighth of every channel a pool
        if (!dspx::plan_runs(nchan, n, max_waves, rn, chunk, 1, nullptr)) return -1;
        unsigned wgs = max_waves / 4;
        dspx::draw_chunks(rn, wgs, nchan, n, 1000.0, dummy);
    } else if (!dspx::plan_runs(nchan, n, max_waves, rn, chunk, 0, drawn ? dummy : nullptr)) {
        return -1;
    }
    if (rn.total_runs > cap) return -1;
    for (unsigned r = 0; r < rn.total_runs; ++r) {
        const dspx::RunSpan sp = dspx::run_span(rn, r);
        spans[3 * r] = sp.chan, spans[3 * r + 1] = sp.c0, spans[3 * r + 2] = sp.c1;
    }
    if (lead_runs) *lead_runs = rn.tickets ? rn.lead_runs : rn.total_runs;
    return rn.total_runs;
}

#ifdef DSPX_M4W_TIMING
extern "C" __attribute__((visibility("default"))) int dspx_debug_m4w_timing(unsigned long long* out, size_t count) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dspx::g_m4w_timing), count * sizeof(unsigned long long));
}
#endif

