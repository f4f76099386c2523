/*
 * dspx — C ABI of the MI355X (gfx950) FIR engine behind dsp::lfilter(b, 1.0, x), with the recursive tail
 * of dsp::lfilter(b, a, x) and the per-scale convolution bank of dsp::cwt on the same kernels.
 *
 * The reference library (Unbinilium/DSP, dsp/include/dsp.hpp) is a header-only C++
 * template library with no FFI of its own; the interface this ABI replaces is the
 * body of its lfilter template.  Each entry point cites the reference lines whose
 * behaviour it reproduces.  Plain pointers and sizes only; `stream` arguments are a
 * hipStream_t passed as void* (NULL = the default stream).
 *
 * All functions return DSPX_OK (0) or a negative dspx_status; dspx_last_error()
 * gives a thread-local message for the most recent failure on the calling thread.
 * The entry points may be called concurrently from several host threads.
 */
#ifndef DSPX_H
#define DSPX_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define DSPX_API __attribute__((visibility("default")))
#else
#define DSPX_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef enum dspx_status {
    DSPX_OK = 0,
    DSPX_ERR_INVALID = -1,   /* bad pointer / size / stride / overlapping buffers */
    DSPX_ERR_NO_DEVICE = -2, /* no usable gfx950 device */
    DSPX_ERR_HIP = -3,       /* a HIP runtime call failed; see dspx_last_error() */
    DSPX_ERR_NOMEM = -4,     /* a device or host allocation failed */
    DSPX_ERR_DOMAIN = -5,    /* where the reference throws std::runtime_error (e.g. a wavelet scale too large) */
    DSPX_ERR_HOST = -6       /* a host-side failure inside the library (e.g. a staging thread could not start) */
} dspx_status;

/* Arithmetic mode of the filter kernels (the `flags` argument). */
#define DSPX_FIR_FMA 0u    /* default: fused multiply-adds, the library picks the pipe: the FP64 matrix
                              pipe (Toeplitz-block v_mfma_f64_4x4x4_4b_f64) for filters of 52 taps or more,
                              the vector pipe (v_fma_f64, taps strictly ascending) below                     */
#define DSPX_FIR_EXACT 1u  /* separately rounded multiply then add per tap, taps ascending, on the vector
                              pipe: bit-identical to the reference loop compiled without FMA contraction
                              (dsp.hpp:361-364); about half the arithmetic rate                              */
#define DSPX_FIR_MFMA 2u   /* force the matrix pipe (ignored together with DSPX_FIR_EXACT)                   */
#define DSPX_FIR_VECTOR 4u /* force the vector pipe                                                          */
#define DSPX_FIR_TILED 8u  /* matrix pipe only: run the workgroup-tiled form of the kernel also where the
                              wave-autonomous forms are the default (filters over 16-byte aligned buffers at any
                              tap count; wavelet banks).  Filters: same results bit for bit.  Wavelet banks: the
                              forms cut a row into different chunks, so their results differ in rounding.  Kept
                              so that the forms can be measured against each other                           */

DSPX_API int dspx_version(void);
/* Number of visible HIP devices (0 if none or the runtime is unusable). */
DSPX_API int dspx_device_count(void);
/* The host-pointer entry points keep their staging (pinned host buffers, device buffers, streams: up to two
   idle sets per device, each sized by the largest call it served) between calls, as the reference keeps
   lfilter_ctx_t::result (dsp.hpp:333, 348-350).  This frees the idle sets; returns how many were freed. */
DSPX_API int dspx_release_staging(void);
DSPX_API const char* dspx_last_error(void);
DSPX_API const char* dspx_status_string(int status);

/* ---- host tap design (no GPU work) -------------------------------------------------
 * C entry points into the library's own instantiation of the header templates in
 * dsp.hpp, so that non-C++ callers get taps that are bit-identical to
 *   firwin<double>(numtaps, cutoffs, pass_zero, scale, HAMMING)   dsp.hpp:190-304
 *   hamming_window<double>(width, alpha, sym)                      dsp.hpp:74-115
 *   sinc<double>(x)                                                dsp.hpp:60-67
 * Arguments the reference throws std::invalid_argument for return DSPX_ERR_INVALID
 * with the reference's message in dspx_last_error().  `out` holds numtaps / width doubles. */
DSPX_API int dspx_firwin_f64(size_t numtaps, const double* cutoffs, size_t ncutoffs, int pass_zero, int scale,
                             double* out);
DSPX_API int dspx_hamming_window_f64(size_t width, double alpha, int sym, double* out);
DSPX_API double dspx_sinc_f64(double x);

/* ---- host-pointer entry points --------------------------------------------------
 * y[i] = sum_{j < min(ntaps, i+1)} taps[j] * x[i-j], taps ascending, zero initial
 * state, output length == input length: the FIR branch (denominator of size 1) of
 *   lfilter(ctx, numerators, denominators, x)            dsp.hpp:342-371
 * reached through lfilter(b, T den, x)                    dsp.hpp:393-396
 * (the value of `den` is never read by the reference, dsp.hpp:366).
 * n == 0 is a no-op; ntaps == 0 writes zeros (the reference's inner loop never runs).
 * x and y must not overlap.  Runs on the current HIP device. */
DSPX_API int dspx_fir_f64(const double* taps, size_t ntaps, const double* x, double* y, size_t n, unsigned flags);

/* nchan independent channels (one reference lfilter call each), channel c at
 * x + c*stride / y + c*stride, stride >= n, all with the same taps. */
DSPX_API int dspx_fir_f64_batched(const double* taps, size_t ntaps, const double* x, double* y, size_t nchan, size_t n,
                         size_t stride, unsigned flags);

/* Same contract, sharded over the first `ngpus` devices (0 = all): contiguous channel
 * ranges when nchan >= ngpus, otherwise contiguous segments of each channel with the
 * preceding ntaps-1 input samples as history.  No inter-device exchange; results are
 * bit-identical for every ngpus.  One staging pipeline per device (pinned buffers, a few copy threads). */
DSPX_API int dspx_fir_f64_multi(const double* taps, size_t ntaps, const double* x, double* y, size_t nchan, size_t n,
                       size_t stride, int ngpus, unsigned flags);

/* Streaming form (no counterpart in the reference, whose lfilter keeps no state between calls,
 * dsp.hpp:356-365): `hist` (nchan x (ntaps-1) doubles, channel c at hist + c*(ntaps-1)) holds the
 * samples that precede x; on return it holds the last ntaps-1 samples of [hist | x], ready for the
 * next chunk.  Runs on the vector pipe unless DSPX_FIR_MFMA is given, so that chunked calls reproduce
 * one long vector-pipe (DSPX_FIR_VECTOR or DSPX_FIR_EXACT) call bit for bit wherever the chunks are cut;
 * on the matrix pipe that holds for chunk lengths that are multiples of 4. */
DSPX_API int dspx_fir_f64_stream(const double* taps, size_t ntaps, double* hist, const double* x, double* y,
                                 size_t nchan, size_t n, size_t stride, unsigned flags);

/* dsp::lfilter with a recursive part (dsp.hpp:342-371 including the tail 366-368):
 *     y[i] = sum_{j < min(nb, i+1)} b[j] * x[i-j]  -  sum_{1 <= j < min(na, i+1)} a[j] * y[i-j]
 * per channel, zero initial state; a[0] is never read (the reference does not normalise by it).  Host
 * pointers, current device, layout as dspx_fir_f64_batched.  na <= 1 is the plain FIR; na may be at most 9.
 * DSPX_FIR_EXACT: the tail runs one lane per channel in sample order with separately rounded products and
 * subtractions, bit-identical to the reference (slow for few channels: the loop is sequential by nature).
 * Default: block-parallel evaluation (per-lane runs joined by a scan over powers of the recurrence's
 * companion matrix) with fused multiply-adds; its rounding differs from the sequential loop's, so parity is
 * to tolerance, and it presumes a filter whose response does not grow without bound -- filters whose matrix
 * powers overflow run on the sequential kernel instead. */
DSPX_API int dspx_lfilter_f64(const double* b, size_t nb, const double* a, size_t na, const double* x, double* y,
                              size_t nchan, size_t n, size_t stride, unsigned flags);

/* Streaming form of dspx_lfilter_f64 (no counterpart in the reference): xhist (nchan x (nb-1)) and yhist
 * (nchan x (na-1)), oldest first, hold the samples and outputs that precede this chunk and are updated for
 * the next one.  The FIR part runs on the vector pipe unless DSPX_FIR_MFMA is given (as dspx_fir_f64_stream).
 * With DSPX_FIR_EXACT chunked calls reproduce one long call bit for bit; in the default mode they agree to
 * rounding (the block-parallel tail rounds by position inside a call). */
DSPX_API int dspx_lfilter_f64_stream(const double* b, size_t nb, const double* a, size_t na, double* xhist,
                                     double* yhist, const double* x, double* y, size_t nchan, size_t n,
                                     size_t stride, unsigned flags);

/* Same contract, sharded over the first `ngpus` devices (0 = all) by contiguous channel ranges.  With a
 * recursive part (na > 1) a channel is never split across devices (its state is a serial dependence), so at
 * most nchan devices work; without one this is dspx_fir_f64_multi. */
DSPX_API int dspx_lfilter_f64_multi(const double* b, size_t nb, const double* a, size_t na, const double* x, double* y,
                                    size_t nchan, size_t n, size_t stride, int ngpus, unsigned flags);

/* Device-resident form: dspx_fir_f64_dev followed by the recursive tail, in place on d_y.  `a` is a HOST
 * pointer (na coefficients, read during the call).  d_xhist as d_hist of dspx_fir_f64_dev; d_yhist (optional):
 * per channel the na-1 outputs that precede y[0], oldest first (channel c at d_yhist + c*yhist_stride).
 * d_work: device scratch of dspx_lfilter_workspace_bytes(na, nchan, n) bytes, 256-byte aligned, needed by the
 * default (block-parallel) tail; it may be reused by later calls on the same stream. */
DSPX_API size_t dspx_lfilter_workspace_bytes(size_t na, size_t nchan, size_t n);
DSPX_API int dspx_lfilter_f64_dev(const double* d_b, size_t nb, const double* a, size_t na, const double* d_x,
                                  size_t x_stride, const double* d_xhist, size_t xhist_stride, const double* d_yhist,
                                  size_t yhist_stride, double* d_y, size_t y_stride, size_t nchan, size_t n,
                                  void* d_work, size_t work_bytes, unsigned flags, void* stream);

/* dspx_lfilter_f64_dev for callers that cannot allocate device memory or bring a workspace (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same
 * arguments, kernels and results (bit for bit) as dspx_lfilter_f64_dev, minus d_work / work_bytes: the tail's scratch
 * (matrix powers, one state per 4096-sample tile of every channel) lives in a DSPX_NOALLOC_SCRATCH_BYTES array inside
 * the library's device code, one copy per GPU for the life of the process.  The call itself makes no hipMalloc /
 * hipFree and uses no allocating library routine.  A call whose tail needs more (dspx_lfilter_workspace_bytes(na,
 * nchan, n) > DSPX_NOALLOC_SCRATCH_BYTES; launches of 1024 channels or more need only 20 KiB whatever n is) returns
 * DSPX_ERR_INVALID before anything is launched.  All calls on one GPU share that GPU's array: calls that overlap in
 * time on one GPU (different streams or threads) race; calls on one stream, or on different GPUs, do not.  Nothing in
 * the array has to start at zero (every call rewrites what it reads).  The FIRST call on each GPU looks the array's
 * device address up (hipGetSymbolAddress, which may load the code object): make one warm-up call per GPU outside any
 * stream capture; every later call on that GPU only checks its arguments and launches kernels. */
#define DSPX_NOALLOC_SCRATCH_BYTES ((size_t)16 << 20)
DSPX_API int dspx_lfilter_f64_dev_noalloc(const double* d_b, size_t nb, const double* a, size_t na, const double* d_x,
                                          size_t x_stride, const double* d_xhist, size_t xhist_stride,
                                          const double* d_yhist, size_t yhist_stride, double* d_y, size_t y_stride,
                                          size_t nchan, size_t n, unsigned flags, void* stream);

/* The sharding plan dspx_fir_f64_multi executes, exposed so that it can be inspected and
 * tested without GPUs: piece i covers channels [chan, chan + nchan) x samples
 * [start, start + len) and is assigned to `device`; the pieces of one device are staged
 * through HBM in order.  A piece with start > 0 is filtered with the min(start, ntaps-1)
 * samples before it as history (zeros before the channel start).  Writes at most `cap`
 * pieces to `out` and returns the total number of pieces (or a negative dspx_status). */
typedef struct dspx_piece {
    int device;
    size_t chan, nchan, start, len;
} dspx_piece;
DSPX_API long dspx_fir_plan(size_t nchan, size_t n, int ngpus, dspx_piece* out, size_t cap);

/* ---- device-resident entry point ------------------------------------------------
 * All pointers are device pointers on the current device; the launch is asynchronous
 * on `stream`.  d_hist, when non-NULL, holds for every channel c the ntaps-1 samples
 * that precede d_x[c*x_stride] (d_hist[c*hist_stride + ntaps-2] is x[-1]); NULL is the
 * reference's zero initial state.  History has no counterpart in the reference (its
 * lfilter keeps no state, dsp.hpp:356-365); it is what makes segment sharding and
 * chunked streaming produce the same samples as one long call.
 * Streams: any number may be in use at once.  For large inputs the matrix-pipe kernels hand their work out through
 * a few device counters that belong to the launch's stream (64 sets per GPU inside the library, left at zero by every
 * launch); hipStreamPerThread, hipStreamLegacy, launches recorded under stream capture and streams beyond the 64th
 * get fixed shares of work per wave instead -- a few per cent slower on batched inputs, the same results. */
DSPX_API int dspx_fir_f64_dev(const double* d_taps, size_t ntaps, const double* d_x, size_t x_stride, const double* d_hist,
                     size_t hist_stride, double* d_y, size_t y_stride, size_t nchan, size_t n, unsigned flags,
                     void* stream);

/* ---- single precision: lfilter<float>                                                dsp.hpp:333-396 with T = float
 * The reference sums in float and forms every product numerators[j] * x[i-j] in the type its operands promote to
 * (dsp.hpp:363): in float for containers of float (dspx_fir_f32), in double for containers of double
 * (dspx_fir_f32_from_f64 -- the form the reference's own test uses, dsp.hpp:1590-1597: double taps and samples, float
 * result).  Multiply, conversion and add are rounded separately, taps ascending: bit-identical to the reference loop
 * compiled without contraction.  Functional, not tuned (the tuned path is double).  Host pointers, current device;
 * the _dev forms take device pointers, nchan channels (channel c at + c*stride), an optional history of ntaps-1
 * samples per channel as dspx_fir_f64_dev, and a stream. */
DSPX_API int dspx_fir_f32(const float* taps, size_t ntaps, const float* x, float* y, size_t n);
DSPX_API int dspx_fir_f32_from_f64(const double* taps, size_t ntaps, const double* x, float* y, size_t n);
/* lfilter<double> over containers of float (dsp.hpp:361-364 with T = double): every product numerators[j] * x[i-j] is
 * formed in float, converted to double and added to a double sum, taps ascending -- bit-identical to the reference. */
DSPX_API int dspx_fir_f64_from_f32(const float* taps, size_t ntaps, const float* x, double* y, size_t n);
DSPX_API int dspx_fir_f64_from_f32_dev(const float* d_taps, size_t ntaps, const float* d_x, size_t x_stride,
                                       const float* d_hist, size_t hist_stride, double* d_y, size_t y_stride,
                                       size_t nchan, size_t n, void* stream);
/* lfilter<float> with a recursive part (dsp.hpp:356-369 with T = float): the FIR part as dspx_fir_f32 /
 * dspx_fir_f32_from_f64, then y_i -= a[j] * y[i-j], j = 1 .. min(na, i+1) - 1, in place (a[0] is never read; na may be
 * at most 9).  The reference evaluates that tail in the type denominators[j] * float promotes to: pass the
 * denominators as doubles (float ones convert exactly) and tail_in_double = 0 for float denominators (float products
 * and differences), 1 for double ones (double products and differences, converted to float).  Separately rounded
 * operations in the reference's order: bit-identical to it.  One lane per channel, samples in order: functional, not
 * fast (the tuned recursive path is dspx_lfilter_f64). */
DSPX_API int dspx_lfilter_f32(const float* b, size_t nb, const double* a, size_t na, int tail_in_double, const float* x,
                              float* y, size_t n);
DSPX_API int dspx_lfilter_f32_from_f64(const double* b, size_t nb, const double* a, size_t na, int tail_in_double,
                                       const double* x, float* y, size_t n);
DSPX_API int dspx_fir_f32_dev(const float* d_taps, size_t ntaps, const float* d_x, size_t x_stride, const float* d_hist,
                              size_t hist_stride, float* d_y, size_t y_stride, size_t nchan, size_t n, void* stream);
DSPX_API int dspx_fir_f32_from_f64_dev(const double* d_taps, size_t ntaps, const double* d_x, size_t x_stride,
                                       const double* d_hist, size_t hist_stride, float* d_y, size_t y_stride,
                                       size_t nchan, size_t n, void* stream);

/* ---- wavelet bank on the FIR kernel ---------------------------------------------------
 * integrate_wavelet<double>(MORLET, precision, lower, upper)   dsp.hpp:934-985   (host arithmetic)
 * psi and x each hold 2^precision doubles. */
DSPX_API int dspx_integrate_wavelet_morlet_f64(size_t precision, double lower, double upper, double* psi, double* x);

/* cwt(signal, scales, wavelet_psi, wavelet_x)                   dsp.hpp:1040-1239
 * `scales` may come in any order; like cwt_ctx_t's constructor (dsp.hpp:1008-1015) the call sorts
 * them ascending, and `result` (nscales x n, row-major, host memory) has one row per sorted scale.
 * Host side (same arithmetic as the reference): per scale the index table
 * floor(j / max(scale*x_step, EPS)) (dsp.hpp:1141-1150), the gathered, reversed wavelet as a tap
 * vector, -sqrt(scale), and the centre-crop offset floor((L-2)/2) (dsp.hpp:1194-1216).
 * GPU side: row r is the FIR of the zero-extended signal with its L_r taps, then
 * (c[i+1] - c[i]) * -sqrt(scale) (dsp.hpp:1185-1191), cropped -- all rows in one launch.
 * Errors: where the reference throws std::invalid_argument -> DSPX_ERR_INVALID; where it throws
 * std::runtime_error ("Selected scale is too large/small", no scales) -> DSPX_ERR_DOMAIN; the
 * reference's message is in dspx_last_error().  A table of a single tap, on which the reference
 * fails an assertion (dsp.hpp:1193), is reported as "too small". 
 * The result of a pass stays in device memory until it is downloaded; when all rows do not fit in half of the
 * free device memory they are transformed in several passes (the environment variable DSPX_CWT_ROWS_PER_PASS
 * caps the rows of a pass further).  Rows do not depend on how they are grouped. */
DSPX_API int dspx_cwt_f64(const double* signal, size_t n, const double* scales, size_t nscales, const double* psi,
                          size_t npsi, const double* wx, size_t nx, double* result, unsigned flags);
/* The same, with the scales sharded over the first `ngpus` devices (0 = all), balanced by the sum of
 * their tap counts; every device reads the whole signal and produces whole rows (no exchange). */
DSPX_API int dspx_cwt_f64_multi(const double* signal, size_t n, const double* scales, size_t nscales,
                                const double* psi, size_t npsi, const double* wx, size_t nx, double* result, int ngpus,
                                unsigned flags);
/* Which device dspx_cwt_f64_multi gives each sorted scale to, and its tap count (no GPU needed). */
DSPX_API int dspx_cwt_plan(const double* scales, size_t nscales, size_t npsi, const double* wx, size_t nx, int ngpus,
                           int* device_of_row, size_t* ntaps_of_row);

/* Device-resident use: the per-scale tables are built once and kept on the current device. */
typedef struct dspx_cwt_bank dspx_cwt_bank;
DSPX_API int dspx_cwt_bank_create(const double* scales, size_t nscales, const double* psi, size_t npsi,
                                  const double* wx, size_t nx, dspx_cwt_bank** bank);
DSPX_API size_t dspx_cwt_bank_rows(const dspx_cwt_bank* bank);
DSPX_API size_t dspx_cwt_bank_total_taps(const dspx_cwt_bank* bank);
/* d_result row r (r-th smallest scale) starts at d_result + r*row_stride; row_stride >= n. */
DSPX_API int dspx_cwt_bank_apply_dev(const dspx_cwt_bank* bank, const double* d_signal, size_t n, double* d_result,
                                     size_t row_stride, unsigned flags, void* stream);
DSPX_API void dspx_cwt_bank_destroy(dspx_cwt_bank* bank);

/* ---- optional exchange step: all-gather of result shards over RCCL ---------------------------------
 * No counterpart in the reference (it has no devices).  The sharded filter needs no collective; callers that want
 * every GPU to hold the whole result gather the shards afterwards: one process per GPU, rank r contributes `count`
 * doubles at d_send and receives the shards of all ranks, in rank order, at d_recv (nranks * count doubles; d_send
 * may be d_recv + rank*count).  Contiguous channel ranges of equal size (dspx_fir_plan with nchan a multiple of the
 * device count) thus land in channel order.  RCCL is loaded at run time (the copy the process already holds, else
 * librccl.so.1).  Rank 0 makes an id (DSPX_COMM_ID_BYTES opaque bytes) and hands it to the other ranks by any
 * means (a file, a socket, MPI, torch.distributed); every rank then calls dspx_comm_create with its current device
 * set -- collectively, like ncclCommInitRank, which it wraps.  dspx_allgather_f64_dev is asynchronous on `stream`. */
#define DSPX_COMM_ID_BYTES 128
typedef struct dspx_comm dspx_comm;
DSPX_API int dspx_comm_unique_id(void* id_bytes);
DSPX_API int dspx_comm_create(const void* id_bytes, int nranks, int rank, dspx_comm** comm);
DSPX_API int dspx_comm_ranks(const dspx_comm* comm);
DSPX_API int dspx_allgather_f64_dev(dspx_comm* comm, const double* d_send, double* d_recv, size_t count, void* stream);
DSPX_API void dspx_comm_destroy(dspx_comm* comm);

/* ---- deterministic synthetic data and checksums (benchmark / parity support) -----
 * d_x[i] = ((mix64(seed + (offset+i+1)*0x9E3779B97F4A7C15) >> 11) * 2^-52) - 1.0,
 * mix64 = the SplitMix64 finaliser; uniform in [-1, 1), exactly representable. */
DSPX_API int dspx_fill_splitmix_f64_dev(double* d_x, size_t n, uint64_t seed, uint64_t offset, void* stream);
/* *d_sum (device uint64, overwritten) = wrapping sum over i of
 * mix64(bits(d_y[i]) + (offset+i+1)*0x9E3779B97F4A7C15).  Partition-independent. */
DSPX_API int dspx_checksum_f64_dev(const double* d_y, size_t n, uint64_t offset, uint64_t* d_sum, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSPX_H */
