// C ABI of the FIR engine (declared in include/dspx.h): argument checking, the
// device-resident launch, and the host-pointer paths that move caller memory through
// double-buffered device staging (one worker thread and stream pair per GPU).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "dsp.hpp"
#include "dspx.h"
#include "dspx_common.h"
#include "fir_kernels.h"

namespace dspx {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

}  // namespace dspx

namespace {

using dspx::aligned16;
using dspx::fail;
using dspx::g_last_error;

constexpr unsigned long long kMaxGrid = 0x7fffffffull;  // workgroups per launch

bool ranges_overlap(const void* a, size_t abytes, const void* b, size_t bbytes) {
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(a), b0 = reinterpret_cast<uintptr_t>(b);
    return a0 < b0 + bbytes && b0 < a0 + abytes;
}

// Launches the filter over device buffers, splitting the channel range if the grid
// would exceed the launch limit.
int launch_dev(const double* d_taps, size_t ntaps, const double* d_x, size_t x_stride, const double* d_hist,
               size_t hist_stride, double* d_y, size_t y_stride, size_t nchan, size_t n, unsigned flags,
               hipStream_t stream) {
    const bool mfma = dspx::use_matrix_pipe(flags, ntaps);
    const int ob = dspx::matrix_row_blocks(ntaps);
    const size_t tile = (size_t)(mfma ? dspx::fir_mfma_tile(ob) : dspx::fir_tile());
    const size_t tiles_per_chan = (n + tile - 1) / tile;
    if (tiles_per_chan > 0xffffffffull) return fail(DSPX_ERR_INVALID, "n too large for one launch: %zu", n);
    const bool vec2 = aligned16(d_x) && aligned16(d_y) && (x_stride % 2 == 0) && (y_stride % 2 == 0);
    const size_t chans_per_launch = std::max<size_t>(1, kMaxGrid / tiles_per_chan);
    if (tiles_per_chan > kMaxGrid) return fail(DSPX_ERR_INVALID, "n too large for one launch: %zu", n);
    for (size_t c0 = 0; c0 < nchan; c0 += chans_per_launch) {
        const size_t nc = std::min(chans_per_launch, nchan - c0);
        dspx::FirShape s;
        s.x_stride = x_stride;
        s.hist_stride = hist_stride;
        s.y_stride = y_stride;
        s.n = (long long)n;
        s.total_tiles = (unsigned)(nc * tiles_per_chan);
        s.tiles_per_chan = (unsigned)tiles_per_chan;
        s.ntaps = (int)ntaps;
        dspx::set_tile_divider(s);
        if (mfma)
            DSPX_HIP(dspx::launch_fir_mfma_f64(d_taps, d_x + c0 * x_stride,
                                               d_hist ? d_hist + c0 * hist_stride : nullptr, d_y + c0 * y_stride, s,
                                               ob, vec2, stream));
        else
            DSPX_HIP(dspx::launch_fir_f64(d_taps, d_x + c0 * x_stride, d_hist ? d_hist + c0 * hist_stride : nullptr,
                                          d_y + c0 * y_stride, s, (flags & DSPX_FIR_EXACT) != 0, vec2, stream));
    }
    return DSPX_OK;
}

int check_common(const double* taps, size_t ntaps, const void* x, const void* y, size_t nchan, size_t n,
                 unsigned flags) {
    if (int rc = dspx::check_flags(flags)) return rc;
    if (ntaps > 0x7fffffffull) return fail(DSPX_ERR_INVALID, "ntaps too large: %zu", ntaps);
    if (ntaps && !taps) return fail(DSPX_ERR_INVALID, "taps is NULL");
    if (nchan && n && (!x || !y)) return fail(DSPX_ERR_INVALID, "x or y is NULL");
    return DSPX_OK;
}

// ---------------------------------------------------------------------------------
// Host-pointer path.  The work of one device is a list of pieces; a piece is `cnt`
// channels starting at channel `c`, samples [s0, s0 + len) of each.  A piece is moved
// to the device as rows of  [ntaps-1 history samples | len samples]  so that pieces
// that start inside a channel see exactly the samples one long call would see.
struct Piece {
    size_t c, cnt, s0, len;
};

constexpr size_t kPieceSamples = size_t(1) << 24;  // samples per staged piece (128 MiB of input)

void cut_pieces(size_t c0, size_t c1, size_t s0, size_t s1, std::vector<Piece>& out) {
    const size_t len = s1 - s0;
    if (len == 0 || c1 == c0) return;
    if (len <= kPieceSamples) {
        const size_t per = std::max<size_t>(1, kPieceSamples / len);
        for (size_t c = c0; c < c1; c += per) out.push_back({c, std::min(per, c1 - c), s0, len});
    } else {
        for (size_t c = c0; c < c1; ++c)
            for (size_t s = s0; s < s1; s += kPieceSamples) out.push_back({c, 1, s, std::min(kPieceSamples, s1 - s)});
    }
}

// Work of each of G devices: contiguous channel ranges when there are at least G channels,
// otherwise contiguous segments of every channel.
std::vector<std::vector<Piece>> make_plan(size_t nchan, size_t n, int G) {
    std::vector<std::vector<Piece>> plan(G);
    if (nchan >= (size_t)G) {
        for (int g = 0; g < G; ++g) cut_pieces(nchan * g / G, nchan * (g + 1) / G, 0, n, plan[g]);
    } else {
        // segment starts are multiples of 4: the matrix-pipe kernel groups taps in fours by an output's
        // distance from the start of its call, so such cuts leave every output's arithmetic unchanged
        auto cut = [&](int g) { return g == G ? n : (n * g / G) & ~size_t(3); };
        for (int g = 0; g < G; ++g) cut_pieces(0, nchan, cut(g), cut(g + 1), plan[g]);
    }
    return plan;
}

struct Slot {
    double *d_in = nullptr, *d_out = nullptr;
    size_t in_cap = 0, out_cap = 0;
    hipStream_t stream = nullptr;
};

// `hist0`, when non-null, holds for every channel the ntaps-1 samples that precede x[0]
// (channel c at hist0 + c*(ntaps-1)); null means zeros.
int run_device(int device, const double* taps, size_t ntaps, const double* hist0, const double* x, double* y,
               size_t stride, const std::vector<Piece>& pieces, unsigned flags) {
    if (pieces.empty()) return DSPX_OK;
    DSPX_HIP(hipSetDevice(device));
    const size_t hl = ntaps ? ntaps - 1 : 0;
    double* d_taps = nullptr;
    Slot slots[2];
    int rc = DSPX_OK;
    auto body = [&]() -> int {
        if (ntaps) {
            DSPX_HIP(hipMalloc(&d_taps, ntaps * sizeof(double)));
            DSPX_HIP(hipMemcpy(d_taps, taps, ntaps * sizeof(double), hipMemcpyHostToDevice));
        }
        for (auto& s : slots) DSPX_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        for (size_t i = 0; i < pieces.size(); ++i) {
            const Piece& p = pieces[i];
            Slot& s = slots[i & 1];
            DSPX_HIP(hipStreamSynchronize(s.stream));  // the slot's previous piece has left the device
            // row pitch: history + samples, rounded to an even count so rows stay 16-byte aligned
            const size_t hl_pad = (hl + 1) & ~size_t(1);
            const size_t in_pitch = hl_pad + ((p.len + 1) & ~size_t(1));
            const size_t out_pitch = (p.len + 1) & ~size_t(1);
            if (s.in_cap < in_pitch * p.cnt) {
                if (s.d_in) DSPX_HIP(hipFree(s.d_in));
                s.d_in = nullptr;
                DSPX_HIP(hipMalloc(&s.d_in, in_pitch * p.cnt * sizeof(double)));
                s.in_cap = in_pitch * p.cnt;
            }
            if (s.out_cap < out_pitch * p.cnt) {
                if (s.d_out) DSPX_HIP(hipFree(s.d_out));
                s.d_out = nullptr;
                DSPX_HIP(hipMalloc(&s.d_out, out_pitch * p.cnt * sizeof(double)));
                s.out_cap = out_pitch * p.cnt;
            }
            // history: the lead samples that exist in the caller's buffer; before them the caller's
            // initial history (samples s0 .. hl-1 of it when lead == s0 < hl), or zeros
            const size_t lead = std::min(hl, p.s0);
            const double* d_hist = nullptr;
            if ((p.s0 > 0 || hist0) && hl > 0) {
                if (lead < hl) {
                    if (hist0)
                        DSPX_HIP(hipMemcpy2DAsync(s.d_in + (hl_pad - hl), in_pitch * sizeof(double),
                                                  hist0 + p.c * hl + p.s0, hl * sizeof(double),
                                                  (hl - lead) * sizeof(double), p.cnt, hipMemcpyHostToDevice,
                                                  s.stream));
                    else
                        DSPX_HIP(hipMemset2DAsync(s.d_in + (hl_pad - hl), in_pitch * sizeof(double), 0,
                                                  (hl - lead) * sizeof(double), p.cnt, s.stream));
                }
                if (lead)
                    DSPX_HIP(hipMemcpy2DAsync(s.d_in + (hl_pad - lead), in_pitch * sizeof(double),
                                              x + p.c * stride + (p.s0 - lead), stride * sizeof(double),
                                              lead * sizeof(double), p.cnt, hipMemcpyHostToDevice, s.stream));
                d_hist = s.d_in + (hl_pad - hl);
            }
            DSPX_HIP(hipMemcpy2DAsync(s.d_in + hl_pad, in_pitch * sizeof(double), x + p.c * stride + p.s0,
                                      stride * sizeof(double), p.len * sizeof(double), p.cnt, hipMemcpyHostToDevice,
                                      s.stream));
            const int lrc = launch_dev(d_taps, ntaps, s.d_in + hl_pad, in_pitch, d_hist, in_pitch, s.d_out, out_pitch,
                                       p.cnt, p.len, flags, s.stream);
            if (lrc != DSPX_OK) return lrc;
            DSPX_HIP(hipMemcpy2DAsync(y + p.c * stride + p.s0, stride * sizeof(double), s.d_out,
                                      out_pitch * sizeof(double), p.len * sizeof(double), p.cnt,
                                      hipMemcpyDeviceToHost, s.stream));
        }
        for (auto& s : slots) DSPX_HIP(hipStreamSynchronize(s.stream));
        return DSPX_OK;
    };
    rc = body();
    const std::string keep = g_last_error;
    for (auto& s : slots) {
        if (s.stream) (void)hipStreamSynchronize(s.stream), (void)hipStreamDestroy(s.stream);
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_out) (void)hipFree(s.d_out);
    }
    if (d_taps) (void)hipFree(d_taps);
    if (rc != DSPX_OK) g_last_error = keep;
    return rc;
}

int host_path(const double* taps, size_t ntaps, const double* hist0, const double* x, double* y, size_t nchan,
              size_t n, size_t stride, int ngpus, unsigned flags) {
    if (int rc = check_common(taps, ntaps, x, y, nchan, n, flags)) return rc;
    if (nchan == 0 || n == 0) return DSPX_OK;
    if (stride < n) return fail(DSPX_ERR_INVALID, "stride (%zu) < n (%zu)", stride, n);
    const size_t span = ((nchan - 1) * stride + n) * sizeof(double);
    if (ranges_overlap(x, span, y, span)) return fail(DSPX_ERR_INVALID, "x and y overlap");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DSPX_ERR_NO_DEVICE, "no HIP device");
    int cur = 0;
    DSPX_HIP(hipGetDevice(&cur));
    if (ngpus < 0) {  // single-device entry points: stay on the caller's current device
        std::vector<Piece> pieces;
        cut_pieces(0, nchan, 0, n, pieces);
        return run_device(cur, taps, ntaps, hist0, x, y, stride, pieces, flags);
    }
    const int G = ngpus == 0 ? ndev : ngpus;
    if (G > ndev) return fail(DSPX_ERR_INVALID, "ngpus (%d) exceeds visible devices (%d)", G, ndev);
    const std::vector<std::vector<Piece>> plan = make_plan(nchan, n, G);
    std::vector<int> rcs(G, DSPX_OK);
    std::vector<std::string> msgs(G);
    std::vector<std::thread> workers;
    for (int g = 0; g < G; ++g)
        workers.emplace_back([&, g] {
            rcs[g] = run_device(g, taps, ntaps, hist0, x, y, stride, plan[g], flags);
            if (rcs[g] != DSPX_OK) msgs[g] = g_last_error;
        });
    for (auto& w : workers) w.join();
    (void)hipSetDevice(cur);
    for (int g = 0; g < G; ++g)
        if (rcs[g] != DSPX_OK) return fail(rcs[g], "device %d: %s", g, msgs[g].c_str());
    return DSPX_OK;
}

}  // namespace

extern "C" {

int dspx_version(void) { return 100; }

int dspx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char* dspx_last_error(void) { return g_last_error.c_str(); }

const char* dspx_status_string(int status) {
    switch (status) {
    case DSPX_OK: return "ok";
    case DSPX_ERR_INVALID: return "invalid argument";
    case DSPX_ERR_NO_DEVICE: return "no usable device";
    case DSPX_ERR_HIP: return "HIP runtime error";
    case DSPX_ERR_NOMEM: return "out of device memory";
    case DSPX_ERR_DOMAIN: return "value outside the supported domain";
    default: return "unknown status";
    }
}

int dspx_firwin_f64(size_t numtaps, const double* cutoffs, size_t ncutoffs, int pass_zero, int scale, double* out) {
    if ((ncutoffs && !cutoffs) || (numtaps && !out)) return fail(DSPX_ERR_INVALID, "NULL pointer");
    try {
        const std::vector<double> c(cutoffs, cutoffs + ncutoffs);
        const std::vector<double> h = dsp::firwin<double>(numtaps, c, pass_zero != 0, scale != 0);
        std::copy(h.begin(), h.end(), out);
    } catch (const std::invalid_argument& e) {
        return fail(DSPX_ERR_INVALID, "%s", e.what());
    }
    return DSPX_OK;
}

int dspx_hamming_window_f64(size_t width, double alpha, int sym, double* out) {
    if (width && !out) return fail(DSPX_ERR_INVALID, "NULL pointer");
    try {
        const std::vector<double> w = dsp::hamming_window<double>(width, alpha, sym != 0);
        std::copy(w.begin(), w.end(), out);
    } catch (const std::invalid_argument& e) {
        return fail(DSPX_ERR_INVALID, "%s", e.what());
    }
    return DSPX_OK;
}

double dspx_sinc_f64(double x) { return dsp::sinc<double>(x); }

int dspx_fir_f64(const double* taps, size_t ntaps, const double* x, double* y, size_t n, unsigned flags) {
    return host_path(taps, ntaps, nullptr, x, y, 1, n, n, -1, flags);
}

int dspx_fir_f64_batched(const double* taps, size_t ntaps, const double* x, double* y, size_t nchan, size_t n,
                         size_t stride, unsigned flags) {
    return host_path(taps, ntaps, nullptr, x, y, nchan, n, stride, -1, flags);
}

int dspx_fir_f64_multi(const double* taps, size_t ntaps, const double* x, double* y, size_t nchan, size_t n,
                       size_t stride, int ngpus, unsigned flags) {
    if (ngpus < 0) return fail(DSPX_ERR_INVALID, "ngpus must be >= 0");
    return host_path(taps, ntaps, nullptr, x, y, nchan, n, stride, ngpus, flags);
}

int dspx_fir_f64_stream(const double* taps, size_t ntaps, double* hist, const double* x, double* y, size_t nchan,
                        size_t n, size_t stride, unsigned flags) {
    const size_t hl = ntaps ? ntaps - 1 : 0;
    if (hl && nchan && !hist) return fail(DSPX_ERR_INVALID, "hist is NULL");
    // chunk lengths are arbitrary, so unless the caller insists the chunks run on the vector pipe,
    // whose per-output arithmetic does not depend on where a call starts
    if (!(flags & DSPX_FIR_MFMA)) flags |= DSPX_FIR_VECTOR;
    if (int rc = host_path(taps, ntaps, hl ? hist : nullptr, x, y, nchan, n, stride, -1, flags)) return rc;
    // the next chunk's history: the last ntaps-1 samples of [hist | x]
    for (size_t c = 0; c < nchan && hl && n; ++c) {
        double* h = hist + c * hl;
        const double* xc = x + c * stride;
        if (n >= hl) {
            std::copy(xc + (n - hl), xc + n, h);
        } else {
            std::copy(h + n, h + hl, h);
            std::copy(xc, xc + n, h + (hl - n));
        }
    }
    return DSPX_OK;
}

long dspx_fir_plan(size_t nchan, size_t n, int ngpus, dspx_piece* out, size_t cap) {
    if (ngpus < 1) return fail(DSPX_ERR_INVALID, "ngpus must be >= 1");
    if (cap && !out) return fail(DSPX_ERR_INVALID, "out is NULL");
    const auto plan = make_plan(nchan, n, ngpus);
    size_t k = 0;
    for (int g = 0; g < ngpus; ++g)
        for (const Piece& p : plan[g]) {
            if (k < cap) out[k] = dspx_piece{g, p.c, p.cnt, p.s0, p.len};
            ++k;
        }
    return (long)k;
}

int dspx_fir_f64_dev(const double* d_taps, size_t ntaps, const double* d_x, size_t x_stride, const double* d_hist,
                     size_t hist_stride, double* d_y, size_t y_stride, size_t nchan, size_t n, unsigned flags,
                     void* stream) {
    if (int rc = check_common(d_taps, ntaps, d_x, d_y, nchan, n, flags)) return rc;
    if (nchan == 0 || n == 0) return DSPX_OK;
    if (nchan > 1 && (x_stride < n || y_stride < n))
        return fail(DSPX_ERR_INVALID, "channel stride smaller than n (%zu): x %zu, y %zu", n, x_stride, y_stride);
    if (d_hist && nchan > 1 && ntaps > 1 && hist_stride < ntaps - 1)
        return fail(DSPX_ERR_INVALID, "hist_stride (%zu) < ntaps-1", hist_stride);
    const size_t xspan = ((nchan - 1) * x_stride + n) * sizeof(double);
    const size_t yspan = ((nchan - 1) * y_stride + n) * sizeof(double);
    if (ranges_overlap(d_x, xspan, d_y, yspan)) return fail(DSPX_ERR_INVALID, "d_x and d_y overlap");
    return launch_dev(d_taps, ntaps, d_x, x_stride, d_hist, hist_stride, d_y, y_stride, nchan, n, flags,
                      static_cast<hipStream_t>(stream));
}

int dspx_fill_splitmix_f64_dev(double* d_x, size_t n, uint64_t seed, uint64_t offset, void* stream) {
    if (n && !d_x) return fail(DSPX_ERR_INVALID, "d_x is NULL");
    DSPX_HIP(dspx::launch_fill_splitmix(d_x, n, seed, offset, static_cast<hipStream_t>(stream)));
    return DSPX_OK;
}

int dspx_checksum_f64_dev(const double* d_y, size_t n, uint64_t offset, uint64_t* d_sum, void* stream) {
    if ((n && !d_y) || !d_sum) return fail(DSPX_ERR_INVALID, "NULL pointer");
    DSPX_HIP(dspx::launch_checksum(d_y, n, offset, reinterpret_cast<unsigned long long*>(d_sum),
                                   static_cast<hipStream_t>(stream)));
    return DSPX_OK;
}

}  // extern "C"
