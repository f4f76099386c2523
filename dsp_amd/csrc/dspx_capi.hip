// C ABI of the FIR engine (declared in include/dspx.h): argument checking, the
// device-resident launch, and the host-pointer paths that move caller memory through
// pinned staging and device buffers in an eight-slot pipeline per GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "dsp.hpp"
#include "dspx.h"
#include "dspx_common.h"
#include "fir_kernels.h"
#include "iir_kernels.h"

namespace dspx {

thread_local std::string g_last_error;

static std::atomic<int> g_alias_devices{0};
bool devices_aliased() { return g_alias_devices.load(std::memory_order_relaxed) != 0; }
int g_alias_devices_exchange(bool on) { return g_alias_devices.exchange(on ? 1 : 0); }

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

using dspx::kMaxGrid;

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
    const size_t tile = (size_t)(mfma ? dspx::fir_mfma_tile() : dspx::fir_tile());
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
                                               vec2, (flags & DSPX_FIR_TILED) != 0, stream));
        else
            DSPX_HIP(dspx::launch_fir_f64(d_taps, d_x + c0 * x_stride, d_hist ? d_hist + c0 * hist_stride : nullptr,
                                          d_y + c0 * y_stride, s, (flags & DSPX_FIR_EXACT) != 0, vec2, stream));
    }
    return DSPX_OK;
}

// The recursive tail of lfilter (dsp.hpp:366-368) for `order` = na - 1 >= 1 coefficients a[1..order].
struct Tail {
    dspx::IirCoef coef{};
    int order = 0;
    bool exact = false, scan = false;
    unsigned lane_plain = 0, tile_plain = 0;  // dspx::IirPlan
    const double* y0 = nullptr;  // host-pointer paths: per channel the `order` outputs before y[0], oldest first
};

// Fills `t` from the denominators; a[0] is never read (dsp.hpp:366 starts at j = 1).
int make_tail(const double* a, size_t na, unsigned flags, Tail& t) {
    t = Tail{};
    if (na <= 1) return DSPX_OK;
    if (!a) return fail(DSPX_ERR_INVALID, "denominators is NULL");
    if (na - 1 > (size_t)dspx::kIirMaxOrder)
        return fail(DSPX_ERR_INVALID, "at most %d denominator coefficients are supported, got %zu",
                    dspx::kIirMaxOrder + 1, na);
    t.order = (int)(na - 1);
    for (int j = 0; j < t.order; ++j) t.coef.a[j] = a[j + 1];
    t.exact = (flags & DSPX_FIR_EXACT) != 0;
    // a recurrence whose state grows without bound (or non-finite coefficients) leaves the block-parallel
    // path without usable matrix powers: such filters run on the sequential kernel
    const dspx::IirPlan plan = dspx::iir_plan(t.coef.a, t.order);
    t.scan = !t.exact && plan.usable;
    t.lane_plain = plan.lane_plain, t.tile_plain = plan.tile_plain;
    return DSPX_OK;
}

size_t tail_workspace_bytes(const Tail& t, size_t nchan, size_t n) {
    return t.order && t.scan ? dspx::iir_workspace_bytes(nchan, n, t.order) : 0;
}

// Workspace the block-parallel tail needs for a call of `nchan` channels x `n` samples: launch_tail() cuts the
// channels into launches that reuse one workspace, so the need is that of the most demanding launch (the last
// one can be the only one with too few channels for the walk kernel, which is the one that needs no tile states).
size_t scan_workspace_bytes(int order, size_t nchan, size_t n) {
    const size_t tiles = (n + dspx::kIirTile - 1) / dspx::kIirTile;
    const size_t per_launch = std::max<size_t>(1, kMaxGrid / std::max<size_t>(tiles, 1));
    size_t need = 0;
    if (nchan >= per_launch) need = dspx::iir_workspace_bytes(per_launch, n, order);
    if (nchan % per_launch) need = std::max(need, dspx::iir_workspace_bytes(nchan % per_launch, n, order));
    return need;
}

// Applies the tail in place to d_y (which holds the FIR part).  d_state: per channel the `order` outputs
// that precede y[0], oldest first (nullable).  d_work: tail_workspace_bytes() for the channels of one launch.
// True when the numerator is short enough for the tail kernels to compute the FIR part themselves.
bool tail_fuses_fir(const Tail& t, size_t nb, unsigned flags) {
    return t.order && t.scan && nb >= 1 && nb <= (size_t)dspx::kIirMaxFusedTaps && !(flags & DSPX_FIR_MFMA);
}

// `front` (nullable): the FIR part is computed from x inside the tail kernels (tail_fuses_fir) instead of being
// expected in d_y.
int launch_tail(const Tail& t, double* d_y, size_t y_stride, const double* d_state, size_t state_stride, size_t nchan,
                size_t n, void* d_work, hipStream_t stream, const dspx::IirFront* front = nullptr) {
    if (t.order == 0 || nchan == 0 || n == 0) return DSPX_OK;
    const size_t tiles = (n + dspx::kIirTile - 1) / dspx::kIirTile;
    if (tiles > kMaxGrid) return fail(DSPX_ERR_INVALID, "n too large for one launch: %zu", n);
    const size_t chans_per_launch = t.scan ? std::max<size_t>(1, kMaxGrid / tiles) : kMaxGrid;
    for (size_t c0 = 0; c0 < nchan; c0 += chans_per_launch) {
        dspx::IirShape s{};
        s.y_stride = y_stride;
        s.state_stride = state_stride;
        s.n = (long long)n;
        s.tiles_per_chan = (unsigned)tiles;
        s.nchan = (unsigned)std::min(chans_per_launch, nchan - c0);
        s.vec2 = aligned16(d_y) && y_stride % 2 == 0;
        s.lane_plain = t.lane_plain, s.tile_plain = t.tile_plain;
        const double* st = d_state ? d_state + c0 * state_stride : nullptr;
        if (t.scan) {  // launches on one stream run in order, so the workspace is reused from launch to launch
            dspx::IirFront f{};
            if (front) {
                f = *front;
                f.x += c0 * f.x_stride;
                if (f.xhist) f.xhist += c0 * f.xhist_stride;
            }
            DSPX_HIP(dspx::launch_iir_scan(t.coef, t.order, d_y + c0 * y_stride, st, s, front ? &f : nullptr, d_work,
                                           stream));
        }
        else
            DSPX_HIP(dspx::launch_iir_sequential(t.coef, t.order, d_y + c0 * y_stride, st, s, t.exact, stream));
    }
    return DSPX_OK;
}

dspx::IirFront make_front(const double* d_b, size_t nb, const double* d_x, size_t x_stride, const double* d_xhist,
                          size_t xhist_stride) {
    dspx::IirFront f{};
    f.x = d_x, f.xhist = d_xhist, f.x_stride = x_stride, f.xhist_stride = xhist_stride;
    f.b = d_b, f.nb = (int)nb;
    f.vec2 = aligned16(d_x) && x_stride % 2 == 0;
    return f;
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

constexpr size_t kMaxPieceSamples = size_t(1) << 21;  // 16 MiB each way: long enough for full link speed
constexpr size_t kMinPieceSamples = size_t(1) << 16;  // below this the per-piece overheads dominate

// Cuts channels [c0, c1) x samples [s0, s1) into pieces: about eight or more of them when the work is large
// enough, so that staging copies, transfers and kernels of different pieces overlap; powers of two, so that
// cuts inside a channel fall on multiples of 4.
void cut_pieces(size_t c0, size_t c1, size_t s0, size_t s1, std::vector<Piece>& out) {
    const size_t len = s1 - s0;
    if (len == 0 || c1 == c0) return;
    size_t piece = kMinPieceSamples;
    while (piece < kMaxPieceSamples && piece * 16 <= (c1 - c0) * len) piece *= 2;
    if (len <= piece) {
        const size_t per = std::max<size_t>(1, piece / len);
        for (size_t c = c0; c < c1; c += per) out.push_back({c, std::min(per, c1 - c), s0, len});
    } else {
        for (size_t c = c0; c < c1; ++c)
            for (size_t s = s0; s < s1; s += piece) out.push_back({c, 1, s, std::min(piece, s1 - s)});
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

// A slot carries one piece at a time: pinned host staging and device buffers for both directions,
// the stream its copies and kernel run on, and the event that marks its result as back on the host.
struct Slot {
    double *pin_in = nullptr, *pin_out = nullptr, *d_in = nullptr, *d_out = nullptr;
    void* d_work = nullptr;          // workspace of the recursive tail
    double* d_tail = nullptr;        // recursive filters: the last outputs of the slot's piece, read by the piece
                                     // that continues its channel (kIirMaxOrder doubles)
    size_t in_cap = 0, out_cap = 0;  // doubles
    size_t work_cap = 0;             // bytes
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
};

constexpr int kSlots = 8;

// Staging of one device, kept between calls (pinned allocations and stream creation cost more than a
// small filter).  One call uses an engine exclusively; concurrent calls on a device get engines of their own.
struct Engine {
    int device = -1;
    Slot slots[kSlots];
};

std::mutex g_pool_mu;
std::vector<Engine*> g_pool;           // idle engines; never freed at process exit (the runtime may be gone by then)
constexpr size_t kIdlePerDevice = 2;

void destroy_engine(Engine* e) {
    if (!e) return;
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(e->device);
    for (auto& s : e->slots) {
        if (s.stream) (void)hipStreamSynchronize(s.stream), dspx::forget_stream(s.stream), (void)hipStreamDestroy(s.stream);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.pin_in) (void)hipHostFree(s.pin_in);
        if (s.pin_out) (void)hipHostFree(s.pin_out);
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.d_work) (void)hipFree(s.d_work);
        if (s.d_tail) (void)hipFree(s.d_tail);
    }
    (void)hipSetDevice(cur);
    delete e;
}

// The device must be current.
int acquire_engine(int device, Engine** out) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i]->device == device) {
                *out = g_pool[i];
                g_pool.erase(g_pool.begin() + (long)i);
                return DSPX_OK;
            }
    }
    Engine* e = new Engine;
    e->device = device;
    auto body = [&]() -> int {
        for (auto& s : e->slots) {
            DSPX_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
            DSPX_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
            DSPX_HIP(hipMalloc((void**)&s.d_tail, dspx::kIirMaxOrder * sizeof(double)));
        }
        return DSPX_OK;
    };
    if (int rc = body()) {
        const std::string keep = g_last_error;
        destroy_engine(e);
        g_last_error = keep;
        return rc;
    }
    *out = e;
    return DSPX_OK;
}

void release_engine(Engine* e, bool healthy) {
    if (healthy) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t same = 0;
        for (Engine* o : g_pool) same += o->device == e->device;
        if (same < kIdlePerDevice) {
            g_pool.push_back(e);
            return;
        }
    }
    destroy_engine(e);
}

int grow_slot(Slot& s, size_t in_doubles, size_t out_doubles, size_t work_bytes) {
    if (s.work_cap < work_bytes) {
        if (s.d_work) DSPX_HIP(hipFree(s.d_work));
        s.d_work = nullptr, s.work_cap = 0;
        DSPX_HIP(hipMalloc(&s.d_work, work_bytes));
        s.work_cap = work_bytes;
    }
    if (s.in_cap < in_doubles) {
        if (s.pin_in) DSPX_HIP(hipHostFree(s.pin_in));
        if (s.d_in) DSPX_HIP(hipFree(s.d_in));
        s.pin_in = s.d_in = nullptr, s.in_cap = 0;
        DSPX_HIP(hipHostMalloc((void**)&s.pin_in, in_doubles * sizeof(double), hipHostMallocDefault));
        DSPX_HIP(hipMalloc((void**)&s.d_in, in_doubles * sizeof(double)));
        s.in_cap = in_doubles;
    }
    if (s.out_cap < out_doubles) {
        if (s.pin_out) DSPX_HIP(hipHostFree(s.pin_out));
        if (s.d_out) DSPX_HIP(hipFree(s.d_out));
        s.pin_out = s.d_out = nullptr, s.out_cap = 0;
        DSPX_HIP(hipHostMalloc((void**)&s.pin_out, out_doubles * sizeof(double), hipHostMallocDefault));
        DSPX_HIP(hipMalloc((void**)&s.d_out, out_doubles * sizeof(double)));
        s.out_cap = out_doubles;
    }
    return DSPX_OK;
}

// Geometry of a staged piece: rows of [pad | ntaps-1 history samples | len samples], 16-byte aligned.
// The taps travel in front of the rows, so a piece needs one upload.
struct PieceGeom {
    size_t hl, hl_pad, in_pitch, out_pitch, taps_pad;
    PieceGeom(size_t ntaps, size_t len)
        : hl(ntaps ? ntaps - 1 : 0), hl_pad((hl + 1) & ~size_t(1)), in_pitch(hl_pad + ((len + 1) & ~size_t(1))),
          out_pitch((len + 1) & ~size_t(1)), taps_pad((ntaps + 1) & ~size_t(1)) {}
};

// What one call on one device shares between its producer and consumer threads.
struct Pipeline {
    int device;
    Engine* eng;
    const double *taps, *hist0, *x;
    double* y;
    size_t ntaps, stride;
    unsigned flags;
    const std::vector<Piece>* pieces;
    const Tail* tail;  // recursive part, order 0 for none
    // download mode (d_src set): the pieces are regions of a result that already sits in device memory, row r of
    // it destined for row row_map[r] of y (or row r when row_map is null); nothing is uploaded or launched
    const double* d_src = nullptr;
    size_t d_src_pitch = 0;
    const size_t* row_map = nullptr;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<unsigned char> state;  // per piece: 0 waiting, 1 on the device, 2 delivered
    int rc = DSPX_OK;                  // first failure
    std::string msg;

    Pipeline(int device_, Engine* eng_, const double* taps_, const double* hist0_, const double* x_, double* y_,
             size_t ntaps_, size_t stride_, unsigned flags_, const std::vector<Piece>* pieces_, const Tail* tail_)
        : device(device_), eng(eng_), taps(taps_), hist0(hist0_), x(x_), y(y_), ntaps(ntaps_), stride(stride_),
          flags(flags_), pieces(pieces_), tail(tail_), state(pieces_->size(), 0) {}

    void set_error(int code) {
        std::lock_guard<std::mutex> lk(mu);
        if (rc == DSPX_OK) rc = code, msg = g_last_error;
        cv.notify_all();
    }
    // Waits until piece `i` reaches `want`; false when the call has failed meanwhile.
    bool wait_for(size_t i, unsigned char want) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return rc != DSPX_OK || state[i] >= want; });
        return rc == DSPX_OK;
    }
    void mark(size_t i, unsigned char v) {
        std::lock_guard<std::mutex> lk(mu);
        state[i] = v;
        cv.notify_all();
    }

    // Copies the piece's rows into pinned staging, then queues upload, filter and download on the slot's stream.
    int produce(size_t i) {
        const Piece& p = (*pieces)[i];
        Slot& s = eng->slots[i % kSlots];
        const PieceGeom g(ntaps, p.len);
        if (d_src) {
            if (int grc = grow_slot(s, 0, g.out_pitch * p.cnt, 0)) return grc;
            DSPX_HIP(hipMemcpy2DAsync(s.pin_out, g.out_pitch * sizeof(double), d_src + p.c * d_src_pitch + p.s0,
                                      d_src_pitch * sizeof(double), p.len * sizeof(double), p.cnt,
                                      hipMemcpyDeviceToHost, s.stream));
            DSPX_HIP(hipEventRecord(s.done, s.stream));
            return DSPX_OK;
        }
        // the outputs that precede a channel's first piece, when the caller supplied them, travel behind the rows
        const bool with_y0 = tail->order && tail->y0 && p.s0 == 0;
        const size_t y0_off = g.taps_pad + g.in_pitch * p.cnt;
        if (int grc = grow_slot(s, y0_off + (with_y0 ? p.cnt * (size_t)tail->order : 0), g.out_pitch * p.cnt,
                                tail_workspace_bytes(*tail, p.cnt, p.len)))
            return grc;
        if (ntaps) std::memcpy(s.pin_in, taps, ntaps * sizeof(double));
        if (with_y0)
            std::memcpy(s.pin_in + y0_off, tail->y0 + p.c * (size_t)tail->order,
                        p.cnt * (size_t)tail->order * sizeof(double));
        // history: the lead samples that exist in the caller's buffer; before them the caller's
        // initial history (samples s0 .. hl-1 of it when lead == s0 < hl), or zeros
        const size_t lead = std::min(g.hl, p.s0);
        const bool with_hist = (p.s0 > 0 || hist0) && g.hl > 0;
        for (size_t r = 0; r < p.cnt; ++r) {
            double* row = s.pin_in + g.taps_pad + r * g.in_pitch;
            const double* xr = x + (p.c + r) * stride;
            if (with_hist) {
                double* h = row + (g.hl_pad - g.hl);
                if (lead < g.hl) {
                    if (hist0)
                        std::memcpy(h, hist0 + (p.c + r) * g.hl + p.s0, (g.hl - lead) * sizeof(double));
                    else
                        std::memset(h, 0, (g.hl - lead) * sizeof(double));
                }
                if (lead) std::memcpy(h + (g.hl - lead), xr + (p.s0 - lead), lead * sizeof(double));
            }
            std::memcpy(row + g.hl_pad, xr + p.s0, p.len * sizeof(double));
        }
        DSPX_HIP(hipMemcpyAsync(s.d_in, s.pin_in, (y0_off + (with_y0 ? p.cnt * (size_t)tail->order : 0)) * sizeof(double),
                                hipMemcpyHostToDevice, s.stream));
        const double* d_rows = s.d_in + g.taps_pad;
        const bool fused = tail_fuses_fir(*tail, ntaps, flags);
        const double* d_xhist = with_hist ? d_rows + (g.hl_pad - g.hl) : nullptr;
        if (!fused)
            if (int lrc = launch_dev(s.d_in, ntaps, d_rows + g.hl_pad, g.in_pitch, d_xhist, g.in_pitch, s.d_out,
                                     g.out_pitch, p.cnt, p.len, flags, s.stream))
                return lrc;
        if (tail->order) {
            // a piece that continues a channel starts from the last outputs of the piece before it, which
            // runs on another stream and left them in its own slot: wait for it (such pieces are issued in
            // order by one thread)
            const double* d_state = with_y0 ? s.d_in + y0_off : nullptr;
            if (p.s0 > 0) {
                const Slot& before = eng->slots[(i - 1) % kSlots];
                DSPX_HIP(hipStreamWaitEvent(s.stream, before.done, 0));
                d_state = before.d_tail;
            }
            const dspx::IirFront front = make_front(s.d_in, ntaps, d_rows + g.hl_pad, g.in_pitch, d_xhist, g.in_pitch);
            if (int trc = launch_tail(*tail, s.d_out, g.out_pitch, d_state, (size_t)tail->order, p.cnt, p.len, s.d_work,
                                      s.stream, fused ? &front : nullptr))
                return trc;
            const bool continues = i + 1 < pieces->size() && (*pieces)[i + 1].s0 > 0;
            if (continues) {  // cnt == 1 and len >= order here
                // the slot's buffer was last read by the piece after this slot's previous piece (piece
                // i + 1 - kSlots, whose event its slot still holds: piece i + 1 is not issued yet)
                if (i >= (size_t)kSlots) DSPX_HIP(hipStreamWaitEvent(s.stream, eng->slots[(i + 1) % kSlots].done, 0));
                DSPX_HIP(hipMemcpyAsync(s.d_tail, s.d_out + (p.len - (size_t)tail->order),
                                        (size_t)tail->order * sizeof(double), hipMemcpyDeviceToDevice, s.stream));
            }
        }
        DSPX_HIP(hipMemcpyAsync(s.pin_out, s.d_out, g.out_pitch * p.cnt * sizeof(double), hipMemcpyDeviceToHost,
                                s.stream));
        DSPX_HIP(hipEventRecord(s.done, s.stream));
        return DSPX_OK;
    }

    // Waits for the piece's result to be in pinned staging and copies it into the caller's buffer.
    int consume(size_t i) {
        const Piece& p = (*pieces)[i];
        Slot& s = eng->slots[i % kSlots];
        const PieceGeom g(ntaps, p.len);
        DSPX_HIP(hipEventSynchronize(s.done));
        for (size_t r = 0; r < p.cnt; ++r)
            std::memcpy(y + (row_map ? row_map[p.c + r] : p.c + r) * stride + p.s0, s.pin_out + r * g.out_pitch,
                        p.len * sizeof(double));
        return DSPX_OK;
    }

    // Producer `k` of `np`: pieces k, k + np, ... in order; a slot is reused once its previous piece is delivered.
    void producer(size_t k, size_t np) {
        if (hipSetDevice(device) != hipSuccess) return set_error(fail(DSPX_ERR_HIP, "hipSetDevice(%d) failed", device));
        for (size_t i = k; i < pieces->size(); i += np) {
            if (i >= (size_t)kSlots && !wait_for(i - kSlots, 2)) return;
            if (int prc = produce(i)) return set_error(prc);
            mark(i, 1);
        }
    }
    void consumer(size_t k, size_t nc) {
        if (hipSetDevice(device) != hipSuccess) return set_error(fail(DSPX_ERR_HIP, "hipSetDevice(%d) failed", device));
        for (size_t i = k; i < pieces->size(); i += nc) {
            if (!wait_for(i, 1)) return;
            if (int crc = consume(i)) return set_error(crc);
            mark(i, 2);
        }
    }
};

// Runs the pieces of `pl` through its engine and gives the engine back.  Small calls run in the calling
// thread; larger ones add threads (up to `max_threads` each way) so that the copies into and out of pinned
// staging overlap the transfers and the kernels.
int run_pipeline(Pipeline& pl, size_t max_threads) {
    const std::vector<Piece>& pieces = *pl.pieces;
    Engine* eng = pl.eng;
    auto body = [&]() -> int {
        if (pieces.size() <= 2) {
            for (size_t i = 0; i < pieces.size(); ++i)
                if (int prc = pl.produce(i)) return prc;
            for (size_t i = 0; i < pieces.size(); ++i)
                if (int crc = pl.consume(i)) return crc;
            return DSPX_OK;
        }
        // copy threads per direction: measured on an MI355X host, 1 / 2 / 3 / 4 threads move 3.2 / 3.7 / 4.1 / 4.3 G
        // samples/s end to end; more gain nothing and take cores from the caller
        const size_t nthreads =
            std::min<size_t>(max_threads, pieces.size() >= 64 ? 4 : pieces.size() >= 32 ? 3 : pieces.size() >= 8 ? 2 : 1);
        size_t producers = pl.d_src ? 1 : nthreads;  // issuing a download costs the host nothing
        if (pl.tail->order)  // pieces that continue a channel must be issued in order, by one thread
            for (const Piece& p : pieces)
                if (p.s0 > 0) producers = 1;
        dspx::Joiner workers;
        try {
            for (size_t k = 0; k < nthreads; ++k)
                workers.threads.emplace_back([&pl, k, nthreads] { pl.consumer(k, nthreads); });
            for (size_t k = 1; k < producers; ++k)
                workers.threads.emplace_back([&pl, k, producers] { pl.producer(k, producers); });
            pl.producer(0, producers);
        } catch (const std::exception& e) {  // a thread could not start: the ones that did see the failure and leave
            pl.set_error(fail(DSPX_ERR_HOST, "staging threads: %s", e.what()));
        }
        workers.join();
        if (pl.rc != DSPX_OK) g_last_error = pl.msg;
        return pl.rc;
    };
    const int rc = body();
    const std::string keep = g_last_error;
    bool healthy = rc == DSPX_OK;
    if (!healthy) {  // leave nothing queued on streams that outlive this call
        healthy = true;
        for (auto& s : eng->slots) healthy = (hipStreamSynchronize(s.stream) == hipSuccess) && healthy;
    }
    release_engine(eng, healthy);
    if (rc != DSPX_OK) g_last_error = keep;
    return rc;
}

// Filters `pieces` on `device`.  `hist0`, when non-null, holds for every channel the ntaps-1 samples that
// precede x[0] (channel c at hist0 + c*(ntaps-1)); null means zeros.
int run_device(int device, const double* taps, size_t ntaps, const double* hist0, const double* x, double* y,
               size_t stride, const std::vector<Piece>& pieces, unsigned flags, size_t max_threads,
               const Tail& tail = Tail{}) {
    if (pieces.empty()) return DSPX_OK;
    DSPX_HIP(hipSetDevice(device));
    Engine* eng = nullptr;
    if (int rc = acquire_engine(device, &eng)) return rc;
    Pipeline pl(device, eng, taps, hist0, x, y, ntaps, stride, flags, &pieces, &tail);
    return run_pipeline(pl, max_threads);  // `device` is a physical device (see dspx::physical_device)
}

int host_path_body(const double* taps, size_t ntaps, const double* hist0, const double* x, double* y, size_t nchan,
                   size_t n, size_t stride, int ngpus, unsigned flags, const Tail& tail) {
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
        return run_device(cur, taps, ntaps, hist0, x, y, stride, pieces, flags, 4, tail);
    }
    int G = ngpus == 0 ? ndev : ngpus;
    const bool aliased = dspx::devices_aliased();  // read once, here
    if (G > dspx::logical_device_limit(aliased, ndev))
        return fail(DSPX_ERR_INVALID, "ngpus (%d) exceeds visible devices (%d)", G, ndev);
    // a recursive filter's state is a serial dependence along a channel: such filters shard by channels only
    if (tail.order && nchan < (size_t)G) G = (int)nchan;
    const std::vector<std::vector<Piece>> plan = make_plan(nchan, n, G);
    std::vector<int> rcs(G, DSPX_OK);
    std::vector<std::string> msgs(G);
    {
        dspx::Joiner workers;  // joins the threads that did start if one cannot (the exception goes on to the entry point)
        for (int g = 0; g < G; ++g)
            workers.threads.emplace_back([&, g] {
                rcs[g] = dspx::guarded([&] {
                    return run_device(dspx::physical_device(aliased, g, ndev), taps, ntaps, hist0, x, y, stride, plan[g], flags,
                                      G > 2 ? 1 : 2, tail);
                });
                if (rcs[g] != DSPX_OK) msgs[g] = g_last_error;
            });
    }
    (void)hipSetDevice(cur);
    for (int g = 0; g < G; ++g)
        if (rcs[g] != DSPX_OK) return fail(rcs[g], "device %d: %s", g, msgs[g].c_str());
    return DSPX_OK;
}


// lfilter over narrow element types: TIN is the type the products are formed in (float or double containers), TOUT the
// result type (float: the reference's lfilter<float>; double with TIN = float: lfilter<double> over containers of float).
template <typename TIN, typename TOUT>
int narrow_dev(const TIN* d_taps, size_t ntaps, const TIN* d_x, size_t x_stride, const TIN* d_hist, size_t hist_stride,
               TOUT* d_y, size_t y_stride, size_t nchan, size_t n, hipStream_t stream) {
    if (ntaps > 0x7fffffffull) return fail(DSPX_ERR_INVALID, "ntaps too large: %zu", ntaps);
    if (ntaps && !d_taps) return fail(DSPX_ERR_INVALID, "taps is NULL");
    if (nchan && n && (!d_x || !d_y)) return fail(DSPX_ERR_INVALID, "x or y is NULL");
    if (nchan == 0 || n == 0) return DSPX_OK;
    if (nchan > 1 && (x_stride < n || y_stride < n))
        return fail(DSPX_ERR_INVALID, "channel stride smaller than n (%zu): x %zu, y %zu", n, x_stride, y_stride);
    if (d_hist && nchan > 1 && ntaps > 1 && hist_stride < ntaps - 1)
        return fail(DSPX_ERR_INVALID, "hist_stride (%zu) < ntaps-1", hist_stride);
    if (ranges_overlap(d_x, ((nchan - 1) * x_stride + n) * sizeof(TIN), d_y, ((nchan - 1) * y_stride + n) * sizeof(TOUT)))
        return fail(DSPX_ERR_INVALID, "d_x and d_y overlap");
    const size_t tile = (size_t)dspx::fir_tile(), tiles_per_chan = (n + tile - 1) / tile;
    if (tiles_per_chan > kMaxGrid) return fail(DSPX_ERR_INVALID, "n too large for one launch: %zu", n);
    const size_t chans_per_launch = std::max<size_t>(1, kMaxGrid / tiles_per_chan);
    for (size_t c0 = 0; c0 < nchan; c0 += chans_per_launch) {
        const size_t nc = std::min(chans_per_launch, nchan - c0);
        dspx::FirShape s{};
        s.x_stride = x_stride, s.hist_stride = hist_stride, s.y_stride = y_stride;
        s.n = (long long)n;
        s.total_tiles = (unsigned)(nc * tiles_per_chan);
        s.tiles_per_chan = (unsigned)tiles_per_chan;
        s.ntaps = (int)ntaps;
        const TIN* h = d_hist ? d_hist + c0 * hist_stride : nullptr;
        if constexpr (std::is_same_v<TOUT, double>)
            DSPX_HIP(dspx::launch_fir_f64_from_f32(d_taps, d_x + c0 * x_stride, h, d_y + c0 * y_stride, s, stream));
        else if constexpr (std::is_same_v<TIN, float>)
            DSPX_HIP(dspx::launch_fir_f32(d_taps, d_x + c0 * x_stride, h, d_y + c0 * y_stride, s, stream));
        else
            DSPX_HIP(dspx::launch_fir_f32_from_f64(d_taps, d_x + c0 * x_stride, h, d_y + c0 * y_stride, s, stream));
    }
    return DSPX_OK;
}

// Host pointers, one channel: plain allocate / copy / filter / copy back (no staging pipeline: these paths are
// functional).  With float results, `a` (na > 1) adds the recursive tail of lfilter<float> in place on the device.
template <typename TIN, typename TOUT>
int narrow_host(const TIN* taps, size_t ntaps, const double* a, size_t na, bool tail_in_double, const TIN* x, TOUT* y,
                size_t n) {
    if (ntaps && !taps) return fail(DSPX_ERR_INVALID, "taps is NULL");
    if (n && (!x || !y)) return fail(DSPX_ERR_INVALID, "x or y is NULL");
    dspx::IirCoef coef{};
    const int order = na > 1 ? (int)(na - 1) : 0;
    if (order) {
        if (!a) return fail(DSPX_ERR_INVALID, "denominators is NULL");
        if (order > dspx::kIirMaxOrder)
            return fail(DSPX_ERR_INVALID, "at most %d denominator coefficients are supported, got %zu", dspx::kIirMaxOrder + 1, na);
        for (int j = 0; j < order; ++j) coef.a[j] = a[j + 1];  // a[0] is never read (dsp.hpp:366 starts at j = 1)
    }
    if (n == 0) return DSPX_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DSPX_ERR_NO_DEVICE, "no HIP device");
    TIN *d_taps = nullptr, *d_x = nullptr;
    TOUT* d_y = nullptr;
    auto body = [&]() -> int {
        DSPX_HIP(hipMalloc((void**)&d_taps, std::max<size_t>(ntaps, 1) * sizeof(TIN)));
        DSPX_HIP(hipMalloc((void**)&d_x, n * sizeof(TIN)));
        DSPX_HIP(hipMalloc((void**)&d_y, n * sizeof(TOUT)));
        if (ntaps) DSPX_HIP(hipMemcpy(d_taps, taps, ntaps * sizeof(TIN), hipMemcpyHostToDevice));
        DSPX_HIP(hipMemcpy(d_x, x, n * sizeof(TIN), hipMemcpyHostToDevice));
        if (int rc = narrow_dev<TIN, TOUT>(d_taps, ntaps, d_x, n, nullptr, 0, d_y, n, 1, n, nullptr)) return rc;
        if constexpr (std::is_same_v<TOUT, float>) {
            if (order) DSPX_HIP(dspx::launch_iir_narrow_tail(coef, order, tail_in_double, d_y, n, 1, (long long)n, nullptr));
        }
        DSPX_HIP(hipMemcpy(y, d_y, n * sizeof(TOUT), hipMemcpyDeviceToHost));
        return DSPX_OK;
    };
    const int rc = body();
    const std::string keep = g_last_error;
    if (d_taps) (void)hipFree(d_taps);
    if (d_x) (void)hipFree(d_x);
    if (d_y) (void)hipFree(d_y);
    if (rc != DSPX_OK) g_last_error = keep;
    return rc;
}

// The host-pointer entry points: no C++ exception (allocation, thread creation) crosses the C ABI.
int host_path(const double* taps, size_t ntaps, const double* hist0, const double* x, double* y, size_t nchan, size_t n,
              size_t stride, int ngpus, unsigned flags, const Tail& tail = Tail{}) {
    return dspx::guarded([&] { return host_path_body(taps, ntaps, hist0, x, y, nchan, n, stride, ngpus, flags, tail); });
}

// Body of dspx_lfilter_f64_dev and dspx_lfilter_f64_dev_noalloc: the FIR part (or nothing, when the tail kernels
// compute it themselves), then the recursive tail in place on d_y with `d_work` as its workspace.  Allocates nothing.
int lfilter_dev(const double* d_b, size_t nb, const double* a, size_t na, const double* d_x, size_t x_stride,
                const double* d_xhist, size_t xhist_stride, const double* d_yhist, size_t yhist_stride, double* d_y,
                size_t y_stride, size_t nchan, size_t n, void* d_work, size_t work_bytes, const char* work_name,
                unsigned flags, void* stream) {
    Tail tail;
    if (int rc = dspx::check_flags(flags)) return rc;
    if (int rc = make_tail(a, na, flags, tail)) return rc;
    // everything that can be refused is refused before the first launch
    const size_t need = tail.order && tail.scan && nchan && n ? scan_workspace_bytes(tail.order, nchan, n) : 0;
    if (need && (!d_work || work_bytes < need))
        return fail(DSPX_ERR_INVALID, "the recursive tail needs %zu bytes of scratch for %zu channel(s) x %zu samples, "
                    "%s holds %zu", need, nchan, n, work_name, d_work ? work_bytes : (size_t)0);
    if (tail.order && d_yhist && nchan > 1 && yhist_stride < (size_t)tail.order)
        return fail(DSPX_ERR_INVALID, "yhist_stride (%zu) < na-1", yhist_stride);
    const bool fused = tail_fuses_fir(tail, nb, flags);
    if (fused) {  // the checks dspx_fir_f64_dev would make; the tail kernels read x themselves
        if (int rc = check_common(d_b, nb, d_x, d_y, nchan, n, flags)) return rc;
        if (nchan > 1 && (x_stride < n || y_stride < n))
            return fail(DSPX_ERR_INVALID, "channel stride smaller than n (%zu): x %zu, y %zu", n, x_stride, y_stride);
        if (d_xhist && nchan > 1 && nb > 1 && xhist_stride < nb - 1)
            return fail(DSPX_ERR_INVALID, "hist_stride (%zu) < ntaps-1", xhist_stride);
        if (nchan && n && ranges_overlap(d_x, ((nchan - 1) * x_stride + n) * sizeof(double), d_y,
                                         ((nchan - 1) * y_stride + n) * sizeof(double)))
            return fail(DSPX_ERR_INVALID, "d_x and d_y overlap");
    } else if (int rc = dspx_fir_f64_dev(d_b, nb, d_x, x_stride, d_xhist, xhist_stride, d_y, y_stride, nchan, n, flags,
                                         stream)) {
        return rc;
    }
    if (tail.order == 0 || nchan == 0 || n == 0) return DSPX_OK;
    const dspx::IirFront front = make_front(d_b, nb, d_x, x_stride, d_xhist, xhist_stride);
    return launch_tail(tail, d_y, y_stride, d_yhist, yhist_stride, nchan, n, d_work, static_cast<hipStream_t>(stream),
                       fused ? &front : nullptr);
}

// ---------------------------------------------------------------------------------
// Scratch of dspx_lfilter_f64_dev_noalloc: the workspace of the block-parallel tail (matrix powers + one state per
// tile) in a device array that is part of the library's code object, so that entry point never allocates.  Every
// device the library runs on holds its own copy at its own address for the life of the process.  `scratch` is a
// host-side stand-in for the symbol, not a device address: host code only ever uses the address
// hipGetSymbolAddress returned for the current device.
constexpr size_t kStaticScratchBytes = DSPX_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxScratchDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxScratchDevices] = {};

// The current device's copy of `scratch` (looked up once per device, on the first call there).
int static_scratch(void** out) {
    int dev = -1;
    DSPX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxScratchDevices)
        return fail(DSPX_ERR_INVALID, "device index %d outside the %d devices the static scratch is kept for", dev,
                    kMaxScratchDevices);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        DSPX_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p || (reinterpret_cast<uintptr_t>(p) & 255u))
            return fail(DSPX_ERR_HIP, "hipGetSymbolAddress(scratch) returned %p on device %d", p, dev);
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return DSPX_OK;
}

}  // namespace

namespace dspx {

int download_rows(int device, const double* d_src, size_t src_pitch, size_t rows, size_t n, double* dst, size_t dst_stride,
                  const size_t* row_map) {
    if (rows == 0 || n == 0) return DSPX_OK;
    DSPX_HIP(hipSetDevice(device));
    std::vector<Piece> pieces;
    cut_pieces(0, rows, 0, n, pieces);
    Engine* eng = nullptr;
    if (int rc = acquire_engine(device, &eng)) return rc;
    const Tail none;
    Pipeline pl(device, eng, nullptr, nullptr, nullptr, dst, 0, dst_stride, 0, &pieces, &none);
    pl.d_src = d_src, pl.d_src_pitch = src_pitch, pl.row_map = row_map;
    return run_pipeline(pl, 4);
}

}  // namespace dspx

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

int dspx_release_staging(void) {
    std::vector<Engine*> idle;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        idle.swap(g_pool);
    }
    for (Engine* e : idle) destroy_engine(e);
    return (int)idle.size();
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
    case DSPX_ERR_HOST: return "host-side failure inside the library";
    default: return "unknown status";
    }
}

int dspx_firwin_f64(size_t numtaps, const double* cutoffs, size_t ncutoffs, int pass_zero, int scale, double* out) {
    if ((ncutoffs && !cutoffs) || (numtaps && !out)) return fail(DSPX_ERR_INVALID, "NULL pointer");
    return dspx::guarded([&] {  // the reference's std::invalid_argument becomes DSPX_ERR_INVALID with its message
        const std::vector<double> c(cutoffs, cutoffs + ncutoffs);
        const std::vector<double> h = dsp::firwin<double>(numtaps, c, pass_zero != 0, scale != 0);
        std::copy(h.begin(), h.end(), out);
        return (int)DSPX_OK;
    });
}

int dspx_hamming_window_f64(size_t width, double alpha, int sym, double* out) {
    if (width && !out) return fail(DSPX_ERR_INVALID, "NULL pointer");
    return dspx::guarded([&] {
        const std::vector<double> w = dsp::hamming_window<double>(width, alpha, sym != 0);
        std::copy(w.begin(), w.end(), out);
        return (int)DSPX_OK;
    });
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

int dspx_lfilter_f64(const double* b, size_t nb, const double* a, size_t na, const double* x, double* y, size_t nchan,
                     size_t n, size_t stride, unsigned flags) {
    if (int rc = dspx::check_flags(flags)) return rc;
    Tail tail;
    if (int rc = make_tail(a, na, flags, tail)) return rc;
    return host_path(b, nb, nullptr, x, y, nchan, n, stride, -1, flags, tail);
}

int dspx_lfilter_f64_multi(const double* b, size_t nb, const double* a, size_t na, const double* x, double* y,
                           size_t nchan, size_t n, size_t stride, int ngpus, unsigned flags) {
    if (ngpus < 0) return fail(DSPX_ERR_INVALID, "ngpus must be >= 0");
    if (int rc = dspx::check_flags(flags)) return rc;
    Tail tail;
    if (int rc = make_tail(a, na, flags, tail)) return rc;
    return host_path(b, nb, nullptr, x, y, nchan, n, stride, ngpus, flags, tail);
}

int dspx_lfilter_f64_stream(const double* b, size_t nb, const double* a, size_t na, double* xhist, double* yhist,
                            const double* x, double* y, size_t nchan, size_t n, size_t stride, unsigned flags) {
    if (int rc = dspx::check_flags(flags)) return rc;
    Tail tail;
    if (int rc = make_tail(a, na, flags, tail)) return rc;
    const size_t hl = nb ? nb - 1 : 0, P = (size_t)tail.order;
    if ((hl && nchan && !xhist) || (P && nchan && !yhist)) return fail(DSPX_ERR_INVALID, "history is NULL");
    if (!(flags & DSPX_FIR_MFMA)) flags |= DSPX_FIR_VECTOR;  // as dspx_fir_f64_stream
    tail.y0 = P ? yhist : nullptr;
    if (int rc = host_path(b, nb, hl ? xhist : nullptr, x, y, nchan, n, stride, -1, flags, tail)) return rc;
    // the next chunk's histories: the last nb-1 samples of [xhist | x], the last na-1 outputs of [yhist | y]
    auto roll = [&](double* hist, size_t w, const double* data) {
        for (size_t c = 0; c < nchan && w && n; ++c) {
            double* h = hist + c * w;
            const double* d = data + c * stride;
            if (n >= w) {
                std::copy(d + (n - w), d + n, h);
            } else {
                std::copy(h + n, h + w, h);
                std::copy(d, d + n, h + (w - n));
            }
        }
    };
    roll(xhist, hl, x);
    roll(yhist, P, y);
    return DSPX_OK;
}

size_t dspx_lfilter_workspace_bytes(size_t na, size_t nchan, size_t n) {
    if (na <= 1 || na - 1 > (size_t)dspx::kIirMaxOrder) return 0;
    return scan_workspace_bytes((int)(na - 1), nchan, n);
}

int dspx_lfilter_f64_dev(const double* d_b, size_t nb, const double* a, size_t na, const double* d_x, size_t x_stride,
                         const double* d_xhist, size_t xhist_stride, const double* d_yhist, size_t yhist_stride,
                         double* d_y, size_t y_stride, size_t nchan, size_t n, void* d_work, size_t work_bytes,
                         unsigned flags, void* stream) {
    return lfilter_dev(d_b, nb, a, na, d_x, x_stride, d_xhist, xhist_stride, d_yhist, yhist_stride, d_y, y_stride, nchan,
                       n, d_work, work_bytes, "the workspace (dspx_lfilter_workspace_bytes)", flags, stream);
}

int dspx_lfilter_f64_dev_noalloc(const double* d_b, size_t nb, const double* a, size_t na, const double* d_x,
                                 size_t x_stride, const double* d_xhist, size_t xhist_stride, const double* d_yhist,
                                 size_t yhist_stride, double* d_y, size_t y_stride, size_t nchan, size_t n,
                                 unsigned flags, void* stream) {
    void* work = nullptr;
    if (int rc = static_scratch(&work)) return rc;
    return lfilter_dev(d_b, nb, a, na, d_x, x_stride, d_xhist, xhist_stride, d_yhist, yhist_stride, d_y, y_stride, nchan,
                       n, work, kStaticScratchBytes,
                       "the library's static scratch (DSPX_NOALLOC_SCRATCH_BYTES); use dspx_lfilter_f64_dev with a workspace",
                       flags, stream);
}

int dspx_fir_f32(const float* taps, size_t ntaps, const float* x, float* y, size_t n) {
    return dspx::guarded([&] { return narrow_host<float, float>(taps, ntaps, nullptr, 0, false, x, y, n); });
}

int dspx_fir_f32_from_f64(const double* taps, size_t ntaps, const double* x, float* y, size_t n) {
    return dspx::guarded([&] { return narrow_host<double, float>(taps, ntaps, nullptr, 0, false, x, y, n); });
}

int dspx_fir_f64_from_f32(const float* taps, size_t ntaps, const float* x, double* y, size_t n) {
    return dspx::guarded([&] { return narrow_host<float, double>(taps, ntaps, nullptr, 0, false, x, y, n); });
}

int dspx_lfilter_f32(const float* b, size_t nb, const double* a, size_t na, int tail_in_double, const float* x, float* y,
                     size_t n) {
    return dspx::guarded([&] { return narrow_host<float, float>(b, nb, a, na, tail_in_double != 0, x, y, n); });
}

int dspx_lfilter_f32_from_f64(const double* b, size_t nb, const double* a, size_t na, int tail_in_double, const double* x,
                              float* y, size_t n) {
    return dspx::guarded([&] { return narrow_host<double, float>(b, nb, a, na, tail_in_double != 0, x, y, n); });
}

int dspx_fir_f64_from_f32_dev(const float* d_taps, size_t ntaps, const float* d_x, size_t x_stride, const float* d_hist,
                              size_t hist_stride, double* d_y, size_t y_stride, size_t nchan, size_t n, void* stream) {
    return narrow_dev<float, double>(d_taps, ntaps, d_x, x_stride, d_hist, hist_stride, d_y, y_stride, nchan, n,
                                     static_cast<hipStream_t>(stream));
}

int dspx_fir_f32_dev(const float* d_taps, size_t ntaps, const float* d_x, size_t x_stride, const float* d_hist,
                     size_t hist_stride, float* d_y, size_t y_stride, size_t nchan, size_t n, void* stream) {
    return narrow_dev<float, float>(d_taps, ntaps, d_x, x_stride, d_hist, hist_stride, d_y, y_stride, nchan, n,
                                    static_cast<hipStream_t>(stream));
}

int dspx_fir_f32_from_f64_dev(const double* d_taps, size_t ntaps, const double* d_x, size_t x_stride,
                              const double* d_hist, size_t hist_stride, float* d_y, size_t y_stride, size_t nchan,
                              size_t n, void* stream) {
    return narrow_dev<double, float>(d_taps, ntaps, d_x, x_stride, d_hist, hist_stride, d_y, y_stride, nchan, n,
                                     static_cast<hipStream_t>(stream));
}

long dspx_fir_plan(size_t nchan, size_t n, int ngpus, dspx_piece* out, size_t cap) {
    if (ngpus < 1) return fail(DSPX_ERR_INVALID, "ngpus must be >= 1");
    if (cap && !out) return fail(DSPX_ERR_INVALID, "out is NULL");
    long total = 0;
    const int rc = dspx::guarded([&] {
        const auto plan = make_plan(nchan, n, ngpus);
        size_t k = 0;
        for (int g = 0; g < ngpus; ++g)
            for (const Piece& p : plan[g]) {
                if (k < cap) out[k] = dspx_piece{g, p.c, p.cnt, p.s0, p.len};
                ++k;
            }
        total = (long)k;
        return (int)DSPX_OK;
    });
    return rc == DSPX_OK ? total : (long)rc;
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

// Test hook, deliberately absent from include/dspx.h (see dspx_common.h): logical devices mapped onto the GPUs that exist.
extern "C" DSPX_API int dspx_test_alias_devices(int enable) {
    return dspx::g_alias_devices_exchange(enable != 0);
}
