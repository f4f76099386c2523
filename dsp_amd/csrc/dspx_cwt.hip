// Wavelet-bank entry points of the C ABI (include/dspx.h): host-side tables of the reference's
// cwt (dsp.hpp:1040-1219) and the bank-mode launch of the FIR kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "dsp.hpp"
#include "dspx.h"
#include "dspx_common.h"
#include "fir_kernels.h"

using dspx::fail;

namespace {

// One sorted scale: the tap vector is psi gathered through its index table, reversed.
struct RowTable {
    double scale;
    double factor;       // -sqrt(scale)
    size_t start;        // first kept sample of the differenced full convolution
    std::vector<size_t> index;  // index[j] = floor(j / max(scale*x_step, EPS)), cut at npsi
};

struct Tables {
    std::vector<RowTable> rows;  // ascending scale
};

// Argument checks and per-scale tables, in the reference's order and with its messages.
// `psi` may be null when only the table shapes are wanted (planning).
int build_tables(const double* scales, size_t nscales, size_t npsi, const double* wx, size_t nx, size_t n,
                 bool check_signal, Tables& out) {
    if (npsi <= 2) return fail(DSPX_ERR_INVALID, "The size of wavelet_psi must be greater than 2.");
    if (nx <= 2) return fail(DSPX_ERR_INVALID, "The size of wavelet_x must be greater than 2.");
    if (nx > npsi) return fail(DSPX_ERR_INVALID, "The size of wavelet_x must be less than or equal to wavelet_psi.");
    if (!wx) return fail(DSPX_ERR_INVALID, "wavelet_x is NULL");
    const double x_step = wx[1] - wx[0];
    const double x_range = wx[nx - 1] - wx[0];
    if (std::abs(x_step) <= dsp::EPS) return fail(DSPX_ERR_INVALID, "The step of wavelet_x must be greater than 0.");
    if (std::abs(x_range) <= dsp::EPS) return fail(DSPX_ERR_INVALID, "The range of wavelet_x must be greater than 0.");
    if (nscales == 0) return fail(DSPX_ERR_DOMAIN, "The size of ctx.scales must be greater than 0.");
    if (!scales) return fail(DSPX_ERR_INVALID, "scales is NULL");
    if (check_signal && n == 0) return fail(DSPX_ERR_INVALID, "The size of signal must be greater than 0.");

    std::vector<double> sorted(scales, scales + nscales);
    std::sort(sorted.begin(), sorted.end());
    const size_t cap = static_cast<size_t>(std::ceil(sorted.back() * x_range)) + 1;
    out.rows.clear();
    out.rows.reserve(nscales);
    for (const double scale : sorted) {
        RowTable row;
        row.scale = scale;
        row.factor = -std::sqrt(scale);
        const size_t last = std::min(static_cast<size_t>(std::ceil(scale * x_range)) + 1, cap);
        const double step = std::max(scale * x_step, dsp::EPS);
        for (size_t j = 0; j < last; ++j) {
            const size_t k = static_cast<size_t>(std::floor(static_cast<double>(j) / step));
            if (k >= npsi) break;
            row.index.push_back(k);
        }
        const size_t len = row.index.size();
        if (len < 1) return fail(DSPX_ERR_DOMAIN, "Selected scale is too large.");
        // the reference computes d = (len - 2) / 2 on the differenced convolution and rejects d < EPS
        if (len <= 2) return fail(DSPX_ERR_DOMAIN, "Selected scale is too small.");
        row.start = (len - 2) / 2;  // floor((len - 2) / 2.0)
        out.rows.push_back(std::move(row));
    }
    return DSPX_OK;
}

// Greedy longest-first assignment of rows to G devices by tap count.
std::vector<int> assign_rows(const Tables& t, int G) {
    std::vector<size_t> order(t.rows.size());
    std::iota(order.begin(), order.end(), size_t(0));
    std::stable_sort(order.begin(), order.end(),
                     [&](size_t a, size_t b) { return t.rows[a].index.size() > t.rows[b].index.size(); });
    std::vector<size_t> load(G, 0);
    std::vector<int> dev(t.rows.size(), 0);
    for (size_t r : order) {
        const int g = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        dev[r] = g;
        load[g] += t.rows[r].index.size();
    }
    return dev;
}

}  // namespace

// Device-side bank: packed taps and row descriptors (longest rows first, so that the launch does
// not end on its longest workgroups), plus which result row each descriptor writes.
struct dspx_cwt_bank {
    int device = 0;
    size_t nrows = 0, total_taps = 0;
    double* d_taps = nullptr;
    dspx::BankRow* d_rows = nullptr;
    std::vector<size_t> result_row;  // descriptor -> row of the caller's result
    std::vector<int> row_taps;       // descriptor -> tap count (descending)
};

namespace {

// Builds the device bank for the rows `pick` of `t`; result_row[i] is what pick[i] writes.
int make_bank(const Tables& t, const double* psi, const std::vector<size_t>& pick,
              const std::vector<size_t>& result_row, dspx_cwt_bank** out) {
    auto* bank = new dspx_cwt_bank();
    std::vector<size_t> order(pick.size());
    std::iota(order.begin(), order.end(), size_t(0));
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        return t.rows[pick[a]].index.size() > t.rows[pick[b]].index.size();
    });
    std::vector<double> taps;
    std::vector<dspx::BankRow> rows;
    for (size_t i : order) {
        const RowTable& r = t.rows[pick[i]];
        const size_t len = r.index.size();
        dspx::BankRow d;
        d.taps_off = taps.size();
        d.out_off = result_row[i];  // row number; the kernel scales it by the row stride
        d.scale = r.factor;
        d.ntaps = (int)len;
        // c[t] = sum_k h[k] * x[t - k]; kept sample i is the difference c[start+i+1] - c[start+i]
        d.advance = (int)r.start;
        d.win_steps = dspx::fir_mfma_window_steps(d.ntaps);
        for (size_t k = 0; k < len; ++k) taps.push_back(psi[r.index[len - 1 - k]]);
        if (taps.size() & 1) taps.push_back(0.0);  // keep every row's taps 16-byte aligned
        rows.push_back(d);
        bank->result_row.push_back(result_row[i]);
        bank->row_taps.push_back(d.ntaps);
    }
    bank->nrows = rows.size();
    bank->total_taps = 0;
    for (const auto& d : rows) bank->total_taps += (size_t)d.ntaps;
    auto body = [&]() -> int {
        DSPX_HIP(hipGetDevice(&bank->device));
        DSPX_HIP(hipMalloc(&bank->d_taps, std::max<size_t>(taps.size(), 1) * sizeof(double)));
        DSPX_HIP(hipMalloc(&bank->d_rows, std::max<size_t>(rows.size(), 1) * sizeof(dspx::BankRow)));
        DSPX_HIP(hipMemcpy(bank->d_taps, taps.data(), taps.size() * sizeof(double), hipMemcpyHostToDevice));
        DSPX_HIP(hipMemcpy(bank->d_rows, rows.data(), rows.size() * sizeof(dspx::BankRow), hipMemcpyHostToDevice));
        return DSPX_OK;
    };
    const int rc = body();
    if (rc != DSPX_OK) {
        dspx_cwt_bank_destroy(bank);
        return rc;
    }
    *out = bank;
    return DSPX_OK;
}

int apply_bank(const dspx_cwt_bank* bank, const double* d_signal, size_t n, double* d_result, size_t row_stride,
               unsigned flags, hipStream_t stream) {
    if (int rc = dspx::check_flags(flags)) return rc;
    if (!bank) return fail(DSPX_ERR_INVALID, "bank is NULL");
    if (n == 0) return fail(DSPX_ERR_INVALID, "The size of signal must be greater than 0.");
    if (!d_signal || !d_result) return fail(DSPX_ERR_INVALID, "NULL pointer");
    if (row_stride < n) return fail(DSPX_ERR_INVALID, "row_stride (%zu) < n (%zu)", row_stride, n);
    const bool vec2 = dspx::aligned16(d_signal) && dspx::aligned16(d_result) && (row_stride % 2 == 0);
    // Each row runs on the pipe its own tap count selects, whatever other rows the bank holds, so that a row's bits
    // do not depend on how scales are sharded.  Rows are stored longest first: matrix-pipe rows, then vector-pipe
    // rows are consecutive runs of the descriptors.
    size_t first = 0;
    for (int part = 0; part < 2 && first < bank->nrows; ++part) {
        const bool mfma = part == 0;
        size_t end = first;
        if (mfma) {
            while (end < bank->nrows && dspx::use_matrix_pipe(flags, (size_t)bank->row_taps[end])) ++end;
        } else {
            end = bank->nrows;
        }
        const size_t count = end - first;
        if (count == 0) continue;
        const size_t stride = (size_t)(mfma ? dspx::fir_mfma_bank_stride() : dspx::fir_bank_stride());
        const size_t tiles_per_row = (n + stride - 1) / stride;
        const size_t run = (tiles_per_row + 7) / 8;  // tiles per row and XCD
        const size_t max_rows = dspx::kMaxGrid / (8 * run);
        if (max_rows == 0) return fail(DSPX_ERR_INVALID, "signal too long for one launch: %zu", n);
        for (size_t r0 = 0; r0 < count; r0 += max_rows) {
            const size_t nr = std::min(max_rows, count - r0);
            dspx::FirShape s{};
            s.y_stride = row_stride;
            s.n = (long long)n;
            s.tiles_per_chan = (unsigned)tiles_per_row;
            // positions of the rows' tiles: every row is cut into 8 runs of `run` tiles, one per XCD (vector kernel: one
            // workgroup per position, those past a row's last tile exit at once; matrix kernel: walked by its workgroups)
            s.total_tiles = (unsigned)(nr * 8 * run);
            dspx::set_divider(s, (unsigned)run);
            if (mfma)
                DSPX_HIP(dspx::launch_fir_mfma_bank_f64(bank->d_taps, bank->d_rows + first + r0, d_signal, d_result,
                                                        s, vec2, (flags & DSPX_FIR_TILED) != 0, stream));
            else
                DSPX_HIP(dspx::launch_fir_bank_f64(bank->d_taps, bank->d_rows + first + r0, d_signal, d_result, s,
                                                   (flags & DSPX_FIR_EXACT) != 0, vec2, stream));
        }
        first = end;
    }
    return DSPX_OK;
}

// Host-pointer transform on one device: the rows `pick` of the sorted scales.  The result of a pass (rows x n
// doubles) lives in device memory until it is downloaded; when all rows do not fit in half of the free device
// memory the rows are transformed in several passes (DSPX_CWT_ROWS_PER_PASS caps the rows of a pass further).
int run_cwt_device(int device, const Tables& t, const double* psi, const std::vector<size_t>& pick,
                   const double* signal, size_t n, double* result, unsigned flags) {
    if (pick.empty()) return DSPX_OK;
    DSPX_HIP(hipSetDevice(device));
    const size_t pitch = (n + 1) & ~size_t(1);
    size_t free_b = 0, total_b = 0;
    DSPX_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t sig_bytes = n * sizeof(double);
    const size_t budget = free_b / 2 > sig_bytes ? free_b / 2 - sig_bytes : 0;
    size_t per_pass = std::max<size_t>(1, std::min(pick.size(), budget / (pitch * sizeof(double))));
    if (const char* e = std::getenv("DSPX_CWT_ROWS_PER_PASS")) {
        const long cap = std::atol(e);
        if (cap > 0) per_pass = std::min(per_pass, (size_t)cap);
    }
    dspx_cwt_bank* bank = nullptr;
    double *d_sig = nullptr, *d_res = nullptr;
    auto body = [&]() -> int {
        DSPX_HIP(hipMalloc(&d_sig, sig_bytes));
        DSPX_HIP(hipMalloc(&d_res, pitch * per_pass * sizeof(double)));
        DSPX_HIP(hipMemcpy(d_sig, signal, sig_bytes, hipMemcpyHostToDevice));
        for (size_t r0 = 0; r0 < pick.size(); r0 += per_pass) {
            const std::vector<size_t> rows(pick.begin() + (long)r0, pick.begin() + (long)std::min(pick.size(), r0 + per_pass));
            std::vector<size_t> local(rows.size());
            std::iota(local.begin(), local.end(), size_t(0));
            if (int rc = make_bank(t, psi, rows, local, &bank)) return rc;
            if (int rc = apply_bank(bank, d_sig, n, d_res, pitch, flags, nullptr)) return rc;
            DSPX_HIP(hipDeviceSynchronize());
            if (int rc = dspx::download_rows(device, d_res, pitch, rows.size(), n, result, n, rows.data())) return rc;
            dspx_cwt_bank_destroy(bank);
            bank = nullptr;
        }
        return DSPX_OK;
    };
    const int rc = body();
    const std::string keep = dspx::g_last_error;
    if (bank) dspx_cwt_bank_destroy(bank);
    if (d_sig) (void)hipFree(d_sig);
    if (d_res) (void)hipFree(d_res);
    if (rc != DSPX_OK) dspx::g_last_error = keep;
    return rc;
}

int cwt_host(const double* signal, size_t n, const double* scales, size_t nscales, const double* psi, size_t npsi,
             const double* wx, size_t nx, double* result, int ngpus, unsigned flags) {
    if (int rc = dspx::check_flags(flags)) return rc;
    if (!psi) return fail(DSPX_ERR_INVALID, "wavelet_psi is NULL");
    Tables t;
    if (int rc = build_tables(scales, nscales, npsi, wx, nx, n, true, t)) return rc;
    if (!signal || !result) return fail(DSPX_ERR_INVALID, "signal or result is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DSPX_ERR_NO_DEVICE, "no HIP device");
    int cur = 0;
    DSPX_HIP(hipGetDevice(&cur));
    if (ngpus < 0) {
        std::vector<size_t> all(t.rows.size());
        std::iota(all.begin(), all.end(), size_t(0));
        return run_cwt_device(cur, t, psi, all, signal, n, result, flags);
    }
    const int G = ngpus == 0 ? ndev : ngpus;
    const bool aliased = dspx::devices_aliased();  // read once, here
    if (G > dspx::logical_device_limit(aliased, ndev))
        return fail(DSPX_ERR_INVALID, "ngpus (%d) exceeds visible devices (%d)", G, ndev);
    const std::vector<int> dev = assign_rows(t, G);
    std::vector<std::vector<size_t>> picks(G);
    for (size_t r = 0; r < dev.size(); ++r) picks[dev[r]].push_back(r);
    std::vector<int> rcs(G, DSPX_OK);
    std::vector<std::string> msgs(G);
    {
        dspx::Joiner workers;  // joins the threads that did start if one cannot
        for (int g = 0; g < G; ++g)
            workers.threads.emplace_back([&, g] {
                rcs[g] = dspx::guarded([&] {
                    return run_cwt_device(dspx::physical_device(aliased, g, ndev), t, psi, picks[g], signal, n, result, flags);
                });
                if (rcs[g] != DSPX_OK) msgs[g] = dspx::g_last_error;
            });
    }
    (void)hipSetDevice(cur);
    for (int g = 0; g < G; ++g)
        if (rcs[g] != DSPX_OK) return fail(rcs[g], "device %d: %s", g, msgs[g].c_str());
    return DSPX_OK;
}

}  // namespace

extern "C" {

int dspx_integrate_wavelet_morlet_f64(size_t precision, double lower, double upper, double* psi, double* x) {
    if (!psi || !x) return fail(DSPX_ERR_INVALID, "NULL pointer");
    return dspx::guarded([&] {
        const auto [p, t] = dsp::integrate_wavelet<double>(dsp::cwt_wavelet_t::MORLET, precision, lower, upper);
        std::copy(p.begin(), p.end(), psi);
        std::copy(t.begin(), t.end(), x);
        return (int)DSPX_OK;
    });
}

int dspx_cwt_f64(const double* signal, size_t n, const double* scales, size_t nscales, const double* psi, size_t npsi,
                 const double* wx, size_t nx, double* result, unsigned flags) {
    return dspx::guarded([&] { return cwt_host(signal, n, scales, nscales, psi, npsi, wx, nx, result, -1, flags); });
}

int dspx_cwt_f64_multi(const double* signal, size_t n, const double* scales, size_t nscales, const double* psi,
                       size_t npsi, const double* wx, size_t nx, double* result, int ngpus, unsigned flags) {
    if (ngpus < 0) return fail(DSPX_ERR_INVALID, "ngpus must be >= 0");
    return dspx::guarded([&] { return cwt_host(signal, n, scales, nscales, psi, npsi, wx, nx, result, ngpus, flags); });
}

int dspx_cwt_plan(const double* scales, size_t nscales, size_t npsi, const double* wx, size_t nx, int ngpus,
                  int* device_of_row, size_t* ntaps_of_row) {
    if (ngpus < 1) return fail(DSPX_ERR_INVALID, "ngpus must be >= 1");
    return dspx::guarded([&] {
        Tables t;
        if (int rc = build_tables(scales, nscales, npsi, wx, nx, 0, false, t)) return rc;
        const std::vector<int> dev = assign_rows(t, ngpus);
        for (size_t r = 0; r < t.rows.size(); ++r) {
            if (device_of_row) device_of_row[r] = dev[r];
            if (ntaps_of_row) ntaps_of_row[r] = t.rows[r].index.size();
        }
        return (int)DSPX_OK;
    });
}

int dspx_cwt_bank_create(const double* scales, size_t nscales, const double* psi, size_t npsi, const double* wx,
                         size_t nx, dspx_cwt_bank** bank) {
    if (!bank) return fail(DSPX_ERR_INVALID, "bank is NULL");
    if (!psi) return fail(DSPX_ERR_INVALID, "wavelet_psi is NULL");
    *bank = nullptr;
    return dspx::guarded([&] {
        Tables t;
        if (int rc = build_tables(scales, nscales, npsi, wx, nx, 0, false, t)) return rc;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DSPX_ERR_NO_DEVICE, "no HIP device");
        std::vector<size_t> all(t.rows.size());
        std::iota(all.begin(), all.end(), size_t(0));
        return make_bank(t, psi, all, all, bank);
    });
}

size_t dspx_cwt_bank_rows(const dspx_cwt_bank* bank) { return bank ? bank->nrows : 0; }
size_t dspx_cwt_bank_total_taps(const dspx_cwt_bank* bank) { return bank ? bank->total_taps : 0; }

int dspx_cwt_bank_apply_dev(const dspx_cwt_bank* bank, const double* d_signal, size_t n, double* d_result,
                            size_t row_stride, unsigned flags, void* stream) {
    return apply_bank(bank, d_signal, n, d_result, row_stride, flags, static_cast<hipStream_t>(stream));
}

void dspx_cwt_bank_destroy(dspx_cwt_bank* bank) {
    if (!bank) return;
    if (bank->d_taps) (void)hipFree(bank->d_taps);
    if (bank->d_rows) (void)hipFree(bank->d_rows);
    delete bank;
}

}  // extern "C"
