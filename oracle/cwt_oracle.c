/*
 * CPU oracle for the wavelet bank that sits on top of the FIR kernel.
 * TEST INFRASTRUCTURE ONLY (see fir_oracle.c for the rules that apply).
 *
 * Parity status: PINNED against oracle/_ref (bit-for-bit) and against the golden
 * literals of the reference's test_integrate_wavelet / test_cwt
 * (dsp.hpp:2096-2783) by tests/test_oracle_cwt.py.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#define ORACLE_EPS 1.0e-20 /* dsp.hpp:39 */

/* dsp.hpp:934-985 — Morlet psi(t) = exp(-t^2/2) cos(5t) on 2^precision points
 * over [lower, upper], replaced by its running sum times the grid step.
 * Returns 0, or -1 for the argument errors the reference throws on. */
int oracle_integrate_wavelet_morlet(size_t precision, double lower, double upper, double* psi, double* x) {
    if (precision <= 2 || precision >= sizeof(size_t) * 8) return -1;
    if (lower >= upper) return -1;
    const size_t n = (size_t)1 << precision;
    const double space = upper - lower;
    const double nm1 = (double)(n - 1);
    for (size_t i = 0; i < n; ++i) {
        const double t = lower + (space * ((double)i / nm1));
        psi[i] = exp(-0.5 * t * t) * cos(5.0 * t);
        x[i] = t;
    }
    const double step = x[1] - x[0];
    double sum = 0.0;
    for (size_t i = 0; i < n; ++i) {
        sum += psi[i];
        psi[i] = sum * step;
    }
    return 0;
}

static int cmp_double(const void* a, const void* b) {
    const double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

/* dsp.hpp:1141-1150 — index table of one scale.  idx must hold idx_cap entries.
 * Returns the table length L. */
size_t oracle_cwt_scale_indices(double scale, double x_step, double x_range, size_t npsi, size_t idx_cap,
                                size_t* idx) {
    size_t end = (size_t)ceil(scale * x_range) + 1;
    if (end > idx_cap) end = idx_cap;
    double sms = scale * x_step;
    if (sms < ORACLE_EPS) sms = ORACLE_EPS; /* std::max(sms, EPS), dsp.hpp:1143 */
    size_t len = 0;
    for (size_t j = 0; j < end; ++j) {
        const size_t k = (size_t)floor((double)j / sms);
        if (k >= npsi) break;
        idx[len++] = k;
    }
    return len;
}

/* dsp.hpp:1040-1219 with the constructor of cwt_ctx_t (dsp.hpp:1008-1015):
 * scales are sorted ascending first; result is nscales x n row-major.
 * Returns 0; -1 for invalid arguments; -2 "scale too large"; -3 "scale too small". */
int oracle_cwt(const double* signal, size_t n, const double* scales_in, size_t nscales, const double* psi,
               size_t npsi, const double* wx, size_t nx, double* result) {
    if (npsi <= 2 || nx <= 2 || nx > npsi) return -1;
    const double x_step = wx[1] - wx[0];
    const double x_range = wx[nx - 1] - wx[0];
    if (fabs(x_step) <= ORACLE_EPS || fabs(x_range) <= ORACLE_EPS) return -1;
    if (nscales == 0) return -2;
    if (n == 0) return -1;

    double* scales = (double*)malloc(sizeof(double) * nscales);
    for (size_t i = 0; i < nscales; ++i) scales[i] = scales_in[i];
    qsort(scales, nscales, sizeof(double), cmp_double);

    const size_t idx_cap = (size_t)ceil(scales[nscales - 1] * x_range) + 1;
    size_t* idx = (size_t*)malloc(sizeof(size_t) * (idx_cap ? idx_cap : 1));
    double* coef = (double*)malloc(sizeof(double) * (n + idx_cap));
    int rc = 0;
    size_t pos = 0;
    for (size_t s = 0; s < nscales && rc == 0; ++s) {
        const double scale = scales[s];
        const size_t len = oracle_cwt_scale_indices(scale, x_step, x_range, npsi, idx_cap, idx);
        if (len < 1) {
            rc = -2;
            break;
        }
        if (len < 2) { /* the reference aborts here (ENSURE_TRUE(len_diff >= n_signal), dsp.hpp:1193) */
            rc = -3;
            break;
        }
        const size_t len_conv = n + len - 1;
        for (size_t j = 0; j < len_conv; ++j) coef[j] = 0.0;
        for (size_t j = 0; j < n; ++j) { /* dsp.hpp:1171-1179 */
            const double sj = signal[j];
            for (size_t k = 0; k < len; ++k) coef[j + k] += sj * psi[idx[len - 1 - k]];
        }
        const double nss = -sqrt(scale);
        for (size_t j = 1; j < len_conv; ++j) { /* dsp.hpp:1185-1191 */
            coef[j - 1] = coef[j] - coef[j - 1];
            coef[j - 1] *= nss;
        }
        const size_t len_diff = len_conv - 1;
        const double d = (double)(len_diff - n) / 2.0;
        if (d < ORACLE_EPS) {
            rc = -3;
            break;
        }
        const size_t start = (size_t)floor(d);
        const size_t end_max = start + n;
        size_t end = len_diff - (size_t)ceil(d);
        if (end > end_max) end = end_max;
        for (size_t j = start; j < end; ++j) result[pos++] = coef[j];
        for (size_t j = end; j < end_max; ++j) result[pos++] = NAN;
    }
    free(scales), free(idx), free(coef);
    return rc;
}
