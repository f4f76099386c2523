// Thin extern "C" driver around the UNMODIFIED reference header, compiled from
// where it lies (/root/reference/dsp/include/dsp.hpp) into oracle/_ref/ by
// oracle/Makefile.  TEST INFRASTRUCTURE ONLY: it validates oracle/fir_oracle.c
// and generates tests/golden/.  No reference source is copied into this file;
// it only calls the reference's public templates.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "dsp.hpp"

namespace {
template <typename F> int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::invalid_argument&) {
        return -1;
    } catch (const std::runtime_error&) {
        return -2;
    } catch (...) {
        return -3;
    }
}
}  // namespace

extern "C" {

double ref_sinc(double x) { return dsp::sinc<double>(x); }

int ref_hamming_window(size_t width, double alpha, int sym, double* out) {
    return guarded([&] {
        auto w = dsp::hamming_window<double>(width, alpha, sym != 0);
        std::memcpy(out, w.data(), w.size() * sizeof(double));
    });
}

int ref_firwin(size_t numtaps, const double* cutoffs, size_t ncut, int pass_zero, int scale, double* out) {
    return guarded([&] {
        std::vector<double> c(cutoffs, cutoffs + ncut);
        auto h = dsp::firwin<double>(numtaps, c, pass_zero != 0, scale != 0);
        std::memcpy(out, h.data(), h.size() * sizeof(double));
    });
}

int ref_firwin_scalar(size_t numtaps, double cutoff, double* out) {
    return guarded([&] {
        auto h = dsp::firwin<double>(numtaps, cutoff);
        std::memcpy(out, h.data(), h.size() * sizeof(double));
    });
}

int ref_lfilter(const double* b, size_t nb, const double* a, size_t na, const double* x, size_t n, double* y) {
    return guarded([&] {
        std::vector<double> vb(b, b + nb), va(a, a + na), vx(x, x + n);
        dsp::lfilter_ctx_t<double> ctx;
        dsp::lfilter<double>(ctx, vb, va, vx);
        if (!ctx.result.empty()) std::memcpy(y, ctx.result.data(), ctx.result.size() * sizeof(double));
    });
}

// lfilter(b, 1.0, x): the scalar-denominator overload named by the hot path.
int ref_lfilter_scalar_den(const double* b, size_t nb, double den, const double* x, size_t n, double* y) {
    return guarded([&] {
        std::vector<double> vb(b, b + nb), vx(x, x + n);
        const auto r = dsp::lfilter<double>(vb, den, vx);
        if (!r.empty()) std::memcpy(y, r.data(), r.size() * sizeof(double));
    });
}

int ref_lfilter_f32(const float* b, size_t nb, const float* x, size_t n, float* y) {
    return guarded([&] {
        std::vector<float> vb(b, b + nb), vx(x, x + n);
        const auto r = dsp::lfilter<float>(vb, 1.0f, vx);
        if (!r.empty()) std::memcpy(y, r.data(), r.size() * sizeof(float));
    });
}

int ref_lfilter_f32_from_f64(const double* b, size_t nb, const double* x, size_t n, float* y) {
    return guarded([&] {
        std::vector<double> vb(b, b + nb), vx(x, x + n);
        const auto r = dsp::lfilter<float>(vb, 1.0f, vx);
        if (!r.empty()) std::memcpy(y, r.data(), r.size() * sizeof(float));
    });
}

// lfilter<double> over containers of float: float products, double sum (dsp.hpp:363 with T = double)
int ref_lfilter_f64_from_f32(const float* b, size_t nb, const float* x, size_t n, double* y) {
    return guarded([&] {
        std::vector<float> vb(b, b + nb), vx(x, x + n);
        const auto r = dsp::lfilter<double>(vb, 1.0, vx);
        if (!r.empty()) std::memcpy(y, r.data(), r.size() * sizeof(double));
    });
}

// lfilter<float> with a recursive part, over containers of float and over containers of double
int ref_lfilter_f32_iir(const float* b, size_t nb, const float* a, size_t na, const float* x, size_t n, float* y) {
    return guarded([&] {
        std::vector<float> vb(b, b + nb), va(a, a + na), vx(x, x + n);
        const auto r = dsp::lfilter<float>(vb, va, vx);
        if (!r.empty()) std::memcpy(y, r.data(), r.size() * sizeof(float));
    });
}

int ref_lfilter_f32_from_f64_iir(const double* b, size_t nb, const double* a, size_t na, const double* x, size_t n, float* y) {
    return guarded([&] {
        std::vector<double> vb(b, b + nb), va(a, a + na), vx(x, x + n);
        const auto r = dsp::lfilter<float>(vb, va, vx);
        if (!r.empty()) std::memcpy(y, r.data(), r.size() * sizeof(float));
    });
}

int ref_integrate_wavelet(size_t precision, double* psi, double* x) {
    return guarded([&] {
        auto [p, t] = dsp::integrate_wavelet<double>(dsp::cwt_wavelet_t::MORLET, precision);
        std::memcpy(psi, p.data(), p.size() * sizeof(double));
        std::memcpy(x, t.data(), t.size() * sizeof(double));
    });
}

// result must hold nscales * n doubles.
int ref_cwt(const double* signal, size_t n, const double* scales, size_t nscales, const double* psi,
            const double* px, size_t npsi, double* result) {
    return guarded([&] {
        std::vector<double> vs(signal, signal + n), sc(scales, scales + nscales), vp(psi, psi + npsi),
          vx(px, px + npsi);
        const auto r = dsp::cwt<double>(vs, sc, vp, vx);
        std::memcpy(result, r.data(), r.size() * sizeof(double));
    });
}

// Runs the reference's own embedded known-answer tests for the hot path
// (they assert() on failure; returning at all means they passed).
int ref_run_hot_path_tests(void) {
    dsp::tests::test_sinc();
    dsp::tests::test_hamming_window();
    dsp::tests::test_firwin();
    dsp::tests::test_lfilter();
    dsp::tests::test_integrate_wavelet();
    dsp::tests::test_cwt();
    return 0;
}

}  // extern "C"
