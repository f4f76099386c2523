// Exercises include/dsp.hpp exactly as a user of the reference header would
// (README.md:13-27 of the reference, and its test_firwin / test_lfilter, dsp.hpp:1476-1599):
// same names, same template arguments, same call shapes.  Results are printed as
// hex floats so that the Python side can compare bits.
//
//   dropin_check design <numtaps> <cutoff>            firwin only (no GPU needed)
//   dropin_check readme                               firwin(100, 0.5) + lfilter over sin(n), n < 1024
//   dropin_check filter <numtaps> <cutoff>  < samples lfilter(firwin(...), 1.0, x), x from stdin
//   dropin_check api                                  overloads, ctx reuse, containers, error behaviour
//   dropin_check cwt <nscales>              < samples integrate_wavelet(MORLET, 10) + cwt over integer scales 1..nscales
//   dropin_check wavelet <precision>                  integrate_wavelet only (no GPU needed)
//   dropin_check narrow <b...> / <a...>     < samples the instantiations over narrow element types: lfilter<double> over
//                                                     vector<float>, lfilter<float> with a recursive part over float and over
//                                                     double containers (b and a from the command line, "/" between them)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "dsp.hpp"

static void dump(const char* tag, const std::vector<double>& v) {
    std::printf("%s %zu\n", tag, v.size());
    for (double d : v) std::printf("%a\n", d);
}

template <typename E, typename F> static bool throws(F&& f) {
    try {
        f();
    } catch (const E&) {
        return true;
    } catch (...) {
        return false;
    }
    return false;
}

#define CHECK(cond)                                                      \
    do {                                                                 \
        if (!(cond)) {                                                   \
            std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond);   \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main(int argc, char** argv) {
    using namespace dsp;
    const std::string mode = argc > 1 ? argv[1] : "";
    if (mode == "design" && argc == 4) {
        dump("taps", firwin<double>(std::strtoul(argv[2], nullptr, 10), std::strtod(argv[3], nullptr)));
        return 0;
    }
    if (mode == "readme") {
        std::vector<double> data(1024);
        for (size_t n = 0; n < data.size(); ++n) data[n] = std::sin(static_cast<double>(n));
        auto w = firwin(100, 0.5);
        auto r = lfilter(w, 1.0, data);
        dump("x", data);
        dump("taps", w);
        dump("y", r);
        return 0;
    }
    if (mode == "filter" && argc == 4) {
        std::vector<double> x;
        std::string tok;
        while (std::cin >> tok) x.push_back(std::strtod(tok.c_str(), nullptr));
        auto w = firwin<double>(std::strtoul(argv[2], nullptr, 10), std::strtod(argv[3], nullptr));
        auto r = lfilter<double>(w, 1.0, x);
        dump("y", r);
        // the single-precision instantiation of the reference's test_lfilter (dsp.hpp:1590-1597): float result over the
        // same double taps and data
        auto rf = lfilter<float>(w, 1.0, x);
        dump("yf", std::vector<double>(rf.begin(), rf.end()));
        return 0;
    }
    if (mode == "narrow") {
        std::vector<double> b, a, x;
        bool second = false;
        for (int i = 2; i < argc; ++i) {
            if (std::strcmp(argv[i], "/") == 0)
                second = true;
            else
                (second ? a : b).push_back(std::strtod(argv[i], nullptr));
        }
        std::string tok;
        while (std::cin >> tok) x.push_back(std::strtod(tok.c_str(), nullptr));
        const std::vector<float> bf(b.begin(), b.end()), af(a.begin(), a.end()), xf(x.begin(), x.end());
        dump("f64_from_f32", lfilter<double>(bf, 1.0, xf));         // float products, double sum (dsp.hpp:363, T = double)
        const auto r1 = lfilter<float>(bf, af, xf);                  // tail in float
        dump("f32_iir", std::vector<double>(r1.begin(), r1.end()));
        const auto r2 = lfilter<float>(b, a, x);                     // products and tail in double, float result
        dump("f32_from_f64_iir", std::vector<double>(r2.begin(), r2.end()));
        dump("f64_from_f32_iir", lfilter<double>(bf, a, xf));        // float products, double sum, double tail
        return 0;
    }
    if (mode == "wavelet" && argc == 3) {
        auto [psi, x] = integrate_wavelet<double>(cwt_wavelet_t::MORLET, std::strtoul(argv[2], nullptr, 10));
        dump("psi", psi);
        dump("x", x);
        return 0;
    }
    if (mode == "cwt" && argc == 3) {  // the call shape of the reference's README.md:44-54 and test_cwt (dsp.hpp:2764-2766)
        std::vector<double> data;
        std::string tok;
        while (std::cin >> tok) data.push_back(std::strtod(tok.c_str(), nullptr));
        std::vector<int> scales(std::strtoul(argv[2], nullptr, 10));
        for (size_t i = 0; i < scales.size(); ++i) scales[i] = static_cast<int>(i) + 1;
        auto [psi, x] = integrate_wavelet<double>(cwt_wavelet_t::MORLET, 10);
        auto r = cwt<double>(data, scales, psi, x);
        dump("cwt", r);
        // the ctx overload gives the same rows and reports the shape
        std::vector<double> fscales(scales.begin(), scales.end());
        cwt_ctx_t<double, std::vector<double>> ctx(fscales);
        cwt(ctx, data, psi, x);
        if (ctx.result != r || ctx.shape != std::vector<size_t>{scales.size(), data.size()}) {
            std::printf("FAIL ctx overload\n");
            return 1;
        }
        return 0;
    }
    if (mode == "api") {
        // argument errors of the design functions (reference dsp.hpp:76-78, 200-215)
        CHECK(throws<std::invalid_argument>([] { firwin<double>(100, std::vector<double>{}); }));
        CHECK(throws<std::invalid_argument>([] { firwin<double>(100, 0.5, false); }));
        CHECK(throws<std::invalid_argument>([] { firwin<double>(1, 0.5); }));
        CHECK(throws<std::invalid_argument>([] { hamming_window<double>(1); }));
        CHECK(firwin<double>(101, 0.5, false).size() == 101);
        CHECK(firwin<float>(33, 0.25f).size() == 33);
        CHECK(std::abs(sinc<double>(0.0) - 1.0) < 1e-15);
        CHECK(throws<std::invalid_argument>([] { integrate_wavelet<double>(cwt_wavelet_t::MORLET, 2); }));
        CHECK(throws<std::invalid_argument>([] { integrate_wavelet<double>(cwt_wavelet_t::MORLET, 10, 1.0, 1.0); }));
        CHECK(integrate_wavelet<double>(cwt_wavelet_t::MORLET, 4).first.size() == 16);
        if (argc > 2 && std::strcmp(argv[2], "nogpu") == 0) {
            std::printf("OK (design only)\n");
            return 0;
        }
        std::vector<double> x(5000);
        for (size_t i = 0; i < x.size(); ++i) x[i] = std::cos(0.37 * static_cast<double>(i));
        auto w = firwin<double>(64, 0.3);
        // the three overloads agree
        auto a = lfilter<double>(w, 1.0, x);
        auto b = lfilter<double>(w, std::vector<double>{1.0}, x);
        lfilter_ctx_t<double> ctx;
        lfilter(ctx, w, std::vector<double>{1.0}, x);
        CHECK(a == b && a == ctx.result && a.size() == x.size());
        // the denominator value is never read (reference dsp.hpp:366)
        CHECK(lfilter<double>(w, 2.0, x) == a);
        // ctx.result is reused when the length matches, resized when it differs
        const double* before = ctx.result.data();
        lfilter(ctx, w, std::array<double, 1>{1.0}, x);
        CHECK(ctx.result.data() == before);
        std::vector<double> shorter(x.begin(), x.begin() + 100);
        lfilter(ctx, w, std::array<double, 1>{1.0}, shorter);
        CHECK(ctx.result.size() == 100);
        for (size_t i = 0; i < 100; ++i) CHECK(ctx.result[i] == a[i]);
        // non-contiguous container, std::array taps
        std::deque<double> dq(x.begin(), x.end());
        CHECK(lfilter<double>(w, 1.0, dq) == a);
        // samples or taps of another arithmetic type: the reference forms each product in double after
        // promoting the operand (dsp.hpp:363), which is what converting the container first does
        {
            std::vector<float> xf(x.size());
            std::vector<double> xf_as_double(x.size());
            std::vector<int> xi(x.size());
            std::vector<double> xi_as_double(x.size());
            for (size_t i = 0; i < x.size(); ++i) {
                xf[i] = static_cast<float>(x[i]);
                xf_as_double[i] = static_cast<double>(xf[i]);
                xi[i] = static_cast<int>(1000.0 * x[i]);
                xi_as_double[i] = static_cast<double>(xi[i]);
            }
            CHECK(lfilter<double>(w, 1.0, xf) == lfilter<double>(w, 1.0, xf_as_double));
            CHECK(lfilter<double>(w, 1.0, xi) == lfilter<double>(w, 1.0, xi_as_double));
            std::vector<float> wf(w.begin(), w.end());
            std::vector<double> wf_as_double(wf.begin(), wf.end());
            CHECK(lfilter<double>(wf, 1.0, x) == lfilter<double>(wf_as_double, 1.0, x));
        }
        // lfilter<float>: float sums; products in double for containers of double, in float for containers of float,
        // each compared bit for bit with the reference's loop written out here (dsp.hpp:356-365)
        {
            std::vector<float> wf(w.begin(), w.end()), xf(x.begin(), x.end());
            std::vector<float> want_dd(x.size()), want_ff(x.size());
            for (size_t i = 0; i < x.size(); ++i) {
                float sum_dd = 0.0f, sum_ff = 0.0f;
                for (size_t j = 0; j < std::min(w.size(), i + 1); ++j) {
                    sum_dd += static_cast<float>(w[j] * x[i - j]);
                    sum_ff += static_cast<float>(wf[j] * xf[i - j]);
                }
                want_dd[i] = sum_dd, want_ff[i] = sum_ff;
            }
            CHECK(lfilter<float>(w, 1.0f, x) == want_dd);
            CHECK(lfilter<float>(wf, 1.0f, xf) == want_ff);
            lfilter_ctx_t<float> fctx;
            lfilter<float>(fctx, wf, std::vector<float>{1.0f}, xf);
            CHECK(fctx.result == want_ff);
            CHECK(lfilter<float>(wf, 1.0f, std::vector<float>{}).empty());
            // a recursive part in single precision (dsp.hpp:366-368 with T = float): the tail in float, term by term
            {
                const std::vector<float> af{2.0f, -0.5f, 0.25f};  // (the first entry is never read)
                std::vector<float> want = want_ff;
                for (size_t i = 0; i < want.size(); ++i)
                    for (size_t j = 1; j < std::min(af.size(), i + 1); ++j) want[i] -= af[j] * want[i - j];
                CHECK(lfilter<float>(wf, af, xf) == want);
                CHECK(throws<std::invalid_argument>([&] { lfilter<float>(wf, std::vector<float>(11, 0.1f), xf); }));  // > 8 poles
            }
            // lfilter<double> over float containers: float products, double sum (dsp.hpp:363 with T = double)
            {
                std::vector<double> want(xf.size());
                for (size_t i = 0; i < xf.size(); ++i) {
                    double sum = 0.0;
                    for (size_t j = 0; j < std::min(wf.size(), i + 1); ++j) sum += static_cast<double>(wf[j] * xf[i - j]);
                    want[i] = sum;
                }
                CHECK(lfilter<double>(wf, 1.0, xf) == want);
            }
        }
        // empty input -> empty output, empty numerators -> zeros
        CHECK(lfilter<double>(w, 1.0, std::vector<double>{}).empty());
        auto z = lfilter<double>(std::vector<double>{}, 1.0, x);
        CHECK(z.size() == x.size());
        for (double v : z) CHECK(v == 0.0);
        // streaming addition: the result does not depend on where the stream is cut (bit for bit),
        // and agrees with the one-shot call to rounding
        {
            std::vector<std::vector<double>> joined(2);
            const std::vector<std::vector<size_t>> cuts = {{0, 1, 30, 63, 64, 2500, 2501, x.size()},
                                                           {0, 777, 778, 4000, x.size()}};
            for (size_t v = 0; v < cuts.size(); ++v) {
                lfilter_stream_t<double> st;
                for (size_t c = 0; c + 1 < cuts[v].size(); ++c) {
                    std::vector<double> chunk(x.begin() + cuts[v][c], x.begin() + cuts[v][c + 1]);
                    lfilter(st, w, chunk);
                    joined[v].insert(joined[v].end(), st.result.begin(), st.result.end());
                }
            }
            CHECK(joined[0] == joined[1] && joined[0].size() == a.size());
            double peak = 0.0, worst = 0.0;
            for (size_t i = 0; i < a.size(); ++i) {
                peak = std::max(peak, std::abs(a[i]));
                worst = std::max(worst, std::abs(a[i] - joined[0][i]));
            }
            CHECK(worst <= 1e-12 * peak);
        }
        // a denominator with a recursive part (reference dsp.hpp:366-368) runs on the GPU too; checked here
        // against the loop written out (den[0] is not read, nothing is normalised by it)
        {
            const std::vector<double> num = {0.2, 0.3, 0.1}, den = {7.0, -1.1, 0.5, -0.05};
            const auto got = lfilter<double>(num, den, x);
            std::vector<double> want(x.size());
            double peak = 0.0, worst = 0.0;
            for (size_t i = 0; i < x.size(); ++i) {
                double sum = 0.0;
                for (size_t j = 0; j < std::min(num.size(), i + 1); ++j) sum += num[j] * x[i - j];
                for (size_t j = 1; j < std::min(den.size(), i + 1); ++j) sum -= den[j] * want[i - j];
                want[i] = sum;
            }
            CHECK(got.size() == x.size());
            for (size_t i = 0; i < x.size(); ++i) {
                peak = std::max(peak, std::abs(want[i]));
                worst = std::max(worst, std::abs(want[i] - got[i]));
            }
            CHECK(worst <= 1e-12 * peak);
            lfilter_ctx_t<double> rctx;
            lfilter(rctx, num, den, x);
            CHECK(rctx.result == got);
            // streamed in chunks with both histories carried
            lfilter_stream_t<double> rst;
            std::vector<double> joined;
            const std::vector<size_t> rcuts = {0, 2, 5, 1000, 1001, 4200, x.size()};
            for (size_t c = 0; c + 1 < rcuts.size(); ++c) {
                std::vector<double> chunk(x.begin() + rcuts[c], x.begin() + rcuts[c + 1]);
                lfilter(rst, num, den, chunk);
                joined.insert(joined.end(), rst.result.begin(), rst.result.end());
            }
            CHECK(joined.size() == want.size() && rst.output_history.size() == 3);
            worst = 0.0;
            for (size_t i = 0; i < x.size(); ++i) worst = std::max(worst, std::abs(want[i] - joined[i]));
            CHECK(worst <= 1e-12 * peak);
            // more than nine denominator coefficients: loud error, no silent CPU path
            CHECK(throws<std::invalid_argument>([&] { lfilter<double>(num, std::vector<double>(10, 0.01), x); }));
        }
        // wavelet bank: error classes of the reference (dsp.hpp:1047-1119, 1153-1155, 1195-1197)
        auto [psi, grid] = integrate_wavelet<double>(cwt_wavelet_t::MORLET, 10);
        CHECK(throws<std::runtime_error>([&] { cwt<double>(x, std::vector<double>{0.0625}, psi, grid); }));   // too small
        CHECK(throws<std::runtime_error>([&] { cwt<double>(x, std::vector<double>{}, psi, grid); }));         // no scales
        CHECK(throws<std::invalid_argument>([&] { cwt<double>(std::vector<double>{}, std::vector<double>{1.0}, psi, grid); }));
        CHECK(throws<std::invalid_argument>([&] { cwt<double>(x, std::vector<double>{1.0}, std::vector<double>{1.0, 2.0}, grid); }));
        auto rows = cwt<double>(x, std::vector<double>{3.0, 1.0}, psi, grid);  // unsorted scales are sorted
        auto row1 = cwt<double>(x, std::vector<double>{1.0}, psi, grid);
        CHECK(rows.size() == 2 * x.size());
        for (size_t i = 0; i < x.size(); ++i) CHECK(rows[i] == row1[i]);
        std::printf("OK\n");
        return 0;
    }
    std::fprintf(stderr, "usage: dropin_check design N C | readme | filter N C | api [nogpu]\n");
    return 2;
}
