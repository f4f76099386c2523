// dsp.hpp — MI355X drop-in for the FIR path of Unbinilium/DSP.
//
// Same namespace, names, template parameters, defaults and error behaviour as the
// reference header for:
//     dsp::firwin        (reference dsp/include/dsp.hpp:190-311)   host tap design
//     dsp::lfilter       (reference dsp.hpp:333-396)               FIR filtering on the GPU
//     dsp::integrate_wavelet (reference dsp.hpp:934-985)           host wavelet table
//     dsp::cwt           (reference dsp.hpp:987-1239)              wavelet bank on the same GPU kernel
// plus what they depend on: dsp::constants (dsp.hpp:34-53), dsp::math::sinc
// (dsp.hpp:60-67), dsp::hamming_window (dsp.hpp:74-115), dsp::types::window_t
// (dsp.hpp:119-121), dsp::lfilter_ctx_t (dsp.hpp:333).
//
// Tap design runs on the host and performs the same IEEE-754 operations in the same
// order as the reference, so the taps are bit-identical.  lfilter hands contiguous
// double buffers to the C ABI in dspx.h (libdspx.so, hand-written gfx950 kernels).
// A denominator with more than one coefficient adds the recursive tail of the
// reference loop (dsp.hpp:366-368) on the GPU as well.
// There is no CPU filtering path in this header: a failing GPU call throws
// std::runtime_error, and inputs the GPU path does not cover are rejected
//   - at compile time when T is neither double nor float, or when T is double and the reference
//     would form the products numerators * x in a type other than double (lfilter<float> sums in
//     float, with products in float or double as the reference forms them, bit-identically),
//   - with std::invalid_argument when the denominator has more than nine
//     coefficients.
//
// Requires C++20.  Link with -ldspx.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <concepts>
#include <cstddef>
#include <iterator>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "dspx.h"

namespace dsp {

namespace constants {

inline constexpr double EPS = 1.0e-20;
inline constexpr double PI = 3.14159265358979323846264338327950288419716939937510582097494459231;
inline constexpr double SQRT_2 = 1.41421356237309504880168872420969807856967187537694807317667973799;
inline constexpr double SQRT_2PI = 2.50662827463100050241576528481104525300698674060993831662992357634;

}  // namespace constants

using namespace constants;

namespace math {

// sin(pi x) / (pi x); arguments closer to zero than eps are moved to eps first.
template <std::floating_point T> constexpr T sinc(T x, T eps = static_cast<T>(EPS)) {
    const T arg = (std::abs(x) < eps ? eps : x) * static_cast<T>(PI);
    return static_cast<T>(std::sin(arg) / arg);
}

}  // namespace math

using namespace math;

namespace types {

enum class window_t {
    HAMMING,
};

}  // namespace types

using namespace types;

namespace traits {

// A container the design / filter functions accept: size() -> size_t, operator[],
// and elements that convert to T without throwing.
template <typename C, typename T>
concept indexable_of = requires(const C& c, std::size_t i) {
    { c.size() } -> std::same_as<std::size_t>;
    c[i];
    typename C::value_type;
} && std::is_nothrow_convertible_v<typename C::value_type, T>;

template <typename C, typename T>
concept iterable_of = requires(const C& c) {
    { c.size() } -> std::same_as<std::size_t>;
    typename C::iterator;
    typename C::const_iterator;
    typename C::value_type;
    c.begin();
    c.end();
} && std::is_nothrow_convertible_v<typename C::value_type, T>;

template <typename C, typename S>
concept contiguous_of = requires(const C& c) {
    { std::data(c) } -> std::convertible_to<const S*>;
} && std::contiguous_iterator<decltype(std::begin(std::declval<const C&>()))>;

template <typename C>
concept contiguous_doubles = contiguous_of<C, double>;

}  // namespace traits

using namespace traits;

// Generalised Hamming window  w[n] = alpha + (1 - alpha) cos(f_n),  f_n on a uniform grid
// over [-pi, pi]: `width` points when sym, else the first `width` of width + 1 points.
template <std::floating_point T> std::vector<T> hamming_window(std::size_t width, T alpha = static_cast<T>(0.54), bool sym = true) {
    if (width < 2) [[unlikely]] throw std::invalid_argument("The width must be greater than or equal to 2.");
    const std::size_t points = sym ? width : width + 1;
    const T first = -PI;
    const T span = std::abs(static_cast<T>(PI) - first);
    const T last_index = points - 1;
    const T coeff[2] = {alpha, static_cast<T>(static_cast<T>(1.0) - alpha)};
    std::vector<T> w(width);
    int n = 0;
    for (auto& wn : w) {
        const T f = first + (span * (static_cast<T>(n++) / last_index));
        T v = static_cast<T>(0.0);
        for (std::size_t k = 0; k < 2; ++k) v += coeff[k] * std::cos(static_cast<T>(k) * f);
        wn = v;
    }
    return w;
}

// Windowed-sinc FIR design.  `cutoffs` are band edges in units of the Nyquist frequency.
// As in the reference, when several band pairs are given only the last pair shapes the
// taps (each pair overwrites the previous one), and the pass band used for scaling is
// the first pair.
template <typename T = double, typename Container>
    requires iterable_of<Container, T>
std::vector<T> firwin(std::size_t numtaps, const Container& cutoffs, bool pass_zero = true, bool scale = true,
                      window_t window = window_t::HAMMING) {
    if (cutoffs.size() == 0) [[unlikely]] throw std::invalid_argument("At least one cutoffs frequency must be given.");
    const bool pass_nyquist = !pass_zero;
    if (pass_nyquist && (numtaps % 2 == 0)) [[unlikely]]
        throw std::invalid_argument("The number of numtaps must be odd when pass_nyquist is true.");
    if (window != window_t::HAMMING) [[unlikely]] throw std::invalid_argument("Unsupported window type.");

    std::vector<T> edges;
    edges.reserve(cutoffs.size() + 2);
    if (pass_zero) edges.push_back(static_cast<T>(0.0));
    for (const auto& c : cutoffs) edges.push_back(static_cast<T>(c));
    if (pass_nyquist) edges.push_back(static_cast<T>(1.0));

    const T centre = static_cast<T>(0.5) * static_cast<T>(numtaps - 1);
    const auto offset = [centre](std::size_t i) { return static_cast<T>(i) - centre; };

    std::vector<T> h(numtaps, static_cast<T>(0.0));
    if (edges.size() >= 2) {
        const std::size_t hi = (edges.size() % 2 == 0) ? edges.size() - 1 : edges.size() - 2;  // last odd index
        const T left = edges[hi - 1], right = edges[hi];
        for (std::size_t j = 0; j < numtaps; ++j) {
            const T m = offset(j);
            h[j] = (right * sinc(right * m)) - (left * sinc(left * m));
        }
    }

    const std::vector<T> win = hamming_window<T>(numtaps);
    for (std::size_t i = 0; i < numtaps; ++i) h[i] *= win[i];

    if (scale) {
        const T left = edges[0], right = edges[1];
        T f0;
        if (left == static_cast<T>(0.0))
            f0 = 0.0;
        else if (right == static_cast<T>(1.0))
            f0 = 1.0;
        else
            f0 = static_cast<T>(0.5) * (left + right);
        T gain = 0.0;
        for (std::size_t i = 0; i < numtaps; ++i) gain += h[i] * std::cos(static_cast<T>(PI) * offset(i) * f0);
        if (std::abs(gain) < static_cast<T>(EPS)) [[unlikely]] gain = EPS;
        for (auto& tap : h) tap /= gain;
    }
    return h;
}

template <std::floating_point T = double>
std::vector<T> firwin(std::size_t numtaps, T cutoff, bool pass_zero = true, bool scale = true,
                      window_t window = window_t::HAMMING) {
    return firwin<T>(numtaps, std::array<T, 1>{cutoff}, pass_zero, scale, window);
}

template <typename T = double> struct lfilter_ctx_t { std::vector<T> result; };

namespace detail {

// Contiguous view of a container as elements of S; copies (converting each element) only when the
// container is not a contiguous array of S (e.g. std::deque<double>, std::vector<float> viewed as double).
template <typename S, typename C> class scalar_view {
  public:
    explicit scalar_view(const C& c) : size_(c.size()) {
        if constexpr (contiguous_of<C, S>) {
            data_ = std::data(c);
        } else {
            copy_.resize(size_);
            for (std::size_t i = 0; i < size_; ++i) copy_[i] = static_cast<S>(c[i]);
            data_ = copy_.data();
        }
    }
    const S* data() const { return data_; }
    std::size_t size() const { return size_; }

  private:
    std::vector<S> copy_;
    const S* data_ = nullptr;
    std::size_t size_;
};
template <typename C> using double_view = scalar_view<double, C>;

}  // namespace detail

// y = lfilter(numerators, denominators, x) with zero initial state, written to ctx.result
// (resized only when its length differs from x's).  denominators[0] is never read.
template <typename T = double, typename Container_1, typename Container_2, typename Container_3>
    requires std::floating_point<T> && indexable_of<Container_1, T> && indexable_of<Container_2, T> &&
             indexable_of<Container_3, T>
void lfilter(lfilter_ctx_t<T>& ctx, const Container_1& numerators, const Container_2& denominators,
             const Container_3& x) {
    // The reference multiplies numerators[j] * x[i-j] and denominators[j] * y[i-j] in the type the operands
    // promote to and only then converts to T (dsp.hpp:363, 367).  Converting the narrower operand to that type
    // first -- which is all that happens to the containers here -- gives the same values.
    using V1 = typename Container_1::value_type;
    using V2 = typename Container_2::value_type;
    using V3 = typename Container_3::value_type;
    using product_t = decltype(std::declval<V1>() * std::declval<V3>());
    static_assert(std::is_same_v<T, double> || std::is_same_v<T, float>, "dsp::lfilter (MI355X): T must be double or float");
    const std::size_t n = x.size();
    if (ctx.result.size() != n) [[unlikely]] ctx.result.resize(n);
    if (n == 0) return;
    int rc;
    if constexpr (std::is_same_v<T, float>) {
        // single precision (dsp.hpp:361-368 with T = float): float sums of products formed in float or in double; the
        // tail y_i -= denominators[j] * y[i-j] in the type denominators[j] * float promotes to
        static_assert(std::is_same_v<product_t, float> || std::is_same_v<product_t, double>,
                      "dsp::lfilter<float> (MI355X): the products numerators * x must be formed in float or double");
        using tail_t = decltype(std::declval<V2>() * std::declval<float>());
        static_assert(std::is_same_v<tail_t, float> || std::is_same_v<tail_t, double>,
                      "dsp::lfilter<float> (MI355X): the products denominators * y must be formed in float or double");
        const detail::scalar_view<product_t, Container_1> b(numerators);
        const detail::scalar_view<product_t, Container_3> in(x);
        if (denominators.size() > 1) {
            // (converted to tail_t first, as the promotion would, then handed over as doubles, which is exact)
            std::vector<double> a(denominators.size());
            for (std::size_t j = 0; j < a.size(); ++j) a[j] = static_cast<double>(static_cast<tail_t>(denominators[j]));
            constexpr int in_double = std::is_same_v<tail_t, double> ? 1 : 0;
            if constexpr (std::is_same_v<product_t, float>)
                rc = dspx_lfilter_f32(b.data(), b.size(), a.data(), a.size(), in_double, in.data(), ctx.result.data(), n);
            else
                rc = dspx_lfilter_f32_from_f64(b.data(), b.size(), a.data(), a.size(), in_double, in.data(), ctx.result.data(), n);
            if (rc == DSPX_ERR_INVALID)
                throw std::invalid_argument(std::string("dsp::lfilter (MI355X): ") + dspx_last_error());
        } else if constexpr (std::is_same_v<product_t, float>) {
            rc = dspx_fir_f32(b.data(), b.size(), in.data(), ctx.result.data(), n);
        } else {
            rc = dspx_fir_f32_from_f64(b.data(), b.size(), in.data(), ctx.result.data(), n);
        }
    } else {
        static_assert((std::is_same_v<product_t, double> || std::is_same_v<product_t, float>) &&
                        std::is_same_v<decltype(std::declval<V2>() * std::declval<double>()), double>,
                      "dsp::lfilter<double> (MI355X): the products numerators * x must be formed in double or float, "
                      "those of denominators * y in double");
        if constexpr (std::is_same_v<product_t, float>) {
            // containers whose elements multiply in float (e.g. vector<float> taps over vector<float> samples): the
            // reference forms float products and adds them to a double sum (dsp.hpp:363)
            const detail::scalar_view<float, Container_1> b(numerators);
            const detail::scalar_view<float, Container_3> in(x);
            if (denominators.size() > 1) {
                // FIR part first, then the tail (double arithmetic throughout) in place of a second FIR pass: a
                // one-tap numerator of 1.0 reproduces its input exactly
                std::vector<double> part(n);
                rc = dspx_fir_f64_from_f32(b.data(), b.size(), in.data(), part.data(), n);
                if (rc == DSPX_OK) {
                    const detail::double_view<Container_2> a(denominators);
                    const double one = 1.0;
                    rc = dspx_lfilter_f64(&one, 1, a.data(), a.size(), part.data(), ctx.result.data(), 1, n, n, DSPX_FIR_EXACT);
                    if (rc == DSPX_ERR_INVALID)
                        throw std::invalid_argument(std::string("dsp::lfilter (MI355X): ") + dspx_last_error());
                }
            } else {
                rc = dspx_fir_f64_from_f32(b.data(), b.size(), in.data(), ctx.result.data(), n);
            }
        } else {
            const detail::double_view<Container_1> b(numerators);
            const detail::double_view<Container_3> in(x);
            if (denominators.size() > 1) {  // the recursive tail, dsp.hpp:366-368
                const detail::double_view<Container_2> a(denominators);
                rc = dspx_lfilter_f64(b.data(), b.size(), a.data(), a.size(), in.data(), ctx.result.data(), 1, n, n,
                                      DSPX_FIR_FMA);
                if (rc == DSPX_ERR_INVALID)
                    throw std::invalid_argument(std::string("dsp::lfilter (MI355X): ") + dspx_last_error());
            } else {
                rc = dspx_fir_f64(b.data(), b.size(), in.data(), ctx.result.data(), n, DSPX_FIR_FMA);
            }
        }
    }
    if (rc != DSPX_OK)
        throw std::runtime_error(std::string("dsp::lfilter (MI355X): ") + dspx_status_string(rc) + ": " +
                                 dspx_last_error());
}

template <typename T = double, typename Container_1, typename Container_2, typename Container_3>
    requires std::floating_point<T> && indexable_of<Container_1, T> && indexable_of<Container_2, T> &&
             indexable_of<Container_3, T>
std::vector<T> lfilter(const Container_1& numerators, const Container_2& denominators, const Container_3& x) {
    lfilter_ctx_t<T> ctx;
    lfilter<T>(ctx, numerators, denominators, x);
    return std::move(ctx.result);
}

template <typename T = double, typename Container_1, typename Container_2>
    requires std::floating_point<T> && indexable_of<Container_1, T> && indexable_of<Container_2, T>
std::vector<T> lfilter(const Container_1& numerators, T denominator, const Container_2& x) {
    return lfilter<T>(numerators, std::array<T, 1>{denominator}, x);
}

// ---- addition (no counterpart in the reference): chunked filtering of one long stream ----
// The reference's lfilter starts every call from a zero state (dsp.hpp:356-365), so a stream
// cut into chunks cannot be filtered chunk by chunk with it.  lfilter_stream_t carries the
// last ntaps-1 input samples from one call to the next.  The chunks run on the vector pipe, so
// the concatenated results do not depend on where the stream is cut (bit for bit); they agree
// with one lfilter call over the whole stream to rounding (that call may use the matrix pipe).
template <typename T = double> struct lfilter_stream_t {
    std::vector<T> history;         // ntaps-1 samples; (re)initialised to zeros when its size does not fit
    std::vector<T> output_history;  // denominators.size()-1 outputs (recursive filters only), likewise
    std::vector<T> result;
};

template <typename T = double, typename Container_1, typename Container_2>
    requires std::floating_point<T> && indexable_of<Container_1, T> && indexable_of<Container_2, T>
void lfilter(lfilter_stream_t<T>& stream, const Container_1& numerators, const Container_2& x) {
    static_assert(std::is_same_v<T, double> && std::is_same_v<typename Container_1::value_type, double> &&
                    std::is_same_v<typename Container_2::value_type, double>,
                  "dsp::lfilter (MI355X): only T = double over containers of double is implemented");
    const std::size_t n = x.size();
    const std::size_t keep = numerators.size() ? numerators.size() - 1 : 0;
    if (stream.history.size() != keep) stream.history.assign(keep, 0.0);
    if (stream.result.size() != n) stream.result.resize(n);
    if (n == 0) return;
    const detail::double_view<Container_1> b(numerators);
    const detail::double_view<Container_2> in(x);
    const int rc = dspx_fir_f64_stream(b.data(), b.size(), stream.history.data(), in.data(), stream.result.data(), 1,
                                       n, n, DSPX_FIR_FMA);
    if (rc != DSPX_OK)
        throw std::runtime_error(std::string("dsp::lfilter (MI355X): ") + dspx_status_string(rc) + ": " +
                                 dspx_last_error());
}

// The same with a recursive part (reference dsp.hpp:366-368): the stream also carries the last
// denominators.size()-1 outputs.  Chunked results agree with one call over the whole stream to rounding.
template <typename T = double, typename Container_1, typename Container_2, typename Container_3>
    requires std::floating_point<T> && indexable_of<Container_1, T> && indexable_of<Container_2, T> &&
             indexable_of<Container_3, T>
void lfilter(lfilter_stream_t<T>& stream, const Container_1& numerators, const Container_2& denominators,
             const Container_3& x) {
    static_assert(std::is_same_v<T, double> && std::is_same_v<typename Container_1::value_type, double> &&
                    std::is_same_v<typename Container_2::value_type, double> &&
                    std::is_same_v<typename Container_3::value_type, double>,
                  "dsp::lfilter (MI355X): only T = double over containers of double is implemented");
    const std::size_t n = x.size();
    const std::size_t keep = numerators.size() ? numerators.size() - 1 : 0;
    const std::size_t keep_out = denominators.size() ? denominators.size() - 1 : 0;
    if (stream.history.size() != keep) stream.history.assign(keep, 0.0);
    if (stream.output_history.size() != keep_out) stream.output_history.assign(keep_out, 0.0);
    if (stream.result.size() != n) stream.result.resize(n);
    if (n == 0) return;
    const detail::double_view<Container_1> b(numerators);
    const detail::double_view<Container_2> a(denominators);
    const detail::double_view<Container_3> in(x);
    const int rc = dspx_lfilter_f64_stream(b.data(), b.size(), a.data(), a.size(), stream.history.data(),
                                           stream.output_history.data(), in.data(), stream.result.data(), 1, n, n,
                                           DSPX_FIR_FMA);
    if (rc == DSPX_ERR_INVALID) throw std::invalid_argument(std::string("dsp::lfilter (MI355X): ") + dspx_last_error());
    if (rc != DSPX_OK)
        throw std::runtime_error(std::string("dsp::lfilter (MI355X): ") + dspx_status_string(rc) + ": " +
                                 dspx_last_error());
}

namespace types {

enum class cwt_wavelet_t {
    MORLET,
};

}  // namespace types

// Samples the wavelet on 2^percision points of [lower, upper] and replaces it by its running
// integral (cumulative sum times the grid step).  Returns {psi, x}.
template <std::floating_point T = double>
std::pair<std::vector<T>, std::vector<T>> integrate_wavelet(cwt_wavelet_t wavelet, std::size_t percision = 10,
                                                            T lower = -8.0, T upper = 8.0) {
    if (percision <= 2 || percision >= sizeof(std::size_t) * 8) [[unlikely]]
        throw std::invalid_argument("The percision must be greater than 2 or less than the number of bits in size_t.");
    if (lower >= upper) [[unlikely]] throw std::invalid_argument("The lower must be less than upper.");
    if (wavelet != cwt_wavelet_t::MORLET) [[unlikely]] throw std::invalid_argument("Unsupported wavelet type.");

    const std::size_t count = std::size_t(1) << percision;
    std::vector<T> psi(count), grid(count);
    const T width = upper - lower;
    const T last = static_cast<T>(count - 1);
    for (std::size_t i = 0; i < count; ++i) {
        const T t = lower + (width * (static_cast<T>(i) / last));
        grid[i] = t;
        psi[i] = std::exp(static_cast<T>(-0.5) * t * t) * std::cos(static_cast<T>(5.0) * t);  // Morlet
    }
    const T step = grid[1] - grid[0];
    T running = 0.0;
    for (auto& p : psi) {
        running += p;
        p = running * step;
    }
    return {std::move(psi), std::move(grid)};
}

// Scales sorted ascending with their -sqrt, result buffer and its shape {scales, samples}.  The
// scratch vectors of the reference's context are kept as members for source compatibility; the
// GPU path does not use them.
template <typename T = double, typename Container = std::vector<T>> struct cwt_ctx_t {
    cwt_ctx_t() = default;
    cwt_ctx_t(const Container& s) : scales(s) {
        std::sort(scales.begin(), scales.end());
        negtive_sqrt_scales.reserve(scales.size());
        for (const auto& v : scales) negtive_sqrt_scales.push_back(static_cast<T>(-std::sqrt(v)));
    }

    Container scales;
    std::vector<T> negtive_sqrt_scales;

    std::vector<std::size_t> psi_arange;
    std::vector<std::size_t> psi_indices;
    std::vector<T> coefficients;

    std::vector<T> result;
    std::vector<std::size_t> shape;
};

// Continuous wavelet transform of `signal` for every scale of ctx: ctx.result holds one row of
// signal.size() coefficients per (sorted) scale, ctx.shape = {scales, samples}.
template <typename T = double, typename P = double, typename Container_1, typename Container_2, typename Container_3,
          typename Container_4>
    requires std::floating_point<T> && std::floating_point<P> && indexable_of<Container_1, P> &&
             indexable_of<Container_2, T> && indexable_of<Container_3, T> && indexable_of<Container_4, T>
void cwt(cwt_ctx_t<T, Container_1>& ctx, const Container_2& signal, const Container_3& wavelet_psi,
         const Container_4& wavelet_x) {
    static_assert(std::is_same_v<T, double> && std::is_same_v<P, double> &&
                    std::is_same_v<typename Container_2::value_type, double> &&
                    std::is_same_v<typename Container_3::value_type, double> &&
                    std::is_same_v<typename Container_4::value_type, double>,
                  "dsp::cwt (MI355X): only T = P = double over containers of double is implemented");
    std::vector<double> scales(ctx.scales.size());
    for (std::size_t i = 0; i < scales.size(); ++i) scales[i] = static_cast<double>(ctx.scales[i]);
    const detail::double_view<Container_2> sig(signal);
    const detail::double_view<Container_3> psi(wavelet_psi);
    const detail::double_view<Container_4> grid(wavelet_x);
    const std::size_t wanted = scales.size() * sig.size();
    std::vector<T> out(wanted);
    const int rc = dspx_cwt_f64(sig.data(), sig.size(), scales.data(), scales.size(), psi.data(), psi.size(),
                                grid.data(), grid.size(), out.data(), DSPX_FIR_FMA);
    if (rc == DSPX_ERR_INVALID) throw std::invalid_argument(dspx_last_error());
    if (rc != DSPX_OK) throw std::runtime_error(dspx_last_error());
    ctx.result = std::move(out);
    ctx.shape = {scales.size(), sig.size()};
}

template <typename T = double, typename P = double, typename Container_1, typename Container_2, typename Container_3,
          typename Container_4>
    requires std::floating_point<T> && std::floating_point<P> && indexable_of<Container_1, T> &&
             indexable_of<Container_2, T> && indexable_of<Container_3, T> && indexable_of<Container_4, T>
std::vector<T> cwt(const Container_1& signal, const Container_2& scales, const Container_3& wavelet_psi,
                   const Container_4& wavelet_x) {
    cwt_ctx_t<T, Container_2> ctx(scales);
    cwt<T, P>(ctx, signal, wavelet_psi, wavelet_x);
    return std::move(ctx.result);
}

}  // namespace dsp
