/*
 * CPU oracle for the FIR hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file restates, in plain C, the arithmetic of the reference header
 * (Unbinilium/DSP, dsp/include/dsp.hpp) for tap design and direct-form
 * filtering.  It exists so that tests, the smoke check and the CPU-baseline
 * leg of the benchmark have an independent checker that travels with the
 * repository.  Nothing in the shipped library links, loads or calls it.
 *
 * Parity status: PINNED.  Each function below is checked bit-for-bit against
 * the reference header compiled in the build container (oracle/_ref) and
 * against the golden literals of the reference's own tests
 * (dsp.hpp:1311-1599, 2096-2783) by tests/test_oracle_*.py.
 *
 * Build: gcc -O2 -ffp-contract=off (no -ffast-math, no -march=native), so
 * every product and sum is a separately rounded IEEE-754 double operation,
 * exactly as the reference compiles under g++ -O2 on x86-64.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#define ORACLE_EPS 1.0e-20 /* dsp.hpp:39 */
#define ORACLE_PI 3.14159265358979323846264338327950288419716939937510582097494459231 /* dsp.hpp:43 */

/* dsp.hpp:60-67 — sinc(x) = sin(pi x)/(pi x); |x| < eps is replaced by eps. */
double oracle_sinc(double x) {
    if (fabs(x) < ORACLE_EPS) x = ORACLE_EPS;
    x *= ORACLE_PI;
    return sin(x) / x;
}

/* dsp.hpp:74-115 — generalised Hamming window, symmetric form (sym=true) or
 * periodic form (sym=false: computed on width+1 points, last one dropped).
 * Returns 0 on success, -1 for width < 2 (the reference throws there). */
int oracle_hamming_window(size_t width, double alpha, int sym, double* w) {
    if (width < 2) return -1;
    const size_t npts = sym ? width : width + 1;
    const double start = -ORACLE_PI;
    const double step = fabs(ORACLE_PI - start);
    const double denom = (double)(npts - 1);
    const double a0 = alpha;
    const double a1 = 1.0 - alpha;
    for (size_t i = 0; i < width; ++i) {
        /* the grid value uses int -> double conversion of the running index (dsp.hpp:91-93) */
        const double f = start + (step * ((double)(int)i / denom));
        double acc = 0.0;
        acc += a0 * cos(0.0 * f); /* k = 0 term, dsp.hpp:102-107 */
        acc += a1 * cos(1.0 * f); /* k = 1 term */
        w[i] = acc;
    }
    return 0;
}

/* dsp.hpp:195-304 — windowed-sinc FIR design.
 * Returns 0 on success; -1 no cutoffs (dsp.hpp:200-206); -2 even numtaps with
 * pass_zero == 0 (dsp.hpp:208-215); -3 numtaps < 2 (hamming_window throws,
 * dsp.hpp:76-78); -4 allocation failure.
 * Quirk preserved: the band loop ASSIGNS h[j] (dsp.hpp:249), so with several
 * band pairs only the last pair contributes. */
int oracle_firwin(size_t numtaps, const double* cutoffs, size_t ncut, int pass_zero, int scale, double* h) {
    if (ncut == 0) return -1;
    const int pass_nyquist = !pass_zero;
    if (pass_nyquist && !(numtaps & 1)) return -2;

    const size_t nb = ncut + (pass_zero ? 1 : 0) + (pass_nyquist ? 1 : 0);
    double* bands = (double*)malloc(sizeof(double) * nb);
    double* m = (double*)malloc(sizeof(double) * (numtaps ? numtaps : 1));
    double* win = (double*)malloc(sizeof(double) * (numtaps ? numtaps : 1));
    if (!bands || !m || !win) {
        free(bands), free(m), free(win);
        return -4;
    }
    size_t k = 0;
    if (pass_zero) bands[k++] = 0.0;
    for (size_t i = 0; i < ncut; ++i) bands[k++] = cutoffs[i];
    if (pass_nyquist) bands[k++] = 1.0;

    const double alpha = 0.5 * (double)(numtaps - 1); /* dsp.hpp:230 */
    for (size_t i = 0; i < numtaps; ++i) m[i] = (double)i - alpha;

    for (size_t i = 0; i < numtaps; ++i) h[i] = 0.0;
    for (size_t i = 1; i < nb; i += 2) { /* dsp.hpp:243-251 */
        const double left = bands[i - 1];
        const double right = bands[i];
        for (size_t j = 0; j < numtaps; ++j) {
            h[j] = (right * oracle_sinc(right * m[j])) - (left * oracle_sinc(left * m[j]));
        }
    }

    if (oracle_hamming_window(numtaps, 0.54, 1, win) != 0) { /* dsp.hpp:258 */
        free(bands), free(m), free(win);
        return -3;
    }
    for (size_t i = 0; i < numtaps; ++i) h[i] *= win[i];

    if (scale) { /* dsp.hpp:275-301; nb >= 2 is asserted there */
        const double left = bands[0];
        const double right = nb >= 2 ? bands[1] : bands[0];
        double f0;
        if (left == 0.0)
            f0 = 0.0;
        else if (right == 1.0)
            f0 = 1.0;
        else
            f0 = 0.5 * (left + right);
        double sum = 0.0;
        for (size_t i = 0; i < numtaps; ++i) sum += h[i] * cos(ORACLE_PI * m[i] * f0);
        if (fabs(sum) < ORACLE_EPS) sum = ORACLE_EPS;
        for (size_t i = 0; i < numtaps; ++i) h[i] /= sum;
    }
    free(bands), free(m), free(win);
    return 0;
}

/* dsp.hpp:342-371 — direct-form filter with zero initial state.
 * y[i] = sum_{j < min(nb, i+1)} b[j] * x[i-j], accumulated j-ascending from 0.0,
 * then y[i] -= a[j] * y[i-j] for 1 <= j < min(na, i+1).  a[0] is never read. */
void oracle_lfilter_f64(const double* b, size_t nb, const double* a, size_t na, const double* x, size_t n,
                        double* y) {
    for (size_t i = 0; i < n; ++i) {
        const size_t ip1 = i + 1;
        const size_t nb_end = nb < ip1 ? nb : ip1;
        const size_t na_end = na < ip1 ? na : ip1;
        double sum = 0.0;
        for (size_t j = 0; j < nb_end; ++j) sum += b[j] * x[i - j];
        y[i] = sum;
        for (size_t j = 1; j < na_end; ++j) y[i] -= a[j] * y[i - j];
    }
}

/* FIR only (denominator of size 1), with optional history: hist[0..nb-2] are the
 * nb-1 samples that precede x[0] (hist[nb-2] is x[-1]).  hist == NULL is the
 * reference's zero state (truncated sums, dsp.hpp:357-364).  Used to check
 * segment-sharded filtering, which has no counterpart in the reference. */
void oracle_fir_f64_hist(const double* b, size_t nb, const double* hist, const double* x, size_t n, double* y) {
    for (size_t i = 0; i < n; ++i) {
        double sum = 0.0;
        for (size_t j = 0; j < nb; ++j) {
            if (j <= i)
                sum += b[j] * x[i - j];
            else if (hist)
                sum += b[j] * hist[(nb - 1) - (j - i)];
            else
                break;
        }
        y[i] = sum;
    }
}

/* The same recurrence as oracle_lfilter_f64 in single precision
 * (reference: lfilter<float> over float containers, dsp.hpp:361-364 with T = float). */
void oracle_lfilter_f32(const float* b, size_t nb, const float* x, size_t n, float* y) {
    for (size_t i = 0; i < n; ++i) {
        const size_t ip1 = i + 1;
        const size_t nb_end = nb < ip1 ? nb : ip1;
        float sum = 0.0f;
        for (size_t j = 0; j < nb_end; ++j) sum += b[j] * x[i - j];
        y[i] = sum;
    }
}

/* lfilter<float> over containers of double (what the reference's own test_lfilter calls, dsp.hpp:1590-1597):
 * the product is formed in double, converted to float, and added to a float sum (dsp.hpp:361-364 with T = float). */
void oracle_lfilter_f32_from_f64(const double* b, size_t nb, const double* x, size_t n, float* y) {
    for (size_t i = 0; i < n; ++i) {
        const size_t ip1 = i + 1;
        const size_t nb_end = nb < ip1 ? nb : ip1;
        float sum = 0.0f;
        for (size_t j = 0; j < nb_end; ++j) sum += (float)(b[j] * x[i - j]);
        y[i] = sum;
    }
}

/* lfilter<double> over containers of float (dsp.hpp:361-364 with T = double): numerators[j] * x[i-j] is a float
 * product, converted to double and added to a double sum. */
void oracle_lfilter_f64_from_f32(const float* b, size_t nb, const float* x, size_t n, double* y) {
    for (size_t i = 0; i < n; ++i) {
        const size_t ip1 = i + 1;
        const size_t nb_end = nb < ip1 ? nb : ip1;
        double sum = 0.0;
        for (size_t j = 0; j < nb_end; ++j) {
            const float prod = b[j] * x[i - j];
            sum += (double)prod;
        }
        y[i] = sum;
    }
}

/* lfilter<float> with a recursive part (dsp.hpp:356-369 with T = float).  The tail `y_i -= denominators[j] * y[i-j]`
 * is evaluated in the type denominators[j] * float promotes to: float for float denominators ... */
void oracle_lfilter_f32_iir(const float* b, size_t nb, const float* a, size_t na, const float* x, size_t n, float* y) {
    for (size_t i = 0; i < n; ++i) {
        const size_t ip1 = i + 1;
        const size_t nb_end = nb < ip1 ? nb : ip1, na_end = na < ip1 ? na : ip1;
        float sum = 0.0f;
        for (size_t j = 0; j < nb_end; ++j) sum += b[j] * x[i - j];
        float yi = sum;
        for (size_t j = 1; j < na_end; ++j) {
            const float prod = a[j] * y[i - j];
            yi = yi - prod;
        }
        y[i] = yi;
    }
}

/* ... and double for double denominators (here over double numerators and samples as well, whose products are
 * formed in double and converted to float before they are added, as in oracle_lfilter_f32_from_f64). */
void oracle_lfilter_f32_from_f64_iir(const double* b, size_t nb, const double* a, size_t na, const double* x, size_t n,
                                     float* y) {
    for (size_t i = 0; i < n; ++i) {
        const size_t ip1 = i + 1;
        const size_t nb_end = nb < ip1 ? nb : ip1, na_end = na < ip1 ? na : ip1;
        float sum = 0.0f;
        for (size_t j = 0; j < nb_end; ++j) sum += (float)(b[j] * x[i - j]);
        float yi = sum;
        for (size_t j = 1; j < na_end; ++j) {
            const double prod = a[j] * (double)y[i - j];
            yi = (float)((double)yi - prod);
        }
        y[i] = yi;
    }
}

/* Deterministic synthetic input shared by CPU and GPU:
 * x[idx] = ((mix(seed + (idx+1)*GAMMA) >> 11) * 2^-52) - 1.0, uniform in [-1, 1),
 * where mix() is the SplitMix64 finaliser.  Every value is exactly representable. */
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void oracle_fill_splitmix_f64(double* x, size_t n, uint64_t seed, uint64_t offset) {
    for (size_t i = 0; i < n; ++i) {
        const uint64_t bits = splitmix64_at(seed, offset + (uint64_t)i) >> 11;
        x[i] = (double)bits * 0x1.0p-52 - 1.0;
    }
}

/* Order-independent 64-bit checksum of a double buffer: wrapping sum of
 * mix(bits(y[i]) ^ mix(offset+i)).  Bit-identical outputs give identical sums for
 * any partition of the index range, so shards can be checked piecewise. */
uint64_t oracle_checksum_f64(const double* y, size_t n, uint64_t offset) {
    uint64_t acc = 0;
    for (size_t i = 0; i < n; ++i) {
        union {
            double d;
            uint64_t u;
        } v;
        v.d = y[i];
        acc += splitmix64_at(v.u, offset + (uint64_t)i);
    }
    return acc;
}
