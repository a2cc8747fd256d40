/*
 * TEST INFRASTRUCTURE - plain-C restatement of the two integer/byte-exact pieces of the hot path, independent of torch:
 *
 *   oracle_normalize   pyfilter/utils.py:49-64   nan_to_num_(nan=-inf, posinf=-inf[, neginf -> lowest]) in place,
 *                                                 max-shifted softmax over the particle axis
 *   oracle_systematic  pyfilter/resampling.py:24-52 and the sequential walk the reference's own known-answer test
 *                      checks it against (tests/test_resampling.py:8-28): cumsum with a double accumulator rounded per
 *                      element, last = 1, idx_i = first j with cdf[j] >= (i + u) / N.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product never does.
 * Layout: column-major per filter, w[b*N + i].
 */
#include <math.h>
#include <stdint.h>

#define DEFINE_ORACLE(T, SUF, LOWEST, EXPF)                                                                   \
    void oracle_normalize_##SUF(T* w, T* W, int64_t N, int64_t B) {                                           \
        for (int64_t b = 0; b < B; ++b) {                                                                      \
            T* c = w + b * N;                                                                                  \
            T m = -INFINITY;                                                                                   \
            for (int64_t i = 0; i < N; ++i) {                                                                  \
                T v = c[i];                                                                                    \
                if (v != v || v == (T)INFINITY) v = -INFINITY;                                                 \
                else if (v == (T)-INFINITY) v = LOWEST;                                                        \
                c[i] = v;                                                                                      \
                if (v > m) m = v;                                                                              \
            }                                                                                                  \
            double s = 0.0;                                                                                    \
            for (int64_t i = 0; i < N; ++i) s += (double)EXPF(c[i] - m);                                       \
            for (int64_t i = 0; i < N; ++i) W[b * N + i] = (T)((double)EXPF(c[i] - m) / s);                    \
        }                                                                                                      \
    }                                                                                                          \
    void oracle_systematic_##SUF(const T* W, const T* u, int64_t* idx, int64_t N, int64_t B) {                \
        for (int64_t b = 0; b < B; ++b) {                                                                      \
            const T* c = W + b * N;                                                                            \
            double acc = 0.0;                                                                                  \
            int64_t i = 0;                                                                                     \
            const T nT = (T)N;                                                                                 \
            for (int64_t j = 0; j < N && i < N; ++j) {                                                         \
                acc += (double)c[j];                                                                           \
                const T cdf = (j == N - 1) ? (T)1 : (T)acc;                                                    \
                while (i < N && !(cdf < ((T)i + u[b]) / nT)) idx[b * N + i++] = j;                             \
            }                                                                                                  \
            while (i < N) idx[b * N + i++] = N - 1;                                                            \
        }                                                                                                      \
    }

DEFINE_ORACLE(float, f32, -3.40282346638528859812e+38f, expf)
DEFINE_ORACLE(double, f64, -1.79769313486231570815e+308, exp)
