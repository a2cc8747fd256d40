// pf_models.hpp - per-particle model arithmetic for the built-in model kinds (device side).
//
// These are the closed forms of SURVEY.md §8(a) row M / a12-a15.  In the reference they are chains of aten ops
// issued by user lambdas + stochproc + torch.distributions:
//   mean_scale / propagate        -> stochproc AffineProcess (call sites proposals/linear.py:41, bootstrap.py:11)
//   obs log-density               -> LinearStateSpaceModel.build_density(x).log_prob(y)   (bootstrap.py:12-14)
//   default APF pre-weight        -> proposals/base.py:69-85 + pre_weight_funcs.py:9-11
//   LinearGaussianObservations    -> proposals/linear.py:38-86 + proposals/utils.py:219-267
// Everything is evaluated in registers; parameters are uniform per column (one filter = one parameter row).
#pragma once
#include "pf_device.hpp"

#define PF_MAXD 3
#define PF_MAXO 3

#include "../../include/pf_amd.h"  // PF_HID_*, PF_OBS_*, PF_PROP_*, PF_FILTER_* kind codes

namespace pf {

#define PF_LOG_SQRT_2PI 0.91893853320467274178

template <int D> struct ObsDim {
    static constexpr int MAXO = (D == 1) ? 1 : PF_MAXO;
};

// Per-column parameter row, layout (in units of T):
//   [ hp0[D] hp1[D] hp2[D] hp3[D] | A[O*D] (row-major O x D) | ob[O] | os[O] ]       NP = 4*D + O*D + 2*O
template <typename T, int D> struct ColParams {
    static constexpr int MAXO = ObsDim<D>::MAXO;
    T hp[4][D];
    T A[MAXO][D];
    T ob[MAXO];
    T os[MAXO];
    T y[MAXO];   // observation of this step
    T yn[MAXO];  // observation of the next step (APF: its first-stage weights are prepared one kernel early)
    int O;

    __device__ __forceinline__ void load(const T* __restrict__ row, int O_, const T* __restrict__ yrow) {
        O = O_;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int d = 0; d < D; ++d) hp[k][d] = row[k * D + d];
        const T* a = row + 4 * D;
#pragma unroll
        for (int o = 0; o < MAXO; ++o) {
            const bool on = o < O_;
#pragma unroll
            for (int d = 0; d < D; ++d) A[o][d] = on ? a[o * D + d] : T(0);
            ob[o] = on ? a[O_ * D + o] : T(0);
            os[o] = on ? a[O_ * D + O_ + o] : T(1);
            y[o] = (on && yrow) ? yrow[o] : T(0);
            yn[o] = T(0);
        }
    }
    __device__ __forceinline__ void load_next(const T* __restrict__ yrow) {
#pragma unroll
        for (int o = 0; o < MAXO; ++o) yn[o] = (o < O && yrow) ? yrow[o] : T(0);
    }
};

struct ModelDesc {
    int hid_kind;
    int obs_kind;
    int obs_dim;  // O >= 1
    double dt;
    double inc_scale;  // scale of the increment distribution: 1 or sqrt(dt)
};

// hidden.mean_scale(x) -> (loc, scale)
template <typename T, int D>
__device__ __forceinline__ void mean_scale(const ModelDesc& md, const ColParams<T, D>& cp, const T (&x)[D],
                                           T (&loc)[D], T (&scale)[D]) {
    const T dt = (T)md.dt;
    switch (md.hid_kind) {
        case PF_HID_LINEAR:
#pragma unroll
            for (int d = 0; d < D; ++d) {
                loc[d] = cp.hp[0][d] + cp.hp[1][d] * x[d];
                scale[d] = cp.hp[2][d];
            }
            break;
        case PF_HID_SINE_EM:
#pragma unroll
            for (int d = 0; d < D; ++d) {
                loc[d] = x[d] + pf_sin(x[d] - cp.hp[0][d]) * dt;
                scale[d] = cp.hp[1][d];
            }
            break;
        case PF_HID_VERHULST_EM:
#pragma unroll
            for (int d = 0; d < D; ++d) {
                loc[d] = x[d] + cp.hp[0][d] * (cp.hp[1][d] - x[d]) * x[d] * dt;
                scale[d] = cp.hp[2][d] * x[d];
            }
            break;
        case PF_HID_LORENZ63_EM:
            if constexpr (D == 3) {
                const T s = cp.hp[0][0], r = cp.hp[1][0], b = cp.hp[2][0];
                const T f0 = -s * (x[0] - x[1]);
                const T f1 = r * x[0] - x[1] - x[0] * x[2];
                const T f2 = x[0] * x[1] - b * x[2];
                loc[0] = x[0] + f0 * dt;
                loc[1] = x[1] + f1 * dt;
                loc[2] = x[2] + f2 * dt;
#pragma unroll
                for (int d = 0; d < D; ++d) scale[d] = cp.hp[3][d];
            }
            break;
        case PF_HID_OU:
        default:
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const T kappa = cp.hp[0][d], gamma = cp.hp[1][d], sigma = cp.hp[2][d];
                const T e = pf_exp(-kappa * dt);
                loc[d] = gamma + (x[d] - gamma) * e;
                scale[d] = sigma * pf_sqrt((T(1) - pf_exp(T(-2) * kappa * dt)) / (T(2) * kappa));
            }
            break;
    }
}

// Per-column constants for the closed-form fast path: scalar state (D = 1), scalar linear-Gaussian observation and a
// state-independent transition scale (every built-in kind except Verhulst).  Everything that does not depend on the
// particle - reciprocals, logs, the optimal-proposal gain - is evaluated once per thread, leaving ~30 flops + one sine
// per particle (the reference spends ~40 aten ops and a batched LU here: SURVEY.md section 8(a) a14).
template <typename T, int D> struct ColConsts {
    // D-generic part: linear observation + state-independent transition scale -> Bootstrap needs no per-particle
    // division / log (`lin_fast`); the scalar closed forms below additionally cover LinearGaussianObservations
    static constexpr int MAXO = ObsDim<D>::MAXO;
    // (vector states always observe linearly - PF_OBS_SV is scalar-only - so this is a compile-time fact)
    static constexpr bool lin_fast = D > 1;
    T oi2s[MAXO], oks;          // 1 / (2 s_o^2) per observation component, sum_o (log s_o + log sqrt(2 pi))
    bool fast;
    T g, inv_g, inc, yb, ybn, a;  // yb / ybn: y - b for this / the next observation
    T i2s, ks;               // observation: 1 / (2 s^2), log s + log sqrt(2 pi)
    T i2inc, kt;             // transition:  1 / (2 inc^2), log inc + log sqrt(2 pi) + log |g|
    T c_loc, c_y, kstd, kq;  // LinearGaussianObservations kernel: mean = c_loc * loc + c_y, std, log std + log sqrt(2 pi)
    T i2c, kc;               // LGO pre-weight: 1 / (2 (s^2 + a^2 g^2)), log sqrt(s^2 + a^2 g^2) + log sqrt(2 pi)
    T ou_e;                  // exp(-kappa dt) for the OU kind
    T ob, ovi, cov;          // observation offset b, 1 / s^2, posterior variance of the LGO kernel (c_y = cov a ovi (y - b))

    __device__ __forceinline__ void prepare(const ModelDesc& md, const ColParams<T, D>& cp) {
        fast = false;
        if (lin_fast) {
            oks = T(0);
#pragma unroll
            for (int o = 0; o < MAXO; ++o) {
                const bool on = o < cp.O;
                oi2s[o] = on ? T(0.5) * pf_rcp_c(cp.os[o] * cp.os[o]) : T(0);
                oks += on ? pf_log_c(cp.os[o]) + T(PF_LOG_SQRT_2PI) : T(0);
            }
        }
        if constexpr (D == 1) {
            if (md.obs_kind != PF_OBS_LINEAR || md.hid_kind == PF_HID_VERHULST_EM || md.hid_kind == PF_HID_USER_AFFINE) return;
            fast = true;
            const T dt = (T)md.dt;
            switch (md.hid_kind) {
                case PF_HID_LINEAR: g = cp.hp[2][0]; break;
                case PF_HID_SINE_EM: g = cp.hp[1][0]; break;
                case PF_HID_LORENZ63_EM: g = cp.hp[3][0]; break;
                default: {  // OU
                    const T kappa = cp.hp[0][0];
                    g = cp.hp[2][0] * pf_sqrt((T(1) - pf_exp(T(-2) * kappa * dt)) / (T(2) * kappa));
                    break;
                }
            }
            ou_e = (md.hid_kind == PF_HID_OU) ? pf_exp(-cp.hp[0][0] * dt) : T(0);
            finish_fast(md, cp);
        }
    }
    // everything of the closed forms that follows from the transition scale g and the observation parameters
    __device__ __forceinline__ void finish_fast(const ModelDesc& md, const ColParams<T, D>& cp) {
        if constexpr (D == 1) {
            inv_g = pf_rcp_c(g);
            inc = (T)md.inc_scale;
            a = cp.A[0][0];
            const T s = cp.os[0];
            yb = cp.y[0] - cp.ob[0];
            ybn = cp.yn[0] - cp.ob[0];
            const T hvi = inv_g * inv_g;
            ovi = pf_rcp_c(s * s);
            ob = cp.ob[0];
            i2s = T(0.5) * ovi;
            ks = pf_log_c(s) + T(PF_LOG_SQRT_2PI);
            i2inc = T(0.5) * pf_rcp_c(inc * inc);
            kt = pf_log_c(inc * pf_abs(g)) + T(PF_LOG_SQRT_2PI);
            cov = pf_rcp_c(hvi + a * ovi * a);
            c_loc = cov * hvi;
            c_y = cov * (a * (ovi * yb));
            kstd = pf_sqrt_c(cov);
            kq = T(0.5) * pf_log_c(cov) + T(PF_LOG_SQRT_2PI);
            const T cvar = s * s + a * (g * g) * a;
            i2c = T(0.5) * pf_rcp_c(cvar);
            kc = T(0.5) * pf_log_c(cvar) + T(PF_LOG_SQRT_2PI);
        }
    }
    // PF_HID_USER_AFFINE with ONE transition scale per column (pf_filter_args.user_scale_per_column) under a linear-Gaussian
    // observation of a scalar state: the closed forms with g = that scale - the one-step mean is the caller's (UserMS::loc,
    // gathered at the parent), everything that does not depend on the particle is evaluated here once per thread instead of per
    // particle (the generic arithmetic: two reciprocals, a square root and two logarithms per particle for the optimal proposal)
    __device__ __forceinline__ void prepare_user(const ModelDesc& md, const ColParams<T, D>& cp, T g_user) {
        if constexpr (D == 1) {
            if (md.obs_kind != PF_OBS_LINEAR) return;
            fast = true;
            g = g_user;
            ou_e = T(0);
            finish_fast(md, cp);
        }
    }

    // the entries that carry an observation, recomputed alone when only y changed since prepare() (the operations of
    // prepare, in its order): the column-persistent kernel keeps one ColConsts per run and calls this once per step
    __device__ __forceinline__ void set_obs(const ColParams<T, D>& cp) {
        if constexpr (D == 1) {
            if (!fast) return;
            yb = cp.y[0] - cp.ob[0];
            ybn = cp.yn[0] - cp.ob[0];
            c_y = cov * (a * (ovi * yb));
        }
    }

    // one-step mean of the scalar state (the scale is `g`)
    __device__ __forceinline__ T loc1(const ModelDesc& md, const ColParams<T, D>& cp, T x) const {
        switch (md.hid_kind) {
            case PF_HID_LINEAR: return cp.hp[0][0] + cp.hp[1][0] * x;
            case PF_HID_SINE_EM: return x + pf_sin(x - cp.hp[0][0]) * (T)md.dt;
            default: return cp.hp[1][0] + (x - cp.hp[1][0]) * ou_e;  // OU
        }
    }
    // log N(y; b + A x, diag(s^2)) with the reciprocals / logs hoisted (any D)
    __device__ __forceinline__ T obs_lp_lin(const ColParams<T, D>& cp, const T (&x)[D], bool next) const {
        T lp = -oks;
#pragma unroll
        for (int o = 0; o < MAXO; ++o) {
            if (o < cp.O) {
                T r = (next ? cp.yn[o] : cp.y[o]) - cp.ob[o];
#pragma unroll
                for (int d = 0; d < D; ++d) r -= cp.A[o][d] * x[d];
                lp -= (r * r) * oi2s[o];
            }
        }
        return lp;
    }
    __device__ __forceinline__ T obs_lp(T x, bool next = false) const {
        const T r = (next ? ybn : yb) - a * x;
        return -(r * r) * i2s - ks;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// The scalar closed forms in canonical shape, one record per column and RUN: they are functions of the parameters alone
// except for the three entries that carry an observation (y_t - b, y_{t+1} - b, c_y), which the step kernel derives from
// the record's (b, 1 / s^2, cov) and its two scalar observation loads.  k_fused_reduce evaluates ColConsts::prepare once
// per run and stores the record; the step kernel reads it with scalar loads (constant address space: the values sit in
// SGPRs, no per-thread recomputation).
//   one-step mean:  A2 == 0:  loc = A0 + (x - A3) * A1      (linear AR: A3 = 0; OU: A0 = A3 = gamma, A1 = e^{-kappa dt})
//                   A2 != 0:  loc = x + A2 * sin(x - A3)    (sine diffusion, A2 = dt)
// The arithmetic is the one ColConsts::loc1 / sample_and_weight / pre_weight perform (same operations, same order).
// ---------------------------------------------------------------------------------------------------------------
enum {
    PK_A0 = 0, PK_A1, PK_A2, PK_A3, PK_G, PK_INC, PK_A, PK_YB, PK_YBN, PK_I2S, PK_KS, PK_CLOC, PK_CY, PK_KSTD, PK_INVG,
    PK_I2INC, PK_KT, PK_KQ, PK_I2C, PK_KC, PK_OB, PK_OVI, PK_COV, PK_USED, PK_N = 24
};
template <typename T> using const_ptr = const __attribute__((address_space(4))) T*;

template <typename T> __device__ __forceinline__ void write_col_pack(const ModelDesc& md, const ColParams<T, 1>& cp,
                                                                     const ColConsts<T, 1>& cc, T* __restrict__ q) {
    T A0 = T(0), A1 = T(0), A2 = T(0), A3 = T(0);
    switch (md.hid_kind) {
        case PF_HID_LINEAR: A0 = cp.hp[0][0]; A1 = cp.hp[1][0]; break;
        case PF_HID_SINE_EM: A2 = (T)md.dt; A3 = cp.hp[0][0]; break;
        default: A0 = cp.hp[1][0]; A3 = cp.hp[1][0]; A1 = cc.ou_e; break;  // OU
    }
    q[PK_A0] = A0; q[PK_A1] = A1; q[PK_A2] = A2; q[PK_A3] = A3;
    q[PK_G] = cc.g; q[PK_INC] = cc.inc; q[PK_A] = cc.a; q[PK_YB] = cc.yb; q[PK_YBN] = cc.ybn;
    q[PK_I2S] = cc.i2s; q[PK_KS] = cc.ks; q[PK_CLOC] = cc.c_loc; q[PK_CY] = cc.c_y; q[PK_KSTD] = cc.kstd;
    q[PK_INVG] = cc.inv_g; q[PK_I2INC] = cc.i2inc; q[PK_KT] = cc.kt; q[PK_KQ] = cc.kq; q[PK_I2C] = cc.i2c; q[PK_KC] = cc.kc;
    q[PK_OB] = cc.ob; q[PK_OVI] = cc.ovi; q[PK_COV] = cc.cov;
}

template <typename T> struct FastCol {
    T A0, A1, A2, A3, g, inc, a, yb, ybn, i2s, ks, c_loc, c_y, kstd, inv_g, i2inc, kt, kq, i2c, kc;
    // y_t / y_next: this step's and the next step's observation (0 when there is none: the terms are then unused)
    __device__ __forceinline__ void load(const_ptr<T> q, T y_t, T y_next) {
        A0 = q[PK_A0]; A1 = q[PK_A1]; A2 = q[PK_A2]; A3 = q[PK_A3];
        g = q[PK_G]; inc = q[PK_INC]; a = q[PK_A];
        i2s = q[PK_I2S]; ks = q[PK_KS]; c_loc = q[PK_CLOC]; kstd = q[PK_KSTD];
        inv_g = q[PK_INVG]; i2inc = q[PK_I2INC]; kt = q[PK_KT]; kq = q[PK_KQ]; i2c = q[PK_I2C]; kc = q[PK_KC];
        const T ob = q[PK_OB];
        yb = y_t - ob;
        ybn = y_next - ob;
        c_y = q[PK_COV] * (a * (q[PK_OVI] * yb));  // (the operations of ColConsts::prepare, in its order)
    }
    // LK: the shape of the one-step mean as a compile-time constant (0 decided by A2 at run time, 1 affine, 2 sine)
    template <int LK = 0> __device__ __forceinline__ T loc(T x) const {
        if constexpr (LK == 1) return A0 + (x - A3) * A1;
        if constexpr (LK == 2) return x + pf_sin_fast(x - A3) * A2;
        if (A2 != T(0)) return x + pf_sin_fast(x - A3) * A2;
        return A0 + (x - A3) * A1;
    }
    __device__ __forceinline__ T obs_lp(T x, bool next = false) const {
        const T r = (next ? ybn : yb) - a * x;
        return -(r * r) * i2s - ks;
    }
    template <int LK = 0> __device__ __forceinline__ T pre_weight(int proposal, T x, bool next = false) const {
        if (proposal == PF_PROP_BOOTSTRAP) return obs_lp(loc<LK>(x), next);
        const T r = (next ? ybn : yb) - a * x;
        return -(r * r) * i2c - kc;
    }
    // APF: the importance weight and the first-stage weight of the ancestor in one go - both need loc(x), and the
    // compiler does not merge two pf_sin evaluations across its large-argument branch
    template <int LK = 0> __device__ __forceinline__ T sample_and_weight_apf(int proposal, T x, T z, T& xn, T& pre) const {
        const T l = loc<LK>(x);
        if (proposal == PF_PROP_BOOTSTRAP) {
            pre = obs_lp(l);
            xn = l + g * (z * inc);
            return obs_lp(xn);
        }
        const T r = yb - a * x;
        pre = -(r * r) * i2c - kc;
        const T km = c_loc * l + c_y;
        xn = km + kstd * z;
        const T eps = (xn - l) * inv_g;
        return obs_lp(xn) + (-(eps * eps) * i2inc - kt) - (-T(0.5) * z * z - kq);
    }
    template <int LK = 0> __device__ __forceinline__ T sample_and_weight(int proposal, T x, T z, T& xn) const {
        const T l = loc<LK>(x);
        if (proposal == PF_PROP_BOOTSTRAP) {
            xn = l + g * (z * inc);
            return obs_lp(xn);
        }
        const T km = c_loc * l + c_y;
        xn = km + kstd * z;
        const T eps = (xn - l) * inv_g;
        return obs_lp(xn) + (-(eps * eps) * i2inc - kt) - (-T(0.5) * z * z - kq);
    }
};

template <typename T> __device__ __forceinline__ T normal_logpdf(T y, T loc, T scale) {
    const T r = y - loc;
    if constexpr (sizeof(T) == 4) {
        // float: hardware reciprocal / log (1 ulp class) instead of the division and libm expansions (~40 instructions)
        const T z = r * pf_rcp_c(scale);
        return T(-0.5) * (z * z) - pf_log_c(scale) - T(PF_LOG_SQRT_2PI);  // (log of a negative scale stays NaN)
    } else {
        return -(r * r) / (T(2) * scale * scale) - pf_log(scale) - T(PF_LOG_SQRT_2PI);
    }
}

// model.build_density(x).log_prob(y)
template <typename T, int D>
__device__ __forceinline__ T obs_logpdf(const ModelDesc& md, const ColParams<T, D>& cp, const T (&x)[D], bool next = false) {
    if (md.obs_kind == PF_OBS_SV) return normal_logpdf(next ? cp.yn[0] : cp.y[0], cp.ob[0], x[0]);
    T lp = T(0);
#pragma unroll
    for (int o = 0; o < ColParams<T, D>::MAXO; ++o) {
        if (o < cp.O) {
            T loc = cp.ob[o];
#pragma unroll
            for (int d = 0; d < D; ++d) loc += cp.A[o][d] * x[d];
            lp += normal_logpdf(next ? cp.yn[o] : cp.y[o], loc, cp.os[o]);
        }
    }
    return lp;
}

// log density of x_new under the transition started at (loc, scale): Normal(0, inc).log_prob(eps) - log|scale|
template <typename T, int D>
__device__ __forceinline__ T transition_logpdf(const ModelDesc& md, const T (&xn)[D], const T (&loc)[D],
                                               const T (&scale)[D]) {
    const T inc = (T)md.inc_scale;
    T lp = T(0);
    if constexpr (sizeof(T) == 4) {
        const T c = pf_log_c(inc) + T(PF_LOG_SQRT_2PI);
        const T i2 = T(0.5) * pf_rcp_c(inc * inc);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const T eps = (xn[d] - loc[d]) * pf_rcp_c(scale[d]);
            lp += -(eps * eps) * i2 - c - pf_log_c(pf_abs(scale[d]));
        }
    } else {
        const T c = pf_log(inc) + T(PF_LOG_SQRT_2PI);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const T eps = (xn[d] - loc[d]) / scale[d];
            lp += -(eps * eps) / (T(2) * inc * inc) - c - pf_log(pf_abs(scale[d]));
        }
    }
    return lp;
}

// lower Cholesky factor of a K x K SPD matrix (K <= 3), fully unrolled
template <typename T, int K> __device__ __forceinline__ void chol_lower(const T (&a)[K][K], T (&l)[K][K]) {
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) l[i][j] = T(0);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        T s = a[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= l[j][k] * l[j][k];
        const T ljj = pf_sqrt_g(s);
        l[j][j] = ljj;
#pragma unroll
        for (int i = j + 1; i < K; ++i) {
            T t = a[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= l[i][k] * l[j][k];
            l[i][j] = pf_div(t, ljj);
        }
    }
}

// inverse of an SPD K x K matrix through its Cholesky factor
template <typename T, int K> __device__ __forceinline__ void spd_inverse(const T (&a)[K][K], T (&inv)[K][K]) {
    T l[K][K];
    chol_lower<T, K>(a, l);
    T li[K][K];  // L^{-1}, lower
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) li[i][j] = T(0);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        li[j][j] = pf_div(T(1), l[j][j]);
#pragma unroll
        for (int i = j + 1; i < K; ++i) {
            T t = T(0);
#pragma unroll
            for (int k = j; k < i; ++k) t -= l[i][k] * li[k][j];
            li[i][j] = pf_div(t, l[i][i]);
        }
    }
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) {
            T t = T(0);
#pragma unroll
            for (int k = 0; k < K; ++k) t += li[k][i] * li[k][j];  // (L^-T L^-1)_{ij}
            inv[i][j] = t;
        }
}

// PF_HID_USER_AFFINE: a particle's one-step mean and transition scale as the caller's callable evaluated them (the fused
// kernels gather them from pf_filter_args.user_loc / user_scale).  Passed BY VALUE-LIKE REFERENCE to the inlined model
// functions so that it stays in registers (a pointer to a local array that is sometimes null kept the array in scratch and
// miscompiled the propagate-only path of the register-starved kernels: tests/test_column_route_gpu.py NaN cases).
template <typename T, int D> struct UserMS {
    bool on;
    T loc[D], scale[D];
    __device__ __forceinline__ static UserMS none() {
        UserMS u;
        u.on = false;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            u.loc[d] = T(0);
            u.scale[d] = T(1);
        }
        return u;
    }
    // planes (D, B, N): `col0` = index of (component 0, this column, particle 0), `plane` = B * N, `i` = the particle
    // `percol` (pf_filter_args.user_scale_per_column): pscale is a (D, B) array, `b` the column, `nb` = B
    __device__ __forceinline__ void gather(const T* __restrict__ ploc, const T* __restrict__ pscale, int64_t col0, int64_t plane, int64_t i,
                                           bool percol = false, int b = 0, int nb = 0) {
        on = true;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            loc[d] = ploc[col0 + d * plane + i];
            scale[d] = percol ? pscale[d * nb + b] : pscale[col0 + d * plane + i];
        }
    }
    // pf_filter_args.user_dt != 0: what was gathered is the drift f(x) of an Euler-Maruyama process - the one-step mean is
    // x + f dt with x the particle the drift belongs to (the parent)
    __device__ __forceinline__ void euler(const T (&x)[D], T dt) {
        if (dt != T(0)) {
#pragma unroll
            for (int d = 0; d < D; ++d) loc[d] = x[d] + loc[d] * dt;
        }
    }
};

// APF first-stage weight  (proposal.pre_weight(y, x))
template <typename T, int D>
__device__ __forceinline__ T pre_weight(const ModelDesc& md, int proposal, const ColParams<T, D>& cp,
                                        const ColConsts<T, D>& cc, const T (&x)[D], bool next = false,
                                        const UserMS<T, D>& um = UserMS<T, D>::none()) {
    if constexpr (D == 1) {
        if (cc.fast) {
            if (proposal == PF_PROP_BOOTSTRAP) return cc.obs_lp(um.on ? um.loc[0] : cc.loc1(md, cp, x[0]), next);  // log p(y | E[x_t | x_{t-1}])
            const T r = (next ? cc.ybn : cc.yb) - cc.a * x[0];                               // LGO: evaluated at x_{t-1} itself
            return -(r * r) * cc.i2c - cc.kc;
        }
    }
    T loc[D], scale[D];
    if (um.on) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            loc[d] = um.loc[d];
            scale[d] = um.scale[d];
        }
    } else {
        mean_scale<T, D>(md, cp, x, loc, scale);
    }
    if (proposal == PF_PROP_BOOTSTRAP) return cc.lin_fast ? cc.obs_lp_lin(cp, loc, next) : obs_logpdf<T, D>(md, cp, loc, next);

    // LinearGaussianObservations.pre_weight: N(y; b + A x_{t-1}, diag(s^2) + A diag(g^2) A^T)  (linear.py:57-86)
    constexpr int MO = ColParams<T, D>::MAXO;
    T cov[MO][MO], r[MO];
#pragma unroll
    for (int o = 0; o < MO; ++o) {
        const bool on = o < cp.O;
        T lo = cp.ob[o];
#pragma unroll
        for (int d = 0; d < D; ++d) lo += cp.A[o][d] * x[d];
        r[o] = on ? ((next ? cp.yn[o] : cp.y[o]) - lo) : T(0);
#pragma unroll
        for (int p = 0; p < MO; ++p) {
            T c = (o == p) ? (on ? cp.os[o] * cp.os[o] : T(1)) : T(0);
#pragma unroll
            for (int d = 0; d < D; ++d) c += cp.A[o][d] * scale[d] * scale[d] * cp.A[p][d];
            cov[o][p] = c;
        }
    }
    if constexpr (MO == 1) {
        return normal_logpdf(r[0], T(0), pf_sqrt_g(cov[0][0]));
    } else {
        T l[MO][MO];
        chol_lower<T, MO>(cov, l);
        // solve L v = r
        T v[MO], quad = T(0), logdet = T(0);
#pragma unroll
        for (int i = 0; i < MO; ++i) {
            T t = r[i];
#pragma unroll
            for (int k = 0; k < i; ++k) t -= l[i][k] * v[k];
            v[i] = pf_div(t, l[i][i]);
            quad += v[i] * v[i];
            logdet += pf_log_g(l[i][i]);
        }
        return -T(0.5) * quad - logdet - T(cp.O) * T(PF_LOG_SQRT_2PI);
    }
}

// proposal.sample_and_weight(y, prediction): new state and importance weight given the draws z
template <typename T, int D>
__device__ __forceinline__ T sample_and_weight(const ModelDesc& md, int proposal, const ColParams<T, D>& cp,
                                               const ColConsts<T, D>& cc, const T (&x)[D], const T (&z)[D], T (&xn)[D],
                                               const UserMS<T, D>& um = UserMS<T, D>::none()) {
    if constexpr (D == 1) {
        if (cc.fast) {
            const T loc = um.on ? um.loc[0] : cc.loc1(md, cp, x[0]);  // (a user-defined process: the caller's one-step mean of the parent)
            if (proposal == PF_PROP_BOOTSTRAP) {
                xn[0] = loc + cc.g * (z[0] * cc.inc);
                return cc.obs_lp(xn[0]);
            }
            const T km = cc.c_loc * loc + cc.c_y;
            xn[0] = km + cc.kstd * z[0];
            const T eps = (xn[0] - loc) * cc.inv_g;
            // log p(y | x') + log p(x' | x) - log q(x')
            return cc.obs_lp(xn[0]) + (-(eps * eps) * cc.i2inc - cc.kt) - (-T(0.5) * z[0] * z[0] - cc.kq);
        }
    }
    T loc[D], scale[D];
    if (um.on) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            loc[d] = um.loc[d];
            scale[d] = um.scale[d];
        }
    } else {
        mean_scale<T, D>(md, cp, x, loc, scale);
    }
    if (proposal == PF_PROP_BOOTSTRAP) {
        const T inc = (T)md.inc_scale;
#pragma unroll
        for (int d = 0; d < D; ++d) xn[d] = loc[d] + scale[d] * (z[d] * inc);
        return cc.lin_fast ? cc.obs_lp_lin(cp, xn, false) : obs_logpdf<T, D>(md, cp, xn);
    }

    // optimal proposal for linear-Gaussian observations (find_optimal_density, proposals/utils.py:219-267):
    //   precision = diag(g^-2) + A^T diag(s^-2) A ; cov = precision^-1 ; mean = cov (g^-2 m + A^T s^-2 (y - b))
    // Vector states take the mean in INNOVATION form, mean = m + cov A^T s^-2 (y - b - A m): algebraically the line above
    // (cov (g^-2 m + A^T s^-2 A m) = m), but without its cancellation - with |m| ~ 25 and a posterior spread of ~ 1
    // (Lorenz-63) the precision form loses 4 - 5 float32 ulps of the new particle, which the transition density's
    // 1 / (2 inc^2 g^2) = 50 turns into 3e-3 of log-weight (measured, round 5: the reference's own float32 run is 1.5e-3
    // from exact arithmetic there; this form stays within the particle's own rounding).  float64 results agree to 1e-13.
    constexpr int MO = ColParams<T, D>::MAXO;
    T hvi[D], prec[D][D], rhs[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        hvi[d] = pf_div(T(1), scale[d] * scale[d]);
        rhs[d] = (D == 1) ? hvi[d] * loc[d] : T(0);
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) prec[i][j] = (i == j) ? hvi[i] : T(0);
#pragma unroll
    for (int o = 0; o < MO; ++o) {
        if (o < cp.O) {
            const T ovi = pf_div(T(1), cp.os[o] * cp.os[o]);
            T innov = cp.y[o] - cp.ob[o];
            if constexpr (D > 1) {
#pragma unroll
                for (int d = 0; d < D; ++d) innov -= cp.A[o][d] * loc[d];
            }
            const T ry = ovi * innov;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                rhs[i] += cp.A[o][i] * ry;
#pragma unroll
                for (int j = 0; j < D; ++j) prec[i][j] += cp.A[o][i] * ovi * cp.A[o][j];
            }
        }
    }
    T cov[D][D], km[D], l[D][D];
    if constexpr (D == 1) {
        cov[0][0] = pf_div(T(1), prec[0][0]);
    } else {
        spd_inverse<T, D>(prec, cov);
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T t = T(0);
#pragma unroll
        for (int j = 0; j < D; ++j) t += cov[i][j] * rhs[j];
        km[i] = t;  // (D > 1: the mean's offset from m)
    }
    chol_lower<T, D>(cov, l);
    T logq = T(0);
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T t = km[i];
#pragma unroll
        for (int j = 0; j <= i; ++j) t += l[i][j] * z[j];
        xn[i] = (D == 1) ? t : loc[i] + t;  // (one rounding at the particle's magnitude)
        logq += -T(0.5) * z[i] * z[i] - pf_log_g(l[i][i]) - T(PF_LOG_SQRT_2PI);
    }
    return obs_logpdf<T, D>(md, cp, xn) + transition_logpdf<T, D>(md, xn, loc, scale) - logq;
}

}  // namespace pf
