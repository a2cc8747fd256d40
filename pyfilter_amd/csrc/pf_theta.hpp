// pf_theta.hpp - the theta-level arithmetic of an SMC^2 / PMMH move as three small kernels.
//
// A rejuvenation moves B theta-particles (10^2 .. 10^4) of P parameters (<= 8): it fits a Gaussian to the weighted particles
// (pyfilter/inference/utils.py:42-76 `construct_mvn`), proposes theta* = mean + L eps in unconstrained space and maps it back
// through the priors' bijections (mcmc/utils.py:48-50, prior.py:47-123), evaluates log prior(theta*) - log prior(theta), fits
// the reverse kernel to theta*, and accepts per particle (mcmc/utils.py:57-70).  As torch operations that is ~130 launches
// of a few microseconds each per move - the host time of a rejuvenation, with the re-filter (1 ms of kernels) waiting
// behind them.  Here: pf_theta_fit (one workgroup: weights, mean, covariance, Cholesky), pf_theta_propose (one thread per
// theta-particle), pf_theta_accept (one workgroup: both kernels' log-densities, the acceptance test, the acceptance rate).
// All arithmetic in double whatever the tensors' type (B x P numbers); values are read and written in the caller's type.
#pragma once

namespace pf {

#define PF_THETA_PAIRS (PF_THETA_MAXP * (PF_THETA_MAXP + 1) / 2)

struct ThetaPriors {  // device-side copy of pf_theta_priors (kernel argument)
    int P;
    int kind[PF_THETA_MAXP];
    double a[PF_THETA_MAXP], b[PF_THETA_MAXP];
};
struct ThetaOut {  // the P parameter tensors (B,) theta* is written to
    void* x[PF_THETA_MAXP];
};

__device__ __forceinline__ double th_softplus(double v) { return v > 0.0 ? v + log1p(exp(-v)) : log1p(exp(v)); }
__device__ __forceinline__ double th_xlogy(double x, double y) { return x == 0.0 ? 0.0 : x * log(y); }

// constrained value x = bijection(u) of prior `kind` and log p(x) + log |dx / du|: the density of u (prior.py:98-123 -
// `TransformedDistribution(prior, biject_to(support).inv).log_prob(u)` with torch's own formulas per family)
__device__ __forceinline__ void theta_prior(int kind, double a, double b, double u, double& x, double& lp) {
    const double half_log_2pi = 0.91893853320467274178;
    switch (kind) {
        case PF_PRIOR_NORMAL: {  // real support: identity
            x = u;
            const double z = (x - a) / b;
            lp = -0.5 * z * z - log(b) - half_log_2pi;
            break;
        }
        case PF_PRIOR_LOGNORMAL: {  // positive support: x = exp(u)
            x = exp(u);
            const double lx = log(x), z = (lx - a) / b;
            lp = (-0.5 * z * z - log(b) - half_log_2pi) - lx + u;
            break;
        }
        case PF_PRIOR_EXPONENTIAL: {  // rate a
            x = exp(u);
            lp = log(a) - a * x + u;
            break;
        }
        case PF_PRIOR_GAMMA: {  // concentration a, rate b
            x = exp(u);
            lp = th_xlogy(a, b) + th_xlogy(a - 1.0, x) - b * x - lgamma(a) + u;
            break;
        }
        case PF_PRIOR_HALFNORMAL: {  // scale a
            x = exp(u);
            const double z = x / a;
            lp = (-0.5 * z * z - log(a) - half_log_2pi) + 0.69314718055994530942 + u;
            break;
        }
        case PF_PRIOR_BETA: {  // concentration1 a, concentration0 b; unit interval: x = sigmoid(u) (clamped as torch clamps it)
            double s = 1.0 / (1.0 + exp(-u));
            s = s < 2.2250738585072014e-308 ? 2.2250738585072014e-308 : (s > 1.0 - 2.220446049250313e-16 ? 1.0 - 2.220446049250313e-16 : s);
            x = s;
            lp = th_xlogy(a - 1.0, x) + th_xlogy(b - 1.0, 1.0 - x) + lgamma(a + b) - lgamma(a) - lgamma(b) - th_softplus(-u) - th_softplus(u);
            break;
        }
        default: {  // PF_PRIOR_UNIFORM: low a, high b; x = a + (b - a) sigmoid(u)
            double s = 1.0 / (1.0 + exp(-u));
            s = s < 2.2250738585072014e-308 ? 2.2250738585072014e-308 : (s > 1.0 - 2.220446049250313e-16 ? 1.0 - 2.220446049250313e-16 : s);
            x = a + (b - a) * s;
            lp = -log(b - a) - th_softplus(-u) - th_softplus(u) + log(fabs(b - a));
            break;
        }
    }
}

// lower Cholesky factor of the P x P matrix c (row-major, PF_THETA_MAXP stride; both in LDS: run-time indices into a thread's
// own arrays would live in scratch); false when it is not positive definite
__device__ __forceinline__ bool theta_cholesky(const double* c, int P, double* l) {
    bool ok = true;
    for (int i = 0; i < P; ++i)
        for (int j = 0; j < P; ++j) l[i * PF_THETA_MAXP + j] = 0.0;
    for (int j = 0; j < P; ++j) {
        double d = c[j * PF_THETA_MAXP + j];
        for (int k = 0; k < j; ++k) d -= l[j * PF_THETA_MAXP + k] * l[j * PF_THETA_MAXP + k];
        if (!(d > 0.0)) ok = false;
        const double dj = sqrt(d);
        l[j * PF_THETA_MAXP + j] = dj;
        for (int i = j + 1; i < P; ++i) {
            double v = c[i * PF_THETA_MAXP + j];
            for (int k = 0; k < j; ++k) v -= l[i * PF_THETA_MAXP + k] * l[j * PF_THETA_MAXP + k];
            l[i * PF_THETA_MAXP + j] = v / dj;
        }
    }
    return ok;
}

// One workgroup: normalised weights of the B log-weights (pyfilter.utils.normalize: NaN / +inf count as -inf; NULL = equal
// weights), weighted mean (P), weighted covariance about it, mean <- mean, chol <- scale * lower Cholesky factor of the
// covariance - or, when that is not positive definite, scale * sqrt(diag) (inference/utils.py:42-57).
template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_theta_fit(const T* __restrict__ values, const T* __restrict__ logw, int64_t B, int P,
                                                        double scale, T* __restrict__ mean_out, T* __restrict__ chol_out) {
    __shared__ double red[PF_THETA_PAIRS * PF_NWAVES];
    __shared__ double redm[PF_NWAVES];
    double mx = -__builtin_huge_val();
    if (logw) {
        for (int64_t i = threadIdx.x; i < B; i += PF_BLOCK) {
            const double v = (double)logw[i];
            const double s = (v != v || v == __builtin_huge_val()) ? -__builtin_huge_val() : v;
            mx = s > mx ? s : mx;
        }
        mx = block_max<double>(mx, redm);
    }
    const bool uniform = !logw || !(mx > -__builtin_huge_val());
    auto weight = [&](int64_t i) -> double {
        if (uniform) return 1.0;
        const double v = (double)logw[i];
        return (v != v || v == __builtin_huge_val()) ? 0.0 : exp(v - mx);
    };
    // sum of the weights and the weighted sums of the values
    double acc[PF_THETA_MAXP + 1];
#pragma unroll
    for (int p = 0; p <= PF_THETA_MAXP; ++p) acc[p] = 0.0;
    for (int64_t i = threadIdx.x; i < B; i += PF_BLOCK) {
        const double w = weight(i);
        acc[PF_THETA_MAXP] += w;
#pragma unroll
        for (int p = 0; p < PF_THETA_MAXP; ++p)
            if (p < P) acc[p] += w * (double)values[i * P + p];
    }
    block_sum<PF_THETA_MAXP + 1>(acc, red);
    const double wsum = acc[PF_THETA_MAXP];
    double m[PF_THETA_MAXP];
#pragma unroll
    for (int p = 0; p < PF_THETA_MAXP; ++p) m[p] = p < P ? acc[p] / wsum : 0.0;
    // weighted covariance about the mean (lower triangle, pair (p, q <= p) at p (p + 1) / 2 + q)
    double cv[PF_THETA_PAIRS];
#pragma unroll
    for (int k = 0; k < PF_THETA_PAIRS; ++k) cv[k] = 0.0;
    for (int64_t i = threadIdx.x; i < B; i += PF_BLOCK) {
        const double w = weight(i) / wsum;
        double c[PF_THETA_MAXP];
#pragma unroll
        for (int p = 0; p < PF_THETA_MAXP; ++p) c[p] = p < P ? (double)values[i * P + p] - m[p] : 0.0;
#pragma unroll
        for (int p = 0; p < PF_THETA_MAXP; ++p)
#pragma unroll
            for (int q = 0; q <= p; ++q) cv[p * (p + 1) / 2 + q] += w * c[p] * c[q];
    }
    __syncthreads();
    block_sum<PF_THETA_PAIRS>(cv, red);
    __shared__ double sc[PF_THETA_MAXP * PF_THETA_MAXP], sl[PF_THETA_MAXP * PF_THETA_MAXP], sm[PF_THETA_MAXP];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int p = 0; p < PF_THETA_MAXP; ++p) {
            sm[p] = m[p];
#pragma unroll
            for (int q = 0; q <= p; ++q) sc[p * PF_THETA_MAXP + q] = sc[q * PF_THETA_MAXP + p] = cv[p * (p + 1) / 2 + q];
        }
        if (!theta_cholesky(sc, P, sl)) {  // not positive definite: the diagonal alone
            for (int p = 0; p < P; ++p)
                for (int q = 0; q < P; ++q) {
                    const double d = sc[p * PF_THETA_MAXP + p];
                    sl[p * PF_THETA_MAXP + q] = (p == q) ? sqrt(d > 0.0 ? d : 0.0) : 0.0;
                }
        }
        for (int p = 0; p < P; ++p) {
            mean_out[p] = (T)sm[p];
            for (int q = 0; q < P; ++q) chol_out[p * P + q] = (T)(scale * sl[p * PF_THETA_MAXP + q]);
        }
    }
}

// One thread per theta-particle: u* = mean + L eps; x* = bijection(u*) into the P parameter tensors; the summed log prior
// of u* (unconstrained space).
template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_theta_propose(ThetaPriors pr, const T* __restrict__ mean, const T* __restrict__ chol,
                                                            const T* __restrict__ eps, int64_t B, T* __restrict__ u_out, ThetaOut out,
                                                            T* __restrict__ prior_out) {
    const int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i >= B) return;
    const int P = pr.P;
    double e[PF_THETA_MAXP], lp = 0.0;
#pragma unroll
    for (int p = 0; p < PF_THETA_MAXP; ++p) e[p] = p < P ? (double)eps[i * P + p] : 0.0;
#pragma unroll
    for (int p = 0; p < PF_THETA_MAXP; ++p) {
        if (p < P) {
            double u = (double)mean[p];
            for (int q = 0; q <= p; ++q) u += (double)chol[p * P + q] * e[q];
            u = (double)(T)u;  // (the value the caller keeps - and every later evaluation starts from - is the stored one)
            double x, l1;
            theta_prior(pr.kind[p], pr.a[p], pr.b[p], u, x, l1);
            lp += l1;
            u_out[i * P + p] = (T)u;
            reinterpret_cast<T*>(out.x[p])[i] = (T)x;
        }
    }
    prior_out[i] = (T)lp;
}

// One workgroup: log_acc = [log q_r(u) - log q_f(u*)] + [prior* - prior] + [ll* - ll]; accepted = log(unif) < log_acc (NaN:
// rejected); rate = mean(accepted).  q_f = N(mean_f, L_f L_f^T) the forward kernel, q_r the reverse one (fit to theta*).
template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_theta_accept(const T* __restrict__ u_cur, const T* __restrict__ u_star,
                                                           const T* __restrict__ mean_f, const T* __restrict__ chol_f,
                                                           const T* __restrict__ mean_r, const T* __restrict__ chol_r,
                                                           const T* __restrict__ prior_cur, const T* __restrict__ prior_star,
                                                           const T* __restrict__ ll_cur, const T* __restrict__ ll_star,
                                                           const T* __restrict__ unif, int64_t B, int P, T* __restrict__ log_acc,
                                                           uint8_t* __restrict__ accepted, T* __restrict__ rate) {
    __shared__ double red[PF_NWAVES];
    // the two kernels' means and factors in LDS (run-time indices into per-thread arrays would live in scratch)
    __shared__ double smf[PF_THETA_MAXP], smr[PF_THETA_MAXP], slf[PF_THETA_MAXP * PF_THETA_MAXP], slr[PF_THETA_MAXP * PF_THETA_MAXP], sh[2];
    if (threadIdx.x < PF_THETA_MAXP * PF_THETA_MAXP) {
        const int p = threadIdx.x / PF_THETA_MAXP, q = threadIdx.x % PF_THETA_MAXP;
        const bool in = p < P && q <= p;
        slf[threadIdx.x] = in ? (double)chol_f[p * P + q] : (p == q ? 1.0 : 0.0);
        slr[threadIdx.x] = in ? (double)chol_r[p * P + q] : (p == q ? 1.0 : 0.0);
        if (q == 0) {
            smf[p] = p < P ? (double)mean_f[p] : 0.0;
            smr[p] = p < P ? (double)mean_r[p] : 0.0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double hf0 = 0.0, hr0 = 0.0;  // sum of the log diagonals
        for (int p = 0; p < P; ++p) {
            hf0 += log(slf[p * PF_THETA_MAXP + p]);
            hr0 += log(slr[p * PF_THETA_MAXP + p]);
        }
        sh[0] = hf0;
        sh[1] = hr0;
    }
    __syncthreads();
    const double hf = sh[0], hr = sh[1];
    const double cst = 0.5 * P * 1.83787706640934548356;  // P / 2 log(2 pi)
    auto log_q = [&](const T* x, const double* m, const double* l, double h) {
        double z[PF_THETA_MAXP], ss = 0.0;
#pragma unroll
        for (int p = 0; p < PF_THETA_MAXP; ++p) {  // forward substitution: z = L^-1 (x - m)  (rows beyond P: identity, z = 0)
            double v = p < P ? (double)x[p] - m[p] : 0.0;
#pragma unroll
            for (int q = 0; q < p; ++q) v -= l[p * PF_THETA_MAXP + q] * z[q];
            z[p] = v / l[p * PF_THETA_MAXP + p];
            ss += z[p] * z[p];
        }
        return -0.5 * ss - h - cst;
    };
    double cnt[1] = {0.0};
    for (int64_t i = threadIdx.x; i < B; i += PF_BLOCK) {
        const double la = (log_q(u_cur + i * P, smr, slr, hr) - log_q(u_star + i * P, smf, slf, hf)) +
                          ((double)prior_star[i] - (double)prior_cur[i]) + ((double)ll_star[i] - (double)ll_cur[i]);
        const T la_t = (T)la;
        log_acc[i] = la_t;
        const bool acc = log((double)unif[i]) < (double)la_t;  // (NaN compares false: a failed proposal is rejected)
        accepted[i] = acc ? 1 : 0;
        cnt[0] += acc ? 1.0 : 0.0;
    }
    block_sum<1>(cnt, red);
    if (threadIdx.x == 0) rate[0] = (T)(cnt[0] / (double)B);
}

// The theta-weights along a block of n observations and their statistics in one launch (sequential/state.py:35-44 applied
// n times): w_path[r] = w0 + (ll[0] + .. + ll[r]) - the running sum accumulated in the tensors' type, observation by
// observation, as `w0 + ll.cumsum(0)` does - and stats[r] = (ESS, every weight finite) of row r (k_theta_ess).  One
// workgroup per row; row r re-adds its r + 1 increments (n <= a few dozen).
template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_theta_path(const T* w0, const T* __restrict__ ll, int64_t B,
                                                         T* w_path, T* __restrict__ stats,  // (n = 1: w_path may BE w0)
                                                         double* host_slot, unsigned long long seq, T* acc = nullptr,
                                                         const int* status = nullptr, int slot_per_row = 0) {
    __shared__ T redm[PF_NWAVES];
    __shared__ double red[3 * PF_NWAVES];
    const int r = blockIdx.x;
    T* row = w_path + (int64_t)r * B;
    // pf_theta_step: the move that produced `ll` reports through `status` - non-zero: it did not happen, nothing is updated
    const int st = status != nullptr ? *status : 0;
    if (st != 0) {
        if (threadIdx.x == 0) {
            stats[2 * (int64_t)r] = T(__builtin_nan(""));
            stats[2 * (int64_t)r + 1] = T(0);
            if (host_slot != nullptr && (slot_per_row || r == (int)gridDim.x - 1)) {
                host_slot += slot_per_row ? 4 * (int64_t)r : 0;
                host_slot[0] = __builtin_nan("");
                host_slot[1] = 0.0;
                reinterpret_cast<unsigned long long*>(host_slot)[3] = (unsigned long long)(unsigned)st;
                __threadfence_system();
                __hip_atomic_store((unsigned long long*)(host_slot + 2), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    if (acc != nullptr && r == 0)  // (pf_theta_step: one row) the filters' running log-likelihood, filters/result.py:130
        for (int64_t i = threadIdx.x; i < B; i += PF_BLOCK) acc[i] = acc[i] + ll[i];
    for (int64_t i = threadIdx.x; i < B; i += PF_BLOCK) {
        T c = ll[i];
        for (int k0 = 1; k0 <= r; k0 += 8) {  // (eight independent loads in flight, then the additions in order)
            T v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (k0 + j <= r) ? ll[(int64_t)(k0 + j) * B + i] : T(0);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j <= r) c = c + v[j];
        }
        row[i] = w0[i] + c;
    }
    // (every thread reads back the entries it wrote itself)
    theta_ess_row<T>(row, B, stats + 2 * (int64_t)r, redm, red);
    // pf_theta_step: the last row's statistics also go to host memory the caller polls (pf_host_alloc: coherent, mapped) - two
    // doubles, then the sequence number with system-scope release, so a host that sees `seq` sees the values
    // pf_theta_path with host rows: EVERY row reports into its own 32-byte slot (a host that polls them in order has row q as soon
    // as row q's workgroup is done - no copy command, no event)
    if (host_slot != nullptr && (slot_per_row || r == (int)gridDim.x - 1) && threadIdx.x == 0) {
        host_slot += slot_per_row ? 4 * (int64_t)r : 0;
        const T* o = stats + 2 * (int64_t)r;  // (thread 0 wrote them)
        host_slot[0] = (double)o[0];
        host_slot[1] = (double)o[1];
        reinterpret_cast<unsigned long long*>(host_slot)[3] = 0ull;
        __threadfence_system();
        __hip_atomic_store((unsigned long long*)(host_slot + 2), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Systematic resampling of the B theta-particles from their log-weights in one launch (kernels/mh.py:52-56 ->
// pyfilter.utils.normalize, resampling.py:24-52): W = softmax of the sanitised weights (NaN / +inf carry none; nothing
// finite: equal weights), cdf = its running sum - accumulated in double, rounded once to the tensors' type, last entry 1 -,
// ancestor of position i = the first j with cdf[j] >= (i + u) / B, at most B - 1.  One workgroup; `cdf` is B scratch values.
template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_theta_resample(const T* __restrict__ logw, int64_t B, double u, int64_t* __restrict__ idx,
                                                             T* __restrict__ cdf) {
    __shared__ double redm[PF_NWAVES];
    __shared__ double red[PF_NWAVES];
    auto clean = [](T v) -> double { return is_nan_or_posinf(v) ? -__builtin_huge_val() : (double)v; };
    double mx = -__builtin_huge_val();
    for (int64_t i = threadIdx.x; i < B; i += PF_BLOCK) {
        const double s = clean(logw[i]);
        mx = s > mx ? s : mx;
    }
    mx = block_max<double>(mx, redm);
    const bool uniform = !(mx > -__builtin_huge_val());
    // contiguous chunks per thread: local sums -> exclusive offsets -> the running sum
    const int64_t chunk = (B + PF_BLOCK - 1) / PF_BLOCK;
    const int64_t lo = (int64_t)threadIdx.x * chunk, hi = lo + chunk < B ? lo + chunk : B;
    double local = 0.0;
    for (int64_t i = lo; i < hi; ++i) local += uniform ? 1.0 : exp(clean(logw[i]) - mx);
    double total;
    double run = block_scan_excl(local, red, total);
    for (int64_t i = lo; i < hi; ++i) {
        run += uniform ? 1.0 : exp(clean(logw[i]) - mx);
        cdf[i] = i == B - 1 ? T(1) : (T)(run / total);
    }
    __threadfence_block();
    __syncthreads();
    for (int64_t i = threadIdx.x; i < B; i += PF_BLOCK) {
        const T p = ((T)i + (T)u) / (T)B;
        int64_t a = 0, n = B;  // lower_bound: the first j with cdf[j] >= p
        while (n > 0) {
            const int64_t h = n >> 1;
            if (cdf[a + h] < p) {
                a += h + 1;
                n -= h + 1;
            } else {
                n = h;
            }
        }
        idx[i] = a < B - 1 ? a : B - 1;
    }
}

}  // namespace pf
