// pf_fused.hpp - the fused SISR / APF time step: TWO kernels per step.
//
//   k_fused_scan   (grid tiles+1 x B) re-reduces the column's per-tile partials; the extra workgroup per column does the
//                                    bookkeeping of the current state (moments row, log-likelihood of the previous step,
//                                    resampling decision); if the column resamples: scans this tile of resampling weights into the
//                                    cdf (fp64 carry, rounded per element) and emits j0[tile'] = the ancestor of the first
//                                    grid position of every position tile whose start falls into this tile's cdf range.
//   k_fused_step   (grid tiles x B)  ancestors of this tile's grid positions (LDS window search from j0) -> gather
//                                    x[anc] -> propagate (Philox / tape) -> weight -> write x', logw', anc -> and, while
//                                    the new state is still in registers, the per-tile partials of the NEXT step
//                                    (online max / sum-exp / sum-exp^2, weighted moments, and - for the APF - the
//                                    first-stage weights against the next observation).
//
// k_fused_reduce produces the partials of the very first state only.  A kernel boundary is the only inter-workgroup
// synchronisation; the step index, "observed" flags and observation rows are kernel arguments set by the host loop, so a
// launch has no dependent-load prologue.
#pragma once

namespace pf {

#define PF_STAMP(a, slot)                                                                       \
    do {                                                                                        \
        if ((a).debug_cut < 0 && !(a).finalize_only && blockIdx.x == (unsigned)(-(a).debug_cut - 1) &&  \
            blockIdx.y == 0 && threadIdx.x == 0)                                                \
            (a).dbg[slot] = (unsigned long long)clock64();                                      \
    } while (0)

template <typename T> struct FusedArgs {
    ModelDesc md;
    const T* params;
    int filter, proposal, resampler;
    Geom g;
    double thr_abs;  // ess_threshold * N
    double logN;
    uint64_t seed;
    const uint64_t* seed_dev;  // optional device word added to `seed` (lets a captured graph draw fresh numbers per replay)
    T* x[2];
    T* logw[2];
    int32_t* anc;
    T* cdf;
    T* pos;      // (B, N) sorted resampling positions (multinomial)
    const T* y;  // (T, y_rows, O)
    int y_rows;
    const T* z_tape;
    const T* u_tape;
    T* means;
    T* vars;
    T* ll_steps;
    T* ll_total;
    double* part;
    ColStat* stat;
    int32_t* poison;  // [2][B]
    int32_t* j0;      // [B][tiles]
    // per launch
    int step;       // local step index: slot = step & 1 is read, the other written
    int obs;        // this step weighs against y[step]
    int obs_next;   // the next step exists and is a weighted step (its first-stage weights are prepared here)
    int finalize_only;
    unsigned long long* dbg;
    int debug_cut;  // development knob (env PF_DEBUG_CUT): kernels return early after stage n; 0 = off
};

template <typename T, int D>
__device__ __forceinline__ void load_col_params(const FusedArgs<T>& a, int b, int step, bool with_y, ColParams<T, D>& cp) {
    const int O = a.md.obs_dim;
    const int NP = 4 * D + O * D + 2 * O;
    cp.load(a.params + (int64_t)b * NP, O,
            with_y ? a.y + ((int64_t)step * a.y_rows + (a.y_rows == 1 ? 0 : b)) * O : nullptr);
}

// Accumulates the per-tile partials of a state from registers.
template <typename T, int D> struct PartialAcc {
    OnlineLse<T> a1, a2;
    double q1, mx[D], mxx[D];
    double es;  // sum of the Exp(1) spacings the next step's sorted-uniform multinomial will use for these particles
    bool poison;
    __device__ __forceinline__ void init() {
        a1.init();
        a2.init();
        q1 = 0.0;
        es = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) mx[d] = mxx[d] = 0.0;
        poison = false;
    }
    // One round of VEC particles: lw sanitised log-weights, x particles, pre (if pre_on) first-stage log-weights of the
    // next step.  The running maxima move at most once per round, so there is one exp per element (+ one per rescale).
    template <int VEC>
    __device__ __forceinline__ void push_round(const T (&lw)[VEC], const T (&x)[D][VEC], bool pre_on, const T (&pre)[VEC]) {
        T m = lw[0];
#pragma unroll
        for (int j = 1; j < VEC; ++j) m = (lw[j] > m) ? lw[j] : m;
        if (m > a1.m) {
            const double rs = (a1.m == -Lim<T>::inf()) ? 0.0 : (double)pf_exp_w(a1.m - m);
            a1.s *= rs;
            q1 *= rs * rs;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                mx[d] *= rs;
                mxx[d] *= rs;
            }
            a1.m = m;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            double e = (lw[j] == -Lim<T>::inf()) ? 0.0 : (double)pf_exp_w(lw[j] - a1.m);
            if (lw[j] != lw[j]) e = (double)lw[j];
            a1.s += e;
            q1 += e * e;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const double xd = (double)x[d][j];
                mx[d] += e * xd;
                mxx[d] += e * xd * xd;
            }
        }
        if (pre_on) {
            T rw[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (pre[j] != pre[j] || pre[j] == Lim<T>::inf()) poison = true;
                rw[j] = sanitize_logw(pre[j] + lw[j]);
            }
            T m2 = rw[0];
#pragma unroll
            for (int j = 1; j < VEC; ++j) m2 = (rw[j] > m2) ? rw[j] : m2;
            if (m2 > a2.m) {
                a2.s *= (a2.m == -Lim<T>::inf()) ? 0.0 : (double)pf_exp_w(a2.m - m2);
                a2.m = m2;
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) a2.s += (rw[j] == -Lim<T>::inf()) ? 0.0 : (double)pf_exp_w(rw[j] - a2.m);
        }
    }
    // workgroup reduction + store in two LDS exchanges (maxima, then every rescaled sum); `red` >= (4 + 2D) * PF_NWAVES
    // doubles, `redm` >= 2 * PF_NWAVES Ts, neither used by anything still in flight
    __device__ __forceinline__ void finish(double* part, int b, int k, int B, int tiles, bool pre_on, double* red, T* redm,
                                           int32_t* poison_slot) {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        const T w1 = wave_max<T>(a1.m), w2 = wave_max<T>(a2.m);
        if (lane == 0) {
            redm[wid] = w1;
            redm[PF_NWAVES + wid] = w2;
        }
        __syncthreads();
        T M1 = redm[0], M2 = redm[PF_NWAVES];
#pragma unroll
        for (int w = 1; w < PF_NWAVES; ++w) {
            M1 = (redm[w] > M1) ? redm[w] : M1;
            M2 = (redm[PF_NWAVES + w] > M2) ? redm[PF_NWAVES + w] : M2;
        }
        const double f1 = exp_diff_t<T>((double)a1.m, (double)M1);
        double sums[4 + 2 * D];
        sums[0] = a1.s * f1;
        sums[1] = q1 * f1 * f1;
        sums[2] = pre_on ? a2.s * exp_diff_t<T>((double)a2.m, (double)M2) : 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            sums[3 + d] = mx[d] * f1;
            sums[3 + D + d] = mxx[d] * f1;
        }
        sums[3 + 2 * D] = es;
#pragma unroll
        for (int q = 0; q < 4 + 2 * D; ++q) {
            const double ws = wave_sum(sums[q]);
            if (lane == 0) red[q * PF_NWAVES + wid] = ws;
        }
        if (poison) atomicOr(poison_slot, 1);
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < 4 + 2 * D; ++q) {
                double r = red[q * PF_NWAVES];
#pragma unroll
                for (int w = 1; w < PF_NWAVES; ++w) r += red[q * PF_NWAVES + w];
                sums[q] = r;
            }
            const int64_t stride = (int64_t)B * tiles;
            const int64_t o = (int64_t)b * tiles + k;
            part[PQ_M1 * stride + o] = (double)M1;
            part[PQ_S1 * stride + o] = sums[0];
            part[PQ_Q1 * stride + o] = sums[1];
            part[PQ_M2 * stride + o] = pre_on ? (double)M2 : -__builtin_huge_val();
            part[PQ_S2 * stride + o] = sums[2];
            part[PQ_E * stride + o] = sums[3 + 2 * D];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                part[(PQ_MX + d) * stride + o] = sums[3 + d];
                part[(PQ_MX + D + d) * stride + o] = sums[3 + D + d];
            }
        }
    }
};

// partials of the state in slot (step & 1) - only needed for the first state of a run
template <typename T, int D, int VEC>
__global__ __launch_bounds__(PF_BLOCK) void k_fused_reduce(FusedArgs<T> a) {
    __shared__ double red[(4 + 2 * D) * PF_NWAVES];
    __shared__ T redm[2 * PF_NWAVES];
    const Geom& g = a.g;
    const int b = blockIdx.y, k = blockIdx.x;
    const int slot = a.step & 1;
    const bool pre_on = a.obs && a.filter == PF_FILTER_APF;
    ColParams<T, D> cp;
    load_col_params<T, D>(a, b, a.step, pre_on, cp);
    ColConsts<T, D> cc;
    cc.prepare(a.md, cp);

    const T* lw_col = a.logw[slot] + (int64_t)b * g.N;
    const T* x_base = a.x[slot];
    PartialAcc<T, D> acc;
    acc.init();
    const int64_t base = (int64_t)k * g.tile_elems;
    for (int r = 0; r < g.rounds_per_tile; ++r) {
        const int64_t i0 = base + (int64_t)r * g.round_elems + threadIdx.x * VEC;
        if (i0 >= g.N) break;
        T lw[VEC], xv[D][VEC];
        if (VEC == 1) lw[0] = lw_col[i0]; else load_vec<T, VEC>(lw_col + i0, lw);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const T* xc = x_base + ((int64_t)d * g.B + b) * g.N + i0;
            if (VEC == 1) xv[d][0] = xc[0]; else load_vec<T, VEC>(xc, xv[d]);
        }
        T pre[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            T xj[D];
#pragma unroll
            for (int d = 0; d < D; ++d) xj[d] = xv[d][j];
            pre[j] = pre_on ? pre_weight<T, D>(a.md, a.proposal, cp, cc, xj) : T(0);
        }
        acc.template push_round<VEC>(lw, xv, pre_on, pre);
        if (a.resampler == PF_RESAMPLE_MULTINOMIAL) {
            T ev[VEC];
            draw_exponentials<T, VEC>(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), PF_STREAM_MULTINOMIAL, (uint32_t)a.step,
                                      (uint64_t)((int64_t)b * g.N + i0), ev);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc.es += (double)ev[j];
        }
    }
    acc.finish(a.part, b, k, g.B, g.tiles, pre_on, red, redm, &a.poison[(a.step & 1) * g.B + b]);
}

// Emits j0[t] = i for every position tile t whose first grid position p_t = (t * tile + u) / N satisfies
// c_prev < p_t <= c  (searchsorted side=left).  The intervals (c_prev, c] of consecutive elements partition (-1, 1], so
// every tile start is claimed by exactly one element.  Candidates come from the real-valued inverse of p_t (cheap early
// out: almost no element contains a tile start), membership from the exact fp test of resampling.py:44-51.
// One heavy particle (degenerate weights) can own the first grid position of hundreds of position tiles; such wide
// intervals are queued in LDS and expanded by the whole workgroup instead of one thread looping over them.
#define PF_WIDE_MAX 32
template <typename T> struct WideEntry {
    T c_prev, c;
    int i, ta, tb;
};
template <typename T>
__device__ __forceinline__ void emit_j0(T c_prev, T c, int64_t i, T u, int64_t N, int tile_elems, int tiles,
                                        int32_t* __restrict__ j0_col, WideEntry<T>* wide, int* wide_cnt) {
    // evaluated in T: c * N carries a relative error of eps, i.e. <= eps * N / tile = eps * tiles in tile units; the fp
    // grid formula itself adds as much again.  delta covers both with a wide margin and is still << 1, so only the rare
    // element whose interval really comes close to a tile start runs the exact test below.
    const T inv = T(1) / T(tile_elems);
    const T delta = T(8) * T(tiles) * (sizeof(T) == 4 ? T(1.1920929e-7) : T(2.220446e-16)) + T(1e-4);
    const T lo = (c_prev * T(N) - u) * inv - delta;
    const T hi = (c * T(N) - u) * inv + delta;
    const T th = floor(hi);
    if (th < lo) return;  // no integer in [lo, hi]
    int64_t ta = (int64_t)ceil(lo), tb = (int64_t)th;
    if (ta < 0) ta = 0;
    if (tb > tiles - 1) tb = tiles - 1;
    if (tb - ta >= 8) {
        const int slot = atomicAdd(wide_cnt, 1);
        if (slot < PF_WIDE_MAX) {
            wide[slot].c_prev = c_prev;
            wide[slot].c = c;
            wide[slot].i = (int)i;
            wide[slot].ta = (int)ta;
            wide[slot].tb = (int)tb;
            return;
        }
    }
    const T nT = T(N);
    for (int64_t t = ta; t <= tb; ++t) {
        const T p = grid_position<T>(t * tile_elems, u, nT);
        if (c_prev < p && p <= c) j0_col[t] = (int32_t)i;
    }
}
template <typename T>
__device__ __forceinline__ void emit_wide(const WideEntry<T>* wide, int n, T u, int64_t N, int tile_elems,
                                          int32_t* __restrict__ j0_col) {
    const T nT = T(N);
    for (int e = 0; e < n; ++e) {
        const WideEntry<T> w = wide[e];
        for (int t = w.ta + (int)threadIdx.x; t <= w.tb; t += PF_BLOCK) {
            const T p = grid_position<T>((int64_t)t * tile_elems, u, nT);
            if (w.c_prev < p && p <= w.c) j0_col[t] = w.i;
        }
    }
}

// Combined column statistics from the per-tile partials.  The first PF_BLOCK tiles' values are passed in registers
// (`early`, loaded at the very top of the kernel so their latency overlaps the tile's own work); further tiles are
// read in the loops.  Two LDS exchanges: maxima, then every rescaled sum.
struct ColCombine {
    double m1, m2, S1, Q1, S2, prefK, prefK1;
    double TE, prefE;  // multinomial: total / prefix (tiles below k) of the Exp(1) spacings
};
#define PF_COMBINE_ITERS (PF_MAX_TILES / PF_BLOCK)  // partial records per thread
struct EarlyPartials {
    double m1[PF_COMBINE_ITERS], s1[PF_COMBINE_ITERS], q1[PF_COMBINE_ITERS], m2[PF_COMBINE_ITERS], s2[PF_COMBINE_ITERS];
};
// the Exp(1)-spacing prefix for the sorted-uniform multinomial (its own small reduction; systematic runs skip it)
template <typename T>
__device__ __forceinline__ void combine_spacings(const FusedArgs<T>& a, int64_t cb, int64_t stride, int k, double* red,
                                                 double& total, double& prefix) {
    double v[2] = {0.0, 0.0};
    for (int t = threadIdx.x; t < a.g.tiles; t += PF_BLOCK) {
        const double e = a.part[PQ_E * stride + cb + t];
        v[0] += e;
        if (t < k) v[1] += e;
    }
    block_sum<2>(v, red);
    total = v[0];
    prefix = v[1];
}
template <typename T>
__device__ __forceinline__ void load_early_partials(const FusedArgs<T>& a, int64_t cb, int64_t stride, bool two,
                                                    EarlyPartials& e) {
#pragma unroll
    for (int it = 0; it < PF_COMBINE_ITERS; ++it) {
        const int t = threadIdx.x + it * PF_BLOCK;
        e.m1[it] = e.m2[it] = -__builtin_huge_val();
        e.s1[it] = e.q1[it] = e.s2[it] = 0.0;
        if (t < a.g.tiles) {
            e.m1[it] = a.part[PQ_M1 * stride + cb + t];
            e.s1[it] = a.part[PQ_S1 * stride + cb + t];
            e.q1[it] = a.part[PQ_Q1 * stride + cb + t];
            if (two) {
                e.m2[it] = a.part[PQ_M2 * stride + cb + t];
                e.s2[it] = a.part[PQ_S2 * stride + cb + t];
            }
        }
    }
}
template <typename T>
__device__ __forceinline__ ColCombine combine_column(const FusedArgs<T>& a, const EarlyPartials& e, int64_t cb,
                                                     int64_t stride, int k, bool two, double* red, double* redm) {
    const Geom& g = a.g;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double m1 = e.m1[0], m2 = e.m2[0];
#pragma unroll
    for (int it = 1; it < PF_COMBINE_ITERS; ++it) {
        m1 = fmax(m1, e.m1[it]);
        m2 = fmax(m2, e.m2[it]);
    }
    m1 = wave_max(m1);
    m2 = wave_max(m2);
    if (lane == 0) {
        redm[wid] = m1;
        redm[PF_NWAVES + wid] = m2;
    }
    __syncthreads();
    m1 = redm[0];
    m2 = redm[PF_NWAVES];
#pragma unroll
    for (int w = 1; w < PF_NWAVES; ++w) {
        m1 = fmax(m1, redm[w]);
        m2 = fmax(m2, redm[PF_NWAVES + w]);
    }
    // S1, Q1, S2 and the resampling-weight prefix below tile k / below tile k+1.  Both prefixes use the same masked
    // loop + reduction tree, so tile k's "next" prefix is bit-identical to what tile k+1 computes as its own.
    double v[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < PF_COMBINE_ITERS; ++it) {
        const int t = threadIdx.x + it * PF_BLOCK;
        if (t < g.tiles) {
            const double f = exp_diff_t<T>(e.m1[it], m1);
            const double s = e.s1[it] * f;
            v[0] += s;
            v[1] += e.q1[it] * f * f;
            double sr = s;
            if (two) {
                sr = e.s2[it] * exp_diff_t<T>(e.m2[it], m2);
                v[2] += sr;
            }
            if (t < k) v[3] += sr;
            if (t < k + 1) v[4] += sr;
        }
    }
    block_sum<5>(v, red);
    ColCombine c;
    c.m1 = m1;
    c.m2 = m2;
    c.S1 = v[0];
    c.Q1 = v[1];
    c.S2 = v[2];
    c.prefK = v[3];
    c.prefK1 = v[4];
    return c;
}

// grid (tiles + 1, B): workgroups k < tiles scan their tile; the extra workgroup k == tiles is the column's bookkeeper
// (moments row, log-likelihood increment, resampling decision) - kept off the scanning workgroups' critical path.
template <typename T, int D, int VEC>
__global__ __launch_bounds__(PF_BLOCK, sizeof(T) == 4 ? 4 : 1) void k_fused_scan(FusedArgs<T> a) {
    __shared__ double red[6 * PF_NWAVES];
    __shared__ double redm[2 * PF_NWAVES];
    __shared__ double red2[2 * D * PF_NWAVES];
    __shared__ double reds[PF_NWAVES];
    __shared__ T lastv[PF_BLOCK + 1];  // every thread's last cdf value of the round (+ the previous round's last one)
    __shared__ WideEntry<T> wide[PF_WIDE_MAX];
    __shared__ int wide_cnt;
    const Geom& g = a.g;
    const int b = blockIdx.y, k = blockIdx.x;
    const int step = a.step;
    const int slot = step & 1;
    const bool obs = !a.finalize_only && a.obs;
    const bool apf = a.filter == PF_FILTER_APF;
    const bool two = apf && obs;  // a second (m2, S2) set of partials is live
    const int64_t stride = (int64_t)g.B * g.tiles;
    const int64_t cb = (int64_t)b * g.tiles;
    if (a.debug_cut == 1) return;
    PF_STAMP(a, 0);
    EarlyPartials early;
    load_early_partials<T>(a, cb, stride, two, early);

    if (k == g.tiles) {
        // ------------------------------------------------ bookkeeper ------------------------------------------------------
        const ColCombine c = combine_column<T>(a, early, cb, stride, 0, two, red, redm);
        const double lse_w = c.m1 + log(c.S1);
        const double ess = c.S1 * c.S1 / c.Q1;
        bool resample = apf ? obs : (ess < a.thr_abs);  // apf.py:29-31 | sisr.py:18-19
        if (a.finalize_only) resample = false;
        double mv[2 * D];
#pragma unroll
        for (int q = 0; q < 2 * D; ++q) {
            mv[q] = 0.0;
            for (int t = threadIdx.x; t < g.tiles; t += PF_BLOCK)
                mv[q] += a.part[(PQ_MX + q) * stride + cb + t] * exp_diff_t<T>(a.part[PQ_M1 * stride + cb + t], c.m1);
        }
        block_sum<2 * D>(mv, red2);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) {  // moments of the current state -> row `step` of filter_means / filter_variance
                const double mu = mv[d] / c.S1;
                double var = mv[D + d] / c.S1 - mu * mu;
                if (var < 0.0) var = 0.0;
                a.means[((int64_t)step * g.B + b) * D + d] = (T)mu;
                a.vars[((int64_t)step * g.B + b) * D + d] = (T)var;
            }
            ColStat st = a.stat[b];
            // log-likelihood increment of the previous step: ll = lse(logw') - base_lse   (0 for unweighted steps)
            if (step > 0 && !st.ll_done) {
                double ll = 0.0;
                const int pslot = (step - 1) & 1;
                if (st.prev_observed) {
                    ll = lse_w - st.base_lse;
                    if (a.poison[pslot * g.B + b]) ll = __builtin_nan("");
                }
                a.poison[pslot * g.B + b] = 0;
                a.ll_steps[(int64_t)(step - 1) * g.B + b] = (T)ll;
                a.ll_total[b] = (T)((double)a.ll_total[b] + ll);
            }
            st.lse_w = lse_w;
            st.resample = resample ? 1 : 0;
            st.ll_done = a.finalize_only ? 1 : 0;
            if (!a.finalize_only) {
                st.prev_observed = obs ? 1 : 0;
                // ll_t = [lse(w') - log N] + [lse(rw) - lse(w)] (apf.py:44) | lse(wi + log W), W = 1/N after resampling
                // else the carried weights (sisr.py:52-55)
                st.base_lse = apf ? a.logN - ((c.m2 + log(c.S2)) - lse_w) : (resample ? a.logN : lse_w);
            }
            a.stat[b] = st;
        }
        return;
    }
    if (a.finalize_only) return;

    // ---------------------------------------------------- scanner ---------------------------------------------------------
    // 1. everything that does not need the column totals: this tile's log-weights (and particles for the APF's
    //    in-register pre-weight), its own maximum, the exponentials and the workgroup-local scan (one round per tile is
    //    the common case; multi-round tiles take the generic loop below)
    const T* lw_col = a.logw[slot] + (int64_t)b * g.N;
    const T* x_base = a.x[slot];
    const int64_t base = (int64_t)k * g.tile_elems;
    const int64_t tile_last = (base + g.tile_elems < g.N ? base + g.tile_elems : g.N) - 1;
    const double mk = a.part[(two ? PQ_M2 : PQ_M1) * stride + cb + k];
    const T tile_max = (T)mk;
    ColParams<T, D> cp;
    ColConsts<T, D> cc;
    cc.fast = false;
    cc.lin_fast = false;
    if (two) {
        load_col_params<T, D>(a, b, step, true, cp);
        cc.prepare(a.md, cp);
    }
    const bool sys = a.resampler == PF_RESAMPLE_SYSTEMATIC;
    const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
    const T ub = !sys ? T(0)
                      : (a.u_tape ? a.u_tape[(int64_t)step * g.B + b]
                                  : uniform_draw<T>(seed, PF_STREAM_UNIFORM, (uint32_t)step, (uint64_t)b));
    T* cdf_col = a.cdf + (int64_t)b * g.N;
    int32_t* j0_col = a.j0 + cb;

    auto tile_exponentials = [&](int64_t i0, bool on, double (&e)[VEC]) -> double {
        T lw[VEC], xv[D][VEC];
        if (on) {
            if (VEC == 1) lw[0] = lw_col[i0]; else load_vec<T, VEC>(lw_col + i0, lw);
            if (two) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const T* xc = x_base + ((int64_t)d * g.B + b) * g.N + i0;
                    if (VEC == 1) xv[d][0] = xc[0]; else load_vec<T, VEC>(xc, xv[d]);
                }
            }
        }
        double local = 0.0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            double ej = 0.0;
            if (on) {
                T rw = lw[j];
                if (two) {  // APF: rw = sanitize(pre_weight(x, y) + logw), recomputed in registers, never stored
                    T xj[D];
#pragma unroll
                    for (int d = 0; d < D; ++d) xj[d] = xv[d][j];
                    rw = sanitize_logw(pre_weight<T, D>(a.md, a.proposal, cp, cc, xj) + lw[j]);
                }
                ej = (rw == -Lim<T>::inf()) ? 0.0 : (double)pf_exp_w(rw - tile_max);
            }
            local += ej;
            e[j] = local;  // thread-local inclusive
        }
        return local;
    };
    // cdf values of one round from the scanned exponentials; stores them and emits the position-tile starts
    auto write_round = [&](int r, int64_t i0, bool on, const double (&e)[VEC], double offset, double Pk, double fk,
                           double Pnext) {
        T outv[VEC];
        if (on) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                double cc_ = Pk + fk * (offset + e[j]);
                if (cc_ > Pnext) cc_ = Pnext;
                T c = (T)cc_;
                // the tile's last element is pinned to T(prefix(k+1)) - the value tile k+1 starts from - and the column's
                // last element to 1 (cumsum[..., -1] = 1, resampling.py:49)
                if (i0 + j == tile_last) c = (i0 + j == g.N - 1) ? T(1) : (T)Pnext;
                outv[j] = c;
            }
            if (VEC == 1) cdf_col[i0] = outv[0]; else store_vec<T, VEC>(cdf_col + i0, outv);
        }
        if (sys) {
            // the *stored* cdf value preceding this thread's first element: previous thread's last value (LDS)
            lastv[threadIdx.x + 1] = on ? outv[VEC - 1] : T(1);
            if (threadIdx.x == 0) {
                wide_cnt = 0;
                if (r == 0) lastv[0] = (k == 0) ? T(-1) : (T)Pk;
            }
            __syncthreads();
            if (on) {
                T c_prev = lastv[threadIdx.x];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    emit_j0<T>(c_prev, outv[j], i0 + j, ub, g.N, g.tile_elems, g.tiles, j0_col, wide, &wide_cnt);
                    c_prev = outv[j];
                }
            }
            __syncthreads();
            const int nw = wide_cnt < PF_WIDE_MAX ? wide_cnt : PF_WIDE_MAX;
            if (nw > 0) emit_wide<T>(wide, nw, ub, g.N, g.tile_elems, j0_col);
            if (threadIdx.x == 0) lastv[0] = lastv[PF_BLOCK];
            if (nw > 0) __syncthreads();
        }
    };

    const bool single = g.rounds_per_tile == 1;
    double e0[VEC], excl0 = 0.0;
    const int64_t i00 = base + threadIdx.x * VEC;
    if (single) {
        double total;
        const double local = tile_exponentials(i00, i00 < g.N, e0);
        excl0 = block_scan_excl(local, reds, total);
    }
    PF_STAMP(a, 1);
    if (!sys) {
        // sorted-uniform multinomial: position i = (sum of Exp(1) spacings up to i) / (sum of all N + 1 spacings) - the
        // order statistics of N iid uniforms; the spacings are regenerated from Philox, scanned like the weights
        double TE, prefE;
        combine_spacings<T>(a, cb, stride, k, red, TE, prefE);
        T tail[1];
        draw_exponentials<T, 1>(seed, PF_STREAM_MULTINOMIAL, (uint32_t)step, (uint64_t)((int64_t)g.B * g.N + b), tail);
        const double inv = 1.0 / (TE + (double)tail[0]);
        double carryE = 0.0;
        T* pos_col = a.pos + (int64_t)b * g.N;
        for (int r = 0; r < g.rounds_per_tile; ++r) {
            const int64_t r0 = base + (int64_t)r * g.round_elems;
            if (r0 >= g.N) break;
            const int64_t i0 = r0 + threadIdx.x * VEC;
            const bool on = i0 < g.N;
            T ev[VEC];
            double incl[VEC], local = 0.0, total;
            if (on) draw_exponentials<T, VEC>(seed, PF_STREAM_MULTINOMIAL, (uint32_t)step, (uint64_t)((int64_t)b * g.N + i0), ev);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                local += on ? (double)ev[j] : 0.0;
                incl[j] = local;
            }
            const double excl = block_scan_excl(local, reds, total);
            if (on) {
                T pv[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) pv[j] = (T)((prefE + carryE + excl + incl[j]) * inv);
                if (VEC == 1) pos_col[i0] = pv[0]; else store_vec<T, VEC>(pos_col + i0, pv);
            }
            carryE += total;
        }
    }

    // 2. the column totals
    const ColCombine c = combine_column<T>(a, early, cb, stride, k, two, red, redm);
    PF_STAMP(a, 2);
    const bool resample = apf ? obs : (c.S1 * c.S1 / c.Q1 < a.thr_abs);
    if (!resample || a.debug_cut == 2) return;
    const double MR = two ? c.m2 : c.m1, SR = two ? c.S2 : c.S1;
    const double fk = exp_diff_t<T>(mk, MR) / SR;
    const double Pk = c.prefK / SR;
    const double Pnext = c.prefK1 / SR;
    PF_STAMP(a, 3);

    // 3. cdf = P_k + f_k * (scan), rounded per element
    if (single) {
        write_round(0, i00, i00 < g.N, e0, excl0, Pk, fk, Pnext);
        PF_STAMP(a, 6);
        return;
    }
    double carry = 0.0;
    for (int r = 0; r < g.rounds_per_tile; ++r) {
        const int64_t r0 = base + (int64_t)r * g.round_elems;
        if (r0 >= g.N) break;
        const int64_t i0 = r0 + threadIdx.x * VEC;
        const bool on = i0 < g.N;
        double e[VEC], total;
        const double local = tile_exponentials(i0, on, e);
        const double excl = block_scan_excl(local, reds, total);
        write_round(r, i0, on, e, carry + excl, Pk, fk, Pnext);
        carry += total;
    }
}

template <typename T, int D, int VEC>
__global__ __launch_bounds__(PF_BLOCK, (sizeof(T) == 4 && D == 1) ? 4 : 1) void k_fused_step(FusedArgs<T> a) {
    constexpr int WIN = SearchWin<T, VEC>::WIN;
    // the particles behind the cdf window are staged in LDS too when they are small (<= 8 B per particle), so the
    // ancestor gather is an LDS read instead of a second dependent global round trip
    constexpr bool XWIN = (sizeof(T) * D <= 8);
    __shared__ __attribute__((aligned(32))) T win[WIN];
    __shared__ __attribute__((aligned(32))) T xwin[XWIN ? D * WIN : VEC];
    __shared__ int sh_j0;
    __shared__ double red[(4 + 2 * D) * PF_NWAVES];
    __shared__ T redm[2 * PF_NWAVES];
    const Geom& g = a.g;
    const int b = blockIdx.y, k = blockIdx.x;
    const int tid = threadIdx.x;
    const int step = a.step;
    const int slot = step & 1;
    const bool obs = a.obs != 0;
    const bool apf = a.filter == PF_FILTER_APF;
    const bool resample = a.stat[b].resample != 0;
    const bool multinomial = a.resampler == PF_RESAMPLE_MULTINOMIAL;
    const bool windowed = resample;  // both resamplers search an LDS window of the cdf (their positions are sorted)
    const bool pre_next = a.obs_next && apf;
    const int N = (int)g.N;
    PF_STAMP(a, 8);
    if (a.debug_cut == 1) return;

    // uniform loads first: the window start and the column's parameter rows
    int j0 = (windowed && !multinomial) ? a.j0[(int64_t)b * g.tiles + k] : 0;
    const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
    ColParams<T, D> cp;
    ColConsts<T, D> cc;
    load_col_params<T, D>(a, b, step, obs, cp);
    if (pre_next) cp.load_next(a.y + ((int64_t)(step + 1) * a.y_rows + (a.y_rows == 1 ? 0 : b)) * a.md.obs_dim);
    const T ub = (!windowed || multinomial) ? T(0)
                           : (a.u_tape ? a.u_tape[(int64_t)step * g.B + b]
                                       : uniform_draw<T>(seed, PF_STREAM_UNIFORM, (uint32_t)step, (uint64_t)b));

    const T* x_in = a.x[slot];
    T* x_out = a.x[slot ^ 1];
    const T* lw_in = a.logw[slot] + (int64_t)b * g.N;
    T* lw_out = a.logw[slot ^ 1] + (int64_t)b * g.N;
    const T* cdf_col = a.cdf + (int64_t)b * g.N;
    int32_t* anc_col = a.anc + (int64_t)b * g.N;
    const T* z_step = a.z_tape ? a.z_tape + (int64_t)step * D * g.B * g.N : nullptr;
    const int64_t base = (int64_t)k * g.tile_elems;
    const T nT = T(N);

    bool poison = false;
    PartialAcc<T, D> acc;
    acc.init();
    PF_STAMP(a, 9);
    const T* pos_col = multinomial ? a.pos + (int64_t)b * g.N : nullptr;
    if (windowed && multinomial) {
        // no position-tile table for the sorted uniforms: one wave finds the window start with a 64-ary search
        if (tid < PF_WAVE) {
            const int q = wave_lower_bound<T>(cdf_col, N, pos_col[base], tid & 63);
            if ((tid & 63) == 0) sh_j0 = q;
        }
        __syncthreads();
        j0 = sh_j0;
    }

    for (int r = 0; r < g.rounds_per_tile; ++r) {
        const int64_t r0 = base + (int64_t)r * g.round_elems;
        if (r0 >= g.N) break;
        const int64_t i0 = r0 + tid * VEC;
        const bool on = i0 < g.N;
        T pv[VEC];
        if (multinomial && windowed && on) {
            if (VEC == 1) pv[0] = pos_col[i0]; else load_vec<T, VEC>(pos_col + i0, pv);
        }

        // ---- 1. issue the window loads (cdf, and the particles behind it) ------------------------------------------
        const int ws = j0 - (j0 % VEC);
        T c0[VEC], c1[VEC], xa[D][VEC], xb[D][VEC];
        const int ja = ws + tid * VEC, jb = ws + (PF_BLOCK + tid) * VEC;
        const bool ina = windowed && ja < N, inb = windowed && jb < N;
        if (ina) {
            if (VEC == 1) c0[0] = cdf_col[ja]; else load_vec<T, VEC>(cdf_col + ja, c0);
            if (XWIN) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const T* xc = x_in + ((int64_t)d * g.B + b) * g.N + ja;
                    if (VEC == 1) xa[d][0] = xc[0]; else load_vec<T, VEC>(xc, xa[d]);
                }
            }
        }
        if (inb) {
            if (VEC == 1) c1[0] = cdf_col[jb]; else load_vec<T, VEC>(cdf_col + jb, c1);
            if (XWIN) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const T* xc = x_in + ((int64_t)d * g.B + b) * g.N + jb;
                    if (VEC == 1) xb[d][0] = xc[0]; else load_vec<T, VEC>(xc, xb[d]);
                }
            }
        }
        T lw_old[VEC];
        if (on && !resample) {
            if (VEC == 1) lw_old[0] = lw_in[i0]; else load_vec<T, VEC>(lw_in + i0, lw_old);
        }

        // ---- 2. work that does not need the window, while those loads are in flight --------------------------------
        if (r == 0) cc.prepare(a.md, cp);
        T zt[VEC][D];
        if (on) {
            if (z_step) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    T zr[VEC];
                    const T* zc = z_step + ((int64_t)d * g.B + b) * g.N + i0;
                    if (VEC == 1) zr[0] = zc[0]; else load_vec<T, VEC>(zc, zr);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) zt[j][d] = zr[j];
                }
            } else {
                draw_normals<T, D, VEC>(seed, PF_STREAM_NORMAL, (uint32_t)step, (uint64_t)((int64_t)b * g.N + i0), zt);
            }
        }
        PF_STAMP(a, 10);

        // ---- 3. ancestors ---------------------------------------------------------------------------------------------
        int idx[VEC];
        if (windowed) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (!ina) c0[j] = Lim<T>::inf();
                if (!inb) c1[j] = Lim<T>::inf();
            }
            if (VEC == 1) { win[tid] = c0[0]; win[PF_BLOCK + tid] = c1[0]; }
            else { store_vec<T, VEC>(win + tid * VEC, c0); store_vec<T, VEC>(win + (PF_BLOCK + tid) * VEC, c1); }
            if (XWIN) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if (ina) { if (VEC == 1) xwin[d * WIN + tid] = xa[d][0]; else store_vec<T, VEC>(xwin + d * WIN + tid * VEC, xa[d]); }
                    if (inb) { if (VEC == 1) xwin[d * WIN + PF_BLOCK + tid] = xb[d][0]; else store_vec<T, VEC>(xwin + d * WIN + (PF_BLOCK + tid) * VEC, xb[d]); }
                }
            }
            __syncthreads();
            // on average one ancestor per position: thread t's first position lands near offset (j0 - ws) + t * VEC
            int guess = (j0 - ws) + tid * VEC;
            if (guess > WIN - 1) guess = WIN - 1;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int64_t i = i0 + j;
                int res = N - 1;
                if (i < N) {
                    const T p = multinomial ? pv[j] : grid_position<T>(i, ub, nT);
                    const int q = window_lower_bound<T, WIN>(win, guess, p);
                    guess = q < WIN ? q : WIN - 1;
                    res = (q < WIN) ? ws + q : thread_lower_bound<T>(cdf_col, ws + WIN < N ? ws + WIN : N, N, p);
                    if (res > N - 1) res = N - 1;
                }
                idx[j] = res;
            }
            if (r + 1 < g.rounds_per_tile) {  // the next round's window starts at this round's last ancestor
                if (tid == PF_BLOCK - 1) sh_j0 = idx[VEC - 1];
                __syncthreads();
                j0 = sh_j0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) idx[j] = (int)((i0 + j < g.N) ? (i0 + j) : (g.N - 1));
        }
        PF_STAMP(a, 11);
        if (on && a.debug_cut == 2) {
            if (VEC == 1) anc_col[i0] = idx[0]; else store_vec<int, VEC>(anc_col + i0, idx);
        }

        // ---- 4. gather, propagate, weight -------------------------------------------------------------------------------
        T xo[D][VEC], lwo[VEC], pre_n[VEC];
        if (on && a.debug_cut != 2) {
            T xr[VEC][D];
            if (!resample) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    T xv[VEC];
                    const T* xc = x_in + ((int64_t)d * g.B + b) * g.N + i0;
                    if (VEC == 1) xv[0] = xc[0]; else load_vec<T, VEC>(xc, xv);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) xr[j][d] = xv[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int q = idx[j] - ws;
                    const bool in_lds = XWIN && windowed && q >= 0 && q < WIN;
#pragma unroll
                    for (int d = 0; d < D; ++d)
                        xr[j][d] = in_lds ? xwin[d * WIN + q] : x_in[((int64_t)d * g.B + b) * g.N + idx[j]];
                }
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                T xn[D];
                T w_new;
                if (obs) {
                    const T wi = sample_and_weight<T, D>(a.md, a.proposal, cp, cc, xr[j], zt[j], xn);
                    if (apf) {
                        // second-stage weight ws - pre_weight(x[anc]) (apf.py:43), the pre-weight recomputed in registers
                        w_new = wi - pre_weight<T, D>(a.md, a.proposal, cp, cc, xr[j]);
                        if (w_new != w_new || w_new == Lim<T>::inf()) poison = true;
                    } else {
                        if (wi != wi || wi == Lim<T>::inf()) poison = true;
                        w_new = resample ? wi : (wi + lw_old[j]);
                    }
                } else {
                    // propagate only (NaN observation / unobserved sub-step): weights carried, ll = 0 (state.py:38-42)
                    sample_and_weight<T, D>(a.md, PF_PROP_BOOTSTRAP, cp, cc, xr[j], zt[j], xn);
                    w_new = resample ? T(0) : lw_old[j];
                }
                lwo[j] = sanitize_logw(w_new);
#pragma unroll
                for (int d = 0; d < D; ++d) xo[d][j] = xn[d];
                // first-stage weight of the next step, while the new particle is still in registers
                pre_n[j] = pre_next ? pre_weight<T, D>(a.md, a.proposal, cp, cc, xn, true) : T(0);
            }
            PF_STAMP(a, 12);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                T* xc = x_out + ((int64_t)d * g.B + b) * g.N + i0;
                if (VEC == 1) xc[0] = xo[d][0]; else store_vec<T, VEC>(xc, xo[d]);
            }
            if (VEC == 1) lw_out[i0] = lwo[0]; else store_vec<T, VEC>(lw_out + i0, lwo);
            if (resample || apf) {  // SISR without resampling keeps the previous ancestors (sisr.py:25-26)
                if (VEC == 1) anc_col[i0] = idx[0]; else store_vec<int, VEC>(anc_col + i0, idx);
            }
            acc.template push_round<VEC>(lwo, xo, pre_next, pre_n);
            if (multinomial) {
                T ev[VEC];
                draw_exponentials<T, VEC>(seed, PF_STREAM_MULTINOMIAL, (uint32_t)(step + 1), (uint64_t)((int64_t)b * g.N + i0), ev);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc.es += (double)ev[j];
            }
            PF_STAMP(a, 13);
        }
        if (windowed && r + 1 < g.rounds_per_tile) __syncthreads();  // the window is rewritten by the next round
    }
    if (poison) atomicOr(&a.poison[(step & 1) * g.B + b], 1);
    if (a.debug_cut == 3) return;
    PF_STAMP(a, 14);
    acc.finish(a.part, b, k, g.B, g.tiles, pre_next, red, redm, &a.poison[((step + 1) & 1) * g.B + b]);
    PF_STAMP(a, 15);
}

}  // namespace pf
