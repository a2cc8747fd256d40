// pf_fused.hpp - the fused SISR / APF time step: TWO kernels per step.
//
//   k_fused_scan   (grid tiles x B)  re-reduces the column's per-tile partials, tile 0 finalises the bookkeeping of the
//                                    current state (moments row, log-likelihood of the previous step, resampling
//                                    decision); if the column resamples: scans this tile of resampling weights into the
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
        if ((a).debug_cut < 0 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)        \
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
    T* x[2];
    T* logw[2];
    int32_t* anc;
    T* cdf;
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
    bool poison;
    __device__ __forceinline__ void init() {
        a1.init();
        a2.init();
        q1 = 0.0;
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
    // workgroup reduction + store; `red` >= (3 + 2D) * PF_NWAVES doubles, `redm` >= PF_NWAVES Ts
    __device__ __forceinline__ void finish(double* part, int b, int k, int B, int tiles, bool pre_on, double* red, T* redm,
                                           int32_t* poison_slot) {
        const T M1 = block_max<T>(a1.m, redm);
        const double f1 = exp_diff_t<T>((double)a1.m, (double)M1);
        double sums[3 + 2 * D];
        sums[0] = a1.s * f1;
        sums[1] = q1 * f1 * f1;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            sums[3 + d] = mx[d] * f1;
            sums[3 + D + d] = mxx[d] * f1;
        }
        T M2 = -Lim<T>::inf();
        sums[2] = 0.0;
        if (pre_on) {  // uniform
            M2 = block_max<T>(a2.m, redm);
            sums[2] = a2.s * exp_diff_t<T>((double)a2.m, (double)M2);
        }
        block_sum<3 + 2 * D>(sums, red);
        if (poison) atomicOr(poison_slot, 1);
        if (threadIdx.x == 0) {
            const int64_t stride = (int64_t)B * tiles;
            const int64_t o = (int64_t)b * tiles + k;
            part[PQ_M1 * stride + o] = (double)M1;
            part[PQ_S1 * stride + o] = sums[0];
            part[PQ_Q1 * stride + o] = sums[1];
            part[PQ_M2 * stride + o] = (double)M2;
            part[PQ_S2 * stride + o] = sums[2];
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
    __shared__ double red[(3 + 2 * D) * PF_NWAVES];
    __shared__ T redm[PF_NWAVES];
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
    }
    acc.finish(a.part, b, k, g.B, g.tiles, pre_on, red, redm, &a.poison[(a.step & 1) * g.B + b]);
}

// Emits j0[t] = i for every position tile t whose first grid position p_t = (t * tile + u) / N satisfies
// c_prev < p_t <= c  (searchsorted side=left).  The intervals (c_prev, c] of consecutive elements partition (-1, 1], so
// every tile start is claimed by exactly one element.  Candidates come from the real-valued inverse of p_t (cheap early
// out: almost no element contains a tile start), membership from the exact fp test of resampling.py:44-51.
template <typename T>
__device__ __forceinline__ void emit_j0(T c_prev, T c, int64_t i, T u, int64_t N, int tile_elems, int tiles,
                                        int32_t* __restrict__ j0_col) {
    // evaluated in T: the products carry a relative error of eps, i.e. <= eps * N / tile (< 0.07 at N = 2^30, tile >= 1024)
    // in tile units - covered by delta
    const T inv = T(1) / T(tile_elems);
    const T delta = T(0.25);
    const T lo = (c_prev * T(N) - u) * inv - delta;
    const T hi = (c * T(N) - u) * inv + delta;
    const T th = floor(hi);
    if (th < lo) return;  // no integer in [lo, hi]
    int64_t ta = (int64_t)ceil(lo), tb = (int64_t)th;
    if (ta < 0) ta = 0;
    if (tb > tiles - 1) tb = tiles - 1;
    const T nT = T(N);
    for (int64_t t = ta; t <= tb; ++t) {
        const T p = grid_position<T>(t * tile_elems, u, nT);
        if (c_prev < p && p <= c) j0_col[t] = (int32_t)i;
    }
}

template <typename T, int D, int VEC>
__global__ __launch_bounds__(PF_BLOCK) void k_fused_scan(FusedArgs<T> a) {
    __shared__ double red[6 * PF_NWAVES];
    __shared__ double redm[2 * PF_NWAVES];
    __shared__ double red2[2 * D * PF_NWAVES];
    __shared__ T lastv[PF_BLOCK + 1];  // every thread's last cdf value of the round (+ the previous round's last one)
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

    // prefetch round 0 of this tile (log-weights, and the particles the APF's in-register pre-weight needs) so the
    // loads are in flight while the partials are combined
    const T* lw_col = a.logw[slot] + (int64_t)b * g.N;
    const T* x_base = a.x[slot];
    const int64_t base = (int64_t)k * g.tile_elems;
    T lw[VEC], xv[D][VEC];
    {
        const int64_t i0 = base + threadIdx.x * VEC;
        if (!a.finalize_only && i0 < g.N) {
            if (VEC == 1) lw[0] = lw_col[i0]; else load_vec<T, VEC>(lw_col + i0, lw);
            if (two) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const T* xc = x_base + ((int64_t)d * g.B + b) * g.N + i0;
                    if (VEC == 1) xv[d][0] = xc[0]; else load_vec<T, VEC>(xc, xv[d]);
                }
            }
        }
    }

    // ---- one pass over the column's partials: both maxima, then both sums + this tile's prefix ----------------------
    double m1 = -__builtin_huge_val(), m2 = -__builtin_huge_val();
    for (int t = threadIdx.x; t < g.tiles; t += PF_BLOCK) {
        m1 = fmax(m1, a.part[PQ_M1 * stride + cb + t]);
        if (two) m2 = fmax(m2, a.part[PQ_M2 * stride + cb + t]);
    }
    {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        m1 = wave_max(m1);
        m2 = wave_max(m2);
        if (lane == 0) { redm[wid] = m1; redm[PF_NWAVES + wid] = m2; }
        __syncthreads();
        m1 = redm[0];
        m2 = redm[PF_NWAVES];
#pragma unroll
        for (int w = 1; w < PF_NWAVES; ++w) { m1 = fmax(m1, redm[w]); m2 = fmax(m2, redm[PF_NWAVES + w]); }
    }
    // S1, Q1, S2, and the resampling-weight prefix below this tile / below the next tile.  Both prefixes use the same
    // masked loop + reduction tree, so tile k's "next" prefix is bit-identical to what tile k+1 computes as its own.
    double v[6] = {0, 0, 0, 0, 0, 0};  // S1, Q1, S2, prefix(k), prefix(k+1), -
    for (int t = threadIdx.x; t < g.tiles; t += PF_BLOCK) {
        const double f = exp_diff_t<T>(a.part[PQ_M1 * stride + cb + t], m1);
        const double s = a.part[PQ_S1 * stride + cb + t] * f;
        v[0] += s;
        v[1] += a.part[PQ_Q1 * stride + cb + t] * f * f;
        double sr = s;
        if (two) {
            sr = a.part[PQ_S2 * stride + cb + t] * exp_diff_t<T>(a.part[PQ_M2 * stride + cb + t], m2);
            v[2] += sr;
        }
        if (t < k) v[3] += sr;
        if (t < k + 1) v[4] += sr;
    }
    block_sum<6>(v, red);
    PF_STAMP(a, 1);
    const double S1 = v[0], Q1 = v[1];
    const double lse_w = m1 + log(S1);
    const double ess = S1 * S1 / Q1;

    bool resample;
    if (apf) resample = obs;            // APF resamples every weighted step (apf.py:29-31)
    else resample = ess < a.thr_abs;    // SISR: ess < ess_threshold * N (sisr.py:18-19)
    if (a.finalize_only) resample = false;

    if (k == 0) {
        // moments of the current state -> row `step` of filter_means / filter_variance
        double mv[2 * D];
#pragma unroll
        for (int q = 0; q < 2 * D; ++q) {
            mv[q] = 0.0;
            for (int t = threadIdx.x; t < g.tiles; t += PF_BLOCK)
                mv[q] += a.part[(PQ_MX + q) * stride + cb + t] * exp_diff_t<T>(a.part[PQ_M1 * stride + cb + t], m1);
        }
        block_sum<2 * D>(mv, red2);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const double mu = mv[d] / S1;
                double var = mv[D + d] / S1 - mu * mu;
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
                if (apf) {
                    // ll_t = [lse(w') - log N] + [lse(rw) - lse(w)]   (apf.py:44)
                    st.base_lse = a.logN - ((m2 + log(v[2])) - lse_w);
                } else {
                    // ll_t = lse(wi + log W): W = 1/N after resampling, else the carried weights (sisr.py:52-55)
                    st.base_lse = resample ? a.logN : lse_w;
                }
            }
            a.stat[b] = st;
        }
    }
    PF_STAMP(a, 2);
    if (!resample || a.debug_cut == 2) return;

    // ---- scan this tile of resampling weights: SISR scans logw, APF scans rw = sanitize(pre_weight(x, y) + logw) ------
    const double MR = two ? m2 : m1, SR = two ? v[2] : S1;
    const double mk = a.part[(two ? PQ_M2 : PQ_M1) * stride + cb + k];
    const double fk = exp_diff_t<T>(mk, MR) / SR;
    const double Pk = v[3] / SR;
    const double Pnext = v[4] / SR;
    const T tile_max = (T)mk;

    ColParams<T, D> cp;
    ColConsts<T, D> cc;
    cc.fast = false;
    if (two) {
        load_col_params<T, D>(a, b, step, true, cp);
        cc.prepare(a.md, cp);
    }
    const bool sys = a.resampler == PF_RESAMPLE_SYSTEMATIC;
    const T ub = !sys ? T(0)
                      : (a.u_tape ? a.u_tape[(int64_t)step * g.B + b]
                                  : uniform_draw<T>(a.seed, PF_STREAM_UNIFORM, (uint32_t)step, (uint64_t)b));

    T* cdf_col = a.cdf + (int64_t)b * g.N;
    int32_t* j0_col = a.j0 + cb;
    double carry = 0.0;
    PF_STAMP(a, 3);
    const int64_t tile_last = (base + g.tile_elems < g.N ? base + g.tile_elems : g.N) - 1;
    for (int r = 0; r < g.rounds_per_tile; ++r) {
        const int64_t r0 = base + (int64_t)r * g.round_elems;
        if (r0 >= g.N) break;
        const int64_t i0 = r0 + threadIdx.x * VEC;
        const bool on = i0 < g.N;
        if (on && r > 0) {
            if (VEC == 1) lw[0] = lw_col[i0]; else load_vec<T, VEC>(lw_col + i0, lw);
            if (two) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const T* xc = x_base + ((int64_t)d * g.B + b) * g.N + i0;
                    if (VEC == 1) xv[d][0] = xc[0]; else load_vec<T, VEC>(xc, xv[d]);
                }
            }
        }
        double e[VEC], local = 0.0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            double ej = 0.0;
            if (on) {
                T rw = lw[j];
                if (two) {
                    T xj[D];
#pragma unroll
                    for (int d = 0; d < D; ++d) xj[d] = xv[d][j];
                    rw = sanitize_logw(pre_weight<T, D>(a.md, a.proposal, cp, cc, xj) + lw[j]);
                }
                ej = (rw == -Lim<T>::inf()) ? 0.0 : (double)pf_exp_w(rw - tile_max);
            }
            local += ej;
            e[j] = local;
        }
        PF_STAMP(a, 4);
        double total;
        const double excl = block_scan_excl(local, red, total);
        PF_STAMP(a, 5);
        T outv[VEC];
        if (on) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                double cc = Pk + fk * (carry + excl + e[j]);
                if (cc > Pnext) cc = Pnext;
                T c = (T)cc;
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
            if (r == 0 && threadIdx.x == 0) lastv[0] = (k == 0) ? T(-1) : (T)Pk;
            __syncthreads();
            if (on) {
                T c_prev = lastv[threadIdx.x];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    emit_j0<T>(c_prev, outv[j], i0 + j, ub, g.N, g.tile_elems, g.tiles, j0_col);
                    c_prev = outv[j];
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) lastv[0] = lastv[PF_BLOCK];
        }
        PF_STAMP(a, 6);
        carry += total;
    }
}

template <typename T, int D, int VEC>
__global__ __launch_bounds__(PF_BLOCK) void k_fused_step(FusedArgs<T> a) {
    __shared__ __attribute__((aligned(32))) T win[SearchWin<T, VEC>::WIN];
    __shared__ int sh_j0;
    __shared__ double red[(3 + 2 * D) * PF_NWAVES];
    __shared__ T redm[PF_NWAVES];
    const Geom& g = a.g;
    const int b = blockIdx.y, k = blockIdx.x;
    const int step = a.step;
    const int slot = step & 1;
    const bool obs = a.obs != 0;
    const bool apf = a.filter == PF_FILTER_APF;
    const bool resample = a.stat[b].resample != 0;
    const bool multinomial = a.resampler == PF_RESAMPLE_MULTINOMIAL;
    const bool pre_next = a.obs_next && apf;
    const int N = (int)g.N;
    PF_STAMP(a, 8);

    ColParams<T, D> cp, cpn;
    ColConsts<T, D> cc, ccn;
    load_col_params<T, D>(a, b, step, obs, cp);
    cc.prepare(a.md, cp);
    ccn.fast = false;
    if (pre_next) {
        load_col_params<T, D>(a, b, step + 1, true, cpn);
        ccn.prepare(a.md, cpn);
    }

    const T* x_in = a.x[slot];
    T* x_out = a.x[slot ^ 1];
    const T* lw_in = a.logw[slot] + (int64_t)b * g.N;
    T* lw_out = a.logw[slot ^ 1] + (int64_t)b * g.N;
    const T* cdf_col = a.cdf + (int64_t)b * g.N;
    int32_t* anc_col = a.anc + (int64_t)b * g.N;
    const T* z_step = a.z_tape ? a.z_tape + (int64_t)step * D * g.B * g.N : nullptr;

    const int64_t base = (int64_t)k * g.tile_elems;
    T ub = T(0);
    if (a.debug_cut == 1) return;
    PF_STAMP(a, 9);
    if (resample && !multinomial) {
        ub = a.u_tape ? a.u_tape[(int64_t)step * g.B + b] : uniform_draw<T>(a.seed, PF_STREAM_UNIFORM, (uint32_t)step, (uint64_t)b);
        if (threadIdx.x == 0) sh_j0 = a.j0[(int64_t)b * g.tiles + k];
        __syncthreads();
    }
    bool poison = false;
    PartialAcc<T, D> acc;
    acc.init();
    PF_STAMP(a, 10);

    for (int r = 0; r < g.rounds_per_tile; ++r) {
        const int64_t r0 = base + (int64_t)r * g.round_elems;
        if (r0 >= g.N) break;
        const int64_t i0 = r0 + threadIdx.x * VEC;
        const bool on = i0 < g.N;
        int idx[VEC];
        if (resample) {
            if (!multinomial) {
                systematic_round<T, VEC>(cdf_col, N, i0, ub, nullptr, win, &sh_j0, idx);
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    idx[j] = N - 1;
                    if (i0 + j < g.N) {
                        const T p = uniform_draw<T>(a.seed, PF_STREAM_MULTINOMIAL, (uint32_t)step, (uint64_t)((int64_t)b * g.N + i0 + j));
                        const int q = thread_lower_bound<T>(cdf_col, 0, N, p);
                        idx[j] = q > N - 1 ? N - 1 : q;
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) idx[j] = (int)((i0 + j < g.N) ? (i0 + j) : (g.N - 1));
        }
        PF_STAMP(a, 11);
        if (!on) continue;
        if (a.debug_cut == 2) {
            if (VEC == 1) anc_col[i0] = idx[0]; else store_vec<int, VEC>(anc_col + i0, idx);
            continue;
        }

        T lw_old[VEC];
        if (!resample) {
            if (VEC == 1) lw_old[0] = lw_in[i0]; else load_vec<T, VEC>(lw_in + i0, lw_old);
        }
        T xo[D][VEC], lwo[VEC], pre_n[VEC], zt[VEC][D];
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
            draw_normals<T, D, VEC>(a.seed, PF_STREAM_NORMAL, (uint32_t)step, (uint64_t)((int64_t)b * g.N + i0), zt);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            T xr[D], xn[D];
#pragma unroll
            for (int d = 0; d < D; ++d) xr[d] = x_in[((int64_t)d * g.B + b) * g.N + idx[j]];
            T w_new;
            if (obs) {
                const T wi = sample_and_weight<T, D>(a.md, a.proposal, cp, cc, xr, zt[j], xn);
                if (apf) {
                    // second-stage weight ws - pre_weight(x[anc]) (apf.py:43), the pre-weight recomputed in registers
                    w_new = wi - pre_weight<T, D>(a.md, a.proposal, cp, cc, xr);
                    if (w_new != w_new || w_new == Lim<T>::inf()) poison = true;
                } else {
                    if (wi != wi || wi == Lim<T>::inf()) poison = true;
                    w_new = resample ? wi : (wi + lw_old[j]);
                }
            } else {
                // propagate only (NaN observation / unobserved sub-step): weights carried, ll = 0 (state.py:38-42)
                sample_and_weight<T, D>(a.md, PF_PROP_BOOTSTRAP, cp, cc, xr, zt[j], xn);
                w_new = resample ? T(0) : lw_old[j];
            }
            lwo[j] = sanitize_logw(w_new);
#pragma unroll
            for (int d = 0; d < D; ++d) xo[d][j] = xn[d];
            // first-stage weight of the next step, while the new particle is still in registers
            pre_n[j] = pre_next ? pre_weight<T, D>(a.md, a.proposal, cpn, ccn, xn) : T(0);
        }
        PF_STAMP(a, 12);
        acc.template push_round<VEC>(lwo, xo, pre_next, pre_n);
        PF_STAMP(a, 13);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            T* xc = x_out + ((int64_t)d * g.B + b) * g.N + i0;
            if (VEC == 1) xc[0] = xo[d][0]; else store_vec<T, VEC>(xc, xo[d]);
        }
        if (VEC == 1) lw_out[i0] = lwo[0]; else store_vec<T, VEC>(lw_out + i0, lwo);
        if (resample || apf) {  // SISR without resampling keeps the previous ancestors (sisr.py:25-26)
            if (VEC == 1) anc_col[i0] = idx[0]; else store_vec<int, VEC>(anc_col + i0, idx);
        }
    }
    if (poison) atomicOr(&a.poison[(step & 1) * g.B + b], 1);
    if (a.debug_cut == 3) return;
    PF_STAMP(a, 14);
    acc.finish(a.part, b, k, g.B, g.tiles, pre_next, red, redm, &a.poison[((step + 1) & 1) * g.B + b]);
    PF_STAMP(a, 15);
}

}  // namespace pf
