// pf_column.hpp - the column-persistent time loop: ONE workgroup per filter runs ALL time steps of a run in ONE launch.
//
// The per-step route (pf_fused.hpp) splits a column into tiles and uses the kernel boundary as its grid-wide
// synchronisation: right for 10^5 .. 10^6 particles per filter, but a filter of a few hundred particles - the reference's
// own operating point: 1 000 theta-particles x 250-400 state particles (examples/stochastic-volatility.ipynb:157,
// tests/inference/test_sequential.py:10-13) - is ONE partly filled tile, and a step then costs a whole launch (17-19 us at
// 1024 x 256..512, profiles/r03_small_n_per_step_route.txt) for ~1 us of work.  Here a filter of N <= 4096 particles
// lives in the registers of one workgroup (VEC particles per thread, N / VEC threads) for the whole run:
//
//   per step   [APF: first-stage weights]  ->  max / sum of the resampling weights (wave DPP + one LDS exchange)
//              ->  cdf = fp64 inclusive scan, rounded once, last value 1 (resampling.py:44-49) -> LDS
//              ->  ancestors: branch-free lower_bound of the VEC grid positions in the LDS cdf (searchsorted, side=left)
//              ->  gather x[anc] from LDS, propagate (Philox / tape), weigh (the per-particle model code of the step kernel)
//              ->  (max, sum e, sum e^2, pivoted moments) of the new weights: moments row, log-likelihood increment
//   no kernel boundary, no per-tile partial tables, no window probing, no HBM traffic but the T result rows; the state is
//   read once and written once per run.  Random numbers are keyed exactly as in the step kernel (seed, stream, step,
//   b N + i), so the two routes consume the SAME draws.
//
// Mirrors the same reference code as the step kernel: sisr.py:14-56, apf.py:16-46, particle/utils.py:7-65,
// resampling.py:24-52, filters/base.py:188-221 (NaN observation -> propagate only).
#pragma once

namespace pf {

#define PFC_MAXW 16       // waves per workgroup (1024 threads)
#define PFC_OBS_WORDS 64  // observed flags of a launch as kernel arguments: 2048 steps per launch (longer runs: several)

struct ColumnRun {
    int t0, n_steps;
    int use_bits;                      // 1: obs_bits (the host's flags, baked into the launch); 0: FusedArgs::obs_dev[t]
    int inline_y;                      // (one-step runs on a shared observation row, use_bits == 0) the flag is read off
                                       // y[t0] itself - "not all-NaN" of its <= 3 values - instead of a byte a launch of its own derived
    uint32_t obs_bits[PFC_OBS_WORDS];  // bit s = step t0 + s weighs against y[t0 + s]
};

// Workgroup barrier for LDS exchanges ONLY: waits for this wave's outstanding LDS operations (lgkmcnt), not for its global
// ones.  __syncthreads() carries a workgroup-scope fence, i.e. an `s_waitcnt vmcnt(0)` - it would make every step wait for
// the result rows thread 0 has just stored (means / variances / log-likelihood: never read in this launch) and for the
// next step's observation that was requested a step ahead precisely so that nobody waits for it.  A single-wave workgroup
// needs no barrier at all (its LDS operations execute in order): a compiler fence.
__device__ __forceinline__ void pfc_barrier(int nw) {
    if (nw > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("" ::: "memory");
}

// exclusive scan of one double per thread across a workgroup of `nw` waves (1 .. 16); a single wave never touches LDS
__device__ __forceinline__ double cb_scan_excl(double v, double* red, int nw, double& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const double incl = wave_scan_incl(v, lane);
    if (nw == 1) {
        total = lane_get(incl, 63);
        return incl - v;
    }
    pfc_barrier(nw);
    if (lane == 63) red[wid] = incl;
    pfc_barrier(nw);
    double off = 0.0, tot = 0.0;
    for (int w = 0; w < nw; ++w) {
        const double s = red[w];
        if (w < wid) off += s;
        tot += s;
    }
    total = tot;
    return off + incl - v;
}

// log of a column sum / reciprocal of one, at the precision the filter's type needs: float filters take the hardware
// log2 / rcp (their results are rounded to float anyway), double filters the exact forms (the parity path)
template <typename T> __device__ __forceinline__ double log_sum(double s) {
    if constexpr (sizeof(T) == 4) return (double)pf_log_g((float)s);
    else return log(s);
}
template <typename T> __device__ __forceinline__ double inv_sum(double s) {
    if constexpr (sizeof(T) == 4) {
        double r = (double)__builtin_amdgcn_rcpf((float)s);
        r = __builtin_fma(__builtin_fma(-s, r, 1.0), r, r);  // one Newton step: ~2^-46
        return r;
    } else {
        return 1.0 / s;
    }
}

// TPB: the launch bound - 256 or 1024 threads per workgroup, each its own instantiation, so that the filters of <= 1024
// particles are not register-limited by the bound the larger ones need (1024 threads: 128 VGPRs and, for D > 1, scratch)
// USER: PF_HID_USER_AFFINE runs (the caller's (loc, scale) planes gathered at the ancestors) are their own instantiation -
// carried as a run-time branch the planes' registers cost every built-in model occupancy (D = 3: 220 -> 256 VGPRs + AGPR
// spills, one wave per SIMD instead of two: 7.7 -> 12.8 us per step at 1024 x 512, measured)
//
// Cross-wave exchanges carry (max, sums relative to that max) records, one per wave - the (max, +) semiring of the step
// kernel's per-tile partials - so a reduction is ONE LDS exchange (write record, barrier, every thread folds the <= 16
// records) instead of a max exchange followed by a sum exchange; a single-wave workgroup (N <= 64 VEC) exchanges nothing.
// Barriers per step: scan records, cdf + particle planes, the new state's records (+ none for SISR steps that keep
// their weights).
// KIND / FILT / PROP: the run's hidden-process kind (scalar state: PF_HID_LINEAR / _SINE_EM / _OU with a linear-Gaussian
// observation - the closed forms - or _VERHULST_EM with the stochastic-volatility observation), filter and proposal as compile-time constants, -1 = run-time values.  Specialised instantiations
// (float, <= 256 threads, Philox normals; Lorenz-63: four particles per lane) drop the model-kind switches, the generic
// (non-closed-form) arithmetic and the other filter's / proposal's paths from the loop: 17 - 19 % per step
// (profiles/r03_column_specialisation_bound.txt); everything else takes the run-time kernel.
// RAGGED: columns of N % VEC != 0 particles (scalar states, four particles per lane) - their own instantiations: carried as a
// run-time branch the ragged paths cost the aligned shapes 3 - 6 % per step (same-box A/B, profiles/r03_column_ragged.txt).
template <typename T, int D, int VEC, int TPB, bool USER, int KIND = -1, int FILT = -1, int PROP = -1, bool RAGGED = false>
__global__ __launch_bounds__(TPB) void k_fused_column(FusedArgs<T> a, ColumnRun run) {
    static_assert(!RAGGED || VEC > 1, "ragged columns: several particles per lane");
    extern __shared__ __attribute__((aligned(16))) unsigned char pfc_lds[];
    const Geom& g = a.g;
    const int N = (int)g.N;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int nw = (int)(blockDim.x >> 6);
    int np2 = 64;
    while (np2 < N) np2 <<= 1;
    // LDS carve-up: cdf (np2 Ts, +inf beyond N) | x planes (D x N Ts) | wave records
    constexpr int KB = 4 + 2 * D;  // the state's record: max, sum e, sum e^2, poison, sum e (x - c)[D], sum e (x - c)^2[D]
    T* const cdfs = reinterpret_cast<T*>(pfc_lds);
    T* const xs = cdfs + np2 + PF_LB_PAD;  // (PF_LB_PAD readable +inf entries behind the cdf: sorted_lower_bound's last read)
    const int NP = ((N + VEC - 1) / VEC) * VEC;  // stride of a particle plane in LDS: N rounded up to the lanes' VEC particles
    double* const recA = reinterpret_cast<double*>(pfc_lds + (((size_t)(np2 + PF_LB_PAD + (size_t)D * NP) * sizeof(T) + 15) & ~(size_t)15));
    double* const recB = recA + 2 * PFC_MAXW;  // [2][PFC_MAXW][KB]: double buffered by step parity

    const bool apf = FILT >= 0 ? (FILT == PF_FILTER_APF) : (a.filter == PF_FILTER_APF);
    const bool multinomial = a.resampler == PF_RESAMPLE_MULTINOMIAL;
    const int proposal = PROP >= 0 ? PROP : a.proposal;
    ModelDesc md = a.md;
    if constexpr (USER) md.hid_kind = PF_HID_USER_AFFINE;
    const int O = md.obs_dim;
    constexpr bool user = USER;
    const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
    const int i0 = tid * VEC;
    const bool on = i0 < N;
    // N % VEC != 0: the last lane holds fewer than VEC particles and a column starts at no vector boundary of the (B, N)
    // planes - the state is then loaded / stored element by element and the per-particle validity `ok[j]` stands in for
    // `on` (an invalid slot carries log-weight -inf, i.e. weight 0, through every sum and scan)
    // (as a RUN-TIME branch these paths had cost the D > 1 kernels registers - D = 2: 164 -> 241 VGPRs; as instantiations of
    // their own - round 4: for every state dimension - they cost the aligned kernels nothing)
    constexpr bool RAG = RAGGED;
    constexpr bool ragged = RAGGED;  // (the host launches the RAGGED instantiations for exactly the N % VEC != 0 columns)
    bool ok[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) ok[j] = RAG ? (i0 + j < N) : on;
    const T nT = T(N);
    const T rcN = T(1) / nT;
    const bool pow2 = (N & (N - 1)) == 0;

    // ---- the incoming state -> registers ---------------------------------------------------------------------------------
    const int slot_in = run.t0 & 1;
    T x[D][VEC], lw[VEC];
    int anc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        lw[j] = -Lim<T>::inf();
        anc[j] = i0 + j;
#pragma unroll
        for (int d = 0; d < D; ++d) x[d][j] = T(0);
    }
    if (on && !ragged) {
        const T* lwc = a.logw[slot_in] + (int64_t)b * N + i0;
        if (VEC == 1) lw[0] = lwc[0]; else load_vec<T, VEC>(lwc, lw);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const T* xc = a.x[slot_in] + ((int64_t)d * g.B + b) * N + i0;
            if (VEC == 1) x[d][0] = xc[0]; else load_vec<T, VEC>(xc, x[d]);
        }
        const int32_t* ac = a.anc + (int64_t)b * N + i0;
        if (VEC == 1) anc[0] = ac[0]; else load_vec<int, VEC>(ac, anc);
    } else if (on) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (ok[j]) {
                lw[j] = a.logw[slot_in][(int64_t)b * N + i0 + j];
                anc[j] = a.anc[(int64_t)b * N + i0 + j];
#pragma unroll
                for (int d = 0; d < D; ++d) x[d][j] = a.x[slot_in][((int64_t)d * g.B + b) * N + i0 + j];
            }
        }
    }
    for (int q = N + tid; q < np2 + PF_LB_PAD; q += blockDim.x) cdfs[q] = Lim<T>::inf();  // (never overwritten)

    // the column's parameters and everything derived from them alone: once per run
    ColParams<T, D> cp;
    ColConsts<T, D> cc;
    load_col_params<T, D>(a, b, run.t0, false, cp);
    if constexpr (KIND >= 0) {  // (the host selects these instantiations for exactly such runs)
        static_assert(!USER && (D == 1) == (KIND != PF_HID_LORENZ63_EM), "specialised column kernels: built-in models");
        md.hid_kind = KIND;
        md.obs_kind = (KIND == PF_HID_VERHULST_EM) ? PF_OBS_SV : PF_OBS_LINEAR;  // (Verhulst: the stochastic-volatility built-in)
        if constexpr (D == 1) md.obs_dim = 1;
    }
    const T* const z_tape = (KIND >= 0) ? nullptr : a.z_tape;  // (specialised runs draw their normals: Philox)
    cc.prepare(md, cp);
    if constexpr (USER) {  // one transition scale per column: the scalar closed forms around the caller's mean (ColConsts::prepare_user)
        if (a.user_scale_percol) cc.prepare_user(md, cp, a.user_scale[b]);
    }
    if constexpr (KIND >= 0) __builtin_assume(cc.fast == (D == 1 && KIND != PF_HID_VERHULST_EM));
    auto y_row = [&](int t) { return a.y + ((int64_t)t * a.y_rows + (a.y_rows == 1 ? 0 : b)) * O; };

    // pivot of the weighted moments: the column's first particle, then (about) the previous state's mean
    T piv[D];
    {
        if (tid == 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) xs[d] = x[d][0];
        }
        pfc_barrier(nw);
#pragma unroll
        for (int d = 0; d < D; ++d) piv[d] = xs[d];
        pfc_barrier(nw);
    }

    // The state's weight family: wave maximum mw1, e1 = exp(lw - mw1), and the column's (M1, S1, Q1, moments) folded from
    // the waves' records.  Within a wave the sums run in T (float: 4 DPP adds per quantity), above it in fp64.
    double M1 = 0.0, S1 = 1.0, Q1 = 1.0;
    T mw1 = T(0);
    T e1[VEC];
    int parity = 0;
    auto reduce_state = [&](bool poison, double (&mom)[2 * D]) -> bool {
        T m = lw[0];
#pragma unroll
        for (int j = 1; j < VEC; ++j) m = lw[j] > m ? lw[j] : m;
        mw1 = wave_max<T>(m);
        T v[2 + 2 * D];
#pragma unroll
        for (int k = 0; k < 2 + 2 * D; ++k) v[k] = T(0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            T ej = (lw[j] == -Lim<T>::inf()) ? T(0) : pf_exp_w(lw[j] - mw1);
            if (lw[j] != lw[j]) ej = lw[j];
            e1[j] = ej;
            v[0] += ej;
            if constexpr (FILT != PF_FILTER_APF) v[1] += ej * ej;  // (an APF never looks at the weights' ESS: FILT known -> not formed)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const T xd = x[d][j] - piv[d];
                v[2 + d] += ej * xd;
                v[2 + D + d] += ej * xd * xd;
            }
        }
#pragma unroll
        for (int k = 0; k < 2 + 2 * D; ++k)
            if (FILT != PF_FILTER_APF || k != 1) v[k] = wave_sum<T>(v[k]);
        bool any = __ballot(poison) != 0ull;
        if (nw == 1) {
            M1 = (double)mw1;
            S1 = (double)v[0];
            Q1 = (double)v[1];
#pragma unroll
            for (int k = 0; k < 2 * D; ++k) mom[k] = (double)v[2 + k];
            pfc_barrier(nw);  // (one wave: orders this step's LDS reads before the next step's writes)
            return any;
        }
        double* rec = recB + (size_t)parity * PFC_MAXW * KB;
        parity ^= 1;
        if (lane == 0) {
            double* r = rec + wid * KB;
            r[0] = (double)mw1;
            r[1] = (double)v[0];
            r[2] = (double)v[1];
            r[3] = any ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < 2 * D; ++k) r[4 + k] = (double)v[2 + k];
        }
        pfc_barrier(nw);
        double M = rec[0];
        for (int w = 1; w < nw; ++w) M = rec[w * KB] > M ? rec[w * KB] : M;
        double s = 0.0, q = 0.0, pz = 0.0;
#pragma unroll
        for (int k = 0; k < 2 * D; ++k) mom[k] = 0.0;
        for (int w = 0; w < nw; ++w) {
            const double* r = rec + w * KB;
            const double f = exp_diff_t<T>(r[0], M);
            s += r[1] * f;
            q += r[2] * f * f;
            pz += r[3];
#pragma unroll
            for (int k = 0; k < 2 * D; ++k) mom[k] += r[4 + k] * f;
        }
        M1 = M;
        S1 = s;
        Q1 = q;
        return pz != 0.0;
    };
    // moments row `row` of filter_means / filter_variance from the sums above (thread 0); every thread moves the pivot
    auto write_moments = [&](int row, const double (&mom)[2 * D]) {
        const double inv = inv_sum<T>(S1);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const double dm = mom[d] * inv;
            if (tid == 0) {
                double var = mom[D + d] * inv - dm * dm;
                if (var < 0.0) var = 0.0;
                a.means[((int64_t)row * g.B + b) * D + d] = (T)((double)piv[d] + dm);
                a.vars[((int64_t)row * g.B + b) * D + d] = (T)var;
            }
            piv[d] = (T)((double)piv[d] + dm);
        }
    };

    {
        double mom[2 * D];
        reduce_state(false, mom);
        write_moments(run.t0, mom);
    }
    T ll_tot = (tid == 0) ? a.ll_total[b] : T(0);
    double lse_w = M1 + log_sum<T>(S1);
    if (run.t0 > 0 && tid == 0) {
        // A run issued in pieces: the piece before this one may have been a per-step piece WITHOUT finalize, whose last
        // move's log-likelihood increment is still pending - the per-step route flushes it one launch later, from the
        // column record it keeps in the workspace (column_bookkeeping).  That launch is this one: ll = lse(logw) - base,
        // NaN when the move's weights were poisoned, 0 for an unweighted move.  (A finalised predecessor - either route -
        // left ll_done = 1.)
        const ColStat st = a.stat[b];
        if (!st.ll_done) {
            const int pslot = (run.t0 - 1) & 3;
            double ll = 0.0;
            if (st.prev_observed) {
                ll = lse_w - st.base_lse;
                if (a.poison[pslot * g.B + b]) ll = __builtin_nan("");
            }
            a.ll_steps[(int64_t)(run.t0 - 1) * g.B + b] = (T)ll;
            ll_tot = (T)((double)ll_tot + ll);
        }
    }

    // the first step's observation and offset; every later step's are requested one step ahead
    // Requested a step ahead and consumed RAW one iteration later (nothing in the requesting iteration touches the values,
    // so no wait for them is placed there): the next step's observation row, systematic offset and observed flag - from
    // clamped, always valid addresses; whether they mean anything is decided when they are used.
    T y_nx[ColParams<T, D>::MAXO], u_nx = T(0);
    unsigned char flag_nx = 0;
    auto request_inputs = [&](int s) {
        const int sc = s < run.n_steps ? s : run.n_steps - 1;
        const int t = run.t0 + sc;
        const T* yr = y_row(t);
#pragma unroll
        for (int o = 0; o < ColParams<T, D>::MAXO; ++o) y_nx[o] = yr[o < O ? o : 0];
        if (a.u_tape) u_nx = a.u_tape[(int64_t)t * g.B + b];
        if (run.inline_y) {  // (one-step run on a shared row: "not all-NaN" read off the row itself)
            bool any = false;
#pragma unroll
            for (int o = 0; o < ColParams<T, D>::MAXO; ++o) any = any || (o < O && !(y_nx[o] != y_nx[o]));
            flag_nx = any ? 1 : 0;
        } else if (!run.use_bits) {
            flag_nx = a.obs_dev[t];
        }
    };
    request_inputs(0);
    uint32_t bits = 0u;

    for (int s = 0; s < run.n_steps; ++s) {
        const int t = run.t0 + s;
        if ((s & 31) == 0) bits = run.obs_bits[s >> 5];  // (one scalar load per 32 steps)
        const bool obs = run.use_bits ? ((bits >> (s & 31)) & 1u) != 0 : flag_nx != 0;
        const bool two = apf && obs;
#pragma unroll
        for (int o = 0; o < ColParams<T, D>::MAXO; ++o) cp.y[o] = (obs && o < O) ? y_nx[o] : T(0);
        cc.set_obs(cp);
        const T u_tape = u_nx;
        request_inputs(s + 1);
        bool poison = false;

        // ---- resampling weights, the decision -----------------------------------------------------------------------------------
        T rw[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) rw[j] = lw[j];
        if (two) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                T xj[D];
#pragma unroll
                for (int d = 0; d < D; ++d) xj[d] = x[d][j];
                UserMS<T, D> um = UserMS<T, D>::none();
                if constexpr (user) {
                    if (ok[j]) {
                        um.gather(a.user_loc, a.user_scale, (int64_t)b * N, (int64_t)g.B * N, i0 + j, a.user_scale_percol != 0, b, g.B);
                        um.euler(xj, a.user_dt);
                    }
                }
                const T pre = pre_weight<T, D>(md, proposal, cp, cc, xj, false, um);
                if (ok[j] && is_nan_or_posinf(pre)) poison = true;
                rw[j] = ok[j] ? sanitize_logw(pre + lw[j]) : -Lim<T>::inf();
            }
        }
        const bool resample = apf ? obs : (S1 * S1 / Q1 < a.thr_abs);  // apf.py:29-31 | sisr.py:18-19
        double base_lse = lse_w;
        int idx[VEC];
        T xr[VEC][D];
        if (resample) {
            // ---- cdf of the normalised resampling weights: wave-local fp64 scans rounded once per element (the step kernel's
            // chunk scans), made column-level by the waves' (max, total) records; last value 1 (resampling.py:44-49) -----------
            T er[VEC], mw = mw1;
            if (two) {
                T m = rw[0];
#pragma unroll
                for (int j = 1; j < VEC; ++j) m = rw[j] > m ? rw[j] : m;
                mw = wave_max<T>(m);
#pragma unroll
                for (int j = 0; j < VEC; ++j) er[j] = (rw[j] == -Lim<T>::inf()) ? T(0) : pf_exp_w(rw[j] - mw);
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) er[j] = e1[j];  // the weights' own family: exp(lw - mw1)
            }
            double incl[VEC], local = 0.0;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                local += ok[j] ? (double)er[j] : 0.0;
                incl[j] = local;
            }
            const double iw = wave_scan_incl(local, lane);
            const double excl = iw - local;
            const double tw = lane_get(iw, 63);
            double Mr = (double)mw, C = 0.0, gq = 1.0, S = tw;
            if (nw > 1) {
                if (lane == 63) {
                    recA[2 * wid] = (double)mw;
                    recA[2 * wid + 1] = tw;
                }
                pfc_barrier(nw);
                Mr = recA[0];
                for (int w = 1; w < nw; ++w) Mr = recA[2 * w] > Mr ? recA[2 * w] : Mr;
                S = 0.0;
                for (int w = 0; w < nw; ++w) {
                    const double f = exp_diff_t<T>(recA[2 * w], Mr);
                    const double v = recA[2 * w + 1] * f;
                    if (w < wid) C += v;
                    if (w == wid) gq = f;
                    S += v;
                }
            }
            const double inv_tot = inv_sum<T>(S);
            if (two) base_lse = a.logN - ((Mr + log_sum<T>(S)) - lse_w);  // apf.py:44
            else base_lse = a.logN;                                        // W = 1 / N after resampling
            if (on) {
                T cv[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const T L = (T)(excl + incl[j]);
                    double c = inv_tot * (C + gq * (double)L);
                    if (c > 1.0) c = 1.0;
                    cv[j] = (i0 + j == N - 1) ? T(1) : (T)c;  // resampling.py:49
                    if (!ok[j]) cv[j] = Lim<T>::inf();         // (slots beyond the column: the search's sentinels)
                }
                if (VEC == 1) cdfs[i0] = cv[0]; else store_vec<T, VEC>(cdfs + i0, cv);
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if (VEC == 1) xs[d * NP + i0] = x[d][0]; else store_vec<T, VEC>(xs + d * NP + i0, x[d]);
                }
            }
            // ---- positions: the systematic grid, or the order statistics of N uniforms (Exp(1) spacings) -------------------
            T pp[VEC];
            if (multinomial) {
                T ev[VEC], tail[1];
#pragma unroll
                for (int j = 0; j < VEC; ++j) ev[j] = T(0);
                if (on && !ragged) {
                    draw_exponentials<T, VEC>(seed, PF_STREAM_MULTINOMIAL, (uint32_t)t, (uint64_t)((int64_t)b * N + i0), ev);
                } else if (on) {  // (no vector boundary: the draws of the elements one by one - the same numbers)
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        T e1v[1] = {T(0)};
                        if (ok[j]) draw_exponentials<T, 1>(seed, PF_STREAM_MULTINOMIAL, (uint32_t)t, (uint64_t)((int64_t)b * N + i0 + j), e1v);
                        ev[j] = e1v[0];
                    }
                }
                draw_exponentials<T, 1>(seed, PF_STREAM_MULTINOMIAL, (uint32_t)t, (uint64_t)((int64_t)g.B * N + b), tail);
                double ei[VEC], el = 0.0, etot;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    el += ok[j] ? (double)ev[j] : 0.0;
                    ei[j] = el;
                }
                const double eexcl = cb_scan_excl(el, recA, nw, etot);  // (its own barriers; recA is free again by then)
                const double invE = 1.0 / (etot + (double)tail[0]);
#pragma unroll
                for (int j = 0; j < VEC; ++j) pp[j] = (T)((eexcl + ei[j]) * invE);
            } else {
                const T u = a.u_tape ? u_tape : uniform_draw<T>(seed, PF_STREAM_UNIFORM, (uint32_t)t, (uint64_t)b);
#pragma unroll
                for (int j = 0; j < VEC; ++j)  // (a power-of-two N: the division is an exact multiplication)
                    pp[j] = pow2 ? (T(i0 + j) + u) * rcN : grid_position<T>(i0 + j, u, nT);
            }
            pfc_barrier(nw);  // cdf and x planes are in LDS
            // ---- ancestors: first q with cdf[q] >= p (searchsorted side = left), all VEC probes of a round in flight -------
            // (measured against the step kernel's inverted grid - closed-form offspring ranges, head scatter, max-scan: three
            // LDS round trips instead of log2 N + 2, but one more barrier and the range arithmetic - on one box: within
            // -4 .. +4 %, slower at 5 of 9 shapes, profiles/r03_column_search_vs_inverted_grid.txt; the search stays)
            int q[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) q[j] = 0;
            // (the rounds unrolled with the positions as BYTE offsets - 16 VALU per round of four positions against 21 for a run-time
            // loop over element indices: sorted_lower_bound, pf_device.hpp)
            sorted_lower_bound<T, VEC, 4096>(cdfs, np2, pp, q);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                idx[j] = q[j] > N - 1 ? N - 1 : q[j];
#pragma unroll
                for (int d = 0; d < D; ++d) xr[j][d] = xs[d * NP + idx[j]];
            }
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                idx[j] = (i0 + j < N) ? i0 + j : N - 1;
#pragma unroll
                for (int d = 0; d < D; ++d) xr[j][d] = x[d][j];
            }
        }

        // ---- draws, propagate, weigh -----------------------------------------------------------------------------------------
        T z[VEC][D];
#pragma unroll
        for (int j = 0; j < VEC; ++j)
#pragma unroll
            for (int d = 0; d < D; ++d) z[j][d] = T(0);
        if (on) {
            if (z_tape) {
                const T* zs = z_tape + (int64_t)t * D * g.B * N;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    T zr[VEC];
                    const T* zc = zs + ((int64_t)d * g.B + b) * N + i0;
                    if (VEC == 1) zr[0] = zc[0];
                    else if (!ragged) load_vec<T, VEC>(zc, zr);
                    else {
#pragma unroll
                        for (int j = 0; j < VEC; ++j) zr[j] = ok[j] ? zc[j] : T(0);
                    }
#pragma unroll
                    for (int j = 0; j < VEC; ++j) z[j][d] = zr[j];
                }
            } else {
                if (!ragged) draw_normals<T, D, VEC>(seed, PF_STREAM_NORMAL, (uint32_t)t, (uint64_t)((int64_t)b * N + i0), z);
                else draw_normals_ragged<T, D, VEC>(seed, PF_STREAM_NORMAL, (uint32_t)t, (uint64_t)((int64_t)b * N + i0), z);
            }
        }
        T lw_new[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            T xn[D], w_new;
            UserMS<T, D> um = UserMS<T, D>::none();
            if constexpr (user) {  // the parent's
                um.gather(a.user_loc, a.user_scale, (int64_t)b * N, (int64_t)g.B * N, idx[j], a.user_scale_percol != 0, b, g.B);
                um.euler(xr[j], a.user_dt);
            }
            if (obs) {
                const T wi = sample_and_weight<T, D>(md, proposal, cp, cc, xr[j], z[j], xn, um);
                if (apf) {
                    w_new = wi - pre_weight<T, D>(md, proposal, cp, cc, xr[j], false, um);  // apf.py:43
                    if (ok[j] && is_nan_or_posinf(w_new)) poison = true;
                } else {
                    if (ok[j] && is_nan_or_posinf(wi)) poison = true;
                    w_new = resample ? wi : (wi + lw[j]);  // sisr.py:52-55
                }
            } else {  // NaN observation: propagate only, weights carried, ll = 0 (particle/state.py:38-42)
                sample_and_weight<T, D>(md, PF_PROP_BOOTSTRAP, cp, cc, xr[j], z[j], xn, um);
                w_new = resample ? T(0) : lw[j];
            }
            lw_new[j] = ok[j] ? sanitize_logw(w_new) : -Lim<T>::inf();
#pragma unroll
            for (int d = 0; d < D; ++d) xr[j][d] = xn[d];
        }
        if (ragged) {  // (slots beyond the column stay at 0: nothing of theirs may ever become inf / NaN and meet a zero weight)
#pragma unroll
            for (int j = 0; j < VEC; ++j)
#pragma unroll
                for (int d = 0; d < D; ++d) xr[j][d] = ok[j] ? xr[j][d] : T(0);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            lw[j] = lw_new[j];
            if (resample || apf) anc[j] = idx[j];  // SISR without resampling keeps its ancestors (sisr.py:25-26)
#pragma unroll
            for (int d = 0; d < D; ++d) x[d][j] = xr[j][d];
        }

        // ---- the new state's sums: moments row t + 1, log-likelihood increment of this step ------------------------------------
        double mom[2 * D];
        const bool any_poison = reduce_state(poison, mom);  // (its barrier also orders this step's LDS reads before the next writes)
        write_moments(t + 1, mom);
        lse_w = M1 + log_sum<T>(S1);
        if (tid == 0) {
            double ll = 0.0;
            if (obs) {
                ll = lse_w - base_lse;
                if (any_poison) ll = __builtin_nan("");
            }
            a.ll_steps[(int64_t)t * g.B + b] = (T)ll;
            ll_tot = (T)((double)ll_tot + ll);
        }
    }

    // ---- the final state -> HBM (the slot the per-step route would have written last) ------------------------------------------
    const int slot_out = (run.t0 + run.n_steps) & 1;
    if (on && ragged) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (ok[j]) {
                a.logw[slot_out][(int64_t)b * N + i0 + j] = lw[j];
                a.anc[(int64_t)b * N + i0 + j] = anc[j];
#pragma unroll
                for (int d = 0; d < D; ++d) a.x[slot_out][((int64_t)d * g.B + b) * N + i0 + j] = x[d][j];
            }
        }
    } else if (on) {
        T* lwc = a.logw[slot_out] + (int64_t)b * N + i0;
        if (VEC == 1) lwc[0] = lw[0]; else store_vec<T, VEC>(lwc, lw);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            T* xc = a.x[slot_out] + ((int64_t)d * g.B + b) * N + i0;
            if (VEC == 1) xc[0] = x[d][0]; else store_vec<T, VEC>(xc, x[d]);
        }
        int32_t* ac = a.anc + (int64_t)b * N + i0;
        if (VEC == 1) ac[0] = anc[0]; else store_vec<int, VEC>(ac, anc);
    }
    if (tid == 0) {
        a.ll_total[b] = ll_tot;
        // the per-step route's per-column record, as a finalised run leaves it (a later call on that route starts from it)
        ColStat st{};
        st.lse_w = lse_w;
        st.ll_done = 1;
        a.stat[b] = st;
#pragma unroll
        for (int q = 0; q < 4; ++q) a.poison[q * g.B + b] = 0;
    }
}

}  // namespace pf
