// pf_fused.hpp - the fused SISR / APF time step: ONE kernel launch per step.
//
//   k_fused_step  (grid (tiles + 1) x B)
//     prologue    every workgroup re-reduces the column's per-tile partials (<= 1024 (max, sum) pairs, L2-hot) into the
//                 tile-prefix table (P_t, f_t) in LDS, decides - SISR - whether the column resamples (ESS test on the
//                 same sums: identical arithmetic in every workgroup, so they all agree), and its first wave finds the
//                 window start j0 of the tile's first position (tile via the table, element via one 64-ary probe of the
//                 tile-local scans).  There is no planning kernel: the former k_fused_plan cost 8.4 of the 24 us of a
//                 step at 2^20 particles while moving no data.
//     body        stages a window of local scans L_i, maps them to cdf values (P_k + f_k L_i, fp64, rounded once), takes
//                 the ancestors from the inverted systematic grid (or an LDS search: multinomial / very large grids),
//                 gathers x[anc] (from LDS when small), propagates (Philox / tape), weighs, stores x', logw', anc - and,
//                 while the new state is in registers: the next step's per-tile partials (max / sum exp / sum exp^2,
//                 pivoted weighted moments, the APF's first-stage weights against y_{t+1}) and the tile-local inclusive
//                 scan L'_i of the next resampling weights.
//     bookkeeper  workgroup `tiles` of every column: moments row of the incoming state, log-likelihood increment of
//                 the previous step, the bases of the next one.  Nothing in the same launch depends on it.
// Multinomial resampling runs the same kernel: the sorted resampling positions (order statistics of N iid uniforms)
// are normalised prefix sums of Exp(1) spacings that the step kernel regenerates per round (Philox + workgroup scan);
// the prologue adds a prefix table of the spacings' tile sums (reduced with the partials one step earlier).
//
// k_fused_reduce produces the partials / local scans of the very first state of a run (and the run's per-column
// records: closed-form constants, the moments' pivot); k_fused_book is the bookkeeper alone, for the last state.  A kernel
// boundary is the only inter-workgroup synchronisation; the step index, "observed" flags and observation rows are kernel
// arguments set by the host loop (or baked into a captured hipGraph), so a launch has no dependent flag-load prologue.
#pragma once

namespace pf {

// the step kernel's grid axes: x = column, y = tile (bookkeepers last in dispatch order); A/B builds can restore (tiles + 1, B)
#define PF_STEP_B blockIdx.x
#define PF_STEP_K blockIdx.y

// Development instrumentation (cycle stamps, early-exit cuts for per-stage PMC profiles) is compiled in only with
// -DPF_DEVTOOLS (tools/pmc_stages.py builds that variant); the production kernels carry none of it.
#ifdef PF_DEVTOOLS
#define PF_CUT(a, n) ((a).debug_cut == (n))
#define PF_STAMP(a, slot)                                                                       \
    do {                                                                                        \
        if ((a).debug_cut < 0 && !(a).finalize_only && PF_STEP_K == (unsigned)(-(a).debug_cut - 1) &&  \
            PF_STEP_B == 0 && threadIdx.x == 0)                                                 \
            (a).dbg[slot] = (unsigned long long)clock64();                                      \
    } while (0)
#else
#define PF_CUT(a, n) false
#ifdef PF_ISA_MARKS  // stage boundaries as comments in the ISA listing (static instruction counts; they pin the schedule)
#define PF_STAMP(a, slot) asm volatile("; PF_MARK " #slot)
#else
#define PF_STAMP(a, slot) do { } while (0)
#endif
#endif

template <typename T> struct FusedArgs {
    ModelDesc md;
    const T* params;
    int filter, proposal, resampler;
    Geom g;
    double thr_abs;  // ess_threshold * N
    double logN;
    T rcN;           // 1 / N rounded to T on the host (the IEEE quotient the kernels would compute: ~10 VALU per thread saved)
    uint64_t seed;
    const uint64_t* seed_dev;  // optional device word added to `seed` (lets a captured graph draw fresh numbers per replay)
    T* x[2];
    T* logw[2];
    int32_t* anc;
    const int32_t* anc_prev;  // state history only: the ancestors of the incoming state (a SISR step that keeps its
                              // weights carries them forward into its own slot), else nullptr
    T* cdf;
    T* pos;      // (B, N) sorted resampling positions (multinomial)
    const T* y;  // (T, y_rows, O)
    int y_rows;
    const T* z_tape;
    const T* u_tape;
    const T* user_loc;    // PF_HID_USER_AFFINE: (D, B, N) one-step means / transition scales of the incoming particles, evaluated
    const T* user_scale;  // by the caller's callable (pf_filter_args.user_loc / user_scale)
    int user_scale_percol;  // user_scale is a (D, B) array: one scale per column and component
    T user_dt;              // != 0: user_loc holds the drift f(x) of an Euler-Maruyama process, the mean is x + f dt (pf_filter_args.user_dt)
    T* means;
    T* vars;
    T* ll_steps;
    T* ll_total;
    double* part;          // per-tile partials, two copies: state q's live in copy q & 1 (the step kernel reads the plan of
    int64_t part_stride;   // state `step` while it writes the partials of state `step + 1`; a replayed step stays exact)
    ColStat* stat;
    int32_t* poison;  // [4][B]: slot s & 3 poisons ll_s (written by the launches of steps s - 1 and s, consumed - and
                      // cleared - by the bookkeeper of launch s + 1: three launches may touch three different slots)
    T* cpack;         // [B][PK_N] the run's closed-form constants of every column (scalar fast path; see FastCol)
    double* piv0;     // [B][PF_MAXD] pivot of the weighted moments of the run's first two states (a particle of the first)
    double* ctab;     // [2][B][tiles * R * 4][2] per-chunk (C, g) of the chunk-local scans (see ChunkScan); double buffered by
    int64_t ctab_stride;  // state parity like the scans themselves
    int t0;           // first step of this run: the partials of states t0, t0 + 1 are taken about piv0, those of a later
                      // state q about the mean of state q - 2 (row q - 2 of `means`: written two launches earlier)
    // per launch
    int step;       // local step index: slot = step & 1 is read, the other written
    int obs;        // this step weighs against y[step]                     (-1: read obs_dev[step]; -2: look at y[step] itself)
    int obs_next;   // the next step exists and is a weighted step (its first-stage weights are prepared here; -1: device)
    const uint8_t* obs_dev;  // optional device flags (pf_filter_args.observed_dev)
    // -2 (one-step runs on a shared observation row, neither flag array given: the online move): "not all-NaN" of the row's <= 3
    // values, evaluated by every workgroup itself - uniform scalar loads - instead of a flag byte a launch of its own derived
    __device__ __forceinline__ bool y_informative() const {
        const T* yr = y + (int64_t)step * md.obs_dim;  // (y_rows == 1)
        bool any = false;
        for (int o = 0; o < md.obs_dim; ++o) any = any || !(yr[o] != yr[o]);
        return any;
    }
    __device__ __forceinline__ bool is_obs() const { return obs >= 0 ? obs != 0 : (obs == -1 ? obs_dev[step] != 0 : y_informative()); }
    __device__ __forceinline__ bool is_obs_next() const { return obs_next >= 0 ? obs_next != 0 : obs_dev[step + 1] != 0; }
    int finalize_only;
    int book_inline;  // the column's bookkeeping is done by its last step workgroup (after its own work) instead of an
                      // extra workgroup per column (development / A-B only since the bookkeepers are dispatched last)
    unsigned kmap;    // block row y -> tile k = ((y & m) << s) + (y >> r), (m, s, r) = bytes 0, 1, 2; 0 = identity.  Workgroups go
                      // to the 8 XCDs round robin by linear id; a single column of 2^q tiles maps row y to tile
                      // (y % 8) (tiles / 8) + y / 8 (m = 7, s = q - 3, r = 3): every XCD works on a CONTIGUOUS eighth of the
                      // column, so the neighbouring tiles whose scans / particles a tile gathers from were written - and are
                      // still cached - by its own XCD's L2 (many columns with B % 8 == 0 have that property by construction:
                      // linear id = b + B k)
    __device__ __forceinline__ int tile_of_row(unsigned y) const {
        return (int)(((y & (kmap & 0xffu)) << ((kmap >> 8) & 0xffu)) + (y >> ((kmap >> 16) & 0xffu)));
    }
    unsigned long long* dbg;
    __device__ __forceinline__ const double* part_r() const { return part + (int64_t)(step & 1) * part_stride; }
    __device__ __forceinline__ double* part_w(int state) const { return part + (int64_t)(state & 1) * part_stride; }
    __device__ __forceinline__ const double* ctab_r(int b) const {
        return ctab + (int64_t)(step & 1) * ctab_stride + (int64_t)b * g.tiles * g.rounds_per_tile * PF_NWAVES * 2;
    }
    __device__ __forceinline__ double* ctab_w(int state, int b) const {
        return ctab + (int64_t)(state & 1) * ctab_stride + (int64_t)b * g.tiles * g.rounds_per_tile * PF_NWAVES * 2;
    }
    // pivot of the weighted-moment partials of state q (see t0), component d of column b
    template <int D> __device__ __forceinline__ T pivot(int q, int b, int d) const {
        return (q - 2 >= t0) ? means[((int64_t)(q - 2) * g.B + b) * D + d] : (T)piv0[(int64_t)b * PF_MAXD + d];
    }
    int debug_cut;  // development knob (env PF_DEBUG_CUT): kernels return early after stage n; 0 = off
};

// `late`: an opaque zero added to the row addresses (the step kernel ties it to a value computed after the ancestor
// search, so the loads - and the registers they fill - cannot be hoisted above it)
template <typename T, int D>
__device__ __forceinline__ void load_col_params(const FusedArgs<T>& a, int b, int step, bool with_y, ColParams<T, D>& cp,
                                                int late = 0) {
    const int O = a.md.obs_dim;
    const int NP = 4 * D + O * D + 2 * O;
    cp.load(a.params + (int64_t)b * NP + late, O,
            with_y ? a.y + ((int64_t)step * a.y_rows + (a.y_rows == 1 ? 0 : b)) * O + late : nullptr);
}

// Accumulates the per-tile partials of a state from registers.  Per-thread accumulators are of the filter's type T: a
// thread sums at most 4 R terms, the workgroup / column sums above it are fp64.  The weighted moments are taken about a
// per-column pivot c (the previous state's mean) - sum e (x - c), sum e (x - c)^2 - so the variance never cancels.
// WQ: also sum e^2 (the ESS of the weights - SISR's resampling test; an APF never looks at it)
template <typename T, int D, bool WQ = true> struct PartialAcc {
    T m1, m2;          // running maxima of logw / rw
    T s1, s2, q1;      // sum e, sum e_rw, sum e^2
    T mx[D], mxx[D];   // sum e (x - c), sum e (x - c)^2
    double es;         // sum of the Exp(1) spacings the next step's sorted-uniform multinomial will use for these particles
    bool poison;
    __device__ __forceinline__ void init() {
        m1 = m2 = -Lim<T>::inf();
        s1 = s2 = q1 = T(0);
        es = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) mx[d] = mxx[d] = T(0);
        poison = false;
    }
    // One round of VEC particles: lw sanitised log-weights, x particles, pre (if pre_on) first-stage log-weights of the
    // next step.  The running maxima move at most once per round, so there is one exp per element (+ one per rescale).
    // e_out: exp(rw_j - thread max) of this round's resampling weights rw (= lw, or lw + pre when pre_on) - for
    // single-round tiles the tile-local scan reuses them (times finish()'s factor) instead of evaluating exp again.
    template <int VEC>
    __device__ __forceinline__ void push_round(const T (&lw)[VEC], const T (&x)[D][VEC], bool pre_on, const T (&pre)[VEC],
                                               const T (&piv)[D], T (&e_out)[VEC]) {
        T m = lw[0];
#pragma unroll
        for (int j = 1; j < VEC; ++j) m = (lw[j] > m) ? lw[j] : m;
        if (m > m1) {
            const T rs = (m1 == -Lim<T>::inf()) ? T(0) : pf_exp_w(m1 - m);
            s1 *= rs;
            if constexpr (WQ) q1 *= rs * rs;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                mx[d] *= rs;
                mxx[d] *= rs;
            }
            m1 = m;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            T e = (lw[j] == -Lim<T>::inf()) ? T(0) : pf_exp_w(lw[j] - m1);
            if (lw[j] != lw[j]) e = lw[j];
            e_out[j] = e;
            s1 += e;
            if constexpr (WQ) q1 += e * e;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const T xd = x[d][j] - piv[d];
                mx[d] += e * xd;
                mxx[d] += e * xd * xd;
            }
        }
        if (pre_on) {
            T rw[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (is_nan_or_posinf(pre[j])) poison = true;
                rw[j] = sanitize_logw(pre[j] + lw[j]);
            }
            T mm = rw[0];
#pragma unroll
            for (int j = 1; j < VEC; ++j) mm = (rw[j] > mm) ? rw[j] : mm;
            if (mm > m2) {
                s2 *= (m2 == -Lim<T>::inf()) ? T(0) : pf_exp_w(m2 - mm);
                m2 = mm;
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const T e = (rw[j] == -Lim<T>::inf()) ? T(0) : pf_exp_w(rw[j] - m2);
                e_out[j] = e;
                s2 += e;
            }
        }
    }
    // workgroup reduction + store in two LDS exchanges (maxima, then every rescaled sum); `red` >= (4 + 2D) * PF_NWAVES
    // doubles, `redm` >= 2 * PF_NWAVES Ts, neither used by anything still in flight.  The within-wave sums run in T (for
    // float: 6 DPP adds per quantity instead of fp64 pairs of moves), everything above a wave is fp64.  WITH_ES: the
    // Exp(1) spacings of the multinomial route are reduced too.  F1 / F2: this thread's factors exp(m - M).
    template <bool WITH_ES>
    __device__ __forceinline__ void finish(double* part, int b, int k, int B, int tiles, bool pre_on, double* red, T* redm,
                                           int32_t* poison_slot, T& M1_out, T& M2_out, T& F1_out, T& F2_out) {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        const T w1 = wave_max<T>(m1), w2 = wave_max<T>(m2);
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
        M1_out = M1;
        M2_out = M2;
        const T f1 = (m1 == -Lim<T>::inf()) ? T(0) : pf_exp_w(m1 - M1);
        const T f2 = (!pre_on || m2 == -Lim<T>::inf()) ? T(0) : pf_exp_w(m2 - M2);
        F1_out = f1;
        F2_out = f2;
        constexpr int NS = 3 + 2 * D;
        T sums[NS];
        sums[0] = s1 * f1;
        sums[1] = WQ ? q1 * f1 * f1 : T(0);
        sums[2] = pre_on ? s2 * f2 : T(0);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            sums[3 + d] = mx[d] * f1;
            sums[3 + D + d] = mxx[d] * f1;
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const T ws = (WQ || q != 1) ? wave_sum(sums[q]) : T(0);
            if (lane == 0) red[q * PF_NWAVES + wid] = (double)ws;
        }
        if constexpr (WITH_ES) {
            const double ws = wave_sum(es);
            if (lane == 0) red[NS * PF_NWAVES + wid] = ws;
        }
        if (poison) atomicOr(poison_slot, 1);
        __syncthreads();
        // the tile's record: lane q of wave 0 adds the four waves' sums of quantity q (same order as a single thread would:
        // identical values) and stores it to its row - one pass of ~10 instructions instead of thread 0 walking NS + 1
        // quantities and 6 + 2 D stores one after the other at the very end of the workgroup's critical path
        if (threadIdx.x < NS + 1) {
            const int q = threadIdx.x;
            double r = 0.0;
            if (q < NS || WITH_ES) {
                r = red[q * PF_NWAVES];
#pragma unroll
                for (int w = 1; w < PF_NWAVES; ++w) r += red[q * PF_NWAVES + w];
            }
            // quantity q -> partial row: 0 sum e, 1 sum e^2, 2 sum e_rw, 3 .. 3 + 2 D - 1 the moments, NS the spacings
            const int row = q == 0 ? PQ_S1 : q == 1 ? PQ_Q1 : q == 2 ? PQ_S2 : q == NS ? PQ_E : PQ_MX + (q - 3);
            const int64_t stride = (int64_t)B * tiles;
            const int64_t o = (int64_t)b * tiles + k;
            part[row * stride + o] = r;
            if (q == 0) {
                part[PQ_M1 * stride + o] = (double)M1;
                part[PQ_M2 * stride + o] = pre_on ? (double)M2 : -__builtin_huge_val();
            }
        }
    }
};

// The scan of a tile's resampling weights is stored CHUNK by chunk (a chunk = the 64 * VEC consecutive particles one wave
// owns in a round): L_i = inclusive scan inside the chunk of exp(rw_i - m_c), m_c the chunk's own maximum - produced by
// wave-level operations alone (no workgroup barrier, no second pass over the tile once its maximum is known; for
// multi-round tiles that second pass used to re-read the whole state).  A chunk's values become tile-level through
// (C_c, g_c) = (sum_{c' < c} t_c' g_c', exp(m_c - M)), written once per tile after its last round (finalize_chunk_table):
//   cdf_i = T(min(P_k + f_k (C_c + g_c L_i), P_{k+1})),
// fp64, rounded once to T; a tile's last element is pinned to T(P_{k+1}) and the column's last one to 1
// (resampling.py:49).  The prologue's probe, the body's window staging and the fallback search all go through this one
// function.  Chunk-local sums are <= 64 VEC, so the stored T values carry their full precision.
#define PF_LDS_CHUNKS PF_WAVE  // tiles of up to this many chunks (16 rounds) keep their raw chunk records in LDS
// Single-round tiles know their maximum before anything is stored: their scans are tile-level (one workgroup scan of
// the exponentials the partial accumulator holds anyway), i.e. (C, g) = (0, 1) everywhere - the kernels of that geometry
// (MULTI = false) neither write nor read a chunk table.
template <typename T>
__device__ __forceinline__ T cdf_from_local(T L, double Pk, double fk, double Pnext, bool tile_last, bool col_last) {
    double c = Pk + fk * (double)L;
    if (c > Pnext) c = Pnext;
    T r = (T)c;
    if (tile_last) r = col_last ? T(1) : (T)Pnext;
    return r;
}
template <typename T>
__device__ __forceinline__ T cdf_from_local(T L, double C, double gq, double Pk, double fk, double Pnext, bool tile_last,
                                            bool col_last) {
    double c = Pk + fk * (C + gq * (double)L);
    if (c > Pnext) c = Pnext;
    T r = (T)c;
    if (tile_last) r = col_last ? T(1) : (T)Pnext;
    return r;
}
template <typename T> struct CdfView {  // searches in the implied cdf of one column (slow path)
    const T* L;
    const double* ptab;  // this column's (tiles + 1) prefixes
    const double* ftab;
    const double* ct;    // this column's per-chunk (C, g) pairs; nullptr for single-round tiles
    int chunk_elems;
    int N;
    int tile_elems;
    int tiles;
    // first i in [from, N) with cdf(i) >= p (N if none), given a tile index kt_lo at or below the tile of `from`.
    // Two levels - the tile by its end value T(P_{kt+1}) (1 for the last tile), then the element inside that tile with
    // the tile's (P, f) held in registers - so there is no index / tile_elems division anywhere.
    __device__ __forceinline__ int lower_bound(int from, T p, int kt_lo) const {
        if (from >= N) return N;
        int lo = kt_lo, hi = tiles - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((T)ptab[mid + 1] < p) lo = mid + 1; else hi = mid;
        }
        for (int kt = lo; kt < tiles; ++kt) {  // (one pass unless `from` lies beyond every qualifying element of tile lo)
            const int first = kt * tile_elems;
            const int end = first + tile_elems < N ? first + tile_elems : N;
            int a = from > first ? from : first, b = end;
            if (a >= b) continue;
            const double Pk = ptab[kt], fk = ftab[kt], Pn = ptab[kt + 1];
            while (a < b) {
                const int mid = (a + b) >> 1;
                T cv;
                if (ct) {
                    const double* cg = ct + 2 * (mid / chunk_elems);
                    cv = cdf_from_local<T>(L[mid], cg[0], cg[1], Pk, fk, Pn, mid == end - 1, mid == N - 1);
                } else {
                    cv = cdf_from_local<T>(L[mid], Pk, fk, Pn, mid == end - 1, mid == N - 1);
                }
                if (cv < p) a = mid + 1; else b = mid;
            }
            if (a < end) return a;
        }
        return N;
    }
};

// One round of the chunk-local scan (see cdf_from_local): rw[j] the thread's VEC resampling log-weights (-inf beyond the
// column), executed by every lane of every wave (wave-level reductions).  Stores L_i and leaves the chunk's raw record
// (m_c, t_c) in LDS (`lds_rec`, tiles of <= PF_LDS_CHUNKS chunks) or in the tile's slice of the chunk table (`raw`).
// WT: the scans are an output plane of the step kernel (written through, store_out); k_fused_reduce re-reads its own
// (single-round tiles fold (C, g) in) and keeps them in the cache.
template <typename T, int VEC, bool WT = false>
__device__ __forceinline__ void chunk_scan_round(const T (&rw)[VEC], bool on, T* __restrict__ l_base, int l_elem, int chunk,
                                                 double* lds_rec, double* raw) {
    T* const l_dst = l_base + l_elem;  // (l_base: the round's first element, uniform across the workgroup)
    const int lane = threadIdx.x & 63;
    T mt = rw[0];
#pragma unroll
    for (int j = 1; j < VEC; ++j) mt = rw[j] > mt ? rw[j] : mt;
    const T mc = wave_max<T>(mt);
    double incl[VEC], local = 0.0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        local += (rw[j] == -Lim<T>::inf()) ? 0.0 : (double)pf_exp_w(rw[j] - mc);
        incl[j] = local;
    }
    const double iw = wave_scan_incl(local, lane);
    if (on) {
        const double excl = iw - local;
        T outv[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) outv[j] = (T)(excl + incl[j]);
        if (VEC == 1) l_dst[0] = outv[0];
        else if constexpr (WT) store_out<T, VEC>(l_base, l_elem, outv);
        else store_vec<T, VEC>(l_dst, outv);
    }
    if (lane == 63) {
        double* rec = lds_rec ? lds_rec + 2 * chunk : raw + 2 * chunk;
        rec[0] = (double)mc;
        rec[1] = iw;
    }
}
// After a tile's last round: raw chunk records (m_c, t_c) -> (C_c, g_c) in the chunk table, and the tile's (max, sum)
// of the resampling weights.  Tiles of <= 64 chunks: one barrier (the records), then wave-level arithmetic (every wave
// redundantly, wave 0 writes).  `ct_tile`: the tile's slice of the chunk table (also holds the raw records of larger tiles).
template <typename T>
__device__ __forceinline__ void finalize_chunk_table(double* ct_tile, const double* lds_rec, int nchunks, double* reds,
                                                     T* redm, double& M_out, double& S_out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();  // the last round's records are in place
    if (lds_rec) {
        const bool valid = lane < nchunks;
        const double mq = valid ? lds_rec[2 * lane] : -__builtin_huge_val();
        const double tq = valid ? lds_rec[2 * lane + 1] : 0.0;
        const double M = (double)wave_max<T>((T)mq);
        const double gq = exp_diff_t<T>(mq, M);
        const double v = tq * gq;
        const double iw = wave_scan_incl(v, lane);
        if (wid == 0 && valid) {
            ct_tile[2 * lane] = iw - v;
            ct_tile[2 * lane + 1] = gq;
        }
        M_out = M;
        S_out = lane_get(iw, 63);
        return;
    }
    T mx = -Lim<T>::inf();
    for (int q = threadIdx.x; q < nchunks; q += PF_BLOCK) mx = (T)ct_tile[2 * q] > mx ? (T)ct_tile[2 * q] : mx;
    const double M = (double)block_max<T>(mx, redm);
    double carry = 0.0;
    for (int q0 = 0; q0 < nchunks; q0 += PF_BLOCK) {
        const int q = q0 + threadIdx.x;
        const bool valid = q < nchunks;
        const double gq = exp_diff_t<T>(valid ? ct_tile[2 * q] : -__builtin_huge_val(), M);
        const double v = valid ? ct_tile[2 * q + 1] * gq : 0.0;
        double total;
        const double excl = block_scan_excl(v, reds, total);
        if (valid) {
            ct_tile[2 * q] = carry + excl;
            ct_tile[2 * q + 1] = gq;
        }
        carry += total;
    }
    M_out = M;
    S_out = carry;
}

// partials of the state in slot (step & 1) - only needed for the first state of a run
template <typename T, int D, int VEC>
__global__ __launch_bounds__(PF_BLOCK) void k_fused_reduce(FusedArgs<T> a) {
    __shared__ double red[(4 + 2 * D) * PF_NWAVES];
    __shared__ T redm[2 * PF_NWAVES];
    const Geom& g = a.g;
    const int b = blockIdx.y, k = blockIdx.x;
    const int slot = a.step & 1;
    const bool pre_on = a.is_obs() && a.filter == PF_FILTER_APF;
    const bool user = a.md.hid_kind == PF_HID_USER_AFFINE;
    ColParams<T, D> cp;
    load_col_params<T, D>(a, b, a.step, pre_on, cp);
    ColConsts<T, D> cc;
    cc.prepare(a.md, cp);
    if (user && a.user_scale_percol) cc.prepare_user(a.md, cp, a.user_scale[b]);  // (scalar states: the closed forms, as in the step kernel)

    const T* lw_col = a.logw[slot] + (int64_t)b * g.N;
    const T* x_base = a.x[slot];
    // the run's per-column records.  Pivot of the weighted moments of the run's first two states: the column's first
    // particle (any value inside the cloud keeps sum e (x - c)^2 - (sum e (x - c))^2 from cancelling); every workgroup
    // loads it, workgroup 0 publishes it for the step kernels / bookkeepers of this run
    T piv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) piv[d] = x_base[((int64_t)d * g.B + b) * g.N];
    if (k == 0 && threadIdx.x == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) a.piv0[(int64_t)b * PF_MAXD + d] = (double)piv[d];
    }
    if constexpr (D == 1) {
        // closed-form constants (FastCol): functions of the parameters only - the step kernels add the observations
        if (k == 0 && threadIdx.x == PF_BLOCK - 1 && a.md.obs_kind == PF_OBS_LINEAR && a.md.hid_kind != PF_HID_VERHULST_EM) {
            ColParams<T, 1> cq;
            ColConsts<T, 1> cd;
            load_col_params<T, 1>(a, b, a.step, false, cq);
            cd.prepare(a.md, cq);
            write_col_pack<T>(a.md, cq, cd, a.cpack + (int64_t)b * PK_N);
        }
    }
    PartialAcc<T, D> acc;
    acc.init();
    __shared__ double reds[PF_NWAVES];
    __shared__ double crec[2 * PF_LDS_CHUNKS];
    const bool use_lds = g.rounds_per_tile * PF_NWAVES <= PF_LDS_CHUNKS;
    double* const ct_tile = a.ctab_w(a.step, b) + 2 * (int64_t)k * g.rounds_per_tile * PF_NWAVES;
    T* const l_col = ((a.step & 1) ? a.pos : a.cdf) + (int64_t)b * g.N;  // double buffered like the state (step parity)
    const int64_t base = (int64_t)k * g.tile_elems;
    int rk = 0;
    for (int r = 0; r < g.rounds_per_tile; ++r) {
        const int64_t r0 = base + (int64_t)r * g.round_elems;
        if (r0 >= g.N) break;
        ++rk;
        const int64_t i0 = r0 + threadIdx.x * VEC;
        const bool on = i0 < g.N;
        T rw[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) rw[j] = -Lim<T>::inf();
        if (on) {
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
                UserMS<T, D> um = UserMS<T, D>::none();
                if (user && pre_on) {  // the particle's own one-step mean / scale (the caller's planes)
                    um.gather(a.user_loc, a.user_scale, (int64_t)b * g.N, (int64_t)g.B * g.N, i0 + j, a.user_scale_percol != 0, b, g.B);
                    um.euler(xj, a.user_dt);
                }
                pre[j] = pre_on ? pre_weight<T, D>(a.md, a.proposal, cp, cc, xj, false, um) : T(0);
                if (pre_on && is_nan_or_posinf(pre[j])) acc.poison = true;
                rw[j] = pre_on ? sanitize_logw(pre[j] + lw[j]) : lw[j];
            }
            T e_unused[VEC];
            acc.template push_round<VEC>(lw, xv, false, pre, piv, e_unused);  // (the resampling family: the chunk scan below)
            if (a.resampler == PF_RESAMPLE_MULTINOMIAL) {
                T ev[VEC];
                draw_exponentials<T, VEC>(a.seed + (a.seed_dev ? *a.seed_dev : 0ull), PF_STREAM_MULTINOMIAL, (uint32_t)a.step,
                                          (uint64_t)((int64_t)b * g.N + i0), ev);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc.es += (double)ev[j];
            }
        }
        chunk_scan_round<T, VEC>(rw, on, l_col + r0, threadIdx.x * VEC, r * PF_NWAVES + (threadIdx.x >> 6), use_lds ? crec : nullptr, ct_tile);
    }
    T M1, M2, F1, F2;
    acc.template finish<true>(a.part_w(a.step), b, k, g.B, g.tiles, false, red, redm, &a.poison[(a.step & 3) * g.B + b], M1, M2, F1, F2);
    double Mc, Sc;
    finalize_chunk_table<T>(ct_tile, use_lds ? crec : nullptr, rk * PF_NWAVES, reds, redm, Mc, Sc);
    if (threadIdx.x == 0) {  // the resampling family's (max, sum) come from the scan
        const int64_t stride = (int64_t)g.B * g.tiles, o = (int64_t)b * g.tiles + k;
        if (pre_on) {
            a.part_w(a.step)[PQ_M2 * stride + o] = Mc;
            a.part_w(a.step)[PQ_S2 * stride + o] = Sc;
        } else {
            a.part_w(a.step)[PQ_S1 * stride + o] = Sc;
        }
    }
    if (g.rounds_per_tile == 1) {
        // the step kernels of single-round tiles read tile-level scans and no chunk table (cdf_from_local): fold (C, g) in
        __syncthreads();  // the chunk table of this tile is written (wave 0)
        const int64_t i0 = base + threadIdx.x * VEC;
        if (i0 < g.N) {
            const double* cg = ct_tile + 2 * (threadIdx.x >> 6);
            const double C = cg[0], gq = cg[1];
            T v[VEC];
            if (VEC == 1) v[0] = l_col[i0]; else load_vec<T, VEC>(l_col + i0, v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = (T)(C + gq * (double)v[j]);
            if (VEC == 1) l_col[i0] = v[0]; else store_vec<T, VEC>(l_col + i0, v);
        }
    }
}

// Column-wide sums of one weight family from the per-tile partials: M = max_t m_t, S = sum_t s_t e_t and - WITH_Q -
// Q = sum_t q_t e_t^2, e_t = exp(m_t - M); TABLE: also the tile-prefix table in LDS, ptl[t] = (sum_{t' < t} s_t' e_t') / S
// (ptl[0] = 0 ... ptl[tiles] ~ 1) and ftl[t] = e_t / S, the scale of tile t's max-shifted local sums.  Thread t owns the
// IT consecutive tiles [t * IT, (t + 1) * IT): one load of the (max, sum) pairs, then two workgroup exchanges (the column
// maximum; the scan whose total is the column sum).  Every workgroup that needs these numbers - the step workgroups of
// a column and its bookkeeper - runs THIS function on the same partials: the values (and the SISR resampling decision
// taken from them) are bit-identical everywhere, which is what lets the step kernel do without a planning kernel.
// redm: PF_NWAVES doubles, reds: (2 + NX) * PF_NWAVES doubles.  Ends with a barrier when TABLE (the table is readable).
#define PF_COMBINE_ITERS (PF_MAX_TILES / PF_BLOCK)  // partial records per thread
// 1 / (the column's sum of shifted weights) - the scale of the tile-prefix table.  double filters (the parity path): the IEEE
// division (~30 instructions).  float filters: v_rcp_f64 + two Newton steps (8 instructions, <= 1 ulp of the quotient in
// double - 2^-29 of a float ulp of any cdf value; every workgroup of a column performs the same operations on the same sums,
// so they still agree bit for bit)
template <typename T> __device__ __forceinline__ double pf_rcp_tot(double tot) {
    if constexpr (sizeof(T) == 4) {
        double r = __builtin_amdgcn_rcp(tot);
        r = __builtin_fma(__builtin_fma(-tot, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-tot, r, 1.0), r, r);
        return r;
    }
    return 1.0 / tot;
}
#define PF_PROBE_STEP 8  // spacing of the prologue's window-start probes (entries)
struct ColSums {
    double M, S, Q;
};
// NX > 0 (the bookkeeper): additionally X[q] = sum_t x_{q,t} e_t for the NX partial rows `slot_x + q` - same load round,
// same exchange (the moments' numerators).
// Two halves, so that a caller can put independent work (the step kernel: its Philox draws) between issuing the loads and
// the first use of what they return.
template <bool WITH_Q, int NX> struct ColPartials {
    double m[PF_COMBINE_ITERS], s[PF_COMBINE_ITERS], q[PF_COMBINE_ITERS];
    double x[NX ? NX : 1][PF_COMBINE_ITERS];
};
template <bool WITH_Q, int NX = 0>
__device__ __forceinline__ void load_col_partials(const double* part, int64_t stride, int64_t cb, int tiles, int slot_m,
                                                  int slot_s, int slot_x, ColPartials<WITH_Q, NX>& r) {
    const int IT = (tiles + PF_BLOCK - 1) / PF_BLOCK;
    if constexpr (PF_COMBINE_ITERS == 4 && NX == 0) {
        // a full table (IT = 4, tiles a multiple of four: 2^20 x 1 and every column of >= 769 tiles): a thread's four records
        // are 32 contiguous, 32-byte aligned bytes of each row - two 16-byte loads per row instead of four 8-byte ones with
        // their 64-bit address arithmetic (the rows start 256-byte aligned: make_ws; stride and cb are multiples of tiles)
        if (IT == 4 && (tiles & 3) == 0) {  // (uniform)
            const int t0 = threadIdx.x * 4;
            if (t0 < tiles) {
                double a2[2], b2[2];
                const double* pm = part + slot_m * stride + cb + t0;
                load_vec<double, 2>(pm, a2); load_vec<double, 2>(pm + 2, b2);
                r.m[0] = a2[0]; r.m[1] = a2[1]; r.m[2] = b2[0]; r.m[3] = b2[1];
                const double* ps = part + slot_s * stride + cb + t0;
                load_vec<double, 2>(ps, a2); load_vec<double, 2>(ps + 2, b2);
                r.s[0] = a2[0]; r.s[1] = a2[1]; r.s[2] = b2[0]; r.s[3] = b2[1];
                if constexpr (WITH_Q) {
                    const double* pq = part + PQ_Q1 * stride + cb + t0;
                    load_vec<double, 2>(pq, a2); load_vec<double, 2>(pq + 2, b2);
                    r.q[0] = a2[0]; r.q[1] = a2[1]; r.q[2] = b2[0]; r.q[3] = b2[1];
                } else {
                    r.q[0] = r.q[1] = r.q[2] = r.q[3] = 0.0;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) { r.m[q] = -__builtin_huge_val(); r.s[q] = 0.0; r.q[q] = 0.0; }
            }
            return;
        }
    }
#pragma unroll
    for (int q = 0; q < PF_COMBINE_ITERS; ++q) {
        const int t = threadIdx.x * IT + q;
        const bool on = q < IT && t < tiles;
        r.m[q] = on ? part[slot_m * stride + cb + t] : -__builtin_huge_val();
        r.s[q] = on ? part[slot_s * stride + cb + t] : 0.0;
        r.q[q] = (WITH_Q && on) ? part[PQ_Q1 * stride + cb + t] : 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) r.x[i][q] = on ? part[(slot_x + i) * stride + cb + t] : 0.0;
    }
}
template <typename T, bool WITH_Q, bool TABLE, int NX = 0>
__device__ __forceinline__ ColSums column_sums(const ColPartials<WITH_Q, NX>& in, int tiles, double* ptl, double* ftl,
                                               double* redm, double* reds, double* X = nullptr) {
#pragma clang fp contract(off)  // the same operations in every instantiation / inlining context (see above)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int IT = (tiles + PF_BLOCK - 1) / PF_BLOCK;
    const double (&mloc)[PF_COMBINE_ITERS] = in.m;
    const double (&sloc)[PF_COMBINE_ITERS] = in.s;
    const double (&qloc)[PF_COMBINE_ITERS] = in.q;
    const double (&xloc)[NX ? NX : 1][PF_COMBINE_ITERS] = in.x;
    // (the maxima are values of T carried as doubles: reduced in T - half the cross-lane traffic for float filters)
    T mymax_t = -Lim<T>::inf();
#pragma unroll
    for (int q = 0; q < PF_COMBINE_ITERS; ++q) mymax_t = (T)mloc[q] > mymax_t ? (T)mloc[q] : mymax_t;
    const double MR = (double)block_max<T>(mymax_t, reinterpret_cast<T*>(redm));
    double incl[PF_COMBINE_ITERS], ef[PF_COMBINE_ITERS], run = 0.0, qs = 0.0;
    double xs[NX ? NX : 1];
#pragma unroll
    for (int i = 0; i < NX; ++i) xs[i] = 0.0;
#pragma unroll
    for (int q = 0; q < PF_COMBINE_ITERS; ++q) {
        ef[q] = exp_diff_t<T>(mloc[q], MR);
        run += sloc[q] * ef[q];
        incl[q] = run;
        if (WITH_Q) qs += qloc[q] * ef[q] * ef[q];
#pragma unroll
        for (int i = 0; i < NX; ++i) xs[i] += xloc[i][q] * ef[q];
    }
    const double incl_w = wave_scan_incl(run, lane);
    const double qw = WITH_Q ? wave_sum(qs) : 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) xs[i] = wave_sum(xs[i]);
    __syncthreads();
    if (lane == 63) reds[wid] = incl_w;
    if (WITH_Q && lane == 0) reds[PF_NWAVES + wid] = qw;
    if (NX && lane == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) reds[(2 + i) * PF_NWAVES + wid] = xs[i];
    }
    __syncthreads();
    double wave_off = 0.0, tot = 0.0, qtot = 0.0;
#pragma unroll
    for (int w = 0; w < PF_NWAVES; ++w) {
        const double sw = reds[w];
        if (w < wid) wave_off += sw;
        tot += sw;
        if (WITH_Q) qtot += reds[PF_NWAVES + w];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        double r = reds[(2 + i) * PF_NWAVES];
#pragma unroll
        for (int w = 1; w < PF_NWAVES; ++w) r += reds[(2 + i) * PF_NWAVES + w];
        X[i] = r;
    }
    if constexpr (TABLE) {
        const double excl = wave_off + incl_w - run;
        const double inv_tot = pf_rcp_tot<T>(tot);  // one reciprocal per thread, not 2 IT divisions
        if (threadIdx.x == 0) ptl[0] = 0.0;
#pragma unroll
        for (int q = 0; q < PF_COMBINE_ITERS; ++q) {
            const int t = threadIdx.x * IT + q;
            if (q < IT && t < tiles) {
                ptl[t + 1] = (excl + incl[q]) * inv_tot;
                ftl[t] = ef[q] * inv_tot;
            }
        }
        __syncthreads();
    }
    return ColSums{MR, tot, qtot};
}
// A column of ONE tile (many filters of a few thousand particles - SMC^2's 1024 x 8192): the table is (0, ~1) and its scale
// 1 / S - the operations column_sums performs on a single record, in its order (so the bookkeeper, which runs column_sums
// on the same record, agrees bit for bit), without the workgroup scan, its exchanges and the three padded records per thread.
template <typename T, bool WITH_Q>
__device__ __forceinline__ ColSums column_sums_one_tile(double m, double s, double q, double* ptl, double* ftl) {
#pragma clang fp contract(off)
    const double MR = (double)(T)m;
    const double ef = exp_diff_t<T>(m, MR);
    const double run = s * ef;
    const double qs = WITH_Q ? q * ef * ef : 0.0;
    const double inv_tot = pf_rcp_tot<T>(run);
    if (threadIdx.x == 0) {
        ptl[0] = 0.0;
        ptl[1] = run * inv_tot;
        ftl[0] = ef * inv_tot;
    }
    __syncthreads();
    return ColSums{MR, run, qs};
}
// A column of 2 .. 64 tiles (64 x 65 536, 128 x 8 192, ...): one record per LANE - every wave computes the sums on its own
// (wave maximum, wave scan: no workgroup exchange, no padded records), wave 0 writes the table.  The operations are the
// ones column_sums performs when thread t holds tile t alone (IT = 1: the other waves contribute exact zeros), in its order,
// so the bookkeeper's column_sums on the same records agrees bit for bit.  Ends with a barrier (the table is readable).
template <typename T, bool WITH_Q>
__device__ __forceinline__ ColSums column_sums_wave(const double* part, int64_t stride, int64_t cb, int tiles, int slot_m,
                                                    int slot_s, double* ptl, double* ftl) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bool on = lane < tiles;
    const double m = on ? part[slot_m * stride + cb + lane] : -__builtin_huge_val();
    const double s = on ? part[slot_s * stride + cb + lane] : 0.0;
    const double q = (WITH_Q && on) ? part[PQ_Q1 * stride + cb + lane] : 0.0;
    const double MR = (double)wave_max<T>((T)m);
    const double ef = exp_diff_t<T>(m, MR);
    double run = 0.0, qs = 0.0;
    run += s * ef;
    if (WITH_Q) qs += q * ef * ef;
    const double incl_w = wave_scan_incl(run, lane);
    const double qtot = WITH_Q ? 0.0 + wave_sum(qs) : 0.0;
    const double tot = 0.0 + lane_get(incl_w, 63);
    const double excl = 0.0 + incl_w - run;
    const double inv_tot = pf_rcp_tot<T>(tot);
    if (wid == 0) {
        if (lane == 0) ptl[0] = 0.0;
        if (on) {
            ptl[lane + 1] = (excl + run) * inv_tot;
            ftl[lane] = ef * inv_tot;
        }
    }
    __syncthreads();
    return ColSums{MR, tot, qtot};
}
// multinomial: exclusive prefixes of the tiles' Exp(1) spacing sums in pel[0 .. tiles) and, one slot behind them
// (pel[tiles + 1]), the grand total including the closing spacing `tail`.  Same thread-to-tile mapping as column_sums.
__device__ __forceinline__ void spacing_table(const double* part, int64_t stride, int64_t cb, int tiles, double tail,
                                              double* pel, double* reds) {
    const int IT = (tiles + PF_BLOCK - 1) / PF_BLOCK;
    double inclE[PF_COMBINE_ITERS], runE = 0.0, totalE;
#pragma unroll
    for (int q = 0; q < PF_COMBINE_ITERS; ++q) {
        const int t = threadIdx.x * IT + q;
        if (q < IT && t < tiles) runE += part[PQ_E * stride + cb + t];
        inclE[q] = runE;
    }
    const double exclE = block_scan_excl(runE, reds, totalE);
    if (threadIdx.x == 0) {
        pel[0] = 0.0;
        pel[tiles + 1] = totalE + tail;
    }
#pragma unroll
    for (int q = 0; q < PF_COMBINE_ITERS; ++q) {
        const int t = threadIdx.x * IT + q;
        if (q < IT && t < tiles) pel[t + 1] = exclE + inclE[q];
    }
    __syncthreads();
}

// The column's bookkeeping, run by one extra workgroup per column: moments of the current state (row `step` of
// filter_means / filter_variance), the log-likelihood increment of the previous step, the bases of this one.
// Reads the partials of state `step` only - nothing the step workgroups of the same launch write or wait for.
template <typename T, int D>
__device__ __forceinline__ void column_bookkeeping(const FusedArgs<T>& a, int b, double* red, double* redm) {
    const Geom& g = a.g;
    const int step = a.step;
    const bool obs = !a.finalize_only && a.is_obs();
    const bool apf = a.filter == PF_FILTER_APF;
    const bool two = apf && obs;
    const int64_t stride = (int64_t)g.B * g.tiles;
    const int64_t cb = (int64_t)b * g.tiles;
    // thread 0's scalar inputs are requested first: their latency overlaps the column sums
    ColStat st{};
    int poisoned = 0;
    double ll_tot = 0.0, piv[D];
    const int pslot = (step - 1) & 3;
    if (threadIdx.x == 0) {
        st = a.stat[b];
        if (step > 0) {
            poisoned = a.poison[pslot * g.B + b];
            ll_tot = (double)a.ll_total[b];
        }
#pragma unroll
        for (int d = 0; d < D; ++d) piv[d] = (double)a.template pivot<D>(step, b, d);  // the pivot the partials were taken about
    }
    double mv[2 * D];
    ColPartials<true, 2 * D> p1;
    ColPartials<false, 0> p2;
    load_col_partials<true, 2 * D>(a.part_r(), stride, cb, g.tiles, PQ_M1, PQ_S1, PQ_MX, p1);
    if (two) load_col_partials<false, 0>(a.part_r(), stride, cb, g.tiles, PQ_M2, PQ_S2, 0, p2);
    const ColSums c1 = column_sums<T, true, false, 2 * D>(p1, g.tiles, nullptr, nullptr, redm, red, mv);
    ColSums c2{0.0, 1.0, 0.0};
    if (two) c2 = column_sums<T, false, false>(p2, g.tiles, nullptr, nullptr, redm, red);
    const double lse_w = c1.M + log(c1.S);
    const double ess = c1.S * c1.S / c1.Q;
    bool resample = apf ? obs : (ess < a.thr_abs);  // apf.py:29-31 | sisr.py:18-19 (the step workgroups: the same test)
    if (a.finalize_only) resample = false;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) {  // moments of the current state -> row `step` of filter_means / filter_variance
            const double dm = mv[d] / c1.S;
            double var = mv[D + d] / c1.S - dm * dm;
            if (var < 0.0) var = 0.0;
            a.means[((int64_t)step * g.B + b) * D + d] = (T)(piv[d] + dm);
            a.vars[((int64_t)step * g.B + b) * D + d] = (T)var;
        }
        // log-likelihood increment of the previous step: ll = lse(logw') - base_lse   (0 for unweighted steps)
        if (step > 0 && !st.ll_done) {
            double ll = 0.0;
            if (st.prev_observed) {
                ll = lse_w - st.base_lse;
                if (poisoned) ll = __builtin_nan("");
            }
            a.poison[pslot * g.B + b] = 0;
            a.ll_steps[(int64_t)(step - 1) * g.B + b] = (T)ll;
            a.ll_total[b] = (T)(ll_tot + ll);
        }
        st.lse_w = lse_w;
        if (!a.finalize_only) st.resample = resample ? 1 : 0;
        st.ll_done = a.finalize_only ? 1 : 0;
        if (!a.finalize_only) {
            st.prev_observed = obs ? 1 : 0;
            // ll_t = [lse(w') - log N] + [lse(rw) - lse(w)] (apf.py:44) | lse(wi + log W), W = 1/N after resampling
            // else the carried weights (sisr.py:52-55)
            st.base_lse = apf ? a.logN - ((c2.M + log(c2.S)) - lse_w) : (resample ? a.logN : lse_w);
        }
        a.stat[b] = st;
    }
}

// The bookkeeper alone (grid (1, B)): flushes the moments / log-likelihood of a run's last state (finalize_only).
template <typename T, int D>
__global__ __launch_bounds__(PF_BLOCK) void k_fused_book(FusedArgs<T> a) {
    __shared__ double red[(2 + 2 * D) * PF_NWAVES];
    __shared__ double redm[PF_NWAVES];
    column_bookkeeping<T, D>(a, blockIdx.y, red, redm);
}

// MODE 0: systematic, ancestors from the inverted grid (grid_count);
// MODE 1: multinomial - the sorted positions are regenerated per round from Exp(1) spacings (Philox + workgroup scan),
//         ancestors by searching the staged window;
// MODE 2: systematic with the searching ancestor stage - float grids beyond 2^22 positions, where the closed form of
//         MODE 0 is not exact.  Compile-time so that no variant carries another's registers.  All three read the cdf as
//         tile-local scans + the prologue's table.
// PROP: the proposal as a compile-time constant (0 Bootstrap, 1 LinearGaussianObservations) or -1 = run-time switch.
// For D > 1 the optimal proposal's 3x3 inverse + Cholesky would otherwise set the register budget of Bootstrap runs too.
// FAST: the scalar closed-form path (ColConsts::fast) is known on the host - as a compile-time constant it removes the
// generic per-particle arithmetic (and its registers) from the fast instantiation and vice versa.
#define PF_WAVES_D1_GENERIC 4
// scalar-state multinomial kernels of multi-round tiles (the spacing scan's registers on top of the search's): at 4 waves
// they spill 130 - 250 B / lane; at 3 (<= 168 VGPRs) none - 60.6 -> 56.6 us per step at 2^22 x 1, 75.8 -> 60.1 at 64 x 65 536
// (profiles/r03_multinomial_scalar_waves.txt).  Single-round tiles stay at 4: 2^20 x 1 APF ran 20.4 -> 25.2 us at 3.
#define PF_WAVES_D1_MN 3
#define PF_WAVES_DN_A 3  // (4 waves = 128 VGPRs spill 128 B / lane once the prologue lives in this kernel: measured slower)
// resident waves per SIMD the register allocation is tuned for (float; measured per variant, tools/kbench.py)
template <typename T, int D, int MODE, int PROP, bool FAST, int SPEC, bool MULTI> struct StepWaves {
    static constexpr int value = sizeof(T) != 4 ? 1
                                 : (D == 1 && MODE == 1 && MULTI) ? PF_WAVES_D1_MN
                                 : D == 1     ? (FAST ? 4 : PF_WAVES_D1_GENERIC)
                                 : PROP == PF_PROP_LGO ? 2
                                 : (MODE == 1 || SPEC == 2) ? PF_WAVES_DN_A
                                 : 3;
};
// SPEC: the per-launch flags as compile-time constants for the two steady states of a run - known on the host when the
// launch is issued - so their branches, registers and dead paths (propagate-only move, tape loads, the other filter's
// weight update) leave the kernel: 0 generic (flags read at run time: NaN observations, the run's last step, tapes,
// device-side flags), 1 APF on an observed step followed by an observed step, 2 SISR on an observed step.
// LDS of one step workgroup (declared by the kernel, shared by the two bodies below)
template <typename T, int D, int VEC> struct StepShared {
    static constexpr int WIN = SearchWin<T, VEC>::WIN;
    static constexpr bool XWIN = (sizeof(T) * D <= 8);
    T* win;
    T* xwin;
    int* sh_j0;
    int* sh_cl;
    int* sh_wm;
    double* red;   // (4 + 2 D) * PF_NWAVES doubles
    double* reds;
    T* redm;
    double* ptl;   // PF_MAX_TILES + 2: the column's tile-prefix table (column_sums); the multinomial prologue first builds
                   // the prefixes of the Exp(1) spacing sums here (tiles + 2 entries) and takes what it needs from them
    double* ftl;   // PF_MAX_TILES
    int* sh_plan;  // 2: window start of the tile's first position, its tile
    double* crec;  // 2 * PF_LDS_CHUNKS: raw chunk records (m_c, t_c) of the scan this launch writes
};
// What the prologue hands to the body.
template <typename T> struct StepPlan {
    int j0, kt0;       // window start of the tile's first position (an index at or before its ancestor) and its tile
    T ub;              // systematic: the column's offset u
    double offE, invE; // multinomial: this tile's offset into the running spacing sum, 1 / the column's total
    bool resample;
    bool have_z0;      // the first round's standard normals were drawn in the prologue (under its load latency)
};
// Prologue of a step workgroup: the column's table from the partials of the incoming state, the resampling decision,
// the window start (wave 0; broadcast through LDS).  Everything here is L2-hot and tiny: <= 16 KB of partials, one
// 64-lane probe of the tile-local scans.
template <typename T, int D, int VEC, int MODE, int SPEC, bool EARLY_Z, bool MULTI>
__device__ __forceinline__ StepPlan<T> step_prologue(const FusedArgs<T>& a, const StepShared<T, D, VEC>& sh, T (&z0)[VEC][D]) {
    const Geom& g = a.g;
    const int b = PF_STEP_B, k = a.tile_of_row(PF_STEP_K);
    const int step = a.step;
    const bool obs = SPEC ? true : a.is_obs();
    const bool apf = SPEC ? (SPEC == 1) : (a.filter == PF_FILTER_APF);
    const bool two = apf && obs;
    StepPlan<T> pl{0, 0, T(0), 0.0, 0.0, false, false};
    PF_STAMP(a, 0);
#ifdef PF_DEVTOOLS
    if (a.debug_cut < 0 && threadIdx.x == 0 && b == 0 && (k % 128) == 0 && k / 128 < 8) a.dbg[16 + k / 128] = wall_clock64();
#endif
    if (apf && !obs) return pl;  // propagate-only move of the APF: identity ancestors, nothing to plan
    constexpr bool multinomial = MODE == 1;
    // issued first, used last: the Philox epoch / the step's systematic offset
    const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
    const T u_taped = (!multinomial && a.u_tape) ? a.u_tape[(int64_t)step * g.B + b] : T(0);
    const int64_t stride = (int64_t)g.B * g.tiles;
    const int64_t cb = (int64_t)b * g.tiles;
    double* const redm_d = sh.red + 2 * PF_NWAVES;
    T e0[1] = {T(0)};
    if constexpr (multinomial) {
        // the resampling positions are the order statistics of N iid uniforms, built from normalised Exp(1) spacings
        // (Philox, regenerated wherever needed); their per-tile sums were reduced with the partials, so the first position
        // of every tile follows from a second prefix table - built in the LDS of the weights' table, before it
        T tail[1];
        draw_exponentials<T, 1>(seed, PF_STREAM_MULTINOMIAL, (uint32_t)step, (uint64_t)((int64_t)g.B * g.N + b), tail);
        spacing_table(a.part_r(), stride, cb, g.tiles, (double)tail[0], sh.ptl, sh.reds);
        pl.offE = sh.ptl[k];
        pl.invE = 1.0 / sh.ptl[g.tiles + 1];
        draw_exponentials<T, 1>(seed, PF_STREAM_MULTINOMIAL, (uint32_t)step, (uint64_t)((int64_t)b * g.N + (int64_t)k * g.tile_elems), e0);
    }
    // the column's partials are requested ...
    ColPartials<true, 0> p1;
    ColPartials<false, 0> p2;
    // Both shortcuts below exist in the MULTI-round instantiations only (1024 x 8192, 64 x 65 536, ...): compiled into the
    // single-round kernels as well they cost the headline shape (2^20 x 1: 1024 tiles, neither branch taken) 0.3 us per
    // step - 14.55 -> 14.9, same box, three alternations (profiles/r03_prologue_shortcuts_single_round_ab.txt) - through
    // nothing but code layout / scheduling; branch hints and out-of-line helpers did not recover it.
    constexpr bool SHORT = MULTI;
    const bool one_tile = SHORT && g.tiles == 1 && a.debug_cut != 77;  // (uniform; PF_DEBUG_CUT=77: the general path - A/B tests)
    double m1 = 0.0, s1 = 0.0, q1 = 0.0;
    if (one_tile) {
        const double* part = a.part_r();
        m1 = part[(two ? PQ_M2 : PQ_M1) * stride + cb];
        s1 = part[(two ? PQ_S2 : PQ_S1) * stride + cb];
        q1 = two ? 0.0 : part[PQ_Q1 * stride + cb];
    }
    const bool few_tiles = SHORT && !one_tile && g.tiles <= PF_WAVE && a.debug_cut != 79;  // (uniform; 79: the general path)
    if (!one_tile && !few_tiles) {
        if (two) load_col_partials<false, 0>(a.part_r(), stride, cb, g.tiles, PQ_M2, PQ_S2, 0, p2);
        else load_col_partials<true, 0>(a.part_r(), stride, cb, g.tiles, PQ_M1, PQ_S1, 0, p1);
    }
    // ... and, while they travel (written by other CUs one launch ago: an Infinity-Cache round trip), the first round's
    // standard normals are drawn: ~150 VALU instructions per thread that depend on nothing but the thread's index
    // (EARLY_Z: only the kernels with registers to spare - holding the draws across the prologue costs the D = 3 and the
    // generic scalar kernels more in spills than the overlap returns: measured)
    if (EARLY_Z && (SPEC || !a.z_tape)) {
        const int64_t i0 = (int64_t)k * g.tile_elems + threadIdx.x * VEC;
        if (i0 < g.N) draw_normals<T, D, VEC>(seed, PF_STREAM_NORMAL, (uint32_t)step, (uint64_t)((int64_t)b * g.N + i0), z0);
        pl.have_z0 = true;
    }
    if (one_tile) {
        if (two) {
            column_sums_one_tile<T, false>(m1, s1, q1, sh.ptl, sh.ftl);
            pl.resample = true;
        } else {
            const ColSums c = column_sums_one_tile<T, true>(m1, s1, q1, sh.ptl, sh.ftl);
            pl.resample = c.S * c.S / c.Q < a.thr_abs;
        }
    } else if (few_tiles) {
        if (two) {
            column_sums_wave<T, false>(a.part_r(), stride, cb, g.tiles, PQ_M2, PQ_S2, sh.ptl, sh.ftl);
            pl.resample = true;
        } else {
            const ColSums c = column_sums_wave<T, true>(a.part_r(), stride, cb, g.tiles, PQ_M1, PQ_S1, sh.ptl, sh.ftl);
            pl.resample = c.S * c.S / c.Q < a.thr_abs;
        }
    } else if (two) {
        column_sums<T, false, true>(p2, g.tiles, sh.ptl, sh.ftl, redm_d, sh.red);
        pl.resample = true;  // apf.py:29-31
    } else {
        const ColSums c = column_sums<T, true, true>(p1, g.tiles, sh.ptl, sh.ftl, redm_d, sh.red);
        pl.resample = c.S * c.S / c.Q < a.thr_abs;  // sisr.py:18-19 (the bookkeeper: the same test)
    }
    PF_STAMP(a, 1);
    if (!pl.resample) return pl;  // (uniform)
    const double* ptl = sh.ptl;
    T p;
    if constexpr (multinomial) {
        p = (T)((pl.offE + (double)e0[0]) * pl.invE);
    } else {
        pl.ub = a.u_tape ? u_taped : uniform_draw<T>(seed, PF_STREAM_UNIFORM, (uint32_t)step, (uint64_t)b);
        p = grid_position<T>((int64_t)k * g.tile_elems, pl.ub, T(g.N));
    }
    if (one_tile && a.debug_cut != 78) {
        // ONE tile: its first position is the column's first - index 0 is "at or before its ancestor" by definition, so the
        // window needs no probing (a dependent round of loads on every workgroup's critical path).  Should the first
        // ancestor lie beyond the first window - the column's weight concentrated at its end - the body walks on window by
        // window, as it does for any stretch of negligible weights.
        pl.j0 = 0;
        pl.kt0 = 0;
        return pl;
    }
    if (threadIdx.x < PF_WAVE) {
        const int lane = threadIdx.x;
        // the tile holding the ancestor: first kt whose end value T(P_{kt+1}) (1 for the last tile) is >= p
        // (64-ary over the LDS table: two probe rounds for 1024 tiles instead of ten dependent reads)
        int lo = 0, hi = g.tiles - 1;
        while (lo < hi) {
            const int len = hi - lo + 1;
            const int st = (len + PF_WAVE - 1) / PF_WAVE;
            int probe = lo + (lane + 1) * st - 1;
            if (probe > hi) probe = hi;
            const bool ge = (probe >= g.tiles - 1) || !((T)ptl[probe + 1] < p);
            const unsigned long long bal = __ballot(ge);
            const int f = bal ? __ffsll((long long)bal) - 1 : PF_WAVE - 1;
            int nhi = lo + (f + 1) * st - 1;
            if (nhi > hi) nhi = hi;
            lo = lo + f * st;
            if (lo > nhi) lo = nhi;
            hi = nhi;
        }
        const int kt = lo;
        const double Pk = ptl[kt], Pn = ptl[kt + 1], fk = sh.ftl[kt];
        const int64_t first = (int64_t)kt * g.tile_elems;
        const int64_t last = (first + g.tile_elems < g.N ? first + g.tile_elems : g.N) - 1;
        const T* l_col = ((step & 1) ? a.pos : a.cdf) + (int64_t)b * g.N;
        // Window start inside tile kt: ONE 64-ary probe round.  The body does not need the exact ancestor of the tile's
        // first position, only an index at or before it whose predecessor lies below p (entries ahead of the ancestor own
        // no position and cost nothing but window capacity) - so the bracket the first round yields (<= tile / 64 entries
        // wide) is enough, and a second, dependent probe round (a further memory latency) is not spent.
        const int64_t len = last + 1 - first;
        auto cdf_at = [&](int64_t q) -> T {
            if constexpr (MULTI) {
                const double* cg = a.ctab_r(b) + 2 * (q / (PF_WAVE * VEC));
                return cdf_from_local<T>(l_col[q], cg[0], cg[1], Pk, fk, Pn, q == last, q == g.N - 1);
            } else {
                return cdf_from_local<T>(l_col[q], Pk, fk, Pn, q == last, q == g.N - 1);
            }
        };
        // The 64 probes sit PF_PROBE_STEP entries apart around where p falls if the tile's weight were spread evenly (the
        // cumulative sum of thousands of weights strays from the straight line by a few dozen entries): 16 cache lines the
        // window load is about to read anyway, and a start known to 8 entries - instead of 64 lines spread over the whole
        // tile (at 2^22 x 1 a quarter of the kernel's algorithmic reads) and a start known to tile / 64.
        int64_t res;
        bool found;
        {
            float fr = (float)((double)p - Pk) * __builtin_amdgcn_rcpf((float)(Pn - Pk));
            fr = __builtin_fminf(__builtin_fmaxf(fr, 0.0f), 1.0f);  // (NaN -> 0)
            int64_t g0 = first + (int64_t)(fr * (float)len) - (PF_WAVE / 2) * PF_PROBE_STEP;
            const int64_t g0max = last - (PF_WAVE - 1) * PF_PROBE_STEP;
            if (g0 > g0max) g0 = g0max;
            if (g0 < first) g0 = first;
            int64_t q = g0 + (int64_t)lane * PF_PROBE_STEP;
            if (q > last) q = last;
            const unsigned long long bal = __ballot(cdf_at(q) >= p);
            const int f = bal ? __ffsll((long long)bal) - 1 : 0;
            found = bal != 0 && (f > 0 || g0 == first);  // a probe below p precedes the first one at or above it
            res = f > 0 ? g0 + (int64_t)(f - 1) * PF_PROBE_STEP + 1 : first;
        }
        if (!found) {  // (uniform) the tile's weight is concentrated: one probe round over the whole tile
            const int64_t st = (len + PF_WAVE - 1) / PF_WAVE;
            int64_t probe = first + (lane + 1) * st - 1;
            if (probe > last) probe = last;
            const unsigned long long bal = __ballot(cdf_at(probe) >= p);
            const int f = bal ? __ffsll((long long)bal) - 1 : PF_WAVE - 1;  // (cdf(last) >= p by the choice of kt)
            res = first + (int64_t)f * st;
            if (res > last) res = last;
        }
        if (lane == 0) {
            sh.sh_plan[0] = (int)res;
            sh.sh_plan[1] = kt;
        }
    }
    __syncthreads();
    pl.j0 = sh.sh_plan[0];
    pl.kt0 = sh.sh_plan[1];
    PF_STAMP(a, 2);
    return pl;
}
// RS: whether this column resamples in this step - a run-time fact for SISR (the bookkeeper's ESS test), so the kernel
// holds both bodies and branches once, uniformly, at its top: the resampling body carries no carried-weights path (old
// log-weights, direct state loads), the other one no ancestor stage.
// MK: the model kinds as compile-time constants where the per-particle arithmetic switches on them.  Generic kernels: 0
// run-time kinds, 1 Verhulst diffusion + stochastic-volatility observation (D = 1) - the switch statements fold, their
// scalar branch instructions (one set per particle and density) vanish (SQ_INSTS_SALU 1798 -> 962 per wave on
// 64 x 65 536), 3 a user-defined affine process (PF_HID_USER_AFFINE: the one-step mean / scale of the PARENT come from the
// caller's planes).  Closed-form kernels (FAST): the shape of the one-step mean, 0 run time, 1 affine, 2 sine.
template <typename T, int D, int VEC, int MODE, int PROP, bool FAST, int SPEC, int MK, bool RS, bool MULTI>
__device__ __forceinline__ void step_body(const FusedArgs<T>& a, const StepShared<T, D, VEC>& sh, const StepPlan<T>& pl,
                                          const T (&z0)[VEC][D]) {
    const int proposal = (PROP >= 0) ? PROP : a.proposal;
    ModelDesc md = a.md;
    if constexpr (!FAST && MK == 1) { md.hid_kind = PF_HID_VERHULST_EM; md.obs_kind = PF_OBS_SV; md.obs_dim = 1; }
    if constexpr (!FAST && MK == 4) { md.hid_kind = PF_HID_LORENZ63_EM; md.obs_kind = PF_OBS_LINEAR; }  // (D = 3)
    constexpr bool USER = !FAST && MK == 3;  // PF_HID_USER_AFFINE (one step per run: no next step is prepared here)
    constexpr int WIN = StepShared<T, D, VEC>::WIN;
    // the window of cdf entries a round stages: 256 * (VEC + V1).  The inverted-grid variant reads 256 entries beyond one
    // per position (its window starts within tile / 64 of the first ancestor; a stretch of negligible weights walks on
    // window by window); the searching variants stage two per position and binary-search them in LDS
    constexpr int V1 = (MODE == 0) ? 1 : VEC;
    constexpr int STAGED = PF_BLOCK * (VEC + V1);
    // the particles behind the cdf window are staged in LDS too when they are small (<= 8 B per particle), so the
    // ancestor gather is an LDS read instead of a second dependent global round trip
    constexpr bool XWIN = StepShared<T, D, VEC>::XWIN;
    T* const win = sh.win;
    T* const xwin = sh.xwin;
    int& sh_j0 = *sh.sh_j0;
    int* const sh_cl = sh.sh_cl;
    int* const sh_wm = sh.sh_wm;
    double* const red = sh.red;
    double* const reds = sh.reds;
    T* const redm = sh.redm;
    int* hd = reinterpret_cast<int*>(win);  // systematic route: heads of the offspring ranges (the cdf window is not staged)
    const Geom& g = a.g;
    const int b = PF_STEP_B, k = a.tile_of_row(PF_STEP_K);
    const int tid = threadIdx.x;
    const int step = a.step;
    const int slot = step & 1;
    const bool obs = SPEC ? true : a.is_obs();
    const bool apf = SPEC ? (SPEC == 1) : (a.filter == PF_FILTER_APF);
    constexpr bool resample = RS;
    constexpr bool multinomial = MODE == 1;
    const bool windowed = resample;  // both resamplers search an LDS window of the cdf (their positions are sorted)
    const bool pre_next = SPEC ? (SPEC == 1) : (a.is_obs_next() && apf);
    const int N = (int)g.N;
    PF_STAMP(a, 8);
    if (PF_CUT(a, 1)) return;

    // the window start of the tile's first position and its tile (prologue; no integer division here)
    int j0 = pl.j0;
    int kt0 = pl.kt0;
    // multinomial: this tile's offset into the running sum of the Exp(1) spacings and the reciprocal of their total
    const double offE = pl.offE, invE = pl.invE;
    double carryE = 0.0;
    const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
    const T* z_step = (!SPEC && a.z_tape) ? a.z_tape + (int64_t)step * D * g.B * g.N : nullptr;

    // FAST: the column's closed-form constants come from the record the bookkeeper wrote (FastCol, loaded right before
    // they are used); otherwise the parameter rows are loaded and reduced here
    ColParams<T, D> cp;
    ColConsts<T, D> cc;
    FastCol<T> fc;
    if constexpr (!FAST && D > 1) {
        load_col_params<T, D>(a, b, step, obs, cp);
        if (pre_next) cp.load_next(a.y + ((int64_t)(step + 1) * a.y_rows + (a.y_rows == 1 ? 0 : b)) * a.md.obs_dim);
    }
    const T ub = pl.ub;

    const T* x_in = a.x[slot];
    const T* lw_in = a.logw[slot] + (int64_t)b * g.N;
#ifdef PF_DEVTOOLS
    T* lw_out = a.logw[slot ^ 1] + (int64_t)b * g.N;
    int32_t* anc_col = a.anc + (int64_t)b * g.N;
#endif
    const T* cdf_col = ((step & 1) ? a.pos : a.cdf) + (int64_t)b * g.N;  // the local scans of this step's parity
    const int64_t base = (int64_t)k * g.tile_elems;
    const T nT = T(N);
    const T rcN = a.rcN;
    const bool pow2 = (N & (N - 1)) == 0;                  // then the grid division is an exact multiplication

    bool poison = false;
    PartialAcc<T, D, SPEC != 1> acc;  // (SPEC = 1: APF steady state - the weights' ESS is never looked at)
    acc.init();
    PF_STAMP(a, 9);
    const double* ptab_col = sh.ptl;  // `cdf` / `pos` hold tile-local scans; the cdf is implied by the prologue's table (LDS)
    const double* ftab_col = sh.ftl;
    CdfView<T> view;
    view.L = cdf_col;
    view.ptab = ptab_col;
    view.ftab = ftab_col;
    const double* const ct_col = MULTI ? a.ctab_r(b) : nullptr;  // per-chunk (C, g) of the scans this step reads
    view.ct = ct_col;
    view.chunk_elems = PF_WAVE * VEC;
    view.N = N;
    view.tile_elems = g.tile_elems;
    view.tiles = g.tiles;

    T e_rw[VEC];
    T rwn[VEC];   // this round's next resampling weights (-inf beyond the column)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        e_rw[j] = T(0);
        rwn[j] = -Lim<T>::inf();
    }
    const bool scan_next = !apf || pre_next;  // a next resampling exists: its weights are scanned by this kernel
    // single-round tiles: the tile maximum is known before anything is stored, so the scan is tile-level (one workgroup scan
    // of the exponentials the partial accumulator already holds) and every chunk's (C, g) is (0, 1); multi-round tiles
    // scan chunk by chunk inside the loop (re-reading the tile once its maximum is known would double the kernel's reads)
    constexpr bool single = !MULTI;
    const bool use_lds = g.rounds_per_tile * PF_NWAVES <= PF_LDS_CHUNKS;
    double* const ct_tile = a.ctab_w(step + 1, b) + 2 * (int64_t)k * g.rounds_per_tile * PF_NWAVES;
    T* const l_next = ((step & 1) ? a.cdf : a.pos) + (int64_t)b * g.N;  // the scans of state step + 1
    int rk = 0;
    for (int r = 0; r < g.rounds_per_tile; ++r) {
        const int64_t r0 = base + (int64_t)r * g.round_elems;
        if (r0 >= g.N) break;
        const int64_t i0 = r0 + tid * VEC;
        const bool on = i0 < g.N;
        if (MODE == 0 && windowed) {  // no head anywhere yet (ordered against the scatter by the barrier below)
            int zero[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) zero[j] = 0;
            if (VEC == 1) hd[tid] = 0; else store_vec<int, VEC>(hd + tid * VEC, zero);
        }
        T pv[VEC];
        if (multinomial && windowed) {
            // this round's sorted resampling positions: the Exp(1) spacings regenerated (the previous step's epilogue summed
            // exactly these draws into the tile partial), scanned across the workgroup, normalised by the column total
            T ev[VEC];
            double incl[VEC], local = 0.0, total;
            if (on) draw_exponentials<T, VEC>(seed, PF_STREAM_MULTINOMIAL, (uint32_t)step, (uint64_t)((int64_t)b * g.N + i0), ev);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                local += on ? (double)ev[j] : 0.0;
                incl[j] = local;
            }
            const double excl = block_scan_excl(local, reds, total);
#pragma unroll
            for (int j = 0; j < VEC; ++j) pv[j] = (T)((offE + carryE + excl + incl[j]) * invE);
            carryE += total;
        }

        // ---- 1. issue the window loads (cdf, and the particles behind it) ------------------------------------------
        // thread t stages entries ws + t * VEC + j (c0) and ws + 256 * VEC + t * V1 + j (c1): see inverse_grid_round
        const int ws = j0 - (j0 % VEC);
        T c0[VEC], c1[V1], xa[D][VEC], xb[D][V1];
        const int ja = ws + tid * VEC, jb = ws + PF_BLOCK * VEC + tid * V1;
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
            if (V1 == 1) c1[0] = cdf_col[jb]; else load_vec<T, V1>(cdf_col + jb, c1);
            if (XWIN) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const T* xc = x_in + ((int64_t)d * g.B + b) * g.N + jb;
                    if (V1 == 1) xb[d][0] = xc[0]; else load_vec<T, V1>(xc, xb[d]);
                }
            }
        }
        T lw_old[VEC];
        if (on && !resample) {
            if (VEC == 1) lw_old[0] = lw_in[i0]; else load_vec<T, VEC>(lw_in + i0, lw_old);
        }

        // ---- 2. work that does not need the window, while those loads are in flight --------------------------------
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
            } else if (r == 0 && pl.have_z0) {
#pragma unroll
                for (int j = 0; j < VEC; ++j)
#pragma unroll
                    for (int d = 0; d < D; ++d) zt[j][d] = z0[j][d];
            } else {
                draw_normals<T, D, VEC>(seed, PF_STREAM_NORMAL, (uint32_t)step, (uint64_t)((int64_t)b * g.N + i0), zt);
            }
        }
        PF_STAMP(a, 10);
#ifdef PF_DEVTOOLS
        if (PF_CUT(a, 4)) {  // development: keep the draws alive, skip the rest of the ROUND (every round runs: the cuts
            if (on) lw_out[i0] = zt[0][0] + zt[VEC - 1][D - 1];  // nest for multi-round tiles too)
            continue;
        }
#endif

        // ---- 3. ancestors ---------------------------------------------------------------------------------------------
        int idx[VEC];
        if (windowed) {
            // local scans -> cdf values of one staged window [w0, w0 + WIN) (a staged vector never straddles a tile:
            // tile % VEC == 0).  kt0 is the tile of the window start (from the prologue, advanced as windows move
            // on); a window spans at most a few tiles, so compares replace the integer divisions.  Indices inside a column
            // fit 32 bits (N <= 2^30).
            auto map_window = [&](int w0, int wja, int wjb, bool wina, bool winb, T (&m0)[VEC], T (&m1)[V1]) {
                const int te = g.tile_elems;
                while ((unsigned)((kt0 + 1) * te) <= (unsigned)w0) ++kt0;  // uniform; only ever advances
                int kta = kt0, ktb = kt0;
                {
                    const unsigned e1 = (unsigned)(kt0 + 1) * te, e2 = e1 + te, e3 = e2 + te;
                    kta += ((unsigned)wja >= e1) + ((unsigned)wja >= e2) + ((unsigned)wja >= e3);
                    ktb += ((unsigned)wjb >= e1) + ((unsigned)wjb >= e2) + ((unsigned)wjb >= e3);
                    if (!wina) kta = kt0;
                    if (!winb) ktb = kt0;
                }
                const double tPa = ptab_col[kta], tNa = ptab_col[kta + 1], tFa = ftab_col[kta];
                const double tPb = ptab_col[ktb], tNb = ptab_col[ktb + 1], tFb = ftab_col[ktb];
                const unsigned ea = (unsigned)(kta + 1) * te, eb = (unsigned)(ktb + 1) * te;
                const int la = (int)(ea < (unsigned)N ? ea : (unsigned)N) - 1;
                const int lb = (int)(eb < (unsigned)N ? eb : (unsigned)N) - 1;
                if constexpr (MULTI) {  // (a staged vector lies inside one chunk: chunk boundaries are multiples of 64 VEC)
                    const double* cga = ct_col + 2 * ((wina ? wja : 0) / (PF_WAVE * VEC));
                    const double* cgb = ct_col + 2 * ((winb ? wjb : 0) / (PF_WAVE * VEC));
                    const double Ca = cga[0], ga = cga[1], Cb = cgb[0], gb = cgb[1];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {  // (only a staged vector's LAST element can be a tile's last: see below)
                        const bool tl = (j == VEC - 1) && (wja + j == la);
                        m0[j] = wina ? cdf_from_local<T>(m0[j], Ca, ga, tPa, tFa, tNa, tl, tl && (wja + j == N - 1)) : Lim<T>::inf();
                    }
#pragma unroll
                    for (int j = 0; j < V1; ++j)
                        m1[j] = winb ? cdf_from_local<T>(m1[j], Cb, gb, tPb, tFb, tNb, wjb + j == lb, wjb + j == N - 1) : Lim<T>::inf();
                } else {
                    // a tile's (and the column's) last element is the LAST element of its staged vector (tile_elems % VEC == 0,
                    // N % VEC == 0 for VEC > 1): the pinned value T(P_{k+1}) / 1 is selected for that one element only
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const bool tl = (j == VEC - 1) && (wja + j == la);
                        m0[j] = wina ? cdf_from_local<T>(m0[j], tPa, tFa, tNa, tl, tl && (wja + j == N - 1)) : Lim<T>::inf();
                    }
#pragma unroll
                    for (int j = 0; j < V1; ++j)
                        m1[j] = winb ? cdf_from_local<T>(m1[j], tPb, tFb, tNb, wjb + j == lb, wjb + j == N - 1) : Lim<T>::inf();
                }
            };
            map_window(ws, ja, jb, ina, inb, c0, c1);
            if (XWIN) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if (ina) { if (VEC == 1) xwin[d * WIN + tid] = xa[d][0]; else store_vec<T, VEC>(xwin + d * WIN + tid * VEC, xa[d]); }
                    if (inb) { if (V1 == 1) xwin[d * WIN + PF_BLOCK * VEC + tid] = xb[d][0]; else store_vec<T, V1>(xwin + d * WIN + PF_BLOCK * VEC + tid * V1, xb[d]); }
                }
            }
            if constexpr (MODE != 0) {
                if (VEC == 1) { win[tid] = c0[0]; win[PF_BLOCK + tid] = c1[0]; }
                else { store_vec<T, VEC>(win + tid * VEC, c0); store_vec<T, VEC>(win + (PF_BLOCK + tid) * VEC, c1); }
                __syncthreads();
                T pp[VEC];
                int qq[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) pp[j] = multinomial ? pv[j] : grid_position<T>(i0 + j, ub, nT);
                window_lower_bound_flat<T, WIN, VEC>(win, pp, qq);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int64_t i = i0 + j;
                    int res = N - 1;
                    if (i < N) {
                        res = (qq[j] < WIN) ? ws + qq[j] : view.lower_bound(ws + WIN < N ? ws + WIN : N, pp[j], kt0);
                        if (res > N - 1) res = N - 1;
                    }
                    idx[j] = res;
                }
            } else {
                // Systematic grid, inverted: no search (inverse_grid_round); positions the window does not reach take a
                // binary search in the implied cdf
                inverse_grid_round<T, VEC, V1>(c0, c1, ws, (int)r0, g.round_elems, N, ub, nT, rcN, pow2, i0, hd, sh_cl, sh_wm,
                                           [&](int it, T (&d0)[VEC], T (&d1)[V1]) -> bool {  // the window after the staged one(s)
                                               const int w0 = ws + it * STAGED;
                                               if (w0 >= N) return false;
                                               const int wja = w0 + tid * VEC, wjb = w0 + PF_BLOCK * VEC + tid * V1;
                                               const bool wina = wja < N, winb = wjb < N;
                                               if (wina) { if (VEC == 1) d0[0] = cdf_col[wja]; else load_vec<T, VEC>(cdf_col + wja, d0); }
                                               if (winb) { if (V1 == 1) d1[0] = cdf_col[wjb]; else load_vec<T, V1>(cdf_col + wjb, d1); }
                                               map_window(w0, wja, wjb, wina, winb, d0, d1);
                                               return true;
                                           },
                                           [&](int64_t i, int from) { return view.lower_bound(from, grid_position<T>(i, ub, nT), from ? kt0 : 0); },
                                           idx);
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
#ifdef PF_DEVTOOLS
        if (on && PF_CUT(a, 2)) {
            if (VEC == 1) anc_col[i0] = idx[0]; else store_vec<int, VEC>(anc_col + i0, idx);
        }
#endif

        // ---- 4. gather, propagate, weight -------------------------------------------------------------------------------
        // the column's constants are fetched only now - their address is tied to the ancestors by an opaque zero - so
        // they occupy registers for the arithmetic below and not across the search
        int late;
        asm volatile("v_readfirstlane_b32 %0, %1\n\ts_and_b32 %0, %0, 0" : "=s"(late) : "v"(idx[0]) : "scc");
        // the same for the output side of the argument block: re-read from the kernarg segment behind the opaque zero, so
        // the destination pointers are not held in (spilled) SGPRs through the prologue and the ancestor stage
        using KArgs = const __attribute__((address_space(4))) FusedArgs<T>;
        KArgs* la = (KArgs*)((const_ptr<char>)__builtin_amdgcn_kernarg_segment_ptr() + late);
        T* const x_out = la->x[slot ^ 1];
#ifndef PF_DEVTOOLS
        T* const lw_out = la->logw[slot ^ 1] + (int64_t)b * g.N;
        int32_t* const anc_col = la->anc + (int64_t)b * g.N;
#endif
        if constexpr (FAST) {
            // scalar loads of the run's record (k_fused_reduce wrote it) and of this / the next step's observation
            const_ptr<T> yq = (const_ptr<T>)(uintptr_t)a.y + late;
            const int yc = a.y_rows == 1 ? 0 : b;
            const T y_t = obs ? yq[(int64_t)step * a.y_rows + yc] : T(0);
            const T y_n = pre_next ? yq[(int64_t)(step + 1) * a.y_rows + yc] : T(0);
            fc.load((const_ptr<T>)(uintptr_t)(a.cpack + (int64_t)b * PK_N) + late, y_t, y_n);
        } else if constexpr (D == 1) {
            load_col_params<T, D>(a, b, step, obs, cp, late);
            if (pre_next) cp.load_next(a.y + ((int64_t)(step + 1) * a.y_rows + (a.y_rows == 1 ? 0 : b)) * a.md.obs_dim + late);
            cc.prepare(md, cp);
            if constexpr (USER) {
                // one transition scale per column (pf_filter_args.user_scale_per_column): the optimal proposal / the weights in
                // closed form around the caller's one-step mean (ColConsts::prepare_user) - the generic arithmetic costs a user-defined
                // model two reciprocals, a square root and two logarithms per PARTICLE
                if (la->user_scale_percol) cc.prepare_user(md, cp, la->user_scale[b]);
            } else {
                __builtin_assume(cc.fast == FAST);
            }
        } else if (r == 0) {  // D > 1: the rows were loaded up front (register room, and no exposed latency here)
            cc.prepare(md, cp);
            __builtin_assume(cc.fast == FAST);
        }
        T xo[D][VEC];
        if (on && !PF_CUT(a, 2)) {
            T lwo[VEC], pre_n[VEC];  // this round's new log-weights / first-stage weights of the next step
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
                    const bool in_lds = XWIN && windowed && q >= 0 && q < STAGED;
#pragma unroll
                    for (int d = 0; d < D; ++d)
                        xr[j][d] = in_lds ? xwin[d * WIN + q] : x_in[((int64_t)d * g.B + b) * g.N + idx[j]];
                }
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                T xn[D];
                T w_new;
                // user-defined affine models (MK = 3): the parent's one-step mean / scale from the caller's planes
                UserMS<T, D> um = UserMS<T, D>::none();
                if constexpr (USER) {
                    um.gather(la->user_loc, la->user_scale, (int64_t)b * g.N, (int64_t)g.B * g.N, idx[j], la->user_scale_percol != 0, b, g.B);
                    um.euler(xr[j], la->user_dt);
                }
                if (obs) {
                    T wi, pre_anc = T(0);
                    if constexpr (FAST) {
                        if (apf) wi = fc.template sample_and_weight_apf<FAST ? MK : 0>(proposal, xr[j][0], zt[j][0], xn[0], pre_anc);
                        else wi = fc.template sample_and_weight<FAST ? MK : 0>(proposal, xr[j][0], zt[j][0], xn[0]);
                    } else {
                        wi = sample_and_weight<T, D>(md, proposal, cp, cc, xr[j], zt[j], xn, um);
                    }
                    if (apf) {
                        // second-stage weight ws - pre_weight(x[anc]) (apf.py:43), the pre-weight recomputed in registers
                        if constexpr (FAST) w_new = wi - pre_anc;
                        else w_new = wi - pre_weight<T, D>(md, proposal, cp, cc, xr[j], false, um);
                        if (is_nan_or_posinf(w_new)) poison = true;
                    } else {
                        if (is_nan_or_posinf(wi)) poison = true;
                        w_new = resample ? wi : (wi + lw_old[j]);
                    }
                } else {
                    // propagate only (NaN observation / unobserved sub-step): weights carried, ll = 0 (state.py:38-42)
                    if constexpr (FAST) fc.template sample_and_weight<FAST ? MK : 0>(PF_PROP_BOOTSTRAP, xr[j][0], zt[j][0], xn[0]);
                    else sample_and_weight<T, D>(md, PF_PROP_BOOTSTRAP, cp, cc, xr[j], zt[j], xn, um);
                    w_new = resample ? T(0) : lw_old[j];
                }
                lwo[j] = sanitize_logw(w_new);
#pragma unroll
                for (int d = 0; d < D; ++d) xo[d][j] = xn[d];
                // first-stage weight of the next step, while the new particle is still in registers
                if constexpr (FAST) pre_n[j] = pre_next ? fc.template pre_weight<FAST ? MK : 0>(proposal, xn[0], true) : T(0);
                else if constexpr (USER) {
                    // (pf_run_hints.prepare_next: the optimal proposal with one transition scale per column - the new particle
                    // and the column's scale are all its first-stage weight reads; `um.scale` IS the column's scale)
                    pre_n[j] = pre_next ? pre_weight<T, D>(md, proposal, cp, cc, xn, true, um) : T(0);
                } else pre_n[j] = pre_next ? pre_weight<T, D>(md, proposal, cp, cc, xn, true) : T(0);
                // keep the scheduler from interleaving all VEC particles' arithmetic: that is what pushes the kernel
                // over its register budget (spills cost real HBM traffic: PMC WRITE_SIZE)
                if (j & 1) __builtin_amdgcn_sched_barrier(0);
            }
            PF_STAMP(a, 12);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                T* xc = x_out + ((int64_t)d * g.B + b) * g.N + r0;  // (the round's first particle: uniform)
                if (VEC == 1) xc[tid] = xo[d][0]; else store_out<T, VEC, !MULTI>(xc, tid * VEC, xo[d]);
            }
            if (VEC == 1) lw_out[i0] = lwo[0]; else store_out<T, VEC, !MULTI>(lw_out + r0, tid * VEC, lwo);
            if (resample || apf) {  // SISR without resampling keeps the previous ancestors (sisr.py:25-26)
                if (VEC == 1) anc_col[i0] = idx[0]; else store_out<int, VEC, !MULTI>(anc_col + r0, tid * VEC, idx);
            } else if (la->anc_prev) {  // ... which, with a state history, means copying them into this state's slot
                const int32_t* ap = la->anc_prev + (int64_t)b * g.N + i0;
                int prev[VEC];
                if (VEC == 1) prev[0] = ap[0]; else load_vec<int, VEC>(ap, prev);
                if (VEC == 1) anc_col[i0] = prev[0]; else store_vec<int, VEC>(anc_col + i0, prev);
            }
            if (PF_CUT(a, 5)) {
                if (pre_next) { if (VEC == 1) lw_out[i0] = pre_n[0]; else store_vec<T, VEC>(lw_out + i0, pre_n); }
                continue;
            }
            T piv[D];  // pivot of the moments of state step + 1: the mean of state step - 1, or the run's record (FusedArgs::pivot)
#pragma unroll
            for (int d = 0; d < D; ++d)
                piv[d] = (step - 1 >= a.t0) ? la->means[((int64_t)(step - 1) * g.B + b) * D + d] : (T)la->piv0[(int64_t)b * PF_MAXD + d];
            // the next resampling weights (scanned chunk by chunk below); the partial accumulator keeps the weights' own
            // family (moments, ll, ESS)
            if constexpr (MULTI) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    if (pre_next && is_nan_or_posinf(pre_n[j])) acc.poison = true;
                    rwn[j] = pre_next ? sanitize_logw(pre_n[j] + lwo[j]) : lwo[j];
                }
            }
            acc.template push_round<VEC>(lwo, xo, single && pre_next, pre_n, piv, e_rw);
            if (multinomial) {
                T ev[VEC];
                draw_exponentials<T, VEC>(seed, PF_STREAM_MULTINOMIAL, (uint32_t)(step + 1), (uint64_t)((int64_t)b * g.N + i0), ev);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc.es += (double)ev[j];
            }
            PF_STAMP(a, 13);
        }
        if (scan_next && !single) {  // (uniform) every lane of every wave: the chunk-local scan is wave-level
            chunk_scan_round<T, VEC, false>(rwn, on, l_next + r0, tid * VEC, r * PF_NWAVES + (tid >> 6), use_lds ? sh.crec : nullptr, ct_tile);
#pragma unroll
            for (int j = 0; j < VEC; ++j) rwn[j] = -Lim<T>::inf();
            ++rk;
        }
        if (windowed && r + 1 < g.rounds_per_tile) __syncthreads();  // the window is rewritten by the next round
    }
    if (poison) atomicOr(&a.poison[(step & 3) * g.B + b], 1);
    if (PF_CUT(a, 3)) return;
    PF_STAMP(a, 14);
    T M1, M2, F1, F2;
    acc.template finish<MODE == 1>(a.part_w(step + 1), b, k, g.B, g.tiles, single && pre_next, red, redm, &a.poison[((step + 1) & 3) * g.B + b], M1, M2, F1, F2);
    if (PF_CUT(a, 6)) return;
    if (scan_next && single) {
        // exp(rw - thread max) and the thread's factor exp(thread max - tile max) are in registers (push_round / finish)
        const int64_t i0 = base + tid * VEC;
        const bool on = i0 < g.N;
        const T f = pre_next ? F2 : F1;
        double e[VEC], local = 0.0, total;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            local += on ? (double)(e_rw[j] * f) : 0.0;
            e[j] = local;
        }
        const double excl = block_scan_excl(local, reds, total);
        if (on) {
            T outv[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) outv[j] = (T)(excl + e[j]);
            if (VEC == 1) l_next[i0] = outv[0]; else store_out<T, VEC, !MULTI>(l_next + base, tid * VEC, outv);
        }
    } else if (scan_next) {
        // the chunk table of the next step's resampling weights, and the tile's (max, sum) of that family from the same sums
        double Mc, Sc;
        finalize_chunk_table<T>(ct_tile, use_lds ? sh.crec : nullptr, rk * PF_NWAVES, reds, redm, Mc, Sc);
        if (tid == 0) {
            const int64_t stride = (int64_t)g.B * g.tiles, o = (int64_t)b * g.tiles + k;
            if (pre_next) {  // APF: rw = first-stage weight + log-weight
                a.part_w(step + 1)[PQ_M2 * stride + o] = Mc;
                a.part_w(step + 1)[PQ_S2 * stride + o] = Sc;
            } else {         // SISR resamples on the weights themselves
                a.part_w(step + 1)[PQ_S1 * stride + o] = Sc;
            }
        }
    }
    PF_STAMP(a, 15);
#ifdef PF_DEVTOOLS
    if (a.debug_cut < 0 && tid == 0 && b == 0 && (k % 128) == 0 && k / 128 < 8) a.dbg[24 + k / 128] = wall_clock64();
#endif
}

template <typename T, int D, int VEC, int MODE, int PROP, bool FAST, int SPEC, int MK, bool MULTI>
__global__ __launch_bounds__(PF_BLOCK, (StepWaves<T, D, MODE, PROP, FAST, SPEC, MULTI>::value)) void k_fused_step(FusedArgs<T> a) {
    using SH = StepShared<T, D, VEC>;
    __shared__ __attribute__((aligned(32))) T win[SH::WIN];
    __shared__ __attribute__((aligned(32))) T xwin[SH::XWIN ? D * SH::WIN : VEC];
    __shared__ int sh_j0;
    __shared__ int sh_cl[2 * PF_NWAVES], sh_wm[PF_NWAVES];
    __shared__ double red[(4 + 2 * D) * PF_NWAVES];
    __shared__ double reds[PF_NWAVES];
    __shared__ T redm[2 * PF_NWAVES];
    __shared__ double ptl[PF_MAX_TILES + 2], ftl[PF_MAX_TILES];
    __shared__ int sh_plan[2];
    __shared__ double crec[MULTI ? 2 * PF_LDS_CHUNKS : 2];
    __shared__ double redb[PF_NWAVES];
    // Grid (B, tiles + 1): x = the column, y = the tile - and y == tiles the column's bookkeeper (scratch: 2 + 2 D rows of `red`).
    // Workgroups are dispatched in linear order (x fastest), so every column's bookkeeper comes AFTER all step workgroups: see
    // filter_run_impl.  (The round-2 layout (tiles + 1, B) interleaved the bookkeepers with the columns: 2 - 3 us per step slower.)
    if (!a.book_inline && PF_STEP_K == (unsigned)a.g.tiles) {
        column_bookkeeping<T, D>(a, PF_STEP_B, red, redb);
        return;
    }
    const SH sh{win, xwin, &sh_j0, sh_cl, sh_wm, red, reds, redm, ptl, ftl, sh_plan, crec};
    T z0[VEC][D];
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int d = 0; d < D; ++d) z0[j][d] = T(0);
    const StepPlan<T> pl = step_prologue<T, D, VEC, MODE, SPEC, (FAST && D == 1 && sizeof(T) == 4), MULTI>(a, sh, z0);
    if constexpr (SPEC == 1) {
        step_body<T, D, VEC, MODE, PROP, FAST, SPEC, MK, true, MULTI>(a, sh, pl, z0);
    } else {
        if (pl.resample) step_body<T, D, VEC, MODE, PROP, FAST, SPEC, MK, true, MULTI>(a, sh, pl, z0);
        else step_body<T, D, VEC, MODE, PROP, FAST, SPEC, MK, false, MULTI>(a, sh, pl, z0);
    }
    if (a.book_inline == 1 && PF_STEP_K == (unsigned)a.g.tiles - 1u)  // (reads the partials of the incoming state only)
        column_bookkeeping<T, D>(a, PF_STEP_B, red, redb);
}

// clears the per-column records of a fresh run (see filter_run_impl)
template <typename W> __global__ __launch_bounds__(PF_BLOCK) void k_zero_words(W* __restrict__ p, size_t n) {
    const size_t i = (size_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (i < n) p[i] = W(0);
}

// "Observation k carries information" (filters/base.py:212: an all-NaN observation is a propagate-only move): one wave per
// step, flag[k] = any element of y[k] is not NaN.  Launched by pf_filter_run itself when the caller passes neither flag
// array, and exported as pf_observed_flags.
template <typename T>
__global__ __launch_bounds__(PF_WAVE) void k_observed_flags(const T* __restrict__ y, int64_t row_elems, uint8_t* __restrict__ out) {
    const T* row = y + (int64_t)blockIdx.x * row_elems;
    bool any = false;
    for (int64_t i = threadIdx.x; i < row_elems; i += PF_WAVE) any |= !(row[i] != row[i]);
    const unsigned long long bal = __ballot(any);
    if (threadIdx.x == 0) out[blockIdx.x] = bal ? 1 : 0;
}

// both of the above in ONE launch - what a fresh self-contained run that derives its flags starts with (an online filter() move is
// such a run per observation: one launch less per move): the first `zero_blocks` workgroups clear, workgroup zero_blocks + k flags y[k]
template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_zero_and_flags(uint32_t* __restrict__ p, size_t n, unsigned zero_blocks, const T* __restrict__ y,
                                                             int64_t row_elems, uint8_t* __restrict__ out) {
    if (blockIdx.x < zero_blocks) {
        const size_t i = (size_t)blockIdx.x * PF_BLOCK + threadIdx.x;
        if (i < n) p[i] = 0u;
        return;
    }
    if (threadIdx.x >= PF_WAVE) return;
    const unsigned k = blockIdx.x - zero_blocks;
    const T* row = y + (int64_t)k * row_elems;
    bool any = false;
    for (int64_t i = threadIdx.x; i < row_elems; i += PF_WAVE) any |= !(row[i] != row[i]);
    const unsigned long long bal = __ballot(any);
    if (threadIdx.x == 0) out[k] = bal ? 1 : 0;
}

// theta-level bookkeeping of SMC^2 in one launch (sequential/state.py:35-44, smc2.py:59-62): effective sample size of B
// log-weights under pyfilter.utils.normalize (NaN / +inf count as -inf; all -inf -> uniform) and whether every weight is
// finite.  One workgroup per row of B weights (B = the number of theta-particles, 10^2 .. 10^5; rows = the observations of
// a speculative block).  out[r][0] = ESS, out[r][1] = 1 if all finite.
template <typename T>
__device__ __forceinline__ void theta_ess_row(const T* w, int64_t B, T* out, T* redm, double* red) {
    T m = -Lim<T>::inf();
    bool finite = true;
    for (int64_t i = threadIdx.x; i < B; i += PF_BLOCK) {
        const T v = w[i];
        finite &= !(is_nan_or_posinf(v) || v == -Lim<T>::inf());
        const T s = is_nan_or_posinf(v) ? -Lim<T>::inf() : v;
        m = s > m ? s : m;
    }
    m = block_max<T>(m, redm);
    double acc[3] = {0.0, 0.0, finite ? 0.0 : 1.0};
    if (m > -Lim<T>::inf()) {
        for (int64_t i = threadIdx.x; i < B; i += PF_BLOCK) {
            const T v = w[i];
            const double e = is_nan_or_posinf(v) ? 0.0 : exp((double)v - (double)m);
            acc[0] += e;
            acc[1] += e * e;
        }
    }
    block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        out[0] = (T)(acc[1] > 0.0 ? acc[0] * acc[0] / acc[1] : (double)B);
        out[1] = acc[2] == 0.0 ? T(1) : T(0);
    }
}
template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_theta_ess(const T* __restrict__ w, int64_t B, T* __restrict__ out) {
    __shared__ T redm[PF_NWAVES];
    __shared__ double red[3 * PF_NWAVES];
    theta_ess_row<T>(w + (int64_t)blockIdx.x * B, B, out + 2 * (int64_t)blockIdx.x, redm, red);
}

}  // namespace pf
