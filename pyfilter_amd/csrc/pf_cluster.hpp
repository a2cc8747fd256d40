// pf_cluster.hpp - the column-CLUSTER persistent time loop: a filter of a few thousand particles (2 048 < N <= 16 384) is
// held in the registers of c = ceil(N / (256 VEC)) workgroups for ALL time steps of a run, ONE launch per run.
//
// Why: such a filter is too large for one workgroup (pf_column.hpp stops at 2 048 particles: sixteen waves of one workgroup
// are issue-bound on one CU) and on the per-step route (pf_fused.hpp) a step costs a whole launch - 12 - 14 us at 128 x 8 192,
// whatever the number of filters, of which the kernel boundary, the table rebuild and the window probe are most.  That is the
// per-rank shape of an 8-GPU SMC^2 (BASELINE configs[4]: 1 024 theta x 8 192 particles sharded 128 per GPU).  A column of
// 8 192 particles only needs its OWN 2 - 8 workgroups to agree once per step, not the whole grid:
//
//   per step   every WAVE publishes one record of the 64 VEC particles it owns (a "chunk"): (max, sum e, sum e^2, moments) of
//              the new log-weights and (max, total) of the next step's resampling weights, next to the chunk-local inclusive
//              scan L_i of those weights and the particles themselves (write-through `sc1` stores into the state buffer the
//              per-step route would have written, and into the scan planes), as 16-byte {3 words, tag} granules - tag = state
//              index + 1 (+ the launch's generation x 4096 when the caller numbers its launches: nothing to clear between them),
//              written by one `sc1` store each, so a record needs no flag and no second drain;
//              wave 0 of every member polls the column's <= 64 records (lane l <-> chunk l, `sc1` loads: L1 bypassed - one poller
//              per workgroup: what a hand-off costs is set by the traffic in the consumer CU's own memory queue), folds them with
//              wave-level DPP operations and broadcasts the fold through LDS: identical arithmetic in every member, so all take
//              the same decisions (ESS test, window) bit for bit.  No workgroup ever waits for a workgroup of ANOTHER column;
//              ancestors: the positions of a member lie in a contiguous range of the cdf; the chunks that can hold their
//              ancestors (conservative bounds from the folded chunk totals) are staged into LDS as cdf values
//              T(inv_tot (C_c + g_c L_i)) - the column kernel's formula, one rounding - with their particles, <= 3 072 entries
//              per pass (more passes when a stretch of negligible weights makes the range longer); branch-free lower_bound
//              of the VEC grid positions (searchsorted, side = left), gather from LDS, propagate, weigh: the per-particle
//              model code of the other two routes; draws keyed (seed, stream, step, b N + i) as there: the SAME numbers.
//   the state is read once and written once per run (its last publish IS the final state); moments rows / log-likelihood
//   increments are written by thread 0 of member 0 from the same folds.
//
// Residency: a member spins on its siblings, so the members of a column must be resident together.  The workgroup ids are
// GROUPED: eight columns share 8 c consecutive ids, member k of column (group, j) has id 8 c group + 8 k + j - its siblings are
// the ids congruent to j mod 8 next to it, i.e. (observed id % 8 placement, in-order dispatch per XCD) they sit on ONE XCD and
// are dispatched back to back.  Whatever else runs on the device - another stream's or another process's cluster launch
// included - the workgroups of a launch that are resident at any moment are therefore whole columns plus at most one partly
// dispatched column per XCD: < c slots of an XCD's >= 128 wait for a slot, all others make progress and free theirs when their
// run ends.  No two launches can hold each other's slots for good (the previous numbering, b + nbp k, dispatched member 0 of
// EVERY column first: two concurrent launches could each own half the chip without one complete column).  The host still sizes
// a launch by the occupancy query (columns per launch = resident slots / c: no workgroup queues behind a whole run of its
// own launch) and runs the columns of a larger batch in consecutive launches.  On one XCD (checked: the first records carry the
// XCC ids) the exchange stays in that XCD's L2: plain stores, L1-bypassing loads; otherwise agent-scope `sc1` on both sides,
// correct under any placement (PF_ROUTE_CLUSTER_SPREAD pins that form in the tests).  HIP promises neither the placement nor
// the dispatch order, so every spin is bounded all the same (ClusterRun.patience polls; the wait for a sibling's FIRST record -
// the one that can include waiting for a slot behind a foreign kernel - gets that plus 1 024 polls per step of the run): a
// launch that cannot make progress poisons its log-likelihoods with NaN, raises the error word and - pf_filter_args.status -
// the caller's status word, and the caller re-issues the piece on the per-step route (same draws, same numbers).
//
// Mirrors the same reference code as the other routes: sisr.py:14-56, apf.py:16-46, particle/utils.py:7-65,
// resampling.py:24-52, filters/base.py:188-221 (NaN observation -> propagate only).
#pragma once

namespace pf {

#define PFK_TPB 256
#define PFK_NW (PFK_TPB / 64)
#define PFK_MAX_CHUNKS 64  // a column's chunk records live one per lane
#define PFK_WIN 3072       // staged window: cdf values + particle planes of this many entries per pass
#define PFK_WIN_P2 4096    // the search's power-of-two bound of it
#define PFK_SPIN_LIMIT (1 << 21)
#ifdef PFK_ISA_MARKS  // stage boundaries as comments in the ISA listing (static instruction counts)
#define PFK_MARK(n) asm volatile("; PFK_MARK " #n)
#else
#define PFK_MARK(n) do { } while (0)
#endif
#define PFK_FOLD (16 + 3 * 64)  // scalars | per chunk: inclusive prefix, exclusive prefix, factor

// pf_filter_observe: the theta update of an SMC^2 observation (pf_theta_step: w += ll, the running total, the (ESS, all finite) pair
// on the device and in the polled host slot) folded into the run's tail - every column's bookkeeper writes its increment through and
// counts itself in; member 0 of the LAST column to arrive does the update for all B columns.  One launch per observation.
struct ClusterTheta {
    int enabled;
    void* w;           // (B) theta log-weights, updated in place
    const void* ll;    // (B) the row of pf_filter_args.ll_steps the run's LAST step writes
    void* stats;       // (2)
    double* slot;      // pf_host_alloc memory or null: {ESS, all finite, seq, status}
    unsigned long long seq;
    void* acc;         // (B) running log-likelihood or null
    unsigned* arrive;  // columns whose bookkeeper is done (workspace; left at zero)
};

struct ClusterRun {
    unsigned char* rec;  // granule records of this launch's columns: [2 (state parity)][nb][NG][64] x 16 B
    int b0, nb, nbp;     // first column, columns of this launch, nb rounded up to a multiple of 8 (grid = nbp * c)
    int c;               // member workgroups per column
    int nchunks;         // waves with particles per column = ceil(N / (64 VEC))
    int* err;            // |= 1: a poll ran out of patience, |= 2: an ancestor fell outside the staged chunks
    int* status;         // pf_filter_args.status (or null): the same bits, never cleared by the library
    int patience;        // polls of one wait before a member gives up
    ClusterTheta th;
    unsigned tag_base;   // the record of state s carries tag_base + s + 1.  0: the records were cleared for this launch.  Else the
                         // caller numbers its launches on this workspace (pf_run_hints.cluster_generation) and tag_base =
                         // generation * 4096: records of earlier launches never match, nothing is cleared
    int spread;          // != 0 (PF_ROUTE_CLUSTER_SPREAD, tests): the members of a column get CONSECUTIVE ids - one per XCD under
                         // the id % 8 placement - so the exchange runs on its placement-independent form (agent-scope `sc1` on both
                         // sides, never the same-XCD fast path)
};

typedef pf_u4 pfk_u4;

// 16-byte agent-scope accesses through a buffer descriptor (pf_device.hpp: `sc1` loads bypass this CU's L1 - another CU's stores
// are never seen there -, `sc1` stores write through to the memory side)
__device__ __forceinline__ pfk_u4 pfk_load16(const void* base, int byte_off) { return ld16_sc1(base, byte_off); }
__device__ __forceinline__ void pfk_store16(void* base, int byte_off, pfk_u4 v) { st16_sc1(base, byte_off, v); }
// VEC consecutive elements of a plane another member wrote (sc1 stores) - L1 bypassed
template <typename T, int VEC> __device__ __forceinline__ void pfk_load_vec(const T* base, int elem, T (&out)[VEC]) {
    constexpr int BYTES = (int)sizeof(T) * VEC;
    static_assert(BYTES % 16 == 0, "cluster kernel: 16-byte lanes");
    Pack<T, VEC> q;
#pragma unroll
    for (int k = 0; k < BYTES / 16; ++k) {
        const pfk_u4 v = pfk_load16(base, elem * (int)sizeof(T) + 16 * k);
        __builtin_memcpy(reinterpret_cast<char*>(&q) + 16 * k, &v, 16);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[j] = q.v[j];
}

template <typename T> struct PfkWords;
template <> struct PfkWords<float> {
    static constexpr int N = 1;
    __device__ __forceinline__ static void put(uint32_t* w, float v) { w[0] = __float_as_uint(v); }
    __device__ __forceinline__ static float get(const uint32_t* w) { return __uint_as_float(w[0]); }
};
template <> struct PfkWords<double> {
    static constexpr int N = 2;
    __device__ __forceinline__ static void put(uint32_t* w, double v) {
        w[0] = (uint32_t)__double2loint(v);
        w[1] = (uint32_t)__double2hiint(v);
    }
    __device__ __forceinline__ static double get(const uint32_t* w) { return __hiloint2double((int)w[1], (int)w[0]); }
};

// KIND / FILT / PROP: as in k_fused_column - the run's hidden-process kind, filter and proposal as compile-time constants
// (-1 = run-time values) for float runs of the built-in scalar models on Philox normals.
// (launch bound: the float kernels of scalar states are held to 128 VGPRs - four workgroups per CU, i.e. 128 columns of 8 192
// particles resident at once; at 129 the sine-diffusion instantiation fitted three and the batch took two launches)
template <typename T, int D, int VEC, int KIND = -1, int FILT = -1, int PROP = -1>
__global__ __launch_bounds__(PFK_TPB, (sizeof(T) == 4 && D == 1) ? (VEC == 4 ? 4 : 2) : 1) void k_fused_cluster(FusedArgs<T> a, ColumnRun run, ClusterRun cr) {
    constexpr int CH = 64 * VEC;        // particles per chunk = per wave
    constexpr int WGE = PFK_TPB * VEC;  // particles per member workgroup
    constexpr int WINC = PFK_WIN / CH;  // chunks per staged window
    constexpr int WROUNDS = (WINC + PFK_NW - 1) / PFK_NW;
    constexpr int NWT = PfkWords<T>::N;
    constexpr int NT = 5 + 2 * D;  // T values of a record: max, sum e, sum e^2, poison bits, moments[2 D], max of the resampling weights
    constexpr int NWORDS = NT * NWT + 2;  // + the chunk's scan total (double)
    constexpr int NG = (NWORDS + 2) / 3;  // granules of three words + tag
    static_assert(PFK_WIN % CH == 0, "window = whole chunks");
    extern __shared__ __attribute__((aligned(16))) unsigned char pfk_lds[];
    T* const cdfs = reinterpret_cast<T*>(pfk_lds);  // PFK_WIN_P2 entries (+inf beyond the staged ones)
    T* const xs = cdfs + PFK_WIN_P2;                // D planes of PFK_WIN entries
    // the folded records of a state, written by wave 0, read by every wave: [2 (state parity)][PFK_FOLD] doubles
    double* const fold_lds = reinterpret_cast<double*>(pfk_lds + (size_t)(PFK_WIN_P2 + D * PFK_WIN) * sizeof(T));

    const Geom& g = a.g;
    const int N = (int)g.N;
    // grouped ids (see the residency note above): id = 8 c group + 8 k + j  <->  member k of column 8 group + j
    const unsigned in_grp = blockIdx.x % (8u * (unsigned)cr.c);
    const int bl = cr.spread ? (int)(blockIdx.x / (unsigned)cr.c) : (int)(8u * (blockIdx.x / (8u * (unsigned)cr.c)) + (in_grp & 7u));
    const int k = cr.spread ? (int)(blockIdx.x % (unsigned)cr.c) : (int)(in_grp >> 3);
    if (bl >= cr.nb) return;  // (padding ids: they only keep the siblings' ids congruent mod 8)
    const int b = cr.b0 + bl;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = cr.nchunks;
    const int ch = k * PFK_NW + wid;  // this wave's chunk
    const bool wave_on = ch < nchunks;

    const bool apf = FILT >= 0 ? (FILT == PF_FILTER_APF) : (a.filter == PF_FILTER_APF);
    const int proposal = PROP >= 0 ? PROP : a.proposal;
    ModelDesc md = a.md;
    const int O = md.obs_dim;
    const uint64_t seed = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
    const int i0 = k * WGE + tid * VEC;
    const bool on = i0 < N;  // (N % VEC == 0: a lane holds VEC particles or none)
    const T nT = T(N);
    const T rcN = T(1) / nT;
    const bool pow2 = (N & (N - 1)) == 0;
    const int64_t colN = (int64_t)b * N;
    auto l_plane = [&](int q) -> T* { return ((q & 1) ? a.pos : a.cdf) + colN; };  // the scans of state q
    auto x_plane = [&](int q, int d) -> T* { return a.x[q & 1] + ((int64_t)d * g.B + b) * N; };

    // ---- the incoming state -> registers -----------------------------------------------------------------------------------
    T x[D][VEC], lw[VEC];
    int anc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        lw[j] = -Lim<T>::inf();
        anc[j] = i0 + j < N ? i0 + j : N - 1;
#pragma unroll
        for (int d = 0; d < D; ++d) x[d][j] = T(0);
    }
    if (on) {
        load_vec<T, VEC>(a.logw[run.t0 & 1] + colN + i0, lw);
#pragma unroll
        for (int d = 0; d < D; ++d) load_vec<T, VEC>(x_plane(run.t0, d) + i0, x[d]);
        load_vec<int, VEC>(a.anc + colN + i0, anc);
    }

    ColParams<T, D> cp;
    ColConsts<T, D> cc;
    load_col_params<T, D>(a, b, run.t0, false, cp);
    if constexpr (KIND >= 0) {
        static_assert((D == 1) == (KIND != PF_HID_LORENZ63_EM), "specialised cluster kernels: built-in models");
        md.hid_kind = KIND;
        md.obs_kind = (KIND == PF_HID_VERHULST_EM) ? PF_OBS_SV : PF_OBS_LINEAR;
        if constexpr (D == 1) md.obs_dim = 1;
    }
    const T* const z_tape = (KIND >= 0) ? nullptr : a.z_tape;
    cc.prepare(md, cp);
    if constexpr (KIND >= 0) __builtin_assume(cc.fast == (D == 1 && KIND != PF_HID_VERHULST_EM));
    auto y_row = [&](int t) { return a.y + ((int64_t)t * a.y_rows + (a.y_rows == 1 ? 0 : b)) * O; };

    // pivot of the weighted moments: the column's first particle, then (about) the previous state's mean - every member moves it
    // by the same folded sums
    T piv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) piv[d] = x_plane(run.t0, d)[0];

    for (int q = PFK_WIN + tid; q < PFK_WIN_P2; q += PFK_TPB) cdfs[q] = Lim<T>::inf();  // (never overwritten: a window stages <= PFK_WIN entries)
    bool dead = false;  // a poll of this workgroup timed out: nothing is waited for any more, the run's log-likelihoods become NaN
    // Every member of the column runs on ONE XCD (checked on the first records, which carry the XCC ids): the payload and the
    // records then stay in that XCD's L2 - plain stores, read by the siblings with L1-bypassing loads - instead of being written
    // through to the memory side and fetched from there.  Until it is known (and whenever the members are spread) both sides use
    // agent-scope `sc1` accesses, which are correct under any placement.
    bool fastx = false;
    const unsigned my_xcc = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;  // HW_REG_XCC_ID

    // ---- publish: the records, scans and particles of the state in registers, as state `s` (local index) of this run -------------
    // rw of the NEXT move (cp.yn / cc.ybn hold its observation): lw + first-stage weights when that move is a weighted APF move
    auto publish = [&](int s, bool two_next, bool poison_w) {
        const int q = run.t0 + s;
        // the particles first: their stores travel while the sums and the scan below are computed (state 0 of a run is where
        // the caller left it)
        if (on && (fastx || s > 0)) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (fastx) store_vec<T, VEC>(x_plane(q, d) + i0, x[d]);
                else store_out<T, VEC, true>(x_plane(q, d), i0, x[d]);
            }
        }
        T m = lw[0];
#pragma unroll
        for (int j = 1; j < VEC; ++j) m = lw[j] > m ? lw[j] : m;
        const T mw1 = wave_max<T>(m);
        T v[2 + 2 * D];
#pragma unroll
        for (int i = 0; i < 2 + 2 * D; ++i) v[i] = T(0);
        T e1[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            // (log-weights are sanitised: never NaN.  An APF never looks at the weights' ESS - FILT known: sum e^2 is not formed)
            const T ej = (lw[j] == -Lim<T>::inf()) ? T(0) : pf_exp_w(lw[j] - mw1);
            e1[j] = ej;
            v[0] += ej;
            if constexpr (FILT != PF_FILTER_APF) v[1] += ej * ej;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const T xd = x[d][j] - piv[d];
                v[2 + d] += ej * xd;
                v[2 + D + d] += ej * xd * xd;
            }
        }
#pragma unroll
        for (int i = 0; i < 2 + 2 * D; ++i)
            if (FILT != PF_FILTER_APF || i != 1) v[i] = wave_sum<T>(v[i]);
        bool poison_pre = false;
        T er[VEC], mw2 = mw1;
        if (two_next) {
            T rw[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                T xj[D];
#pragma unroll
                for (int d = 0; d < D; ++d) xj[d] = x[d][j];
                const T pre = pre_weight<T, D>(md, proposal, cp, cc, xj, true);
                if (on && is_nan_or_posinf(pre)) poison_pre = true;
                rw[j] = on ? sanitize_logw(pre + lw[j]) : -Lim<T>::inf();
            }
            T mm = rw[0];
#pragma unroll
            for (int j = 1; j < VEC; ++j) mm = rw[j] > mm ? rw[j] : mm;
            mw2 = wave_max<T>(mm);
#pragma unroll
            for (int j = 0; j < VEC; ++j) er[j] = (rw[j] == -Lim<T>::inf()) ? T(0) : pf_exp_w(rw[j] - mw2);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) er[j] = e1[j];
        }
        double incl[VEC], local = 0.0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            local += on ? (double)er[j] : 0.0;
            incl[j] = local;
        }
        const double iw = wave_scan_incl(local, lane);
        const double excl = iw - local;
        const double tw = lane_get(iw, 63);
        const unsigned pbits = ((__ballot(poison_w) != 0ull) ? 1u : 0u) | ((__ballot(poison_pre) != 0ull) ? 2u : 0u) | (my_xcc << 4);
        if (on) {
            T lv[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) lv[j] = (T)(excl + incl[j]);
            if (fastx) store_vec<T, VEC>(l_plane(q) + i0, lv);
            else store_out<T, VEC, true>(l_plane(q), i0, lv);
        }
        // the payload must have left this wave before its record can be seen (asm: the compiler drops a builtin wait it can
        // prove redundant - MI355X_MICROARCH.md, compiler hazard)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wave_on) {
            uint32_t w[3 * NG];
#pragma unroll
            for (int i = 0; i < 3 * NG; ++i) w[i] = 0u;
            PfkWords<T>::put(w + 0 * NWT, mw1);
            PfkWords<T>::put(w + 1 * NWT, v[0]);
            PfkWords<T>::put(w + 2 * NWT, v[1]);
            PfkWords<T>::put(w + 3 * NWT, (T)pbits);
#pragma unroll
            for (int i = 0; i < 2 * D; ++i) PfkWords<T>::put(w + (4 + i) * NWT, v[2 + i]);
            PfkWords<T>::put(w + (4 + 2 * D) * NWT, mw2);
            PfkWords<double>::put(w + NT * NWT, tw);
            pfk_u4 gv = {0u, 0u, 0u, cr.tag_base + (unsigned)(s + 1)};
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                if (lane == gi) {
                    gv.x = w[3 * gi];
                    gv.y = w[3 * gi + 1];
                    gv.z = w[3 * gi + 2];
                }
            }
            if (lane < NG) {
                unsigned char* base = cr.rec + ((size_t)((s & 1) * cr.nb + bl) * NG) * (64 * 16);
                if (fastx) *reinterpret_cast<pfk_u4*>(base + (lane * 64 + ch) * 16) = gv;
                else pfk_store16(base, (lane * 64 + ch) * 16, gv);
            }
        }
    };

    // ---- poll + fold: the column's records of state `s`, one per lane of WAVE 0 (one poller per workgroup: what a hand-off costs
    // is set by the traffic in the consumer CU's own memory queue); the other waves wait at the barrier and read the fold from LDS
    struct Fold {
        double M1, S1, Q1, M2, S2;
        double mom[2 * D];
        bool poison_w, poison_pre;
        const double *ci, *cex, *gg;  // per chunk, in LDS: inclusive / exclusive prefix of the rescaled chunk totals, the factors
    };
    auto poll_fold = [&](int s) -> Fold {
        double* const fl = fold_lds + (s & 1) * PFK_FOLD;
        if (wid == 0) {
            const unsigned char* base = cr.rec + ((size_t)((s & 1) * cr.nb + bl) * NG) * (64 * 16);
            pfk_u4 gr[NG];
            const unsigned want = cr.tag_base + (unsigned)(s + 1);
            // (the first records of a run may have to wait for a sibling's SLOT - behind whatever else occupies the device -
            // not only for its arithmetic: that wait is given time in proportion to the run)
            const int limit = (s == 0) ? cr.patience + ((run.n_steps < (1 << 19) ? run.n_steps : (1 << 19)) << 10) : cr.patience;
            int spins = 0;
            for (;;) {
                // (one poller wave per workgroup: all NG rows every time - a second round trip for the rest of a record would cost
                // more than the extra loads)
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) gr[gi] = pfk_load16(base, (gi * 64 + lane) * 16);
                bool ok = true;
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) ok = ok && gr[gi].w == want;
                ok = ok || lane >= nchunks || dead;
                if (cr.patience < 0) {  // (tests: every member gives up at its first wait, whatever it finds)
                    dead = true;
                    if (lane == 0) {
                        atomicOr(cr.err, 1);
                        if (cr.status) atomicOr(cr.status, 1);
                    }
                    break;
                }
                if (__ballot(ok) == ~0ull) break;
                if (++spins > limit) {
                    dead = true;
                    if (lane == 0) {
                        atomicOr(cr.err, 1);
                        if (cr.status) atomicOr(cr.status, 1);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                asm volatile("" ::: "memory");
            }
            uint32_t w[3 * NG];
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                w[3 * gi] = gr[gi].x;
                w[3 * gi + 1] = gr[gi].y;
                w[3 * gi + 2] = gr[gi].z;
            }
            const bool have = lane < nchunks;
            const T mw1 = have ? PfkWords<T>::get(w + 0 * NWT) : -Lim<T>::inf();
            const T s1 = have ? PfkWords<T>::get(w + 1 * NWT) : T(0);
            const T q1 = have ? PfkWords<T>::get(w + 2 * NWT) : T(0);
            const unsigned pb = have ? (unsigned)PfkWords<T>::get(w + 3 * NWT) : (my_xcc << 4);
            const T mw2 = have ? PfkWords<T>::get(w + (4 + 2 * D) * NWT) : -Lim<T>::inf();
            const double tw = have ? PfkWords<double>::get(w + NT * NWT) : 0.0;
            const T M1 = wave_max<T>(mw1);
            const T f1 = (mw1 == -Lim<T>::inf()) ? T(0) : pf_exp_w(mw1 - M1);
            const double S1 = (double)wave_sum<T>(s1 * f1);
            const double Q1 = (double)wave_sum<T>(q1 * f1 * f1);
            double mom[2 * D];
#pragma unroll
            for (int i = 0; i < 2 * D; ++i) {
                const T mi = have ? PfkWords<T>::get(w + (4 + i) * NWT) : T(0);
                mom[i] = (double)wave_sum<T>(mi * f1);
            }
            const bool pw = __ballot((pb & 1u) != 0u) != 0ull, pp_ = __ballot((pb & 2u) != 0u) != 0ull;
            const bool one_xcd = __ballot((pb >> 4) != my_xcc) == 0ull;
            const T M2 = wave_max<T>(mw2);
            const double gg = exp_diff_t<T>((double)mw2, (double)M2);
            const double vv = tw * gg;
            const double ci = wave_scan_incl(vv, lane);
            const double S2 = lane_get(ci, 63);
            fl[16 + lane] = ci;
            fl[16 + 64 + lane] = ci - vv;
            fl[16 + 128 + lane] = gg;
            if (lane == 0) {
                fl[0] = (double)M1;
                fl[1] = S1;
                fl[2] = Q1;
                fl[3] = (double)M2;
                fl[4] = S2;
                fl[5] = (double)((pw ? 1 : 0) | (pp_ ? 2 : 0) | (one_xcd ? 4 : 0) | (dead ? 8 : 0));
#pragma unroll
                for (int i = 0; i < 2 * D; ++i) fl[6 + i] = mom[i];
            }
        }
        pfc_barrier(PFK_NW);
        Fold f;
        f.M1 = fl[0];
        f.S1 = fl[1];
        f.Q1 = fl[2];
        f.M2 = fl[3];
        f.S2 = fl[4];
        const int flags = (int)fl[5];
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) f.mom[i] = fl[6 + i];
        f.poison_w = (flags & 1) != 0;
        f.poison_pre = (flags & 2) != 0;
        if (s == 0) fastx = (flags & 4) != 0 && !cr.spread;
        dead = dead || (flags & 8) != 0;
        f.ci = fl + 16;
        f.cex = fl + 16 + 64;
        f.gg = fl + 16 + 128;
        return f;
    };

    // moments row `row` from a fold (thread 0 of member 0 writes; every thread moves the pivot)
    auto write_moments = [&](int row, const Fold& f) {
        const double inv = inv_sum<T>(f.S1);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const double dm = f.mom[d] * inv;
            if (k == 0 && tid == 0) {
                double var = f.mom[D + d] * inv - dm * dm;
                if (var < 0.0) var = 0.0;
                a.means[((int64_t)row * g.B + b) * D + d] = (T)((double)piv[d] + dm);
                a.vars[((int64_t)row * g.B + b) * D + d] = (T)var;
            }
            piv[d] = (T)((double)piv[d] + dm);
        }
    };

    // ---- observed flags / observations, one step ahead ---------------------------------------------------------------------------
    auto obs_flag = [&](int s) -> bool {
        if (s >= run.n_steps) return false;
        if (run.use_bits) return ((run.obs_bits[s >> 5] >> (s & 31)) & 1u) != 0;
        if (run.inline_y) {  // (one-step run, shared row)
            const T* yr = a.y + (int64_t)(run.t0 + s) * O;
            bool any = false;
            for (int o = 0; o < O; ++o) any = any || !(yr[o] != yr[o]);
            return any;
        }
        return a.obs_dev[run.t0 + s] != 0;
    };
    auto load_yn = [&](int s, bool obs) {  // cp.yn <- the observation of local step s (0 when it carries none)
        const T* yr = y_row(run.t0 + (s < run.n_steps ? s : run.n_steps - 1));
#pragma unroll
        for (int o = 0; o < ColParams<T, D>::MAXO; ++o) cp.yn[o] = (obs && o < O) ? yr[o < O ? o : 0] : T(0);
    };

    bool obs_nx = obs_flag(0);
    load_yn(0, obs_nx);
    cc.set_obs(cp);
    publish(0, apf && obs_nx, false);

    T ll_tot = T(0);
    double base_prev = 0.0;
    bool obs_prev = false, prepoison_prev = false;
    const bool book = (k == 0 && tid == 0);
    if (book) ll_tot = a.ll_total[b];

    for (int s = 0; s < run.n_steps; ++s) {
        const int t = run.t0 + s;
        const bool obs = obs_nx;
        const bool two = apf && obs;
#pragma unroll
        for (int o = 0; o < ColParams<T, D>::MAXO; ++o) cp.y[o] = cp.yn[o];
        obs_nx = obs_flag(s + 1);
        load_yn(s + 1, obs_nx);
        cc.set_obs(cp);

        PFK_MARK(10_loop_top);
        // ---- draws first: they depend on nothing the siblings produce and hide under the hand-off ------------------------------
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
                    load_vec<T, VEC>(zs + ((int64_t)d * g.B + b) * N + i0, zr);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) z[j][d] = zr[j];
                }
            } else {
                draw_normals<T, D, VEC>(seed, PF_STREAM_NORMAL, (uint32_t)t, (uint64_t)(colN + i0), z);
            }
        }
        // the column's systematic offset: the caller's tape, or ONE Philox chain per workgroup (wave 0, under the hand-off; the
        // other waves read it with the fold) - every thread drawing the same number cost ~100 VALU per wave and step
        T u = T(0);
        if (a.u_tape) u = a.u_tape[(int64_t)t * g.B + b];
        else if (wid == 0) fold_lds[(s & 1) * PFK_FOLD + 15] = (double)uniform_draw<T>(seed, PF_STREAM_UNIFORM, (uint32_t)t, (uint64_t)b);

        PFK_MARK(20_draws_done);
        // ---- the state's records: moments row t, log-likelihood increment of the previous move, the decision -------------------
        const Fold f = poll_fold(s);
        if (!a.u_tape) u = (T)fold_lds[(s & 1) * PFK_FOLD + 15];
        write_moments(t, f);
        const double lse_w = f.M1 + log_sum<T>(f.S1);
        if (book) {
            if (s > 0) {
                double ll = 0.0;
                if (obs_prev) {
                    ll = lse_w - base_prev;
                    if (f.poison_w || prepoison_prev) ll = __builtin_nan("");
                }
                a.ll_steps[(int64_t)(t - 1) * g.B + b] = (T)ll;
                ll_tot = (T)((double)ll_tot + ll);
            } else if (run.t0 > 0) {
                // a run issued in pieces: the per-step piece before this one may have left its last increment pending (see
                // k_fused_column)
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
        }
        PFK_MARK(30_fold_books_done);
        const bool resample = apf ? obs : (f.S1 * f.S1 / f.Q1 < a.thr_abs);  // apf.py:29-31 | sisr.py:18-19
        double base_lse = lse_w;
        bool poison = false;
        int idx[VEC];
        T xr[VEC][D];
        if (resample) {
            const double inv_tot = inv_sum<T>(f.S2);
            if (two) base_lse = a.logN - ((f.M2 + log_sum<T>(f.S2)) - lse_w);  // apf.py:44
            else base_lse = a.logN;
            T pp[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) pp[j] = pow2 ? (T(i0 + j) + u) * rcN : grid_position<T>(i0 + j, u, nT);
            // the chunks that can hold this member's ancestors: those whose end is not certainly below its first position ..
            // the first whose end is certainly not below its last (chunk ends from the folded totals, a relative margin for the
            // rounding of the staged values)
            const int first = k * WGE, last = (first + WGE < N ? first + WGE : N) - 1;
            const double p_lo = (double)(pow2 ? (T(first) + u) * rcN : grid_position<T>(first, u, nT));
            const double p_hi = (double)(pow2 ? (T(last) + u) * rcN : grid_position<T>(last, u, nT));
            const double ce = inv_tot * f.ci[lane];
            const double margin = sizeof(T) == 4 ? 1e-5 : 1e-11;
            int w0 = __popcll(__ballot(lane < nchunks && ce * (1.0 + margin) < p_lo));
            int w1 = __popcll(__ballot(lane < nchunks && ce * (1.0 - margin) < p_hi));
            w1 = w1 > nchunks - 1 ? nchunks - 1 : w1;
            w0 = w0 > w1 ? w1 : w0;
            bool done[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                done[j] = !on;
                idx[j] = N - 1;
#pragma unroll
                for (int d = 0; d < D; ++d) xr[j][d] = T(0);
            }
            for (int wa = w0;; wa += WINC) {
                const int wb = (wa + WINC - 1 < w1) ? wa + WINC - 1 : w1;
                const int nwin = wb - wa + 1;
                const int len = (nwin * CH < N - wa * CH) ? nwin * CH : N - wa * CH;
                int np2 = 64;
                while (np2 < len) np2 <<= 1;
                PFK_MARK(40_window_bounds_done);
                pfc_barrier(PFK_NW);  // (the previous readers of the window are done)
                // ---- stage: cdf = T(inv_tot (C_c + g_c L_i)), last value of the column 1 (resampling.py:44-49) -----------------
                T lv[WROUNDS][VEC], xv[WROUNDS][D][VEC];
#pragma unroll
                for (int r = 0; r < WROUNDS; ++r) {
                    const int c_ = wa + r * PFK_NW + wid;
                    const int e0 = c_ * CH + lane * VEC;
                    if (c_ <= wb && e0 < N) {
                        pfk_load_vec<T, VEC>(l_plane(t), e0, lv[r]);
#pragma unroll
                        for (int d = 0; d < D; ++d) pfk_load_vec<T, VEC>(x_plane(t, d), e0, xv[r][d]);
                    }
                }
#pragma unroll
                for (int r = 0; r < WROUNDS; ++r) {
                    const int c_ = wa + r * PFK_NW + wid;
                    const int e0 = c_ * CH + lane * VEC;
                    const int cs = c_ < PFK_MAX_CHUNKS ? c_ : PFK_MAX_CHUNKS - 1;
                    const double C = f.cex[cs], gq = f.gg[cs];  // (wave-uniform index: broadcast reads)
                    if (c_ <= wb && e0 < N) {
                        T cv[VEC];
#pragma unroll
                        for (int j = 0; j < VEC; ++j) {
                            double c = inv_tot * (C + gq * (double)lv[r][j]);
                            if (c > 1.0) c = 1.0;
                            cv[j] = (e0 + j == N - 1) ? T(1) : (T)c;
                        }
                        const int o = (c_ - wa) * CH + lane * VEC;
                        store_vec<T, VEC>(cdfs + o, cv);
#pragma unroll
                        for (int d = 0; d < D; ++d) store_vec<T, VEC>(xs + d * PFK_WIN + o, xv[r][d]);
                    }
                }
                // (+inf behind the staged entries: [len, PFK_WIN) here, 16 bytes per thread and pass; [PFK_WIN, PFK_WIN_P2) once per launch)
                for (int q = len + tid * VEC; q < PFK_WIN; q += PFK_TPB * VEC) {
                    T infs[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) infs[j] = Lim<T>::inf();
                    store_vec<T, VEC>(cdfs + q, infs);
                }
                pfc_barrier(PFK_NW);
                PFK_MARK(50_staged);
                // ---- first q with cdf[q] >= p (searchsorted side = left), all VEC probes of a round in flight; the rounds are
                // unrolled with the positions as BYTE offsets, so a probe is one ds_read with an immediate offset and a round costs
                // compare + select + add per position (as a run-time loop over element indices: 21 VALU per round for four positions,
                // a third of the kernel's instructions); rounds above the staged length are skipped (uniform) ------------------------
                int q[VEC];
                sorted_lower_bound<T, VEC, PFK_WIN_P2>(cdfs, np2, pp, q);
                PFK_MARK(60_search_loop_done);
                const bool col_end = wb == nchunks - 1;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const bool hit = q[j] < len;
                    if (!done[j] && (hit || col_end)) {
                        const int qq = hit ? q[j] : len - 1;
                        idx[j] = wa * CH + qq;
#pragma unroll
                        for (int d = 0; d < D; ++d) xr[j][d] = xs[d * PFK_WIN + qq];
                        done[j] = true;
                    }
                }
                if (wb >= w1) break;
            }
            bool miss = false;
#pragma unroll
            for (int j = 0; j < VEC; ++j) miss = miss || !done[j];
            if (__ballot(miss) != 0ull && lane == 0) {
                atomicOr(cr.err, 2);
                if (cr.status) atomicOr(cr.status, 2);
            }
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                idx[j] = (i0 + j < N) ? i0 + j : N - 1;
#pragma unroll
                for (int d = 0; d < D; ++d) xr[j][d] = x[d][j];
            }
        }

        PFK_MARK(70_ancestors_done);
        // ---- propagate, weigh (the per-particle model code of the other routes) -----------------------------------------------------
        T lw_new[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            T xn[D], w_new;
            if (obs) {
                const T wi = sample_and_weight<T, D>(md, proposal, cp, cc, xr[j], z[j], xn);
                if (apf) {
                    w_new = wi - pre_weight<T, D>(md, proposal, cp, cc, xr[j], false);  // apf.py:43
                    if (on && is_nan_or_posinf(w_new)) poison = true;
                } else {
                    if (on && is_nan_or_posinf(wi)) poison = true;
                    w_new = resample ? wi : (wi + lw[j]);  // sisr.py:52-55
                }
            } else {  // NaN observation: propagate only, weights carried, ll = 0 (particle/state.py:38-42)
                sample_and_weight<T, D>(md, PF_PROP_BOOTSTRAP, cp, cc, xr[j], z[j], xn);
                w_new = resample ? T(0) : lw[j];
            }
            lw_new[j] = on ? sanitize_logw(w_new) : -Lim<T>::inf();
#pragma unroll
            for (int d = 0; d < D; ++d) xr[j][d] = on ? xn[d] : T(0);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            lw[j] = lw_new[j];
            if (resample || apf) anc[j] = idx[j];  // SISR without resampling keeps its ancestors (sisr.py:25-26)
#pragma unroll
            for (int d = 0; d < D; ++d) x[d][j] = xr[j][d];
        }
        PFK_MARK(80_model_done);
        base_prev = base_lse;
        obs_prev = obs;
        prepoison_prev = f.poison_pre;
        publish(s + 1, apf && obs_nx, poison);
        PFK_MARK(90_published);
    }

    // ---- the final state: log-weights and ancestors (its particles are its last publish) ------------------------------------------
    const int q_out = run.t0 + run.n_steps;
    if (on) {
        store_vec<T, VEC>(a.logw[q_out & 1] + colN + i0, lw);
        store_vec<int, VEC>(a.anc + colN + i0, anc);
    }
    if (k == 0) {  // member 0 closes the books: the last state's moments row, the last move's increment, the column record
        const Fold f = poll_fold(run.n_steps);
        write_moments(q_out, f);
        const double lse_w = f.M1 + log_sum<T>(f.S1);
        if (book) {
            double ll = 0.0;
            if (obs_prev) {
                ll = lse_w - base_prev;
                if (f.poison_w || prepoison_prev) ll = __builtin_nan("");
            }
            if (cr.th.enabled) st1_sc1<T>(a.ll_steps, (int)((int64_t)(q_out - 1) * g.B + b), (T)ll);  // (read by another column's workgroup)
            else a.ll_steps[(int64_t)(q_out - 1) * g.B + b] = (T)ll;
            ll_tot = (T)((double)ll_tot + ll);
            if (dead || *(volatile int*)cr.err) ll_tot = (T)__builtin_nan("");
            a.ll_total[b] = ll_tot;
            ColStat st{};
            st.lse_w = lse_w;
            st.ll_done = 1;
            a.stat[b] = st;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) a.poison[qq * g.B + b] = 0;
        }
        if (cr.th.enabled) {
            int* const lastf = reinterpret_cast<int*>(pfk_lds);  // (the window planes are free: every reader passed the fold's barrier)
            if (book) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the increment has left before the arrival can be seen
                const unsigned old = __hip_atomic_fetch_add(cr.th.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *lastf = (old == (unsigned)g.B - 1u) ? 1 : 0;
            }
            __syncthreads();
            if (*lastf) {
                __syncthreads();
                const int B = g.B;
                T* const w = reinterpret_cast<T*>(cr.th.w);
                T* const acc = reinterpret_cast<T*>(cr.th.acc);
                const T* const llr = reinterpret_cast<const T*>(cr.th.ll);
                T* const stats = reinterpret_cast<T*>(cr.th.stats);
                T* const redm = reinterpret_cast<T*>(pfk_lds + 64);
                double* const red = reinterpret_cast<double*>(pfk_lds + 128);
                const int err = __hip_atomic_load(cr.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (err == 0) {  // (pf_theta_step's arithmetic: k_theta_path, row 0)
                    for (int i = tid; i < B; i += PFK_TPB) {
                        const T v = ld1_sc1<T>(llr, i);
                        w[i] = w[i] + v;
                        if (acc) acc[i] = acc[i] + v;
                    }
                    theta_ess_row<T>(w, (int64_t)B, stats, redm, red);
                } else if (tid == 0) {  // a launch that gave up: nothing is updated, the slot says so
                    stats[0] = T(__builtin_nan(""));
                    stats[1] = T(0);
                }
                if (tid == 0) {
                    __hip_atomic_store(cr.th.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cr.th.slot != nullptr) {
                        cr.th.slot[0] = err == 0 ? (double)stats[0] : __builtin_nan("");
                        cr.th.slot[1] = err == 0 ? (double)stats[1] : 0.0;
                        reinterpret_cast<unsigned long long*>(cr.th.slot)[3] = (unsigned long long)(unsigned)err;
                        __threadfence_system();
                        __hip_atomic_store((unsigned long long*)(cr.th.slot + 2), cr.th.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
        }
    }
}

}  // namespace pf
