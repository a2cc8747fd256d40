// pf_kernels.hip - hand-written HIP kernels (gfx950 / CDNA4, wave64) + the C ABI of libpfamd.so.
//
// One workgroup (256 threads) owns one *tile* of one *column* (= one filter of the batch dim); a tile is R rounds of
// 256*VEC consecutive particles (pf_device.hpp).  Every per-column global dependency (max / sum / scan total) is
// carried through per-tile *partials* in the workspace that the next kernel re-reduces at its start, so no kernel
// needs inter-workgroup communication inside a launch (the kernel boundary is the only synchronisation).
//
// HBM-bound by design: no MFMA anywhere - there is no dense contraction on this path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <type_traits>

#include "../../include/pf_amd.h"
#include "pf_device.hpp"
#include "pf_models.hpp"
#include "pf_philox.hpp"

namespace pf {

// ---------------------------------------------------------------------------------------------------------------
// geometry
// ---------------------------------------------------------------------------------------------------------------
#define PF_MAX_TILES 1024
#define PF_TARGET_WGS 1024
#define PF_AUTO_FLAGS 128  // steps whose observed flags pf_filter_run derives itself (workspace slot)

struct Geom {
    int64_t N;
    int B;
    int vec;            // 4 when N % 4 == 0 else 1
    int round_elems;    // 256 * vec
    int rounds_per_tile;
    int tile_elems;
    int tiles;          // per column
};

// `target`: workgroups per launch the tile size aims at (pf_run_hints.tile_target; 0 = PF_TARGET_WGS)
static inline Geom make_geom(int64_t N, int64_t B, int64_t target = 0) {
    Geom g;
    g.N = N;
    g.B = (int)B;
    g.vec = (N % 4 == 0) ? 4 : 1;
    g.round_elems = PF_BLOCK * g.vec;
    const int64_t rounds_total = (N + g.round_elems - 1) / g.round_elems;
    // Tile size: every workgroup pays a fixed price (column combine, constants, reductions), so tiles grow until the
    // grid is down to ~PF_TARGET_WGS workgroups (4 per CU) - but never more than PF_MAX_TILES tiles per column.
    int min_r = (g.vec == 4) ? 1 : 4;  // >= 1024-particle tiles
    if (target <= 0) target = PF_TARGET_WGS;
    int64_t r = (rounds_total * B) / target;
    if (r > rounds_total) r = rounds_total;
    const int64_t r_cap = (rounds_total + PF_MAX_TILES - 1) / PF_MAX_TILES;
    if (r < r_cap) r = r_cap;
    if (r < min_r) r = min_r;
    g.rounds_per_tile = (int)r;
    g.tile_elems = g.rounds_per_tile * g.round_elems;
    g.tiles = (int)((N + g.tile_elems - 1) / g.tile_elems);
    return g;
}

// per-column bookkeeping that survives between steps (lives in the workspace)
struct ColStat {
    double lse_w;      // log sum exp of the current log-weights
    double base_lse;   // ll_t = lse(logw'_t) - base_lse   (see DESIGN.md "log-likelihood bookkeeping")
    int resample;      // this step resamples this column
    int prev_observed; // the previous step was a weighted (observed) step
    int ll_done;       // the previous step's log-likelihood was already flushed by a finalize-only pass
    int pad;
};

// workspace carve-up (all offsets 256-byte aligned)
struct WsLayout {
    size_t off_part;   // double partials[2][(6 + 2D)][B][tiles]
    size_t part_elems;
    size_t off_stat;   // ColStat[B]
    size_t off_poison; // int32 [4][B]
    size_t off_ctr;    // int32 [4] (reserved) | at +64: uint8 [PF_AUTO_FLAGS] observed flags derived on the device
    size_t off_dbg;    // uint64 [32]: development timestamps (clock64) of workgroup (0, 0)
    size_t off_cpack;  // T [B][PK_N] (sized for double): the run's closed-form records
    size_t off_piv0;   // double [B][PF_MAXD]: the run's moment pivots
    size_t off_ctab;   // double [2][B][tiles * rounds_per_tile * 4][2]: per-chunk (offset, factor) of the chunk-local scans
    size_t ctab_elems;
    size_t off_clu;    // cluster route (pf_cluster.hpp; columns of PF_CLUSTER_MIN_N < N <= PF_CLUSTER_MAX_N particles): int32 error
                       // word (256 B) | granule records [2][B][PF_CLUSTER_NG][64] x 16 B; absent (clu_bytes = 0) otherwise
    size_t clu_bytes;
    size_t off_tree;   // T [B][cdf_tree_total(N)]: the cdf sampled at every 16th, 256th, ... entry (the stand-alone multinomial's search tables)
    size_t total;
};

// pf_multinomial's search tables (the "cdf tree"): level l holds the LAST cdf entry of every block of 16^(l+1) entries (the column's last
// entry, 1, closes every level); levels are padded to 16 entries, the top one has at most 16.  N <= 2^30: at most 7 levels, N / 15
// entries in all.  Sizes and offsets are recomputed where they are used (a handful of scalar shifts): kept in per-thread arrays indexed
// by a run-time level they were promoted to LDS / scratch - 18 KB of LDS in k_scan.
__host__ __device__ static inline int cdf_tree_levels(int64_t N) {
    int levels = 0;
    int64_t n = N;
    do {
        n = (n + 15) >> 4;
        ++levels;
    } while (n > 16 && levels < 8);
    return levels;
}
__host__ __device__ static inline void cdf_tree_level(int64_t N, int l, int& size, int& off) {
    int64_t n = (N + 15) >> 4;
    int o = 0;
    for (int i = 0; i < l; ++i) {
        o += (int)((n + 15) & ~(int64_t)15);
        n = (n + 15) >> 4;
    }
    size = (int)n;
    off = o;
}
__host__ __device__ static inline int cdf_tree_total(int64_t N) {  // entries per column
    int size, off;
    cdf_tree_level(N, cdf_tree_levels(N) - 1, size, off);
    return off + ((size + 15) & ~15);
}

// the cluster route's column sizes: above what one workgroup holds (pf_column.hpp), at most 64 chunks of 256 particles
#define PF_CLUSTER_MIN_N 2048
#define PF_CLUSTER_MAX_N 16384
#define PF_CLUSTER_NG 8  // granules per chunk record, the largest instantiation (double, D = 3: 24 words)

static inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

static inline WsLayout make_ws(const Geom& g, int D) {
    WsLayout w;
    size_t o = 0;
    w.off_part = o;  // two copies (the fused pipeline double-buffers them by state parity)
    w.part_elems = (size_t)(6 + 2 * D) * g.B * g.tiles;
    o = align256(o + 2 * sizeof(double) * w.part_elems);
    w.off_stat = o;
    o = align256(o + sizeof(ColStat) * (size_t)g.B);
    w.off_poison = o;
    o = align256(o + sizeof(int32_t) * 4 * (size_t)g.B);
    w.off_ctr = o;
    o = align256(o + 64 + PF_AUTO_FLAGS);
    w.off_dbg = o;
    o = align256(o + 256);
    w.off_cpack = o;
    o = align256(o + sizeof(double) * (size_t)g.B * 24);
    w.off_piv0 = o;
    o = align256(o + sizeof(double) * (size_t)g.B * PF_MAXD);
    w.off_ctab = o;
    w.ctab_elems = (size_t)g.B * g.tiles * g.rounds_per_tile * PF_NWAVES * 2;
    o = align256(o + 2 * sizeof(double) * w.ctab_elems);
    w.off_clu = o;
    // (16 KB per column - reserved only for batches the route can take in a handful of launches: <= 8 192 member workgroups)
    w.clu_bytes = (g.N > PF_CLUSTER_MIN_N && g.N <= PF_CLUSTER_MAX_N && g.N % 4 == 0 && ((g.N + 1023) / 1024) * (int64_t)g.B <= 8192)
                      ? 256 + (size_t)2 * g.B * PF_CLUSTER_NG * 64 * 16 : 0;
    o = align256(o + w.clu_bytes);
    w.off_tree = o;  // pf_multinomial: 16-ary search tables over the cdf (cdf_tree_*), sized for double
    o = align256(o + sizeof(double) * (size_t)g.B * cdf_tree_total(g.N));
    w.total = o;
    return w;
}

// Upper bound of make_ws(...).total over every tile geometry make_geom can produce for (N, B) (any `target`): the partials
// are largest with the most tiles per column (the smallest tiles), the chunk table never holds more than
// rounds_total + one tile's rounds per column.
static inline size_t ws_bound(int64_t N, int64_t B, int D) {
    Geom g = make_geom(N, B, (int64_t)1 << 40);  // a huge target = the smallest tiles = the most tiles per column
    const int64_t rounds_total = (N + g.round_elems - 1) / g.round_elems;
    const size_t most_tiles = make_ws(g, D).total;
    g.tiles = 1;
    g.rounds_per_tile = (int)(2 * rounds_total + 2);  // tiles * rounds_per_tile <= rounds_total + rounds_per_tile <= 2 rounds_total
    const size_t most_chunks = make_ws(g, D).total;
    return most_tiles + most_chunks;
}

// partial slots
// E: sum of the tile's Exp(1) spacings (sorted-uniform multinomial); MX[d] at 6+d, MXX[d] at 6+D+d
enum { PQ_M1 = 0, PQ_S1 = 1, PQ_Q1 = 2, PQ_M2 = 3, PQ_S2 = 4, PQ_E = 5, PQ_MX = 6 };

// ---------------------------------------------------------------------------------------------------------------
// wave-cooperative lower_bound over a non-decreasing array: first j in [0, n) with c[j] >= p (clamped to n-1).
// 64-ary search: each round the 64 lanes probe 64 equally spaced elements and a ballot picks the sub-range.
// ---------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ int wave_lower_bound(const T* __restrict__ c, int n, T p, int lane) {
    int lo = 0, hi = n;
    while (hi - lo > PF_WAVE) {
        const int len = hi - lo;
        const int step = (len + PF_WAVE - 1) / PF_WAVE;
        int probe = lo + (lane + 1) * step - 1;
        if (probe > hi - 1) probe = hi - 1;
        const bool ge = c[probe] >= p;
        const unsigned long long bal = __ballot(ge);
        if (bal == 0ull) return n - 1;  // p above every element (or NaNs): clamp
        const int f = __ffsll((long long)bal) - 1;
        int nhi = lo + (f + 1) * step;
        if (nhi > hi) nhi = hi;
        lo = lo + f * step;
        hi = nhi;
    }
    const int idx = lo + lane;
    const bool ge = (idx < hi) ? (c[idx] >= p) : true;
    const unsigned long long bal = __ballot(ge);
    const int f = __ffsll((long long)bal) - 1;
    int r = lo + f;
    return r > n - 1 ? n - 1 : r;
}

// plain per-thread lower_bound on global memory in [lo, hi)
template <typename T> __device__ __forceinline__ int thread_lower_bound(const T* __restrict__ c, int lo, int hi, T p) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (c[mid] < p) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// searchsorted position of the systematic grid: (i + u) / N evaluated exactly as resampling.py:44-46 does in T
template <typename T> __device__ __forceinline__ T grid_position(int64_t i, T u, T n_as_t) { return (T(i) + u) / n_as_t; }

// ---------------------------------------------------------------------------------------------------------------
// The systematic grid inverted: K(c) = #{ i in [0, N) : grid_position(i) <= c }.  With it the ancestor of position i is
// the entry j with K(cdf_{j-1}) <= i < K(cdf_j) - the same relation searchsorted(side=left) defines - and no search
// is needed: every cdf entry computes its own offspring range in closed form (branch-free, so weight degeneracy does
// not make lanes diverge).  Exactness: the candidate floor(c N - u) + 1 is at most one off the true K (both the fma
// and the rounding of grid_position move the decision for at most one i while N * eps <= 1/4, i.e. N <= 2^22 in
// float, any N in double), so evaluating the grid position - with exactly the arithmetic of grid_position - at the two
// neighbouring indices settles it.  Larger float grids walk from the candidate to the exact boundary (step kernel).
// POW2: N is a power of two - the division is an exact multiplication by `rcN` = 1 / N.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, bool POW2> __device__ __forceinline__ T grid_value(T x_plus_u, T nT, T rcN) {
    return POW2 ? x_plus_u * rcN : x_plus_u / nT;
}
template <typename T, bool POW2> __device__ __forceinline__ int grid_count(T c, T u, T nT, T rcN, int N) {
    T t = __builtin_fma(c, nT, -u);
    t = __builtin_fmin(__builtin_fmax(t, T(-1)), nT);  // +inf (beyond the column) -> N; NaN -> -1
    const T fl = __builtin_floor(t);
    const T pa = grid_value<T, POW2>(fl + u, nT, rcN), pb = grid_value<T, POW2>((fl + T(1)) + u, nT, rcN);
    const int K = (int)fl + ((pa <= c) ? 1 : 0) + ((pb <= c) ? 1 : 0);
    return K < 0 ? 0 : (K > N ? N : K);
}
// offspring counts relative to the round's first position r0, clamped to the round: [0, RE]
template <typename T, int VEC, bool POW2>
__device__ __forceinline__ void grid_counts_local(const T (&c)[VEC], T u, T nT, T rcN, int N, int r0, int RE,
                                                  int (&out)[VEC]) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int k = grid_count<T, POW2>(c[j], u, nT, rcN, N) - r0;
        out[j] = k < 0 ? 0 : (k > RE ? RE : k);
    }
}

// One round of 256 * VEC consecutive systematic grid positions [r0, r0 + RE) against the 2 * 256 * VEC cdf entries
// starting at ws that the threads hold in registers (c0: entries ws + tid * VEC + j, c1: the same + 256 * VEC; +inf
// beyond the column).  Every entry computes how many of the round's positions lie at or below it (grid_count); entry q
// owns positions [K_{q-1}, K_q) and writes q + 1 at the head of that range in `hd`; a running maximum over the round's
// positions spreads the heads.  No search, no divergence.  `hd` (RE + 64 ints) must have its first RE entries zeroed before
// the call (the first barrier inside orders that against the scatter); `fallback(i, from)` resolves positions the
// window does not reach (from = first index not staged, or 0 when - defensively - no head precedes the position).
// sh_cl: 2 * PF_NWAVES ints, sh_wm: PF_NWAVES ints.  Three barriers.
#define PF_MAX_WINDOWS 12
// `next_window(it, d0, d1)` stages the cdf entries [ws + it * S, ws + (it + 1) * S) the same way (returns false when
// the column ends before them).  It is only called when the windows so far do not account for all RE positions - a
// stretch of negligible weights - and lets the workgroup walk on window by window (up to PF_MAX_WINDOWS) before the
// remaining positions fall back to per-position binary searches, whose ~20 dependent loads would set the duration of
// the whole kernel.
// V1: entries per thread of the window's second part.  A window is S = 256 * (VEC + V1) entries: thread t holds entries
// t * VEC + j (c0) and 256 * VEC + t * V1 + j (c1).  V1 = VEC is the 2 x 256 x VEC window of the stand-alone resampler;
// the fused step kernel uses V1 = 1 - 256 * VEC positions rarely need more than 256 * (VEC + 1) entries when the window
// starts within tile / 64 of the first ancestor, and every entry staged is an entry read, mapped and counted.
template <typename T, int VEC, int V1, typename NextWindow, typename Fallback>
__device__ __forceinline__ void inverse_grid_round(const T (&c0)[VEC], const T (&c1)[V1], int ws, int r0i, int RE, int N,
                                                   T ub, T nT, T rcN, bool pow2, int64_t i0, int* hd, int* sh_cl, int* sh_wm,
                                                   NextWindow&& next_window, Fallback&& fallback, int (&idx)[VEC]) {
    constexpr int S = PF_BLOCK * (VEC + V1);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int dump = RE + lane;
    // counts of one staged window -> heads; returns the number of positions accounted for so far
    auto scatter_window = [&](const T (&w0)[VEC], const T (&w1)[V1], int qbase, int covered_before) -> int {
        int cn0[VEC], cn1[V1];
        if (pow2) {
            grid_counts_local<T, VEC, true>(w0, ub, nT, rcN, N, r0i, RE, cn0);
            grid_counts_local<T, V1, true>(w1, ub, nT, rcN, N, r0i, RE, cn1);
        } else {
            grid_counts_local<T, VEC, false>(w0, ub, nT, rcN, N, r0i, RE, cn0);
            grid_counts_local<T, V1, false>(w1, ub, nT, rcN, N, r0i, RE, cn1);
        }
        if (lane == 63) {
            sh_cl[wid] = cn0[VEC - 1];
            sh_cl[PF_NWAVES + wid] = cn1[V1 - 1];
        }
        __syncthreads();  // the wave-boundary counts are visible; `hd` is zeroed
        int pv0 = wave_prev(cn0[VEC - 1], 0), pv1 = wave_prev(cn1[V1 - 1], 0);
        if (lane == 0) {
            pv0 = wid ? sh_cl[wid - 1] : covered_before;  // entries before the first window own no position of this round
            pv1 = sh_cl[PF_NWAVES + wid - 1];              // wave 0: the first part's last entry
        }
        const int covered = sh_cl[2 * PF_NWAVES - 1];
        // branch-free scatter: entries without offspring in this round write to a per-lane dump slot behind the RE heads
        // (exec-mask juggling per conditional store costs ~5 scalar instructions, a v_cndmask one vector instruction)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int lo0 = j ? cn0[j - 1] : pv0;
            hd[(cn0[j] > lo0) ? lo0 : dump] = qbase + tid * VEC + j + 1;
        }
#pragma unroll
        for (int j = 0; j < V1; ++j) {
            const int lo1 = j ? cn1[j - 1] : pv1;
            hd[(cn1[j] > lo1) ? lo1 : dump] = qbase + PF_BLOCK * VEC + tid * V1 + j + 1;
        }
        return covered;
    };
    int covered = scatter_window(c0, c1, 0, 0);  // positions of this round the window(s) account for
    int windows = 1;
    for (; windows < PF_MAX_WINDOWS && covered < RE; ++windows) {  // uniform: `covered` comes from LDS
        __syncthreads();                                            // everyone has read sh_cl
        T d0[VEC], d1[V1];
        if (!next_window(windows, d0, d1)) break;
        covered = scatter_window(d0, d1, windows * S, covered);
    }
    __syncthreads();
    int h[VEC];
    if (VEC == 1) h[0] = hd[tid]; else load_vec<int, VEC>(hd + tid * VEC, h);
#pragma unroll
    for (int j = 1; j < VEC; ++j) h[j] = imax(h[j], h[j - 1]);
    const int inc = wave_scan_max(h[VEC - 1]);
    if (lane == 63) sh_wm[wid] = inc;
    __syncthreads();
    int carry = wave_prev(inc, 0);
#pragma unroll
    for (int w = 0; w < PF_NWAVES - 1; ++w) carry = (w < wid) ? imax(carry, sh_wm[w]) : carry;
    const int64_t beyond = (int64_t)ws + (int64_t)windows * S;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int64_t i = i0 + j;
        const int q = imax(carry, h[j]);
        int res = ws + q - 1;
        if (i < N && (tid * VEC + j >= covered || q == 0)) res = fallback(i, (q == 0) ? 0 : (int)(beyond < N ? beyond : N));
        idx[j] = (i < N && res < N) ? res : N - 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Window search shared by the stand-alone resampler and the fused step kernel.
// For one round of 256*VEC consecutive grid positions: stage cdf[j0, j0 + WIN) in LDS, every thread lower_bounds its
// VEC positions inside the window (falling back to a global binary search beyond it), and the ancestor of the
// round's last position becomes the next round's window start (ancestors are non-decreasing).
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int VEC> struct SearchWin {
    static constexpr int WIN = 2 * PF_BLOCK * VEC;
};

// Branch-free lower_bound of VEC values in the LDS window (WIN a power of two): log2(WIN) + 1 rounds of "probe, compare,
// advance" with all VEC probes of a round in flight together.  The same instruction stream for every lane - no
// exec-mask juggling (the galloping search above spends as many scalar as vector instructions on divergent loops).
// Returns positions in [0, WIN] (WIN = beyond the window).
template <typename T, int WIN, int VEC>
__device__ __forceinline__ void window_lower_bound_flat(const T* win, const T (&p)[VEC], int (&out)[VEC]) {
    static_assert((WIN & (WIN - 1)) == 0, "window size must be a power of two");
    // positions as BYTE offsets: a probe is one ds_read with an immediate offset, a round compare + select + add per position
    // (element indices cost a shift and an add more per probe: 21 against 16 VALU per round of four positions)
    const unsigned char* const wb = reinterpret_cast<const unsigned char*>(win);
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[j] = 0;
#pragma unroll
    for (int step = WIN / 2; step >= 1; step >>= 1) {
        T v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = *reinterpret_cast<const T*>(wb + out[j] + (step - 1) * (int)sizeof(T));
#pragma unroll
        for (int j = 0; j < VEC; ++j) out[j] += (v[j] < p[j]) ? step * (int)sizeof(T) : 0;
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        out[j] += (*reinterpret_cast<const T*>(wb + out[j]) < p[j]) ? (int)sizeof(T) : 0;
        out[j] /= (int)sizeof(T);
    }
}

template <typename T, int VEC>
__device__ __forceinline__ void systematic_round(const T* __restrict__ cdf_col, int N, int64_t i0, T u,
                                                 const T* __restrict__ u_elem, T* win, int* sh_j0, int (&idx)[VEC]) {
    constexpr int WIN = SearchWin<T, VEC>::WIN;  // 2 * 256 * VEC cdf values cover the round's 256 * VEC positions
    const int tid = threadIdx.x;
    const int j0 = *sh_j0;
    const int ws = j0 - (j0 % VEC);  // window start, aligned to the vector width (N % VEC == 0)
    // stage the window: two vector loads per thread, both issued before the first LDS store; +inf beyond the column
    {
        T v0[VEC], v1[VEC];
        const int ja = ws + tid * VEC, jb = ws + (PF_BLOCK + tid) * VEC;
        const bool ina = ja < N, inb = jb < N;
        if (ina) { if (VEC == 1) v0[0] = cdf_col[ja]; else load_vec<T, VEC>(cdf_col + ja, v0); }
        if (inb) { if (VEC == 1) v1[0] = cdf_col[jb]; else load_vec<T, VEC>(cdf_col + jb, v1); }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (!ina) v0[j] = Lim<T>::inf();
            if (!inb) v1[j] = Lim<T>::inf();
        }
        if (VEC == 1) { win[tid] = v0[0]; win[PF_BLOCK + tid] = v1[0]; }
        else { store_vec<T, VEC>(win + tid * VEC, v0); store_vec<T, VEC>(win + (PF_BLOCK + tid) * VEC, v1); }
    }
    __syncthreads();
    const T nT = T(N);
    T pp[VEC];
    int qq[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) pp[j] = (i0 + j < N) ? grid_position<T>(i0 + j, u_elem ? u_elem[i0 + j] : u, nT) : T(0);
    window_lower_bound_flat<T, WIN, VEC>(win, pp, qq);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int64_t i = i0 + j;
        int res = N - 1;
        if (i < N) {
            res = (qq[j] < WIN) ? ws + qq[j] : thread_lower_bound<T>(cdf_col, ws + WIN < N ? ws + WIN : N, N, pp[j]);
            if (res > N - 1) res = N - 1;
        }
        idx[j] = res;
    }
    __syncthreads();  // everyone is done with `win` and has read sh_j0
    if (tid == PF_BLOCK - 1) *sh_j0 = idx[VEC - 1];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// stand-alone primitives
// ---------------------------------------------------------------------------------------------------------------

// per-tile online (max, sum exp, sum exp^2) of log-weights; optional in-place sanitise
// (the primitives' bodies are device functions of (column b, tile k): the kernels below run one per workgroup; the one-launch
// variants for columns of ONE tile - k_resample_one_tile / k_normalize_one_tile - run them back to back in one workgroup)
template <typename T, int VEC>
__device__ __forceinline__ void reduce_logw_body(T* __restrict__ logw, int sanitize, const uint8_t* colmask,
                                                 double* __restrict__ part, const Geom& g, int b, int k) {
    __shared__ double red[4 * PF_NWAVES];
    __shared__ T redm[PF_NWAVES];
    if (colmask && !colmask[b]) return;
    T* col = logw + (int64_t)b * g.N;
    OnlineLse<T> acc;
    acc.init();
    double q = 0.0;
    const int64_t base = (int64_t)k * g.tile_elems;
    for (int r = 0; r < g.rounds_per_tile; ++r) {
        const int64_t i0 = base + (int64_t)r * g.round_elems + threadIdx.x * VEC;
        if (i0 >= g.N) break;
        T v[VEC];
        if (VEC == 1) v[0] = col[i0]; else load_vec<T, VEC>(col + i0, v);
        bool changed = false;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const T s = sanitize_logw(v[j]);
            if (sanitize) { changed |= !(s == v[j]); v[j] = s; }
            double rs, e;
            acc.push(v[j], rs, e);
            q = q * rs * rs + e * e;
        }
        if (sanitize && changed) {
            if (VEC == 1) col[i0] = v[0]; else store_vec<T, VEC>(col + i0, v);
        }
    }
    const T M = block_max<T>(acc.m, redm);
    const double f = exp_diff((double)acc.m, (double)M);
    double sums[2] = {acc.s * f, q * f * f};
    block_sum<2>(sums, red);
    if (threadIdx.x == 0) {
        const int64_t stride = (int64_t)g.B * g.tiles;
        const int64_t o = (int64_t)b * g.tiles + k;
        part[PQ_M1 * stride + o] = (double)M;
        part[PQ_S1 * stride + o] = sums[0];
        part[PQ_Q1 * stride + o] = sums[1];
    }
}
template <typename T, int VEC>
__global__ __launch_bounds__(PF_BLOCK) void k_reduce_logw(T* __restrict__ logw, int sanitize, const uint8_t* colmask,
                                                          double* __restrict__ part, Geom g) {
    reduce_logw_body<T, VEC>(logw, sanitize, colmask, part, g, blockIdx.y, blockIdx.x);
}

// combine the (m, s[, q]) partials of one column; every thread gets the results
struct ColLse {
    double M, S, Q;
    double prefix;  // sum of the rescaled tile sums strictly before tile k (un-normalised)
};
// (T: the filter's arithmetic type - float columns take the fast float exp for the tile factors exp(m_t - M), as the fused kernels'
// tables do: the factors multiply float-precision tile sums, and four double-precision exp() per thread were most of what a
// workgroup of k_normalize_write / k_scan spent at 1 024 tiles per column)
template <typename T = double>
__device__ __forceinline__ ColLse combine_partials(const double* __restrict__ part, int slot_m, int slot_s, int slot_q,
                                                   int b, int k, int B, int tiles, double* red, double* redm) {
    const int64_t stride = (int64_t)B * tiles;
    const double* pm = part + slot_m * stride + (int64_t)b * tiles;
    const double* ps = part + slot_s * stride + (int64_t)b * tiles;
    const double* pq = (slot_q >= 0) ? part + slot_q * stride + (int64_t)b * tiles : nullptr;
    double m = -__builtin_huge_val();
    for (int t = threadIdx.x; t < tiles; t += PF_BLOCK) m = fmax(m, pm[t]);
    const double M = block_max<double>(m, redm);
    double v[3] = {0.0, 0.0, 0.0};
    for (int t = threadIdx.x; t < tiles; t += PF_BLOCK) {
        const double f = exp_diff_t<T>(pm[t], M);
        const double s = ps[t] * f;
        v[0] += s;
        if (pq) v[1] += pq[t] * f * f;
        if (t < k) v[2] += s;
    }
    block_sum<3>(v, red);
    ColLse r;
    r.M = M;
    r.S = v[0];
    r.Q = v[1];
    r.prefix = v[2];
    return r;
}

template <typename T, int VEC>
__device__ __forceinline__ void normalize_write_body(const T* __restrict__ logw, T* __restrict__ W, T* __restrict__ lse,
                                                     T* __restrict__ ess, const double* __restrict__ part, const Geom& g, int b, int k) {
    __shared__ double red[4 * PF_NWAVES];
    __shared__ double redm[PF_NWAVES];
    // (the sums of squares serve the ESS, which tile 0's workgroup alone reports: the others skip a third of the records)
    const ColLse c = combine_partials<T>(part, PQ_M1, PQ_S1, (k == 0 && ess) ? PQ_Q1 : -1, b, k, g.B, g.tiles, red, redm);
    if (k == 0 && threadIdx.x == 0) {
        if (lse) lse[b] = (T)(c.M + log(c.S));
        if (ess) ess[b] = (T)(c.S * c.S / c.Q);
    }
    if (!W) return;
    const T* col = logw + (int64_t)b * g.N;
    T* out = W + (int64_t)b * g.N;
    const T M = (T)c.M;
    const T S = (T)c.S;
    const int64_t base = (int64_t)k * g.tile_elems;
    for (int r = 0; r < g.rounds_per_tile; ++r) {
        const int64_t i0 = base + (int64_t)r * g.round_elems + threadIdx.x * VEC;
        if (i0 >= g.N) break;
        T v[VEC];
        if (VEC == 1) v[0] = col[i0]; else load_vec<T, VEC>(col + i0, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = pf_exp(v[j] - M) / S;
        if (VEC == 1) out[i0] = v[0]; else store_vec<T, VEC>(out + i0, v);
    }
}
template <typename T, int VEC>
__global__ __launch_bounds__(PF_BLOCK) void k_normalize_write(const T* __restrict__ logw, T* __restrict__ W,
                                                              T* __restrict__ lse, T* __restrict__ ess,
                                                              const double* __restrict__ part, Geom g) {
    normalize_write_body<T, VEC>(logw, W, lse, ess, part, g, blockIdx.y, blockIdx.x);
}
// pf_normalize for columns of ONE tile: both passes in one launch (same arithmetic, same values)
template <typename T, int VEC>
__global__ __launch_bounds__(PF_BLOCK) void k_normalize_one_tile(T* __restrict__ logw, T* __restrict__ W, T* __restrict__ lse,
                                                                 T* __restrict__ ess, double* __restrict__ part, Geom g) {
    reduce_logw_body<T, VEC>(logw, 1, nullptr, part, g, blockIdx.y, 0);
    __threadfence_block();
    __syncthreads();  // the tile's record and the sanitised log-weights (this workgroup's own writes) are visible
    normalize_write_body<T, VEC>(logw, W, lse, ess, part, g, blockIdx.y, 0);
}

// per-tile fp64 sums of already-normalised weights (systematic, normalized=True path)
template <typename T, int VEC>
__device__ __forceinline__ void tile_sum_body(const T* __restrict__ W, const uint8_t* colmask, double* __restrict__ part,
                                              const Geom& g, int b, int k) {
    __shared__ double red[PF_NWAVES];
    if (colmask && !colmask[b]) return;
    const T* col = W + (int64_t)b * g.N;
    double s[1] = {0.0};
    const int64_t base = (int64_t)k * g.tile_elems;
    for (int r = 0; r < g.rounds_per_tile; ++r) {
        const int64_t i0 = base + (int64_t)r * g.round_elems + threadIdx.x * VEC;
        if (i0 >= g.N) break;
        T v[VEC];
        if (VEC == 1) v[0] = col[i0]; else load_vec<T, VEC>(col + i0, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[0] += (double)v[j];
    }
    block_sum<1>(s, red);
    if (threadIdx.x == 0) {
        const int64_t stride = (int64_t)g.B * g.tiles;
        part[PQ_M1 * stride + (int64_t)b * g.tiles + k] = 0.0;  // "max" 0 -> exp_diff = 1
        part[PQ_S1 * stride + (int64_t)b * g.tiles + k] = s[0];
    }
}
template <typename T, int VEC>
__global__ __launch_bounds__(PF_BLOCK) void k_tile_sum(const T* __restrict__ W, const uint8_t* colmask,
                                                       double* __restrict__ part, Geom g) {
    tile_sum_body<T, VEC>(W, colmask, part, g, blockIdx.y, blockIdx.x);
}

// Scan of one tile: cdf_i = T( P_k + f_k * sum_{j <= i in tile} e_j ), carried in fp64 and rounded per element -
// the same value torch's CPU cumsum produces (double accumulator, per-element round; SURVEY.md §0 finding 1).
//   FROM_W   : e_j = W_j,                 f_k = 1,                     P_k = sum of previous tile sums
//   otherwise: e_j = exp(logw_j - m_k),   f_k = exp(m_k - M) / S,      P_k = normalised prefix
template <typename T, int VEC, bool FROM_W>
__device__ __forceinline__ void scan_tile(const T* __restrict__ src_col, T* __restrict__ cdf_col, const Geom& g, int k,
                                          T tile_max, double Pk, double fk, double Pnext, double* red, const T (&v_first)[VEC],
                                          T* __restrict__ tree_col = nullptr) {
    const int tree_levels = tree_col ? cdf_tree_levels(g.N) : 0;
    double carry = 0.0;
    const int64_t base = (int64_t)k * g.tile_elems;
    for (int r = 0; r < g.rounds_per_tile; ++r) {
        const int64_t r0 = base + (int64_t)r * g.round_elems;
        if (r0 >= g.N) break;  // uniform across the workgroup
        const int64_t i0 = r0 + threadIdx.x * VEC;
        const bool on = i0 < g.N;
        T v[VEC];
        double e[VEC];
        if (r == 0) {  // (loaded by the caller BEFORE it combined the tile records: one round trip to memory instead of two in a row)
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[j] = v_first[j];
        } else
        if (on) {
            if (VEC == 1) v[0] = src_col[i0]; else load_vec<T, VEC>(src_col + i0, v);
        }
        double local = 0.0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            double ej = 0.0;
            if (on) ej = FROM_W ? (double)v[j] : ((v[j] == -Lim<T>::inf()) ? 0.0 : (double)pf_exp_w(v[j] - tile_max));
            local += ej;
            e[j] = local;  // thread-local inclusive
        }
        double total;
        const double excl = block_scan_excl(local, red, total);
        if (on) {
            T outv[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                double c = Pk + fk * (carry + excl + e[j]);
                if (!FROM_W && c > Pnext) c = Pnext;
                outv[j] = (i0 + j == g.N - 1) ? T(1) : (T)c;  // cumsum[..., -1] = 1.0  (resampling.py:49)
            }
            if (VEC == 1) cdf_col[i0] = outv[0]; else store_vec<T, VEC>(cdf_col + i0, outv);
            if (tree_col) {  // (pf_multinomial) the entries that close a block of 16, 256, ... - and the column's last one every level
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int64_t i = i0 + j;
                    int64_t n = (g.N + 15) >> 4;  // level 0's size; `o` its offset
                    int o = 0;
                    if (i == g.N - 1) {
                        for (int l = 0; l < tree_levels; ++l) {
                            tree_col[o + n - 1] = outv[j];
                            o += (int)((n + 15) & ~(int64_t)15);
                            n = (n + 15) >> 4;
                        }
                    } else if (((i + 1) & 15) == 0) {
                        int64_t m = (i + 1) >> 4;
                        tree_col[m - 1] = outv[j];
                        for (int l = 1; l < tree_levels && (m & 15) == 0; ++l) {
                            o += (int)((n + 15) & ~(int64_t)15);
                            n = (n + 15) >> 4;
                            m >>= 4;
                            tree_col[o + m - 1] = outv[j];
                        }
                    }
                }
            }
        }
        carry += total;
    }
}

template <typename T, int VEC, bool FROM_W>
__device__ __forceinline__ void scan_body(const T* __restrict__ src, T* __restrict__ cdf, const uint8_t* colmask,
                                          const double* __restrict__ part, const Geom& g, int b, int k, T* __restrict__ tree = nullptr) {
    __shared__ double red[4 * PF_NWAVES];
    __shared__ double redm[PF_NWAVES];
    if (colmask && !colmask[b]) return;
    // the tile's first round of elements: issued now, used after the tile records are combined
    T v_first[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) v_first[j] = T(0);
    {
        const int64_t i0 = (int64_t)k * g.tile_elems + threadIdx.x * VEC;
        if (i0 < g.N) {
            const T* sc = src + (int64_t)b * g.N;
            if (VEC == 1) v_first[0] = sc[i0]; else load_vec<T, VEC>(sc + i0, v_first);
        }
    }
    ColLse c;
    if constexpr (FROM_W) {
        // normalised weights: every tile record is (max 0, plain sum) - the prefix is the plain sum of the tile sums before k, in
        // combine_partials' order (its factors exp(0 - 0) are exactly 1: identical values) without the maxima's exchange and
        // four double-precision exp() per thread: k_scan at 2^20 x 1 (1 024 tile records per workgroup) 10.3 -> see
        // profiles/r05_primitives_baseline_shapes.txt
        const double* ps = part + PQ_S1 * ((int64_t)g.B * g.tiles) + (int64_t)b * g.tiles;
        double v[2] = {0.0, 0.0};
        for (int t = threadIdx.x; t < g.tiles; t += PF_BLOCK) {
            const double sv = ps[t];
            v[0] += sv;
            if (t < k) v[1] += sv;
        }
        block_sum<2>(v, red);
        c.M = 0.0;
        c.S = v[0];
        c.Q = 0.0;
        c.prefix = v[1];
    } else {
        c = combine_partials<T>(part, PQ_M1, PQ_S1, -1, b, k, g.B, g.tiles, red, redm);
    }
    const int64_t stride = (int64_t)g.B * g.tiles;
    const double mk = part[PQ_M1 * stride + (int64_t)b * g.tiles + k];
    const double sk = part[PQ_S1 * stride + (int64_t)b * g.tiles + k];
    double Pk, fk, Pnext;
    if (FROM_W) {
        Pk = c.prefix;
        fk = 1.0;
        Pnext = Pk + sk;
    } else {
        fk = exp_diff_t<T>(mk, c.M) / c.S;  // (the factor combine_partials<T> gives this tile inside the later tiles' prefixes)
        Pk = c.prefix / c.S;
        Pnext = Pk + sk * fk;
    }
    scan_tile<T, VEC, FROM_W>(src + (int64_t)b * g.N, cdf + (int64_t)b * g.N, g, k, (T)mk, Pk, fk, Pnext, red, v_first,
                              tree ? tree + (int64_t)b * cdf_tree_total(g.N) : nullptr);
}
template <typename T, int VEC, bool FROM_W>
__global__ __launch_bounds__(PF_BLOCK) void k_scan(const T* __restrict__ src, T* __restrict__ cdf,
                                                   const uint8_t* colmask, const double* __restrict__ part, Geom g, T* tree) {
    scan_body<T, VEC, FROM_W>(src, cdf, colmask, part, g, blockIdx.y, blockIdx.x, tree);
}

// ancestors from the cdf: systematic grid (u per column) or iid uniforms (multinomial)
// MN: iid multinomial draws (pf_multinomial) instead of the systematic grid - a template parameter so that the systematic
// instantiations do not carry the draws' registers (four 16-entry groups in flight per thread)
template <typename T, int VEC, bool MN>
__device__ __forceinline__ void search_body(const T* __restrict__ cdf, const T* __restrict__ u, int u_per_elem,
                                            const T* __restrict__ v, uint64_t seed, uint32_t step,
                                            const uint8_t* colmask, int32_t* __restrict__ idx, const Geom& g, int force_search,
                                            int b, int k, const T* __restrict__ tree = nullptr) {
    __shared__ __attribute__((aligned(32))) T win[SearchWin<T, VEC>::WIN];
    __shared__ int sh_j0;
    if (colmask && !colmask[b]) return;
    const T* col = cdf + (int64_t)b * g.N;
    int32_t* out = idx + (int64_t)b * g.N;
    const int N = (int)g.N;
    const int64_t base = (int64_t)k * g.tile_elems;
    const int lane = threadIdx.x & 63;

    if constexpr (!MN) {
        const T* u_elem = u_per_elem ? u + (int64_t)b * g.N : nullptr;
        const T ub = u_per_elem ? u_elem[base < g.N ? base : 0] : u[b];
        // (the whole workgroup bracketing the answer with 256 or 1 024 counted probes per round - two or three dependent round
        // trips instead of one wave's four or five - measured SLOWER: k_search 6.3 -> 8.3 / 9.8 us at 2^20 x 1, 13.6 -> 15.9 / 21.9
        // at 64 x 65 536: 1 024 workgroups x 1 024 probes is traffic, and two barriers per round; profiles/r05h_primitives_ab.txt)
        if (threadIdx.x < PF_WAVE) {
            const int j0 = wave_lower_bound<T>(col, N, grid_position<T>(base, ub, T(N)), lane);
            if (lane == 0) sh_j0 = j0;
        }
        __syncthreads();
        // one u per column and a grid the closed form is exact for: ancestors from the inverted grid, no search
        const bool inverse = !u_per_elem && !force_search && !(sizeof(T) == 4 && N > (1 << 22));
        if (inverse) {
            __shared__ int sh_cl[2 * PF_NWAVES], sh_wm[PF_NWAVES];
            int* hd = reinterpret_cast<int*>(win);
            const int tid = threadIdx.x;
            const T nT = T(N), rcN = T(1) / nT;
            const bool pow2 = (N & (N - 1)) == 0;
            for (int r = 0; r < g.rounds_per_tile; ++r) {
                const int64_t r0 = base + (int64_t)r * g.round_elems;
                if (r0 >= g.N) break;
                const int64_t i0 = r0 + tid * VEC;
                const int j0 = sh_j0;
                const int ws = j0 - (j0 % VEC);
                {
                    int zero[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) zero[j] = 0;
                    if (VEC == 1) hd[tid] = 0; else store_vec<int, VEC>(hd + tid * VEC, zero);
                }
                // the window: 256 (VEC + 1) entries from the round's exact start (the fused step kernel's geometry, round 5 - the
                // ancestors of 256 VEC consecutive grid positions span at most 256 VEC + 1 entries; every staged entry is an entry
                // read and counted: 2 x 256 VEC cost k_search a third more per round); a stretch the window does not reach walks on
                // window by window inside inverse_grid_round
                constexpr int V1 = 1;
                constexpr int STAGED = PF_BLOCK * (VEC + V1);
                T c0[VEC], c1[V1];
                const int ja = ws + tid * VEC, jb = ws + PF_BLOCK * VEC + tid * V1;
                if (ja < N) { if (VEC == 1) c0[0] = col[ja]; else load_vec<T, VEC>(col + ja, c0); }
                if (jb < N) c1[0] = col[jb];
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (!(ja < N)) c0[j] = Lim<T>::inf();
                if (!(jb < N)) c1[0] = Lim<T>::inf();
                int res[VEC];
                inverse_grid_round<T, VEC, V1>(c0, c1, ws, (int)r0, g.round_elems, N, ub, nT, rcN, pow2, i0, hd, sh_cl, sh_wm,
                                           [&](int it, T (&d0)[VEC], T (&d1)[V1]) -> bool {
                                               const int w0 = ws + it * STAGED;
                                               if (w0 >= N) return false;
                                               const int wja = w0 + tid * VEC, wjb = w0 + PF_BLOCK * VEC + tid * V1;
#pragma unroll
                                               for (int j = 0; j < VEC; ++j) d0[j] = Lim<T>::inf();
                                               d1[0] = Lim<T>::inf();
                                               if (wja < N) { if (VEC == 1) d0[0] = col[wja]; else load_vec<T, VEC>(col + wja, d0); }
                                               if (wjb < N) d1[0] = col[wjb];
                                               return true;
                                           },
                                           [&](int64_t i, int from) { return thread_lower_bound<T>(col, from, N, grid_position<T>(i, ub, nT)); },
                                           res);
                if (i0 < g.N) {
                    if (VEC == 1) out[i0] = res[0]; else store_vec<int, VEC>(out + i0, res);
                }
                if (tid == PF_BLOCK - 1) sh_j0 = res[VEC - 1];  // next round's window starts at this round's last ancestor
                __syncthreads();
            }
            return;
        }
        for (int r = 0; r < g.rounds_per_tile; ++r) {
            const int64_t r0 = base + (int64_t)r * g.round_elems;
            if (r0 >= g.N) break;
            const int64_t i0 = r0 + threadIdx.x * VEC;
            int res[VEC];
            systematic_round<T, VEC>(col, N, i0, ub, u_elem, win, &sh_j0, res);
            if (i0 < g.N) {
                if (VEC == 1) out[i0] = res[0]; else store_vec<int, VEC>(out + i0, res);
            }
        }
    } else {
        // iid draws (the reference's torch.multinomial order: unsorted), each a lower_bound over the whole column.  A bisection of
        // N entries moves a 64-byte sector per 4-byte probe, ~10 of them per draw beyond what the caches hold, and that traffic -
        // not the chain of dependent loads - is what bounds it (4M x 1: 358 us; a two-level bisection with the thread's four
        // draws side by side was SLOWER, profiles/r06_systematic_two_launches.txt).  So the search is 16-ary over tables the scan
        // kernel leaves (cdf_tree_*: the cdf at every 16th, 256th, ... entry): a level is ONE aligned 64-byte group of 16 entries per
        // draw, counted in registers; the top levels (<= 4 096 entries) live in the caches, the 256-stride one in L2, and a draw
        // touches one or two lines beyond them instead of ten.
        const int levels = cdf_tree_levels(g.N);
        const T* tcol = tree + (int64_t)b * cdf_tree_total(g.N);
        // entries of the 16-group at `row` that are < p; `nv` of them exist (a level's padding / the column's end: never counted)
        auto count16 = [&](const T* __restrict__ row, int nv, T p, bool vector_ok) -> int {
            int cnt = 0;
            if (vector_ok) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    T e[4];
                    load_vec<T, 4>(row + 4 * q4, e);
#pragma unroll
                    for (int i = 0; i < 4; ++i) cnt += (4 * q4 + i < nv && e[i] < p) ? 1 : 0;
                }
            } else {
                for (int i = 0; i < nv; ++i) cnt += (row[i] < p) ? 1 : 0;
            }
            return cnt;
        };
        // The top levels - as many as the window array holds (2 048 / 512 entries) - are copied into LDS once per workgroup: every
        // lane of a global load names its own line, and at 2^20 x 1 those per-lane requests (20 of them per draw through five levels,
        // ~1 per clock and CU) were what bounded the kernel; LDS serves the same 16-entry groups without them.
        constexpr int WIN = SearchWin<T, VEC>::WIN;
        int lds_from = levels;  // levels [lds_from, levels) sit in `win`, the top one first, each padded to 16 entries
        {
            int cum = 0;
            for (int l = levels - 1; l >= 0; --l) {
                int size, off;
                cdf_tree_level(g.N, l, size, off);
                const int pad = (size + 15) & ~15;
                if (cum + pad > WIN) break;
                for (int i = threadIdx.x; i < pad; i += PF_BLOCK) win[cum + i] = tcol[off + i];
                cum += pad;
                lds_from = l;
            }
        }
        __syncthreads();
        for (int r = 0; r < g.rounds_per_tile; ++r) {
            const int64_t i0 = base + (int64_t)r * g.round_elems + threadIdx.x * VEC;
            if (i0 >= g.N) break;
            T p[VEC];
            int grp[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int64_t e = (int64_t)b * g.N + i0 + j;
                p[j] = v ? v[e] : uniform_draw<T>(seed, PF_STREAM_MULTINOMIAL, step, (uint64_t)e);
                grp[j] = 0;
            }
            for (int l = levels - 1, cum = 0; l >= lds_from; --l) {
                int size, off;
                cdf_tree_level(g.N, l, size, off);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int at = grp[j] * 16;
                    const int nv = size - at > 16 ? 16 : size - at;
                    const T* row = win + cum + at;
                    int cnt = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) cnt += (i < nv && row[i] < p[j]) ? 1 : 0;
                    grp[j] = at + cnt < size ? at + cnt : size - 1;
                }
                cum += (size + 15) & ~15;
            }
            for (int l = lds_from - 1; l >= 0; --l) {  // (the thread's VEC draws level by level: independent loads side by side)
                int size, off;
                cdf_tree_level(g.N, l, size, off);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int at = grp[j] * 16;  // (< size: the clamp below)
                    const int nv = size - at > 16 ? 16 : size - at;
                    const int nxt = at + count16(tcol + off + at, nv, p[j], true);
                    // (a group's last entry is >= p by the level above, so nxt names an entry of this level - unless NaNs broke the
                    // order: the clamp keeps every address inside its table)
                    grp[j] = nxt < size ? nxt : size - 1;
                }
            }
            int res[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int at = grp[j] * 16;  // (< N)
                const int nv = N - at > 16 ? 16 : N - at;
                const int a = at + count16(col + at, nv, p[j], VEC == 4 && nv == 16);
                res[j] = a > N - 1 ? N - 1 : a;
            }
            if (VEC == 1) out[i0] = res[0]; else store_vec<int, VEC>(out + i0, res);
        }
    }
}
template <typename T, int VEC, bool MN>
__global__ __launch_bounds__(PF_BLOCK) void k_search(const T* __restrict__ cdf, const T* __restrict__ u,
                                                     int u_per_elem, const T* __restrict__ v, uint64_t seed,
                                                     uint32_t step, const uint8_t* colmask, int32_t* __restrict__ idx,
                                                     Geom g, int force_search, const T* tree) {
    search_body<T, VEC, MN>(cdf, u, u_per_elem, v, seed, step, colmask, idx, g, force_search, blockIdx.y, blockIdx.x, tree);
}
// ---------------------------------------------------------------------------------------------------------------
// systematic(W) WITHOUT a materialised cdf (pf_systematic with cdf == NULL): two launches instead of three, 12 bytes per particle
// instead of 21.  The three-launch form is a chain tile sums -> cdf in memory -> search; here the cdf values a workgroup needs
// are rebuilt where they are used, from the weights themselves:
//   k_chunk_scan   every tile scans its weights once (fp64) and leaves, per CHUNK of 256 particles (one wave x 4), the sum of the
//                  tile's weights before the chunk (`cb`), and the tile's sum;
//   k_chunk_search every workgroup (a tile of grid positions) builds the column's tile-prefix table in LDS (<= 1 024 tile sums),
//                  finds the chunk its first position falls into - a count over the table, a count over that tile's `cb` -,
//                  and then per round stages FIVE chunks of WEIGHTS from there: a wave re-scans its chunk (the same DPP scan in the
//                  same lanes as k_chunk_scan) and has its 256 cdf values in registers,
//                         cdf_j = T( P_tile + ( cb_chunk + ( lanes before + own elements up to j ) ) ),   cdf_{N-1} = 1,
//                  one rounding to T per element like k_scan's; the ancestors come from the inverted systematic grid
//                  (grid_count: every entry's offspring range in closed form, heads scattered into LDS, a running maximum) as in
//                  k_search.  A stretch the five chunks do not cover walks on; a window that brings no progress (a long stretch of
//                  weightless particles) JUMPS: the chunk of the first position not yet covered is found like the first one.
// A cdf value is a deterministic function of (column, j) - whichever workgroup evaluates it gets the same bits -, so ancestors are
// consistent across tiles; against k_scan's values they differ in the association of the fp64 sum only (exact for float weights
// of ordinary dynamic range: tests/test_primitives_gpu.py).
// ---------------------------------------------------------------------------------------------------------------
#define PF_CHUNK (PF_WAVE * 4)
#define PF_CHUNK_WINDOW 5  // chunks staged per window: 256 (alignment slack) + 1 024 positions + 1 <= 1 280 entries

// what an element adds to the running sum: the weight itself, or exp(logw - tile maximum) (k_scan's addend: 0 for -inf)
template <typename T, bool FROM_W> __device__ __forceinline__ double chunk_addend(T v, T mk, bool on) {
    if constexpr (FROM_W) return (double)v;
    return (on && v != -Lim<T>::inf()) ? (double)pf_exp_w(v - mk) : 0.0;
}

// FROM_W: normalised weights, e_j = W_j.  Otherwise log-weights (pf_systematic_logw): sanitised IN PLACE (utils.py:57), then
// e_j = exp(logw_j - m_k) against the tile's own maximum m_k - the tile record is (m_k, sum e), as k_reduce_logw leaves it.
template <typename T, bool FROM_W>
__global__ __launch_bounds__(PF_BLOCK) void k_chunk_scan(T* __restrict__ W, const uint8_t* colmask, double* __restrict__ part,
                                                         double* __restrict__ cb, Geom g, int nchunks) {
    __shared__ double red[PF_NWAVES];
    __shared__ T redm[PF_NWAVES];
    const int b = blockIdx.y, k = blockIdx.x;
    if (colmask && !colmask[b]) return;
    T* col = W + (int64_t)b * g.N;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t base = (int64_t)k * g.tile_elems;
    T mk = T(0);
    if constexpr (!FROM_W) {
        T m = -Lim<T>::inf();
        for (int r = 0; r < g.rounds_per_tile; ++r) {
            const int64_t i0 = base + (int64_t)r * g.round_elems + threadIdx.x * 4;
            if (i0 >= g.N) break;
            T v[4];
            load_vec<T, 4>(col + i0, v);
            bool changed = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const T sv = sanitize_logw(v[j]);
                changed |= !(sv == v[j]);
                v[j] = sv;
                m = sv > m ? sv : m;
            }
            if (changed) store_vec<T, 4>(col + i0, v);  // (read back below by the thread that wrote it)
        }
        mk = block_max<T>(m, redm);
    }
    double carry = 0.0;
    for (int r = 0; r < g.rounds_per_tile; ++r) {
        const int64_t r0 = base + (int64_t)r * g.round_elems;
        if (r0 >= g.N) break;
        const int64_t i0 = r0 + threadIdx.x * 4;
        const bool on = i0 < g.N;
        T v[4] = {T(0), T(0), T(0), T(0)};
        if (on) load_vec<T, 4>(col + i0, v);
        double run = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) run += chunk_addend<T, FROM_W>(v[j], mk, on);
        double total;
        const double excl = block_scan_excl(run, red, total);
        const int64_t c = r0 / PF_CHUNK + wid;
        if (lane == 0 && c < nchunks) cb[(int64_t)b * nchunks + c] = carry + excl;
        carry += total;
    }
    if (threadIdx.x == 0) {
        const int64_t stride = (int64_t)g.B * g.tiles;
        part[PQ_M1 * stride + (int64_t)b * g.tiles + k] = FROM_W ? 0.0 : (double)mk;
        part[PQ_S1 * stride + (int64_t)b * g.tiles + k] = carry;
    }
}

// the 4 cdf values lane `lane` holds of chunk c (entries c * 256 + lane * 4 + j); +inf beyond the column.  All 64 lanes call.
// Two phases so that a wave with two chunks to rebuild (the window's fifth) has both loads in flight before it scans the first.
template <typename T> struct ChunkLoad {
    T v[4];
    double basec;
    bool live, in;
    int j0;
};
template <typename T>
__device__ __forceinline__ ChunkLoad<T> chunk_load(const T* __restrict__ col, const double* __restrict__ cbcol, int c, int nchunks, int N,
                                                   int lane) {
    ChunkLoad<T> q;
    q.j0 = c * PF_CHUNK + lane * 4;
    q.live = c < nchunks;  // wave-uniform
    q.in = q.live && q.j0 < N;
#pragma unroll
    for (int j = 0; j < 4; ++j) q.v[j] = T(0);
    if (q.in) load_vec<T, 4>(col + q.j0, q.v);
    q.basec = q.live ? cbcol[c] : 0.0;
    return q;
}
// (P, f, Pn, m): the chunk's tile - prefix, factor (1 for weights), the next tile's prefix (the clamp k_scan applies to log-weight
// tiles: the exps of a tile never carry it past the next one's start), maximum
template <typename T, bool FROM_W>
__device__ __forceinline__ void chunk_cdf4(const ChunkLoad<T>& q, double P, double f, double Pn, T m, int N, int lane, T (&out)[4]) {
    double e[4];
    double run = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        run += chunk_addend<T, FROM_W>(q.v[j], m, q.in);
        e[j] = run;
    }
    const double wex = wave_scan_incl(run, lane) - run;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double cv = FROM_W ? P + (q.basec + (wex + e[j])) : P + f * (q.basec + (wex + e[j]));
        if (!FROM_W && cv > Pn) cv = Pn;
        out[j] = q.in ? ((q.j0 + j == N - 1) ? T(1) : (T)cv) : Lim<T>::inf();
    }
}

#ifdef PF_DEVTOOLS  // (the instrumented build: cycle stamps of the middle workgroup of column 0, tools/chunk_search_stages.py)
#define PF_CSTAMP_ARG , unsigned long long* dbg
#define PF_CSTAMP(slot)                                                                                       \
    do {                                                                                                      \
        if (dbg && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x == (unsigned)g.tiles / 2) dbg[slot] = (unsigned long long)clock64(); \
    } while (0)
#else
#define PF_CSTAMP_ARG
#define PF_CSTAMP(slot) do { } while (0)
#endif
template <typename T, bool FROM_W>
__global__ __launch_bounds__(PF_BLOCK) void k_chunk_search(const T* __restrict__ W, const T* __restrict__ u, const uint8_t* colmask,
                                                           const double* __restrict__ part, const double* __restrict__ cb,
                                                           int32_t* __restrict__ idx, Geom g, int nchunks PF_CSTAMP_ARG) {
    __shared__ double ptab[PF_MAX_TILES + 4];
    __shared__ double ftab[FROM_W ? 1 : PF_MAX_TILES];  // log-weights: the tiles' factors exp(m_t - M) / S and maxima
    __shared__ T mtab[FROM_W ? 1 : PF_MAX_TILES];
    __shared__ T redm[PF_NWAVES];
    __shared__ double red[PF_NWAVES];
    __shared__ int hd[PF_BLOCK * 4 + PF_WAVE];
    __shared__ __attribute__((aligned(32))) T c1buf[PF_CHUNK];
    __shared__ int sh_cl[2 * PF_NWAVES], sh_wm[PF_NWAVES], sh_cnt[PF_NWAVES], sh_j0;
    const int b = blockIdx.y, k = blockIdx.x;
    if (colmask && !colmask[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int N = (int)g.N, tiles = g.tiles;
    PF_CSTAMP(0);
    const int cpt = g.rounds_per_tile * (g.round_elems / PF_CHUNK);  // chunks per tile (<= PF_BLOCK: the host checks)
    const T* col = W + (int64_t)b * g.N;
    const double* cbcol = cb + (int64_t)b * nchunks;
    int32_t* out = idx + (int64_t)b * g.N;
    const int64_t base = (int64_t)k * g.tile_elems;
    // ---- the column's tile-prefix table: P_t = sum of the tile sums before t (the same bits in every workgroup) ----
    // (log-weights: of the tile sums rescaled to the column's maximum, over their total - the normalised prefix k_scan uses)
    {
        const int64_t stride = (int64_t)g.B * tiles;
        const double* ps = part + PQ_S1 * stride + (int64_t)b * tiles;
        double s[4], f[4] = {1.0, 1.0, 1.0, 1.0};
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = (tid * 4 + j < tiles) ? ps[tid * 4 + j] : 0.0;
        if constexpr (!FROM_W) {
            const double* pm = part + PQ_M1 * stride + (int64_t)b * tiles;
            double m[4];
            T mx = -Lim<T>::inf();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m[j] = (tid * 4 + j < tiles) ? pm[tid * 4 + j] : -__builtin_huge_val();
                mx = (T)m[j] > mx ? (T)m[j] : mx;
            }
            const double M = (double)block_max<T>(mx, redm);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f[j] = exp_diff_t<T>(m[j], M);
                s[j] *= f[j];
                mtab[tid * 4 + j] = (T)m[j];
            }
        }
        double total;
        const double excl = block_scan_excl((s[0] + s[1]) + (s[2] + s[3]), red, total);
        const double inv = FROM_W ? 1.0 : 1.0 / total;
        ptab[tid * 4 + 0] = excl * inv;
        ptab[tid * 4 + 1] = (excl + s[0]) * inv;
        ptab[tid * 4 + 2] = (excl + (s[0] + s[1])) * inv;
        ptab[tid * 4 + 3] = (excl + ((s[0] + s[1]) + s[2])) * inv;
        if (tid == PF_BLOCK - 1) ptab[PF_MAX_TILES] = FROM_W ? total : 1.0;  // (the end of a column of PF_MAX_TILES tiles)
        if constexpr (!FROM_W) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ftab[tid * 4 + j] = f[j] * inv;
        }
    }
    __syncthreads();
    PF_CSTAMP(1);
    const T ub = u[b];
    const T nT = T(N), rcN = T(1) / nT;
    const bool pow2 = (N & (N - 1)) == 0;
    // workgroup-wide count of up to four predicates per thread (uniform result; ballots, no cross-lane data movement)
    auto block_count = [&](bool p0, bool p1, bool p2, bool p3) -> int {
        const int w = (__popcll(__ballot(p0)) + __popcll(__ballot(p1))) + (__popcll(__ballot(p2)) + __popcll(__ballot(p3)));
        __syncthreads();  // (sh_cnt's previous readers)
        if (lane == 0) sh_cnt[wid] = w;
        __syncthreads();
        return (sh_cnt[0] + sh_cnt[1]) + (sh_cnt[2] + sh_cnt[3]);
    };
    // the chunk holding the first entry with cdf >= p: the last tile, then the last chunk of it, that starts below p
    auto find_chunk = [&](T p) -> int {
        bool below[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) below[j] = tid * 4 + j < tiles && (T)ptab[tid * 4 + j] < p;
        int t = block_count(below[0], below[1], below[2], below[3]) - 1;
        t = t < 0 ? 0 : t;
        const int c = t * cpt + tid;
        const bool in = tid < cpt && c < nchunks;
        const double v = in ? cbcol[c] : 0.0;
        int cc = block_count(in && (T)(ptab[t] + (FROM_W ? v : ftab[FROM_W ? 0 : t] * v)) < p, false, false, false) - 1;
        cc = cc < 0 ? 0 : cc;
        return t * cpt + cc;
    };
    const int RE = g.round_elems;
    for (int r = 0; r < g.rounds_per_tile; ++r) {
        const int64_t r0 = base + (int64_t)r * RE;
        if (r0 >= g.N) break;
        const int64_t i0 = r0 + tid * 4;
        const int need = (g.N - r0 < RE) ? (int)(g.N - r0) : RE;
        int chunk = (r == 0) ? find_chunk(grid_position<T>(r0, ub, nT)) : sh_j0 / PF_CHUNK;
        PF_CSTAMP(2);
        {
            const int zero[4] = {0, 0, 0, 0};
            store_vec<int, 4>(hd + tid * 4, zero);
        }
        const int dump = RE + lane;
        int covered = 0, stalled = 0;
        for (;;) {  // every branch below is workgroup-uniform
            T c0[4], c1[4];
            // (a window of five chunks from offset `off` in tile t0 reaches at most into tile t0 + 1: off + 4 < 2 cpt, cpt >= 4)
            const int t0 = chunk / cpt, off = chunk - t0 * cpt;
            const ChunkLoad<T> la = chunk_load<T>(col, cbcol, chunk + wid, nchunks, N, lane);
            ChunkLoad<T> lb;
            if (wid == 0) lb = chunk_load<T>(col, cbcol, chunk + 4, nchunks, N, lane);
            // (entries past the last tile are beyond the column: +inf whatever the tables hold - the index is only kept in range)
            const int ta = t0 + (off + wid >= cpt ? 1 : 0) < tiles ? t0 + (off + wid >= cpt ? 1 : 0) : tiles - 1;
            const int tb = t0 + (off + 4 >= cpt ? 1 : 0) < tiles ? t0 + (off + 4 >= cpt ? 1 : 0) : tiles - 1;
            chunk_cdf4<T, FROM_W>(la, ptab[ta], FROM_W ? 1.0 : ftab[FROM_W ? 0 : ta], ptab[ta + 1], FROM_W ? T(0) : mtab[FROM_W ? 0 : ta], N, lane, c0);
            if (wid == 0) {
                chunk_cdf4<T, FROM_W>(lb, ptab[tb], FROM_W ? 1.0 : ftab[FROM_W ? 0 : tb], ptab[tb + 1], FROM_W ? T(0) : mtab[FROM_W ? 0 : tb], N, lane, c1);
                store_vec<T, 4>(c1buf + lane * 4, c1);
            }
            __syncthreads();  // c1buf is written (and, first window: hd is zeroed)
            PF_CSTAMP(3);
            T d1[1] = {c1buf[tid]};
            int cn0[4], cn1[1];
            if (pow2) {
                grid_counts_local<T, 4, true>(c0, ub, nT, rcN, N, (int)r0, RE, cn0);
                grid_counts_local<T, 1, true>(d1, ub, nT, rcN, N, (int)r0, RE, cn1);
            } else {
                grid_counts_local<T, 4, false>(c0, ub, nT, rcN, N, (int)r0, RE, cn0);
                grid_counts_local<T, 1, false>(d1, ub, nT, rcN, N, (int)r0, RE, cn1);
            }
            if (lane == 63) {
                sh_cl[wid] = cn0[3];
                sh_cl[PF_NWAVES + wid] = cn1[0];
            }
            __syncthreads();
            int pv0 = wave_prev(cn0[3], 0), pv1 = wave_prev(cn1[0], 0);
            if (lane == 0) {
                pv0 = wid ? sh_cl[wid - 1] : covered;  // entries before the window own no position not covered yet
                pv1 = sh_cl[PF_NWAVES + wid - 1];      // (wave 0: the first part's last entry)
            }
            const int covered_now = sh_cl[2 * PF_NWAVES - 1];
            const int q0 = chunk * PF_CHUNK + tid * 4 + 1, q1 = (chunk + 4) * PF_CHUNK + tid + 1;  // entry index + 1
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lo = j ? cn0[j - 1] : pv0;
                hd[(cn0[j] > lo) ? lo : dump] = q0 + j;
            }
            hd[(cn1[0] > pv1) ? pv1 : dump] = q1;
            stalled = (covered_now > covered) ? 0 : stalled + 1;
            covered = covered_now > covered ? covered_now : covered;
            PF_CSTAMP(4);
            if (covered >= need || stalled >= 2 || (int64_t)(chunk + PF_CHUNK_WINDOW) * PF_CHUNK >= g.N) break;
            __syncthreads();  // everyone has read sh_cl / c1buf
            // no progress: the next entry that owns a position is far away - look it up; else walk on
            chunk = stalled ? find_chunk(grid_position<T>(r0 + covered, ub, nT)) : chunk + PF_CHUNK_WINDOW;
        }
        __syncthreads();
        int h[4];
        load_vec<int, 4>(hd + tid * 4, h);
#pragma unroll
        for (int j = 1; j < 4; ++j) h[j] = imax(h[j], h[j - 1]);
        const int inc = wave_scan_max(h[3]);
        if (lane == 63) sh_wm[wid] = inc;
        __syncthreads();
        int carry = wave_prev(inc, 0);
#pragma unroll
        for (int w = 0; w < PF_NWAVES - 1; ++w) carry = (w < wid) ? imax(carry, sh_wm[w]) : carry;
        int res[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = imax(carry, h[j]);
            const bool ok = (tid * 4 + j < covered) && q > 0 && q <= N;  // (not covered: NaN weights - the clamp searchsorted gives)
            res[j] = ok ? q - 1 : N - 1;
        }
        if (i0 < g.N) store_vec<int, 4>(out + i0, res);
        if (tid == PF_BLOCK - 1) sh_j0 = res[3];  // the next round's window starts at this round's last ancestor
        __syncthreads();
        PF_CSTAMP(5);
    }
}
#undef PF_CSTAMP
#undef PF_CSTAMP_ARG

// systematic / multinomial resampling of columns of ONE tile (filters of up to a few thousand particles, or many filters:
// 1 024 x 8 192 is one 8-round tile per column) in ONE launch: tile record -> scan -> ancestors, the three kernels' bodies back
// to back in the column's workgroup (same arithmetic: identical cdf and ancestors; the cdf goes through memory between
// the stages exactly as between the launches, it is just never re-read from another CU)
template <typename T, int VEC, bool FROM_W, bool MN>
__global__ __launch_bounds__(PF_BLOCK) void k_resample_one_tile(T* __restrict__ src, const T* __restrict__ u, int u_per_elem,
                                                                const T* __restrict__ v, uint64_t seed,
                                                                uint32_t step, const uint8_t* colmask, T* __restrict__ cdf,
                                                                int32_t* __restrict__ idx, double* __restrict__ part, Geom g, T* tree) {
    const int b = blockIdx.y;
    if (FROM_W) tile_sum_body<T, VEC>(src, colmask, part, g, b, 0);
    else reduce_logw_body<T, VEC>(src, 1, colmask, part, g, b, 0);
    __threadfence_block();
    __syncthreads();
    scan_body<T, VEC, FROM_W>(src, cdf, colmask, part, g, b, 0, MN ? tree : nullptr);
    __threadfence_block();
    __syncthreads();
    search_body<T, VEC, MN>(cdf, u, u_per_elem, v, seed, step, colmask, idx, g, 0, b, 0, tree);
}

template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_gather(const T* __restrict__ x, const int32_t* __restrict__ idx,
                                                     const uint8_t* colmask, T* __restrict__ out, int64_t N, int B,
                                                     int D) {
    const int b = blockIdx.y;
    const bool on = !colmask || colmask[b];
    for (int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x; i < N; i += (int64_t)gridDim.x * PF_BLOCK) {
        const int64_t a = on ? (int64_t)idx[(int64_t)b * N + i] : i;
        for (int d = 0; d < D; ++d) {
            const int64_t o = ((int64_t)d * B + b) * N;
            out[o + i] = x[o + a];
        }
    }
}

// Whole-column moves along the batch dim (FilterResult / ParticleFilterCorrection resample + exchange).  A column is a
// contiguous run of N elements; a workgroup moves PF_COLCHUNK bytes of one column of one plane with 16-byte accesses
// (8 / 4-byte ones when the column size or the base addresses are not 16-byte multiples).  idx == nullptr: identity
// (exchange); mask == nullptr: every column.
#define PF_COLCHUNK (PF_BLOCK * 16 * 8)
typedef unsigned int pf_v4u __attribute__((ext_vector_type(4)));
typedef unsigned int pf_v2u __attribute__((ext_vector_type(2)));
template <typename V>
__global__ __launch_bounds__(PF_BLOCK) void k_columns_move(const char* __restrict__ src, const int64_t* __restrict__ idx,
                                                           const uint8_t* __restrict__ mask, char* __restrict__ dst,
                                                           int64_t col_bytes, int B) {
    const int b = blockIdx.y, p = blockIdx.z;
    if (mask && !mask[b]) return;
    int64_t from = idx ? idx[b] : (int64_t)b;
    if (from < 0) from += B;  // torch-style negative indices
    const V* s = reinterpret_cast<const V*>(src + ((int64_t)p * B + from) * col_bytes);
    V* d = reinterpret_cast<V*>(dst + ((int64_t)p * B + b) * col_bytes);
    const int64_t n = col_bytes / (int64_t)sizeof(V);
    const int64_t per_wg = PF_COLCHUNK / (int64_t)sizeof(V);
    const int64_t lo = (int64_t)blockIdx.x * per_wg;
    const int64_t hi = lo + per_wg < n ? lo + per_wg : n;
    // eight independent 16-byte loads in flight per thread before the first store
    for (int64_t i = lo + threadIdx.x; i < hi; i += 8 * PF_BLOCK) {
        V v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (i + q * PF_BLOCK < hi) v[q] = __builtin_nontemporal_load(s + i + q * PF_BLOCK);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (i + q * PF_BLOCK < hi) __builtin_nontemporal_store(v[q], d + i + q * PF_BLOCK);
    }
}

// log_likelihood partials: online max of v with companion sum W * exp(v - max)
template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_loglik_part(const T* __restrict__ v, const T* __restrict__ W,
                                                          double* __restrict__ part, Geom g) {
    __shared__ double red[PF_NWAVES];
    __shared__ T redm[PF_NWAVES];
    const int b = blockIdx.y, k = blockIdx.x;
    const int64_t base = (int64_t)k * g.tile_elems;
    const int64_t end = (base + g.tile_elems < g.N) ? base + g.tile_elems : g.N;
    T m = -Lim<T>::inf();
    double s = 0.0;
    bool nan_seen = false;
    for (int64_t i = base + threadIdx.x; i < end; i += PF_BLOCK) {
        const T vi = v[(int64_t)b * g.N + i];
        const double wi = W ? (double)W[(int64_t)b * g.N + i] : 1.0 / (double)g.N;
        if (vi != vi) nan_seen = true;
        if (vi > m) {
            s *= (m == -Lim<T>::inf()) ? 0.0 : (double)pf_exp(m - vi);
            m = vi;
        }
        s += (vi == -Lim<T>::inf()) ? 0.0 : wi * (double)pf_exp(vi - m);
    }
    if (nan_seen) s = __builtin_nan("");
    const T M = block_max<T>(m, redm);
    double sums[1] = {s * exp_diff((double)m, (double)M)};
    if (nan_seen) sums[0] = __builtin_nan("");
    block_sum<1>(sums, red);
    if (threadIdx.x == 0) {
        const int64_t stride = (int64_t)g.B * g.tiles;
        part[PQ_M1 * stride + (int64_t)b * g.tiles + k] = (double)M;
        part[PQ_S1 * stride + (int64_t)b * g.tiles + k] = sums[0];
    }
}

template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_loglik_final(const double* __restrict__ part, T* __restrict__ out,
                                                           Geom g) {
    __shared__ double red[4 * PF_NWAVES];
    __shared__ double redm[PF_NWAVES];
    const int b = blockIdx.x;
    const ColLse c = combine_partials<T>(part, PQ_M1, PQ_S1, -1, b, 0, g.B, g.tiles, red, redm);
    if (threadIdx.x == 0) out[b] = (T)(c.M + log(c.S));
}

// moments partials from normalised weights: sum W, sum W x_d, sum W x_d^2
template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_moments_part(const T* __restrict__ x, const T* __restrict__ W,
                                                           double* __restrict__ part, Geom g, int D) {
    __shared__ double red[(1 + 2 * PF_MAXD) * PF_NWAVES];
    const int b = blockIdx.y, k = blockIdx.x;
    const int64_t base = (int64_t)k * g.tile_elems;
    const int64_t end = (base + g.tile_elems < g.N) ? base + g.tile_elems : g.N;
    double acc[1 + 2 * PF_MAXD];
#pragma unroll
    for (int q = 0; q < 1 + 2 * PF_MAXD; ++q) acc[q] = 0.0;
    for (int64_t i = base + threadIdx.x; i < end; i += PF_BLOCK) {
        const double w = (double)W[(int64_t)b * g.N + i];
        acc[0] += w;
#pragma unroll
        for (int d = 0; d < PF_MAXD; ++d) {
            if (d < D) {
                const double xv = (double)x[((int64_t)d * g.B + b) * g.N + i];
                acc[1 + d] += w * xv;
                acc[1 + PF_MAXD + d] += w * xv * xv;
            }
        }
    }
    block_sum<1 + 2 * PF_MAXD>(acc, red);
    if (threadIdx.x == 0) {
        const int64_t stride = (int64_t)g.B * g.tiles;
        const int64_t o = (int64_t)b * g.tiles + k;
#pragma unroll
        for (int q = 0; q < 1 + 2 * PF_MAXD; ++q) part[q * stride + o] = acc[q];
    }
}

template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_moments_final(const double* __restrict__ part, T* __restrict__ mean,
                                                            T* __restrict__ var, Geom g, int D) {
    __shared__ double red[(1 + 2 * PF_MAXD) * PF_NWAVES];
    const int b = blockIdx.x;
    const int64_t stride = (int64_t)g.B * g.tiles;
    double acc[1 + 2 * PF_MAXD];
#pragma unroll
    for (int q = 0; q < 1 + 2 * PF_MAXD; ++q) {
        acc[q] = 0.0;
        for (int t = threadIdx.x; t < g.tiles; t += PF_BLOCK) acc[q] += part[q * stride + (int64_t)b * g.tiles + t];
    }
    block_sum<1 + 2 * PF_MAXD>(acc, red);
    if (threadIdx.x == 0) {
        for (int d = 0; d < D; ++d) {
            const double mu = acc[1 + d];  // sum W x  (the reference does not divide by sum W)
            const double v = acc[1 + PF_MAXD + d] - 2.0 * mu * acc[1 + d] + mu * mu * acc[0];
            mean[(int64_t)b * D + d] = (T)mu;
            var[(int64_t)b * D + d] = (T)(v < 0.0 ? 0.0 : v);
        }
    }
}

// elementwise model kernels (un-fused path)
template <typename T, int D>
__global__ __launch_bounds__(PF_BLOCK) void k_pre_weight(ModelDesc md, const T* __restrict__ params, int proposal,
                                                         const T* __restrict__ x, const T* __restrict__ y,
                                                         int y_rows, T* __restrict__ out, int64_t N, int B) {
    const int b = blockIdx.y;
    const int O = md.obs_dim;
    const int NP = 4 * D + O * D + 2 * O;
    ColParams<T, D> cp;
    cp.load(params + (int64_t)b * NP, O, y + (int64_t)(y_rows == 1 ? 0 : b) * O);
    ColConsts<T, D> cc;
    cc.prepare(md, cp);
    for (int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x; i < N; i += (int64_t)gridDim.x * PF_BLOCK) {
        T xv[D];
#pragma unroll
        for (int d = 0; d < D; ++d) xv[d] = x[((int64_t)d * B + b) * N + i];
        out[(int64_t)b * N + i] = pre_weight<T, D>(md, proposal, cp, cc, xv);
    }
}

template <typename T, int D>
__device__ __forceinline__ void draw_z(const T* __restrict__ z, uint64_t seed, uint32_t step, int64_t N, int B, int b,
                                       int64_t i, T (&zv)[D]) {
    if (z) {
#pragma unroll
        for (int d = 0; d < D; ++d) zv[d] = z[((int64_t)d * B + b) * N + i];
    } else {
        NormalDraw<T, D>::draw(seed, PF_STREAM_NORMAL, step, (uint64_t)((int64_t)b * N + i), zv);
    }
}

template <typename T, int D>
__global__ __launch_bounds__(PF_BLOCK) void k_sample_and_weight(ModelDesc md, const T* __restrict__ params,
                                                                int proposal, int weigh, const T* __restrict__ x,
                                                                const T* __restrict__ y, int y_rows,
                                                                const T* __restrict__ z, uint64_t seed, uint32_t step,
                                                                T* __restrict__ x_out, T* __restrict__ w_out,
                                                                int64_t N, int B) {
    const int b = blockIdx.y;
    const int O = md.obs_dim;
    const int NP = 4 * D + O * D + 2 * O;
    ColParams<T, D> cp;
    cp.load(params + (int64_t)b * NP, O, (weigh && y) ? y + (int64_t)(y_rows == 1 ? 0 : b) * O : nullptr);
    ColConsts<T, D> cc;
    cc.prepare(md, cp);
    for (int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x; i < N; i += (int64_t)gridDim.x * PF_BLOCK) {
        T xv[D], zv[D], xn[D];
#pragma unroll
        for (int d = 0; d < D; ++d) xv[d] = x[((int64_t)d * B + b) * N + i];
        draw_z<T, D>(z, seed, step, N, B, b, i, zv);
        const T w = sample_and_weight<T, D>(md, weigh ? proposal : PF_PROP_BOOTSTRAP, cp, cc, xv, zv, xn);
#pragma unroll
        for (int d = 0; d < D; ++d) x_out[((int64_t)d * B + b) * N + i] = xn[d];
        if (weigh && w_out) w_out[(int64_t)b * N + i] = w;
    }
}

template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_initial_sample(double m0a, double m0b, double m0c, double s0a,
                                                             double s0b, double s0c, const T* __restrict__ z,
                                                             uint64_t seed, T* __restrict__ x, int64_t N, int B,
                                                             int D) {
    const int b = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x; i < N; i += (int64_t)gridDim.x * PF_BLOCK) {
        T zv[4] = {T(0), T(0), T(0), T(0)};
        if (!z) NormalDraw<T, 4>::draw(seed, PF_STREAM_INIT, 0u, (uint64_t)((int64_t)b * N + i), zv);
#pragma unroll
        for (int d = 0; d < PF_MAXD; ++d) {
            if (d < D) {
                const int64_t o = ((int64_t)d * B + b) * N + i;
                const T zz = z ? z[o] : zv[d];
                x[o] = (T)(d == 0 ? m0a : (d == 1 ? m0b : m0c)) + (T)(d == 0 ? s0a : (d == 1 ? s0b : s0c)) * zz;
            }
        }
    }
}

// ... with one initial mean / scale per filter: element (b, d) at m0[b * mb + d * md] (strides in elements, 0 = broadcast)
template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_initial_sample_cols(const T* __restrict__ m0, int64_t mb, int64_t md, const T* __restrict__ s0,
                                                                  int64_t sb, int64_t sd, const T* __restrict__ z, uint64_t seed,
                                                                  T* __restrict__ x, int64_t N, int B, int D) {
    const int b = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x; i < N; i += (int64_t)gridDim.x * PF_BLOCK) {
        T zv[4] = {T(0), T(0), T(0), T(0)};
        if (!z) NormalDraw<T, 4>::draw(seed, PF_STREAM_INIT, 0u, (uint64_t)((int64_t)b * N + i), zv);
#pragma unroll
        for (int d = 0; d < PF_MAXD; ++d) {
            if (d < D) {
                const int64_t o = ((int64_t)d * B + b) * N + i;
                const T zz = z ? z[o] : zv[d];
                // (two roundings - product, then sum; not contracted into an fma - the torch expression `m + s * z` this
                // replaces, to the last bit)
                {
#pragma clang fp contract(off)
                    const T sz = s0[b * sb + d * sd] * zz;
                    x[o] = m0[b * mb + d * md] + sz;
                }
            }
        }
    }
}

// Test support (pf_debug_draw_normals): the standard normals the fused step kernel draws for steps step0 .. - the same
// draw_normals<T, D, VEC> call, addressed as the step kernel addresses it (thread = VEC consecutive particles).
template <typename T, int D, int VEC>
__global__ __launch_bounds__(PF_BLOCK) void k_debug_normals(uint64_t seed, uint32_t step0, T* __restrict__ out, int64_t N,
                                                            int B) {
    const int b = blockIdx.y;
    const uint32_t s = blockIdx.z;
    const int64_t i0 = ((int64_t)blockIdx.x * PF_BLOCK + threadIdx.x) * VEC;
    if (i0 >= N) return;
    T zt[VEC][D];
    draw_normals<T, D, VEC>(seed, PF_STREAM_NORMAL, step0 + s, (uint64_t)((int64_t)b * N + i0), zt);
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            if (i0 + j < N) out[(((int64_t)s * D + d) * B + b) * N + i0 + j] = zt[j][d];
}

// ---------------------------------------------------------------------------------------------------------------
// smoothing over a recorded state history (pyfilter/filters/particle/base.py:105-157), S states, time-major:
//   x_hist (S, D, B, N), logw_hist / anc_hist (S, B, N); anc_hist[t] = the ancestors in state t - 1 of state t's particles.
// ---------------------------------------------------------------------------------------------------------------

// "fl" (_do_sample_fl, :136-152): every particle of the last state walks its ancestral line backwards.  One thread per
// trajectory; the walk is a chain of dependent gathers (latency-bound, N B chains in flight hide it).
template <typename T>
__global__ __launch_bounds__(PF_BLOCK) void k_trace_ancestors(const T* __restrict__ x_hist,
                                                              const int32_t* __restrict__ anc_hist, T* __restrict__ out,
                                                              int64_t S, int64_t N, int B, int D) {
    const int b = blockIdx.y;
    const int64_t plane = (int64_t)B * N;
    for (int64_t i = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x; i < N; i += (int64_t)gridDim.x * PF_BLOCK) {
        int64_t idx = i;
        for (int64_t t = S - 1; t >= 0; --t) {
            for (int d = 0; d < D; ++d)
                out[(t * D + d) * plane + (int64_t)b * N + i] = x_hist[(t * D + d) * plane + (int64_t)b * N + idx];
            if (t > 0) idx = anc_hist[t * plane + (int64_t)b * N + idx];
        }
    }
}

// "ffbs" (_do_sample_ffbs, :105-134): backward simulation.  Trajectory j holds x_{t+1}^{(j)} and draws its state-t particle
// from Categorical(logits_i = logw_t^{(i)} + log p(x_{t+1}^{(j)} | x_t^{(i)})) - an N x N evaluation per step that the
// reference materialises as an (N, N, [B]) tensor.  Here: one thread per trajectory for the WHOLE backward pass
// (trajectories are independent given the recorded states); the workgroup stages 256 candidates (one-step mean, scale and
// weight of particle i) in LDS and every thread scans them twice - (max, sum exp) of its logits, then the inverse-CDF walk
// with ONE uniform per (trajectory, step) (tape `u` (S - 1, B, N) or Philox).  O(N^2 S) flops, O(N S) memory.
#define PF_STREAM_SMOOTH 5
template <typename T, int D>
__global__ __launch_bounds__(PF_BLOCK) void k_ffbs(ModelDesc md, const T* __restrict__ params, const T* __restrict__ x_hist,
                                                    const T* __restrict__ logw_hist, const T* __restrict__ x_last,
                                                    const T* __restrict__ u, uint64_t seed, T* __restrict__ out, int64_t S,
                                                    int64_t N, int B) {
    __shared__ T s_loc[D][PF_BLOCK], s_i2[D][PF_BLOCK], s_c[PF_BLOCK];
    const int b = blockIdx.y;
    const int O = md.obs_dim;
    const int NP = 4 * D + O * D + 2 * O;
    ColParams<T, D> cp;
    cp.load(params + (int64_t)b * NP, O, nullptr);
    const int64_t plane = (int64_t)B * N;
    const int64_t j = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    const bool on = j < N;
    const T inc = (T)md.inc_scale;
    T xj[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        xj[d] = on ? x_last[((int64_t)d * B + b) * N + j] : T(0);
        if (on) out[((S - 1) * D + d) * plane + (int64_t)b * N + j] = xj[d];
    }
    const int64_t tiles = (N + PF_BLOCK - 1) / PF_BLOCK;
    for (int64_t t = S - 2; t >= 0; --t) {
        const T* xt = x_hist + t * D * plane;
        const T* wt = logw_hist + t * plane + (int64_t)b * N;
        // candidate i of a tile: its one-step mean / scale (model.hidden.build_density(state), :112) and its log-weight
        auto stage = [&](int64_t tile) {
            const int64_t i = tile * PF_BLOCK + threadIdx.x;
            T xi[D], loc[D], sc[D];
#pragma unroll
            for (int d = 0; d < D; ++d) xi[d] = i < N ? xt[((int64_t)d * B + b) * N + i] : T(0);
            mean_scale<T, D>(md, cp, xi, loc, sc);
            T c = i < N ? sanitize_logw(wt[i]) : -Lim<T>::inf();
            if (c == Lim<T>::lowest()) c = -Lim<T>::inf();  // a -inf weight stays out of the draw
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const T sd = sc[d] * inc;
                s_loc[d][threadIdx.x] = loc[d];
                s_i2[d][threadIdx.x] = T(0.5) / (sd * sd);
                c -= pf_log(pf_abs(sd)) + T(PF_LOG_SQRT_2PI);
            }
            s_c[threadIdx.x] = c;
        };
        auto logit = [&](int q) {
            T l = s_c[q];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const T r = xj[d] - s_loc[d][q];
                l -= r * r * s_i2[d][q];
            }
            return l;
        };
        // pass 1: running (max, sum exp) over all candidates
        T m = -Lim<T>::inf();
        double ssum = 0.0;
        for (int64_t tile = 0; tile < tiles; ++tile) {
            __syncthreads();
            stage(tile);
            __syncthreads();
            const int cnt = (int)((tile + 1) * PF_BLOCK <= N ? PF_BLOCK : N - tile * PF_BLOCK);
            for (int q = 0; q < cnt; ++q) {
                const T l = logit(q);
                if (l > m) {
                    ssum = (m == -Lim<T>::inf()) ? 0.0 : ssum * (double)pf_exp(m - l);
                    m = l;
                }
                if (l != -Lim<T>::inf()) ssum += (double)pf_exp(l - m);
            }
        }
        // pass 2: inverse CDF - the first candidate whose running sum reaches u * total
        T uj;
        if (u) uj = on ? u[t * plane + (int64_t)b * N + j] : T(0);
        else uj = uniform_draw<T>(seed, PF_STREAM_SMOOTH, (uint32_t)t, (uint64_t)((int64_t)b * N + (on ? j : 0)));
        const double target = (double)uj * ssum;
        double run = 0.0;
        int64_t pick = -1, last_pos = 0;
        for (int64_t tile = 0; tile < tiles; ++tile) {
            __syncthreads();
            stage(tile);
            __syncthreads();
            const int cnt = (int)((tile + 1) * PF_BLOCK <= N ? PF_BLOCK : N - tile * PF_BLOCK);
            for (int q = 0; q < cnt; ++q) {
                const T l = logit(q);
                if (l != -Lim<T>::inf()) {
                    const double e = (double)pf_exp(l - m);
                    if (e > 0.0) last_pos = tile * PF_BLOCK + q;
                    run += e;
                    if (pick < 0 && run > target) pick = tile * PF_BLOCK + q;
                }
            }
        }
        if (pick < 0) pick = last_pos;  // u = 1 - eps against a rounded-down total
        if (on) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                xj[d] = xt[((int64_t)d * B + b) * N + pick];
                out[(t * D + d) * plane + (int64_t)b * N + j] = xj[d];
            }
        }
    }
}

}  // namespace pf

#include "pf_fused.hpp"
#include "pf_column.hpp"
#include "pf_cluster.hpp"

namespace pf {
// pf_filter_observe -> the route that carries the run: what the theta update needs, for the length of that one call on that one
// thread (the cluster route folds it into its launch and says so; every other route leaves it to a pf_theta_step launch)
struct ThetaFold {
    void* w;
    const void* ll;
    void* stats;
    void* slot;
    uint64_t seq;
    void* acc;
    int folded;
};
extern thread_local ThetaFold* tls_theta_fold;
}  // namespace pf

// =================================================================================================================
// C ABI
// =================================================================================================================
using namespace pf;

#define PF_CHECK_LAUNCH()                       \
    do {                                        \
        hipError_t e_ = hipGetLastError();      \
        if (e_ != hipSuccess) return (int)e_;   \
    } while (0)

static inline bool bad_shape(int64_t N, int64_t B) { return N < 1 || B < 1 || N > (int64_t)1 << 30 || B > 65535; }

// `fused`: the call is a fused filter run - the only place a user-defined affine process (whose mean / scale planes
// travel in pf_filter_args) can be evaluated; the stand-alone model kernels take built-in kinds only
static inline int check_model(const pf_model* m, bool fused = false) {
    if (!m || !m->params) return PF_EINVAL;
    if (m->hid_kind == PF_HID_USER_AFFINE && !fused) return PF_EUNSUPPORTED;
    if (m->dim < 1 || m->dim > PF_MAXD || m->obs_dim < 1 || m->obs_dim > PF_MAXO) return PF_EUNSUPPORTED;
    if (m->dim == 1 && m->obs_dim != 1) return PF_EUNSUPPORTED;
    if (m->hid_kind < 0 || m->hid_kind > PF_HID_USER_AFFINE) return PF_EUNSUPPORTED;
    if (m->hid_kind == PF_HID_LORENZ63_EM && m->dim != 3) return PF_EUNSUPPORTED;
    if (m->obs_kind != PF_OBS_LINEAR && m->obs_kind != PF_OBS_SV) return PF_EUNSUPPORTED;
    if (m->obs_kind == PF_OBS_SV && m->dim != 1) return PF_EUNSUPPORTED;
    return PF_OK;
}

static inline ModelDesc to_desc(const pf_model* m) {
    ModelDesc d;
    d.hid_kind = m->hid_kind;
    d.obs_kind = m->obs_kind;
    d.obs_dim = m->obs_dim;
    d.dt = m->dt;
    d.inc_scale = m->inc_scale;
    return d;
}

// The fused-run instantiation matrix compiles as several translation units (the build runs them in parallel):
//   -DPF_TU_NO_F64 -DPF_TU_NO_F32DN -DPF_TU_NO_F32D1 : the main unit - C ABI and the stand-alone primitives
//   -DPF_TU_F32D1_ONLY -DPF_TU_VEC=4|1 : only the float32 fused kernels of scalar states for one vector width + entry
//   -DPF_TU_F32DN_ONLY              : only the float32 fused kernels of D > 1 states + their entry (pf_run_f32_dn)
//   -DPF_TU_F64_ONLY                : only the float64 fused kernels + their entry               (pf_run_f64)
//   ... each of the kernel units additionally with -DPF_TU_MULTI=0|1: only the kernels of single-round / multi-round
//   tiles (the MULTI template argument of k_fused_step; entries carry the suffix _m0 / _m1)
// Without any of the macros the file is a single self-contained unit.
#if defined(PF_TU_COLUMN_F32) || defined(PF_TU_COLUMN_F64) || defined(PF_TU_CLUSTER_F32) || defined(PF_TU_CLUSTER_F64)
// the column-persistent / column-cluster kernels of one arithmetic type, nothing else
#define PF_TU_NO_API
#define PF_TU_NO_F64
#define PF_TU_NO_F32DN
#define PF_TU_NO_F32D1
#endif
#if defined(PF_TU_NO_F64) || defined(PF_TU_NO_F32DN) || defined(PF_TU_NO_F32D1) || defined(PF_TU_F64_ONLY) || defined(PF_TU_F32DN_ONLY) || defined(PF_TU_F32D1_ONLY)
#define PF_TU_SPLIT  // a split build: the column kernels live in their own units (PF_TU_COLUMN_F32 / _F64)
#endif
#if defined(PF_TU_F64_ONLY) || defined(PF_TU_F32DN_ONLY) || defined(PF_TU_F32D1_ONLY)
#define PF_TU_NO_API
#endif
#ifndef PF_TU_NO_API
#ifndef PF_SOURCE_SHA256
#define PF_SOURCE_SHA256 "unknown"
#endif
#define PF_STR2(x) #x
#define PF_STR(x) PF_STR2(x)
namespace pf { thread_local ThetaFold* tls_theta_fold = nullptr; }
extern "C" const char* pf_version(void) { return "pfamd 0.2.0 (gfx950) abi " PF_STR(PF_ABI_VERSION) " src:" PF_SOURCE_SHA256; }
extern "C" int pf_abi_version(void) { return PF_ABI_VERSION; }

extern "C" const char* pf_error_string(int code) {
    switch (code) {
        case PF_OK: return "ok";
        case PF_EINVAL: return "invalid argument";
        case PF_EWORKSPACE: return "workspace too small";
        case PF_EUNSUPPORTED: return "unsupported configuration";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}



#ifdef PF_DEVTOOLS  // (the instrumented build only - tools/pmc_stages.py --build: where the workspace keeps the development timestamps)
extern "C" int pf_debug_offset(int64_t N, int64_t B, size_t* off) {
    if (!off || bad_shape(N, B)) return PF_EINVAL;
    *off = make_ws(make_geom(N, B), PF_MAXD).off_dbg;
    return PF_OK;
}
#endif

extern "C" int pf_workspace_bytes(int64_t N, int64_t B, int64_t D, size_t* bytes) {
    if (!bytes || bad_shape(N, B) || D < 1 || D > PF_MAXD) return PF_EINVAL;
    *bytes = ws_bound(N, B, PF_MAXD);
    return PF_OK;
}

#define PF_DISPATCH_T_VEC(dtype, vec, CALL)                       \
    if (dtype == PF_F32) {                                        \
        if (vec == 4) { CALL(float, 4) } else { CALL(float, 1) }  \
    } else if (dtype == PF_F64) {                                 \
        if (vec == 4) { CALL(double, 4) } else { CALL(double, 1) }\
    } else return PF_EINVAL;

extern "C" int pf_normalize(void* logw, void* W, void* lse, void* ess, int64_t N, int64_t B, int dtype, void* ws,
                            size_t ws_bytes, void* stream) {
    if (!logw || !ws || bad_shape(N, B)) return PF_EINVAL;
    const Geom g = make_geom(N, B);
    const WsLayout wl = make_ws(g, PF_MAXD);
    if (ws_bytes < wl.total) return PF_EWORKSPACE;
    double* part = (double*)((char*)ws + wl.off_part);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(g.tiles, g.B);
#define CALL(T, V)                                                                                                   \
    if (g.tiles == 1) { /* one tile per column: both passes in one launch */                                          \
        hipLaunchKernelGGL((k_normalize_one_tile<T, V>), grid, dim3(PF_BLOCK), 0, st, (T*)logw, (T*)W, (T*)lse,      \
                           (T*)ess, part, g);                                                                        \
    } else {                                                                                                         \
        hipLaunchKernelGGL((k_reduce_logw<T, V>), grid, dim3(PF_BLOCK), 0, st, (T*)logw, 1, (const uint8_t*)nullptr, \
                           part, g);                                                                                 \
        hipLaunchKernelGGL((k_normalize_write<T, V>), grid, dim3(PF_BLOCK), 0, st, (const T*)logw, (T*)W, (T*)lse,   \
                           (T*)ess, (const double*)part, g);                                                         \
    }
    PF_DISPATCH_T_VEC(dtype, g.vec, CALL)
#undef CALL
    PF_CHECK_LAUNCH();
    return PF_OK;
}

// pf_systematic without a cdf (k_chunk_scan + k_chunk_search): columns of several tiles of whole 4-vectors, one u per column, a
// tile's chunk bases within one workgroup's reach, and a grid the closed-form inversion is exact for (float: N <= 2^22)
static inline bool cdf_free_applies(const Geom& g, int dtype, int u_per_elem) {
    if (dtype != PF_F32 && dtype != PF_F64) return false;
    return g.tiles > 1 && g.vec == 4 && !u_per_elem && g.rounds_per_tile * (g.round_elems / PF_CHUNK) <= PF_BLOCK &&
           g.N < ((int64_t)1 << 31) - 4096 && !(dtype == PF_F32 && g.N > ((int64_t)1 << 22));
}
extern "C" int pf_systematic_cdf_free(int64_t N, int64_t B, int dtype, int u_per_element, int* yes) {
    if (!yes || bad_shape(N, B)) return PF_EINVAL;
    *yes = cdf_free_applies(make_geom(N, B), dtype, u_per_element) ? 1 : 0;
    return PF_OK;
}

static int systematic_impl(void* src, bool from_w, const void* u, int u_per_elem, const void* v, int multinomial, uint64_t seed,
                           uint32_t step, const uint8_t* colmask, void* cdf, int32_t* idx, int64_t N, int64_t B,
                           int dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!src || !idx || !ws || bad_shape(N, B) || (!multinomial && !u)) return PF_EINVAL;
    const Geom g = make_geom(N, B);
    const WsLayout wl = make_ws(g, PF_MAXD);
    if (ws_bytes < wl.total) return PF_EWORKSPACE;
    double* part = (double*)((char*)ws + wl.off_part);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(g.tiles, g.B);
    if (!cdf) {  // no cdf wanted: the two-launch form where it applies (pf_systematic_cdf_free), nothing else
        if (multinomial || !cdf_free_applies(g, dtype, u_per_elem)) return PF_EINVAL;
        double* cb = (double*)((char*)ws + wl.off_ctab);
        const int nchunks = (int)((N + PF_CHUNK - 1) / PF_CHUNK);
#ifdef PF_DEVTOOLS  // (the instrumented build: cycle stamps of the middle workgroup of column 0, tools/chunk_search_stages.py)
#define PF_CHUNK_DBG , (unsigned long long*)((char*)ws + wl.off_dbg)
#else
#define PF_CHUNK_DBG
#endif
#define PF_CHUNK_CALL(T, FW)                                                                                                 \
    hipLaunchKernelGGL((k_chunk_scan<T, FW>), grid, dim3(PF_BLOCK), 0, st, (T*)src, colmask, part, cb, g, nchunks);           \
    hipLaunchKernelGGL((k_chunk_search<T, FW>), grid, dim3(PF_BLOCK), 0, st, (const T*)src, (const T*)u, colmask,             \
                       (const double*)part, (const double*)cb, idx, g, nchunks PF_CHUNK_DBG);
        if (dtype == PF_F32) {
            if (from_w) { PF_CHUNK_CALL(float, true) } else { PF_CHUNK_CALL(float, false) }
        } else {
            if (from_w) { PF_CHUNK_CALL(double, true) } else { PF_CHUNK_CALL(double, false) }
        }
#undef PF_CHUNK_CALL
#undef PF_CHUNK_DBG
        PF_CHECK_LAUNCH();
        return PF_OK;
    }
    void* tree = multinomial ? (void*)((char*)ws + wl.off_tree) : nullptr;  // (the iid draws' 16-ary search tables, written by the scan)
#define CALL_MN(T, V, MN)                                                                                            \
    if (g.tiles == 1) { /* one tile per column: record -> scan -> ancestors in one launch */                          \
        if (from_w)                                                                                                  \
            hipLaunchKernelGGL((k_resample_one_tile<T, V, true, MN>), grid, dim3(PF_BLOCK), 0, st, (T*)src,          \
                               (const T*)u, u_per_elem, (const T*)v, seed, step, colmask, (T*)cdf, idx, part, g,     \
                               (T*)tree);                                                                            \
        else                                                                                                         \
            hipLaunchKernelGGL((k_resample_one_tile<T, V, false, MN>), grid, dim3(PF_BLOCK), 0, st, (T*)src,         \
                               (const T*)u, u_per_elem, (const T*)v, seed, step, colmask, (T*)cdf, idx, part, g,     \
                               (T*)tree);                                                                            \
    } else {                                                                                                         \
        if (from_w) {                                                                                                \
            hipLaunchKernelGGL((k_tile_sum<T, V>), grid, dim3(PF_BLOCK), 0, st, (const T*)src, colmask, part, g);    \
            hipLaunchKernelGGL((k_scan<T, V, true>), grid, dim3(PF_BLOCK), 0, st, (const T*)src, (T*)cdf, colmask,   \
                               (const double*)part, g, (T*)tree);                                                    \
        } else {                                                                                                     \
            hipLaunchKernelGGL((k_reduce_logw<T, V>), grid, dim3(PF_BLOCK), 0, st, (T*)src, 1, colmask, part, g);    \
            hipLaunchKernelGGL((k_scan<T, V, false>), grid, dim3(PF_BLOCK), 0, st, (const T*)src, (T*)cdf, colmask,  \
                               (const double*)part, g, (T*)tree);                                                    \
        }                                                                                                            \
        hipLaunchKernelGGL((k_search<T, V, MN>), grid, dim3(PF_BLOCK), 0, st, (const T*)cdf, (const T*)u,            \
                           u_per_elem, (const T*)v, seed, step, colmask, idx, g, /*force_search*/ 0, (const T*)tree);\
    }
#define CALL(T, V) if (multinomial) { CALL_MN(T, V, true) } else { CALL_MN(T, V, false) }
    PF_DISPATCH_T_VEC(dtype, g.vec, CALL)
#undef CALL
#undef CALL_MN
    PF_CHECK_LAUNCH();
    return PF_OK;
}

extern "C" int pf_systematic(const void* W, const void* u, int u_per_element, const uint8_t* colmask, void* cdf,
                             int32_t* idx, int64_t N, int64_t B, int dtype, void* ws, size_t ws_bytes, void* stream) {
    return systematic_impl((void*)W, true, u, u_per_element, nullptr, 0, 0, 0, colmask, cdf, idx, N, B, dtype, ws,
                           ws_bytes, stream);
}

extern "C" int pf_systematic_logw(void* logw, const void* u, int u_per_element, const uint8_t* colmask, void* cdf,
                                  int32_t* idx, int64_t N, int64_t B, int dtype, void* ws, size_t ws_bytes,
                                  void* stream) {
    return systematic_impl(logw, false, u, u_per_element, nullptr, 0, 0, 0, colmask, cdf, idx, N, B, dtype, ws,
                           ws_bytes, stream);
}

extern "C" int pf_multinomial(const void* W, const void* v, uint64_t seed, uint32_t step, const uint8_t* colmask,
                              void* cdf, int32_t* idx, int64_t N, int64_t B, int dtype, void* ws, size_t ws_bytes,
                              void* stream) {
    return systematic_impl((void*)W, true, nullptr, 0, v, 1, seed, step, colmask, cdf, idx, N, B, dtype, ws, ws_bytes,
                           stream);
}

static inline int ew_blocks(int64_t N) {
    int64_t nb = (N + PF_BLOCK - 1) / PF_BLOCK;
    return (int)(nb > 2048 ? 2048 : nb);
}

extern "C" int pf_gather(const void* x, const int32_t* idx, const uint8_t* colmask, void* out, int64_t N, int64_t B,
                         int64_t D, int dtype, void* stream) {
    if (!x || !idx || !out || bad_shape(N, B) || D < 1 || x == out) return PF_EINVAL;
    const dim3 grid(ew_blocks(N), (int)B);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PF_F32)
        hipLaunchKernelGGL((k_gather<float>), grid, dim3(PF_BLOCK), 0, st, (const float*)x, idx, colmask, (float*)out, N, (int)B, (int)D);
    else if (dtype == PF_F64)
        hipLaunchKernelGGL((k_gather<double>), grid, dim3(PF_BLOCK), 0, st, (const double*)x, idx, colmask, (double*)out, N, (int)B, (int)D);
    else return PF_EINVAL;
    PF_CHECK_LAUNCH();
    return PF_OK;
}

static int columns_move(const void* src, const int64_t* idx, const uint8_t* mask, void* dst, int64_t N, int64_t B,
                        int64_t planes, int elem_bytes, void* stream) {
    if (!src || !dst || bad_shape(N, B) || planes < 1 || planes > 65535 || (elem_bytes != 4 && elem_bytes != 8)) return PF_EINVAL;
    const int64_t col_bytes = N * elem_bytes;
    const dim3 grid((unsigned)((col_bytes + PF_COLCHUNK - 1) / PF_COLCHUNK), (unsigned)B, (unsigned)planes);
    hipStream_t st = (hipStream_t)stream;
    const uintptr_t al = (uintptr_t)src | (uintptr_t)dst | (uintptr_t)col_bytes;
    const char* s = (const char*)src;
    char* d = (char*)dst;
    if ((al & 15) == 0) hipLaunchKernelGGL((k_columns_move<pf_v4u>), grid, dim3(PF_BLOCK), 0, st, s, idx, mask, d, col_bytes, (int)B);
    else if ((al & 7) == 0) hipLaunchKernelGGL((k_columns_move<pf_v2u>), grid, dim3(PF_BLOCK), 0, st, s, idx, mask, d, col_bytes, (int)B);
    else hipLaunchKernelGGL((k_columns_move<uint32_t>), grid, dim3(PF_BLOCK), 0, st, s, idx, mask, d, col_bytes, (int)B);
    PF_CHECK_LAUNCH();
    return PF_OK;
}

extern "C" int pf_columns_gather(const void* src, const int64_t* idx, void* dst, int64_t N, int64_t B, int64_t planes,
                                 int elem_bytes, void* stream) {
    if (!idx || src == dst) return PF_EINVAL;  // out of place only: a gather may read columns it has already overwritten
    return columns_move(src, idx, nullptr, dst, N, B, planes, elem_bytes, stream);
}

extern "C" int pf_columns_exchange(void* dst, const void* src, const uint8_t* mask, int64_t N, int64_t B, int64_t planes,
                                   int elem_bytes, void* stream) {
    if (!mask) return PF_EINVAL;
    return columns_move(src, nullptr, mask, dst, N, B, planes, elem_bytes, stream);
}

extern "C" int pf_loglik(const void* v, const void* W, void* out, int64_t N, int64_t B, int dtype, void* ws,
                         size_t ws_bytes, void* stream) {
    if (!v || !out || !ws || bad_shape(N, B)) return PF_EINVAL;
    const Geom g = make_geom(N, B);
    const WsLayout wl = make_ws(g, PF_MAXD);
    if (ws_bytes < wl.total) return PF_EWORKSPACE;
    double* part = (double*)((char*)ws + wl.off_part);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(g.tiles, g.B);
    if (dtype == PF_F32) {
        hipLaunchKernelGGL((k_loglik_part<float>), grid, dim3(PF_BLOCK), 0, st, (const float*)v, (const float*)W, part, g);
        hipLaunchKernelGGL((k_loglik_final<float>), dim3(g.B), dim3(PF_BLOCK), 0, st, (const double*)part, (float*)out, g);
    } else if (dtype == PF_F64) {
        hipLaunchKernelGGL((k_loglik_part<double>), grid, dim3(PF_BLOCK), 0, st, (const double*)v, (const double*)W, part, g);
        hipLaunchKernelGGL((k_loglik_final<double>), dim3(g.B), dim3(PF_BLOCK), 0, st, (const double*)part, (double*)out, g);
    } else return PF_EINVAL;
    PF_CHECK_LAUNCH();
    return PF_OK;
}

extern "C" int pf_moments(const void* x, const void* W, void* mean, void* var, int64_t N, int64_t B, int64_t D,
                          int dtype, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !W || !mean || !var || !ws || bad_shape(N, B) || D < 1 || D > PF_MAXD) return PF_EINVAL;
    const Geom g = make_geom(N, B);
    const WsLayout wl = make_ws(g, PF_MAXD);
    if (ws_bytes < wl.total) return PF_EWORKSPACE;
    double* part = (double*)((char*)ws + wl.off_part);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(g.tiles, g.B);
    if (dtype == PF_F32) {
        hipLaunchKernelGGL((k_moments_part<float>), grid, dim3(PF_BLOCK), 0, st, (const float*)x, (const float*)W, part, g, (int)D);
        hipLaunchKernelGGL((k_moments_final<float>), dim3(g.B), dim3(PF_BLOCK), 0, st, (const double*)part, (float*)mean, (float*)var, g, (int)D);
    } else if (dtype == PF_F64) {
        hipLaunchKernelGGL((k_moments_part<double>), grid, dim3(PF_BLOCK), 0, st, (const double*)x, (const double*)W, part, g, (int)D);
        hipLaunchKernelGGL((k_moments_final<double>), dim3(g.B), dim3(PF_BLOCK), 0, st, (const double*)part, (double*)mean, (double*)var, g, (int)D);
    } else return PF_EINVAL;
    PF_CHECK_LAUNCH();
    return PF_OK;
}



#define PF_DISPATCH_T_D(dtype, D, CALL)                                                     \
    if (dtype == PF_F32) {                                                                  \
        if (D == 1) { CALL(float, 1) } else if (D == 2) { CALL(float, 2) } else { CALL(float, 3) }      \
    } else if (dtype == PF_F64) {                                                           \
        if (D == 1) { CALL(double, 1) } else if (D == 2) { CALL(double, 2) } else { CALL(double, 3) }   \
    } else return PF_EINVAL;

extern "C" int pf_pre_weight(const pf_model* model, int proposal, const void* x, const void* y, int64_t y_rows,
                             void* out, int64_t N, int64_t B, int dtype, void* stream) {
    int rc = check_model(model);
    if (rc) return rc;
    if (!x || !y || !out || bad_shape(N, B) || (y_rows != 1 && y_rows != B)) return PF_EINVAL;
    if (proposal == PF_PROP_LGO && model->obs_kind != PF_OBS_LINEAR) return PF_EUNSUPPORTED;
    const ModelDesc md = to_desc(model);
    const dim3 grid(ew_blocks(N), (int)B);
    hipStream_t st = (hipStream_t)stream;
#define CALL(T, DD)                                                                                                  \
    hipLaunchKernelGGL((k_pre_weight<T, DD>), grid, dim3(PF_BLOCK), 0, st, md, (const T*)model->params, proposal,    \
                       (const T*)x, (const T*)y, (int)y_rows, (T*)out, N, (int)B);
    PF_DISPATCH_T_D(dtype, model->dim, CALL)
#undef CALL
    PF_CHECK_LAUNCH();
    return PF_OK;
}

extern "C" int pf_sample_and_weight(const pf_model* model, int proposal, int weigh, const void* x, const void* y,
                                    int64_t y_rows, const void* z, uint64_t seed, uint32_t step, void* x_out,
                                    void* w_out, int64_t N, int64_t B, int dtype, void* stream) {
    int rc = check_model(model);
    if (rc) return rc;
    if (!x || !x_out || bad_shape(N, B) || (weigh && (!y || !w_out)) || (y_rows != 1 && y_rows != B)) return PF_EINVAL;
    if (proposal == PF_PROP_LGO && model->obs_kind != PF_OBS_LINEAR) return PF_EUNSUPPORTED;
    const ModelDesc md = to_desc(model);
    const dim3 grid(ew_blocks(N), (int)B);
    hipStream_t st = (hipStream_t)stream;
#define CALL(T, DD)                                                                                                  \
    hipLaunchKernelGGL((k_sample_and_weight<T, DD>), grid, dim3(PF_BLOCK), 0, st, md, (const T*)model->params,       \
                       proposal, weigh, (const T*)x, (const T*)y, (int)y_rows, (const T*)z, seed, step, (T*)x_out,   \
                       (T*)w_out, N, (int)B);
    PF_DISPATCH_T_D(dtype, model->dim, CALL)
#undef CALL
    PF_CHECK_LAUNCH();
    return PF_OK;
}

extern "C" int pf_initial_sample(const double* m0, const double* s0, const void* z, uint64_t seed, void* x, int64_t N,
                                 int64_t B, int64_t D, int dtype, void* stream) {
    if (!m0 || !s0 || !x || bad_shape(N, B) || D < 1 || D > PF_MAXD) return PF_EINVAL;
    double m[3] = {0, 0, 0}, s[3] = {0, 0, 0};
    for (int d = 0; d < D; ++d) { m[d] = m0[d]; s[d] = s0[d]; }
    const dim3 grid(ew_blocks(N), (int)B);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PF_F32)
        hipLaunchKernelGGL((k_initial_sample<float>), grid, dim3(PF_BLOCK), 0, st, m[0], m[1], m[2], s[0], s[1], s[2], (const float*)z, seed, (float*)x, N, (int)B, (int)D);
    else if (dtype == PF_F64)
        hipLaunchKernelGGL((k_initial_sample<double>), grid, dim3(PF_BLOCK), 0, st, m[0], m[1], m[2], s[0], s[1], s[2], (const double*)z, seed, (double*)x, N, (int)B, (int)D);
    else return PF_EINVAL;
    PF_CHECK_LAUNCH();
    return PF_OK;
}


extern "C" int pf_initial_sample_cols(const void* m0, int64_t m0_stride_b, int64_t m0_stride_d, const void* s0, int64_t s0_stride_b,
                                      int64_t s0_stride_d, const void* z, uint64_t seed, void* x, int64_t N, int64_t B, int64_t D,
                                      int dtype, void* stream) {
    if (!m0 || !s0 || !x || bad_shape(N, B) || D < 1 || D > PF_MAXD || m0_stride_b < 0 || m0_stride_d < 0 || s0_stride_b < 0 ||
        s0_stride_d < 0)
        return PF_EINVAL;
    const dim3 grid(ew_blocks(N), (int)B);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PF_F32)
        hipLaunchKernelGGL((k_initial_sample_cols<float>), grid, dim3(PF_BLOCK), 0, st, (const float*)m0, m0_stride_b, m0_stride_d,
                           (const float*)s0, s0_stride_b, s0_stride_d, (const float*)z, seed, (float*)x, N, (int)B, (int)D);
    else if (dtype == PF_F64)
        hipLaunchKernelGGL((k_initial_sample_cols<double>), grid, dim3(PF_BLOCK), 0, st, (const double*)m0, m0_stride_b, m0_stride_d,
                           (const double*)s0, s0_stride_b, s0_stride_d, (const double*)z, seed, (double*)x, N, (int)B, (int)D);
    else return PF_EINVAL;
    PF_CHECK_LAUNCH();
    return PF_OK;
}

extern "C" int pf_observed_flags(const void* y, int64_t steps, int64_t row_elems, int dtype, uint8_t* out, void* stream) {
    if (!y || !out || steps < 0 || row_elems < 1 || steps > 0x7fffffff) return PF_EINVAL;
    if (steps == 0) return PF_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PF_F32) hipLaunchKernelGGL((k_observed_flags<float>), dim3((unsigned)steps), dim3(PF_WAVE), 0, st, (const float*)y, row_elems, out);
    else if (dtype == PF_F64) hipLaunchKernelGGL((k_observed_flags<double>), dim3((unsigned)steps), dim3(PF_WAVE), 0, st, (const double*)y, row_elems, out);
    else return PF_EINVAL;
    PF_CHECK_LAUNCH();
    return PF_OK;
}

// ---- theta-level kernels (pf_theta.hpp) ---------------------------------------------------------------------------------
#include "pf_theta.hpp"
extern "C" int pf_theta_fit(const void* values, const void* logw, int64_t B, int32_t P, double scale, int dtype, void* mean,
                            void* chol, void* stream) {
    if (!values || !mean || !chol || B < 1 || P < 1 || P > PF_THETA_MAXP) return PF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PF_F32)
        hipLaunchKernelGGL((k_theta_fit<float>), dim3(1), dim3(PF_BLOCK), 0, st, (const float*)values, (const float*)logw, B, (int)P, scale,
                           (float*)mean, (float*)chol);
    else if (dtype == PF_F64)
        hipLaunchKernelGGL((k_theta_fit<double>), dim3(1), dim3(PF_BLOCK), 0, st, (const double*)values, (const double*)logw, B, (int)P,
                           scale, (double*)mean, (double*)chol);
    else return PF_EINVAL;
    PF_CHECK_LAUNCH();
    return PF_OK;
}

extern "C" int pf_theta_propose(const pf_theta_priors* priors, const void* mean, const void* chol, const void* eps, int64_t B,
                                int dtype, void* u_out, void* const* x_out, void* prior_out, void* stream) {
    if (!priors || !mean || !chol || !eps || !u_out || !x_out || !prior_out || B < 1 || priors->P < 1 || priors->P > PF_THETA_MAXP)
        return PF_EINVAL;
    ThetaPriors pr;
    ThetaOut out;
    pr.P = priors->P;
    for (int p = 0; p < PF_THETA_MAXP; ++p) {
        pr.kind[p] = p < pr.P ? priors->kind[p] : 0;
        pr.a[p] = p < pr.P ? priors->a[p] : 0.0;
        pr.b[p] = p < pr.P ? priors->b[p] : 1.0;
        out.x[p] = p < pr.P ? x_out[p] : nullptr;
        if (p < pr.P && (!x_out[p] || pr.kind[p] < 0 || pr.kind[p] > PF_PRIOR_UNIFORM)) return PF_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((B + PF_BLOCK - 1) / PF_BLOCK));
    if (dtype == PF_F32)
        hipLaunchKernelGGL((k_theta_propose<float>), grid, dim3(PF_BLOCK), 0, st, pr, (const float*)mean, (const float*)chol,
                           (const float*)eps, B, (float*)u_out, out, (float*)prior_out);
    else if (dtype == PF_F64)
        hipLaunchKernelGGL((k_theta_propose<double>), grid, dim3(PF_BLOCK), 0, st, pr, (const double*)mean, (const double*)chol,
                           (const double*)eps, B, (double*)u_out, out, (double*)prior_out);
    else return PF_EINVAL;
    PF_CHECK_LAUNCH();
    return PF_OK;
}

extern "C" int pf_theta_accept(const void* u_cur, const void* u_star, const void* mean_f, const void* chol_f, const void* mean_r,
                               const void* chol_r, const void* prior_cur, const void* prior_star, const void* ll_cur,
                               const void* ll_star, const void* unif, int64_t B, int32_t P, int dtype, void* log_acc,
                               uint8_t* accepted, void* rate, void* stream) {
    if (!u_cur || !u_star || !mean_f || !chol_f || !mean_r || !chol_r || !prior_cur || !prior_star || !ll_cur || !ll_star || !unif ||
        !log_acc || !accepted || !rate || B < 1 || P < 1 || P > PF_THETA_MAXP)
        return PF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define CALL(T)                                                                                                               \
    hipLaunchKernelGGL((k_theta_accept<T>), dim3(1), dim3(PF_BLOCK), 0, st, (const T*)u_cur, (const T*)u_star, (const T*)mean_f, \
                       (const T*)chol_f, (const T*)mean_r, (const T*)chol_r, (const T*)prior_cur, (const T*)prior_star,         \
                       (const T*)ll_cur, (const T*)ll_star, (const T*)unif, B, (int)P, (T*)log_acc, accepted, (T*)rate);
    if (dtype == PF_F32) { CALL(float) } else if (dtype == PF_F64) { CALL(double) } else return PF_EINVAL;
#undef CALL
    PF_CHECK_LAUNCH();
    return PF_OK;
}

extern "C" int pf_theta_path(const void* w0, const void* ll, int64_t n, int64_t B, int dtype, void* w_path, void* stats,
                             void* host_rows, uint64_t seq, const int32_t* status, void* stream) {
    if (!w0 || !ll || !w_path || !stats || B < 1 || n < 0 || n > 65535 || ((uintptr_t)host_rows & 7) != 0) return PF_EINVAL;
    if (n == 0) return PF_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PF_F32)
        hipLaunchKernelGGL((k_theta_path<float>), dim3((unsigned)n), dim3(PF_BLOCK), 0, st, (const float*)w0, (const float*)ll, B,
                           (float*)w_path, (float*)stats, (double*)host_rows, (unsigned long long)seq, (float*)nullptr, (const int*)status, 1);
    else if (dtype == PF_F64)
        hipLaunchKernelGGL((k_theta_path<double>), dim3((unsigned)n), dim3(PF_BLOCK), 0, st, (const double*)w0, (const double*)ll, B,
                           (double*)w_path, (double*)stats, (double*)host_rows, (unsigned long long)seq, (double*)nullptr, (const int*)status, 1);
    else return PF_EINVAL;
    PF_CHECK_LAUNCH();
    return PF_OK;
}

extern "C" int pf_theta_step(void* w, const void* ll, int64_t B, int dtype, void* stats, void* host_slot, uint64_t seq, void* acc,
                             const int32_t* status, void* stream) {
    if (!w || !ll || !stats || B < 1 || ((uintptr_t)host_slot & 7) != 0) return PF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PF_F32)
        hipLaunchKernelGGL((k_theta_path<float>), dim3(1), dim3(PF_BLOCK), 0, st, (const float*)w, (const float*)ll, B, (float*)w,
                           (float*)stats, (double*)host_slot, (unsigned long long)seq, (float*)acc, (const int*)status);
    else if (dtype == PF_F64)
        hipLaunchKernelGGL((k_theta_path<double>), dim3(1), dim3(PF_BLOCK), 0, st, (const double*)w, (const double*)ll, B, (double*)w,
                           (double*)stats, (double*)host_slot, (unsigned long long)seq, (double*)acc, (const int*)status);
    else return PF_EINVAL;
    PF_CHECK_LAUNCH();
    return PF_OK;
}

extern "C" int pf_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return PF_EINVAL;
    *out = nullptr;
    // coherent (fine-grained) + mapped: a device store with system scope is visible to a polling host thread while the stream runs on
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocCoherent | hipHostMallocMapped);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return (int)e;  // (a HIP error code, like a failed launch)
    }
    memset(*out, 0, bytes);
    return PF_OK;
}

extern "C" int pf_host_free(void* p) {
    if (!p) return PF_OK;
    const hipError_t e = hipHostFree(p);
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess ? PF_OK : (int)e;
}

extern "C" int pf_theta_resample(const void* logw, int64_t B, double u, int dtype, int64_t* ancestors, void* cdf_scratch,
                                 void* stream) {
    if (!logw || !ancestors || !cdf_scratch || B < 1 || !(u >= 0.0 && u <= 1.0)) return PF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PF_F32)
        hipLaunchKernelGGL((k_theta_resample<float>), dim3(1), dim3(PF_BLOCK), 0, st, (const float*)logw, B, u, ancestors, (float*)cdf_scratch);
    else if (dtype == PF_F64)
        hipLaunchKernelGGL((k_theta_resample<double>), dim3(1), dim3(PF_BLOCK), 0, st, (const double*)logw, B, u, ancestors, (double*)cdf_scratch);
    else return PF_EINVAL;
    PF_CHECK_LAUNCH();
    return PF_OK;
}

extern "C" int pf_theta_ess(const void* logw, int64_t rows, int64_t B, int dtype, void* out, void* stream) {
    if (!logw || !out || B < 1 || rows < 0 || rows > 0x7fffffff) return PF_EINVAL;
    if (rows == 0) return PF_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)rows);
    if (dtype == PF_F32) hipLaunchKernelGGL((k_theta_ess<float>), grid, dim3(PF_BLOCK), 0, st, (const float*)logw, B, (float*)out);
    else if (dtype == PF_F64) hipLaunchKernelGGL((k_theta_ess<double>), grid, dim3(PF_BLOCK), 0, st, (const double*)logw, B, (double*)out);
    else return PF_EINVAL;
    PF_CHECK_LAUNCH();
    return PF_OK;
}



// ---- smoothing ---------------------------------------------------------------------------------------------------------
extern "C" int pf_smooth_fixed_lag(const void* x_hist, const int32_t* anc_hist, void* out, int64_t S, int64_t N, int64_t B,
                                   int64_t D, int dtype, void* stream) {
    if (!x_hist || !anc_hist || !out || S < 1 || bad_shape(N, B) || D < 1 || D > PF_MAXD) return PF_EINVAL;
    const dim3 grid(ew_blocks(N), (int)B);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PF_F32)
        hipLaunchKernelGGL((k_trace_ancestors<float>), grid, dim3(PF_BLOCK), 0, st, (const float*)x_hist, anc_hist, (float*)out, S, N, (int)B, (int)D);
    else if (dtype == PF_F64)
        hipLaunchKernelGGL((k_trace_ancestors<double>), grid, dim3(PF_BLOCK), 0, st, (const double*)x_hist, anc_hist, (double*)out, S, N, (int)B, (int)D);
    else return PF_EINVAL;
    PF_CHECK_LAUNCH();
    return PF_OK;
}

extern "C" int pf_smooth_ffbs(const pf_model* model, const void* x_hist, const void* logw_hist, const void* x_last,
                              const void* u, uint64_t seed, void* out, int64_t S, int64_t N, int64_t B, int dtype,
                              void* stream) {
    if (!model || !x_hist || !logw_hist || !x_last || !out || S < 1 || bad_shape(N, B)) return PF_EINVAL;
    int rc = check_model(model);
    if (rc) return rc;
    const ModelDesc md = to_desc(model);
    const dim3 grid((unsigned)((N + PF_BLOCK - 1) / PF_BLOCK), (int)B);
    hipStream_t st = (hipStream_t)stream;
#define CALL(T, DD)                                                                                                         \
    hipLaunchKernelGGL((k_ffbs<T, DD>), grid, dim3(PF_BLOCK), 0, st, md, (const T*)model->params, (const T*)x_hist,          \
                       (const T*)logw_hist, (const T*)x_last, (const T*)u, seed, (T*)out, S, N, (int)B);
    PF_DISPATCH_T_D(dtype, model->dim, CALL)
#undef CALL
    PF_CHECK_LAUNCH();
    return PF_OK;
}

// ---- test support ----------------------------------------------------------------------------------------------------
extern "C" int pf_debug_draw_normals(uint64_t seed, uint32_t step0, int64_t n_steps, void* out, int64_t N, int64_t B,
                                     int64_t D, int dtype, void* stream) {
    if (!out || bad_shape(N, B) || D < 1 || D > PF_MAXD || n_steps < 1 || n_steps > 65535) return PF_EINVAL;
    const Geom g = make_geom(N, B);
    const dim3 grid((unsigned)((N + g.round_elems - 1) / g.round_elems), (unsigned)B, (unsigned)n_steps);
    hipStream_t st = (hipStream_t)stream;
#define CALL(T, DD, V) hipLaunchKernelGGL((k_debug_normals<T, DD, V>), grid, dim3(PF_BLOCK), 0, st, seed, step0, (T*)out, N, (int)B)
#define CALL_D(T, V) do { if (D == 1) CALL(T, 1, V); else if (D == 2) CALL(T, 2, V); else CALL(T, 3, V); } while (0)
    if (dtype == PF_F32) { if (g.vec == 4) CALL_D(float, 4); else CALL_D(float, 1); }
    else if (dtype == PF_F64) { if (g.vec == 4) CALL_D(double, 4); else CALL_D(double, 1); }
    else return PF_EINVAL;
#undef CALL_D
#undef CALL
    PF_CHECK_LAUNCH();
    return PF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// fused loop
// ---------------------------------------------------------------------------------------------------------------
#endif  // !PF_TU_NO_API

// Test support: which step-kernel instantiation each launch of the calling thread's most recent fused runs selected
// (pf_debug_launch_trace).  A per-thread ring, written on the host at launch time - nothing a kernel ever reads.
#define PF_TRACE_LEN 2048
#define PF_TRACE_FIELDS 10
struct LaunchTrace {
    int32_t rec[PF_TRACE_LEN][PF_TRACE_FIELDS];
    uint64_t count;
};
LaunchTrace& launch_trace();
static inline void trace_launch(int step, int tbytes, int d, int vec, int mode, int prop, int fast, int spec, int mk, int multi) {
    LaunchTrace& t = launch_trace();
    int32_t* r = t.rec[t.count % PF_TRACE_LEN];
    r[0] = step; r[1] = tbytes; r[2] = d; r[3] = vec; r[4] = mode; r[5] = prop; r[6] = fast; r[7] = spec; r[8] = mk; r[9] = multi;
    ++t.count;
}
#ifndef PF_TU_NO_API
LaunchTrace& launch_trace() {
    static thread_local LaunchTrace t = {};
    return t;
}
extern "C" int pf_debug_launch_trace(int32_t* out, int max_records) {
    if (!out || max_records < 0) return PF_EINVAL;
    const LaunchTrace& t = launch_trace();
    const uint64_t have = t.count < PF_TRACE_LEN ? t.count : PF_TRACE_LEN;
    const int n = (uint64_t)max_records < have ? max_records : (int)have;
    for (int i = 0; i < n; ++i)  // oldest of the last n first
        for (int f = 0; f < PF_TRACE_FIELDS; ++f) out[i * PF_TRACE_FIELDS + f] = t.rec[(t.count - n + i) % PF_TRACE_LEN][f];
    return n;
}
#endif

// the launch arguments every fused kernel shares, from the C ABI's argument block
template <typename T>
static FusedArgs<T> make_fused_args(const pf_filter_args* A, const Geom& g, const WsLayout& wl, int64_t t0) {
    FusedArgs<T> a;
    a.md = to_desc(&A->model);
    a.params = (const T*)A->model.params;
    a.filter = A->filter;
    a.proposal = A->proposal;
    a.resampler = A->resampler;
    a.g = g;
    a.thr_abs = A->ess_threshold * (double)A->N;
    a.logN = log((double)A->N);
    a.rcN = T(1) / T(A->N);
    a.seed = A->seed;
    a.seed_dev = (const uint64_t*)A->step_counter;
    a.x[0] = (T*)A->x[0];
    a.x[1] = (T*)A->x[1];
    a.logw[0] = (T*)A->logw[0];
    a.logw[1] = (T*)A->logw[1];
    a.anc = A->anc;
    a.anc_prev = nullptr;
    a.cdf = (T*)A->cdf;
    a.pos = (T*)A->pos;
    a.y = (const T*)A->y;
    a.y_rows = (int)A->y_rows;
    a.z_tape = (const T*)A->z_tape;
    a.u_tape = (const T*)A->u_tape;
    a.user_loc = (const T*)A->user_loc;
    a.user_scale = (const T*)A->user_scale;
    a.user_scale_percol = A->user_scale_per_column != 0 ? 1 : 0;
    a.user_dt = (T)A->user_dt;
    a.means = (T*)A->means;
    a.vars = (T*)A->vars;
    a.ll_steps = (T*)A->ll_steps;
    a.ll_total = (T*)A->ll_total;
    a.part = (double*)((char*)A->ws + wl.off_part);
    a.part_stride = (int64_t)wl.part_elems;
    a.stat = (ColStat*)((char*)A->ws + wl.off_stat);
    a.poison = (int32_t*)((char*)A->ws + wl.off_poison);
    a.dbg = (unsigned long long*)((char*)A->ws + wl.off_dbg);
    a.cpack = (T*)((char*)A->ws + wl.off_cpack);
    a.piv0 = (double*)((char*)A->ws + wl.off_piv0);
    a.ctab = (double*)((char*)A->ws + wl.off_ctab);
    a.ctab_stride = (int64_t)wl.ctab_elems;
    static_assert(PK_N == 24, "workspace layout reserves 24 slots per column record");
    a.finalize_only = 0;
    a.t0 = (int)t0;
    a.debug_cut = 0;
#ifdef PF_DEVTOOLS  // (the instrumented build of tools/pmc_stages.py: stage cuts / cycle stamps selected per process)
    if (const char* dc = getenv("PF_DEBUG_CUT")) a.debug_cut = atoi(dc);
#endif
    return a;
}

// Columns of fewer tiles than this keep their books inline (the column's last step workgroup, after its own work).  Since
// the bookkeepers are dispatched LAST (the grid's slowest axis is the tile index, see below) they cost nothing on the critical path and inline lost at every
// shape measured, single-tile columns included (1 024 x 8 192: 65.5 -> 59.1 us per step; 256 x 8 192 27.3 -> 22.1;
// profiles/r04c_step_kernel_book_inline_threshold_ab.txt): 1 = never.  (Round 2's rule was 8.)
#define PF_BOOK_INLINE_TILES 1
template <typename T, int D, int VEC, bool MULTI>
static int filter_run_impl(const pf_filter_args* A, const Geom& g, const WsLayout& wl, int64_t t0, int64_t n_steps,
                           int finalize, hipStream_t st, float* kernel_ms) {
    FusedArgs<T> a = make_fused_args<T>(A, g, wl, t0);
    const uint8_t* observed = A->observed;  // host array

    const dim3 grid_tiles(g.tiles, g.B), block(PF_BLOCK);
    // the step kernel: one workgroup per tile + one bookkeeper per column, dispatched after all step workgroups
    // (PF_BOOK_INLINE=0/1 overrides in the development build: 1 = the column's last step workgroup keeps the books)
    a.book_inline = g.tiles < PF_BOOK_INLINE_TILES ? 1 : 0;
#ifdef PF_DEVTOOLS
    if (const char* bi = getenv("PF_BOOK_INLINE")) a.book_inline = atoi(bi);  // (2: nobody keeps the books - timing experiments)
#endif
    // Grid (B, tiles + 1), x = the column: blocks are dispatched in linear order and a 2^20-particle step fills every
    // resident slot of the chip (1 024 = 4 per CU: 33 KB of LDS, 113 VGPRs) - with the tile index slowest the bookkeepers
    // (y == tiles) come after ALL step workgroups and fill slots as they free up; as block (tiles, b) of a (tiles + 1, B) grid
    // they sat between the columns, took slots first, and the last columns' step workgroups started 2 - 3 us late
    // (profiles/r04c_step_kernel_bookkeepers_last_ab.txt).  B = 1 is the same linear order either way.
    a.kmap = 0u;
    if (g.B == 1 && g.tiles >= 16 && (g.tiles & (g.tiles - 1)) == 0) {  // one column of 2^q tiles: an eighth of it per XCD
        unsigned q = 0;
        while ((1 << q) < g.tiles) ++q;
        a.kmap = 7u | ((q - 3u) << 8) | (3u << 16);
    }
    const dim3 grid(g.B, g.tiles + (a.book_inline ? 0 : 1));
    // neither flag array given: the flags are derived from y on the device, into the workspace (runs of <= PF_AUTO_FLAGS steps)
    const bool auto_flags = !A->observed && !A->observed_dev && n_steps > 0;
    // ... of ONE step on a shared observation row (the online move): every kernel looks at the row itself (FusedArgs::is_obs, -2) -
    // no launch that derives a flag byte
    const bool inline_flag = auto_flags && n_steps == 1 && A->y_rows == 1;
    uint8_t* const auto_fl = (uint8_t*)A->ws + wl.off_ctr + 64;
    const int64_t auto_row = A->y_rows * (int64_t)A->model.obs_dim;
    if (t0 == 0) {
        // fresh filter: no previous step to account for (column records + poison flags)
        // (a kernel, not hipMemsetAsync: captured as a memset node the fill stopped clearing these records after ~195
        // replays of the same executable graph on ROCm 7.2 - every log-likelihood of the run came back NaN, "poisoned" -
        // tools/graph_replays.py)
        const size_t words = (wl.off_ctr - wl.off_stat) / sizeof(uint32_t);  // (256-byte aligned regions)
        const unsigned zb = (unsigned)((words + PF_BLOCK - 1) / PF_BLOCK);
        if (auto_flags && !inline_flag)  // (the flags ride along: one launch)
            hipLaunchKernelGGL((k_zero_and_flags<T>), dim3(zb + (unsigned)n_steps), dim3(PF_BLOCK), 0, st,
                               (uint32_t*)((char*)A->ws + wl.off_stat), words, zb, (const T*)A->y + t0 * auto_row, auto_row, auto_fl);
        else
            hipLaunchKernelGGL((k_zero_words<uint32_t>), dim3(zb), dim3(PF_BLOCK), 0, st, (uint32_t*)((char*)A->ws + wl.off_stat), words);
    }
    // state history: slot pointers per launch (the kernels keep addressing "buffer step & 1 is read, the other written")
    const int64_t ring = A->ring >= 3 ? A->ring : 0;
    auto place = [&](int64_t t) {  // launch of step t: reads state t, writes state t + 1
        if (!ring) return;
        const int64_t rs = t % ring, wsl = (t + 1) % ring, bn = (int64_t)g.B * g.N;
        a.x[t & 1] = (T*)A->x[0] + rs * D * bn;
        a.x[(t + 1) & 1] = (T*)A->x[0] + wsl * D * bn;
        a.logw[t & 1] = (T*)A->logw[0] + rs * bn;
        a.logw[(t + 1) & 1] = (T*)A->logw[0] + wsl * bn;
        a.anc = A->anc + wsl * bn;
        a.anc_prev = A->anc + rs * bn;
    };
    place(t0);
    // partials of the incoming state (afterwards every step kernel leaves the partials of the state it wrote)
    a.step = (int)t0;
    const bool dev_flags = A->observed_dev != nullptr || auto_flags;  // the kernels read the flags themselves
    a.obs_dev = A->observed_dev;
    if (auto_flags && !inline_flag) {
        if (t0 != 0)
            hipLaunchKernelGGL((k_observed_flags<T>), dim3((unsigned)n_steps), dim3(PF_WAVE), 0, st, (const T*)A->y + t0 * auto_row, auto_row, auto_fl);
        a.obs_dev = auto_fl - t0;  // (indexed by the absolute step)
    }
    const int flag_mode = inline_flag ? -2 : -1;
    a.obs = n_steps > 0 ? (dev_flags ? flag_mode : (observed[t0] != 0)) : 0;
    a.obs_next = 0;
    // (pf_run_hints.resume: the previous call on this argument block ended with a SISR step that left the partials and local
    // scans of exactly this state in the workspace - the pass is redundant)
    // (an APF leaves them when its last step ran with pf_run_hints.prepare_next: the caller's promise)
    const bool resumed = A->hints.resume != 0 && t0 > 0 && A->ring < 3;
    const bool prepare_next = A->hints.prepare_next != 0 && A->filter == PF_FILTER_APF && !finalize && n_steps > 0;
    if (!resumed) hipLaunchKernelGGL((k_fused_reduce<T, D, VEC>), grid_tiles, block, 0, st, a);

    // ancestor stage of the step kernel: 0 inverted grid (systematic), 1 multinomial, 2 systematic by search - float
    // grids beyond 2^22 positions, where the closed form is not exact (PF_FORCE_SEARCH=1 selects it for testing)
    const bool force_search = A->hints.ancestor_search != 0;
    const int mode = (A->resampler == PF_RESAMPLE_MULTINOMIAL)
                         ? 1
                         : ((sizeof(T) == 4 && (g.N > ((int64_t)1 << 22) || force_search)) ? 2 : 0);
    // steady-state specialisation of this launch (float only: the double kernels are the parity path): see SPEC
    auto spec_of = [&]() -> int {
        if (sizeof(T) != 4 || a.z_tape || a.obs != 1) return 0;
        if (a.md.hid_kind == PF_HID_USER_AFFINE && a.filter == PF_FILTER_APF) return 0;  // (its steady state is not instantiated)
        if (a.filter == PF_FILTER_APF) return a.obs_next == 1 ? 1 : 0;
        return 2;
    };
    auto launch_step_as = [&](auto prop_c, auto fast_c) {
        constexpr int PROP = decltype(prop_c)::value;
        constexpr bool FAST = decltype(fast_c)::value;
        auto go = [&](auto mode_c, auto spec_c) {
            constexpr int MODE = decltype(mode_c)::value;
            constexpr int SPEC = decltype(spec_c)::value;
            auto launch = [&](auto mk_c) {
                constexpr int MK = decltype(mk_c)::value;
                trace_launch((int)a.step, (int)sizeof(T), D, VEC, MODE, PROP, FAST ? 1 : 0, SPEC, MK, MULTI ? 1 : 0);
                hipLaunchKernelGGL((k_fused_step<T, D, VEC, MODE, PROP, FAST, SPEC, MK, MULTI>), grid, block, 0, st, a);
            };
            // model kinds folded at compile time for the stochastic-volatility built-in (float runs; for Lorenz-63 the
            // same specialisation measured no gain)
            if constexpr (!FAST) {  // user-defined affine process: the parent's (loc, scale) come from the caller's planes
                if (a.md.hid_kind == PF_HID_USER_AFFINE) {
                    // (one step per run: no next step, so the APF steady-state specialisation never applies - not instantiated)
                    if constexpr (SPEC != 1) launch(std::integral_constant<int, 3>{});
                    return;
                }
            }
            if constexpr (sizeof(T) == 4 && !FAST && D == 1) {
                if (a.md.hid_kind == PF_HID_VERHULST_EM && a.md.obs_kind == PF_OBS_SV) return launch(std::integral_constant<int, 1>{});
            }
            if constexpr (sizeof(T) == 4 && !FAST && D == 3) {  // Lorenz-63
                if (a.md.hid_kind == PF_HID_LORENZ63_EM && a.md.obs_kind == PF_OBS_LINEAR)
                    return launch(std::integral_constant<int, 4>{});
            }
            if constexpr (sizeof(T) == 4 && FAST && D == 1) {  // shape of the one-step mean of the closed-form models
                if (a.md.hid_kind == PF_HID_SINE_EM) return launch(std::integral_constant<int, 2>{});
                return launch(std::integral_constant<int, 1>{});
            }
            launch(std::integral_constant<int, 0>{});
        };
        auto with_mode = [&](auto mode_c) {
            if constexpr (sizeof(T) == 4) {
                const int sp = spec_of();
                if (sp == 1) return go(mode_c, std::integral_constant<int, 1>{});
                // (the SISR specialisation spills in the multinomial variant and in the closed-form kernels: measured
                // slower than the generic kernel there)
                if (sp == 2) return go(mode_c, std::integral_constant<int, 2>{});
            }
            go(mode_c, std::integral_constant<int, 0>{});
        };
        if (mode == 0) with_mode(std::integral_constant<int, 0>{});
        else if (mode == 1) with_mode(std::integral_constant<int, 1>{});
        else if constexpr (sizeof(T) == 4) go(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
    };
    auto launch_step = [&]() {
        // scalar closed-form models: the proposal is a run-time switch inside one lean kernel (FAST); everything else gets
        // the proposal as a template constant so that Bootstrap runs do not carry the optimal proposal's registers
        bool fast = false;
        if constexpr (D == 1)
            fast = a.md.obs_kind == PF_OBS_LINEAR && a.md.hid_kind != PF_HID_VERHULST_EM && a.md.hid_kind != PF_HID_USER_AFFINE;
        if (fast) {
            if constexpr (D == 1) {
                if (a.proposal == PF_PROP_BOOTSTRAP) launch_step_as(std::integral_constant<int, PF_PROP_BOOTSTRAP>{}, std::true_type{});
                else launch_step_as(std::integral_constant<int, PF_PROP_LGO>{}, std::true_type{});
            }
        } else if (a.proposal == PF_PROP_BOOTSTRAP) {
            launch_step_as(std::integral_constant<int, PF_PROP_BOOTSTRAP>{}, std::false_type{});
        } else {
            launch_step_as(std::integral_constant<int, PF_PROP_LGO>{}, std::false_type{});
        }
    };
    hipEvent_t ev_loop[2] = {nullptr, nullptr};
    if (kernel_ms) {
        // measurement variant: HIP events on the caller's stream around the whole step loop
        for (auto& e : ev_loop)
            if (hipEventCreate(&e) != hipSuccess) return (int)hipGetLastError();
        (void)hipEventRecord(ev_loop[0], st);
    }
    for (int64_t s = 0; s < n_steps; ++s) {
        const int64_t t = t0 + s;
        a.step = (int)t;
        place(t);
        a.obs = dev_flags ? flag_mode : (observed[t] != 0);
        a.obs_next = (s + 1 < n_steps) ? (dev_flags ? -1 : (observed[t + 1] != 0)) : (prepare_next ? 1 : 0);
#ifdef PF_DEVTOOLS
        // stage cuts on ONE launch (the last but one step) when PF_DEBUG_CUT_AT_END is set: the state entering it is
        // valid, so per-dispatch PMC rows of that launch profile the stages on real data
        static const bool cut_at_end = getenv("PF_DEBUG_CUT_AT_END") != nullptr;
        const int cut_all = a.debug_cut;
        if (cut_at_end && cut_all > 0 && s != n_steps - 2) a.debug_cut = 0;
#endif
        launch_step();
#ifdef PF_DEVTOOLS
        a.debug_cut = cut_all;
#endif
    }
    if (kernel_ms) (void)hipEventRecord(ev_loop[1], st);
    if (finalize) {
        a.step = (int)(t0 + n_steps);
        a.obs = a.obs_next = 0;
        a.finalize_only = 1;
        hipLaunchKernelGGL((k_fused_book<T, D>), dim3(1, g.B), block, 0, st, a);
    }
    if (kernel_ms) {
        hipError_t se = hipStreamSynchronize(st);
        if (se != hipSuccess) return (int)se;
        float loop_ms = 0.f;
        (void)hipEventElapsedTime(&loop_ms, ev_loop[0], ev_loop[1]);
        for (auto& e : ev_loop) (void)hipEventDestroy(e);
        // one kernel per step: the in-sequence time of a step IS the step kernel's launch-to-launch duration
        const float per_step = n_steps > 0 ? loop_ms / (float)n_steps : 0.f;
        kernel_ms[0] = per_step;
        kernel_ms[1] = 0.f;  // (the planning kernel of earlier versions: folded into the step kernel's prologue)
        kernel_ms[2] = per_step;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PF_OK : (int)e;
}

// ---- the column-persistent route (pf_column.hpp): filters of a few hundred .. a few thousand particles -----------------
// One launch per run (per PFC_OBS_WORDS * 32 steps): no reduce / bookkeeping launches, no per-column records.
static inline int column_threads(int64_t N, int vec) {
    const int64_t need = (N + vec - 1) / vec;
    return (int)(((need + PF_WAVE - 1) / PF_WAVE) * PF_WAVE);
}
// Particles per lane on the column route: four - for scalar states also when N % 4 != 0 (the per-step geometry's
// one-particle lanes need four times the waves per filter, and past 256 of them the 1024-thread kernel: 1 000 x 333 ran
// 16.3 us per step against 5.4 for 1 000 x 400): the kernel handles the ragged last lane and the unaligned columns itself
// (pf_column.hpp: `ragged`).  D > 1 keeps the geometry's width.  The state's layout in HBM and the Philox addressing do not
// depend on it.  (One particle per lane for ALIGNED columns measured <= 16 % faster below 512 filters x 256 particles and
// up to 3x slower above: profiles/r03_column_vec1_vs_vec4.txt - not adopted.)
static inline int column_vec(const pf_filter_args* /*A*/, const Geom& /*g*/) { return 4; }
static inline size_t column_lds_bytes(int64_t N, int D, size_t tsize, int vec) {
    int64_t np2 = 64;
    while (np2 < N) np2 <<= 1;
    const int64_t NP = ((N + vec - 1) / vec) * vec;  // (the kernel's padded plane stride)
    const size_t planes = (((size_t)(np2 + PF_LB_PAD + (int64_t)D * NP) * tsize) + 15) & ~(size_t)15;  // (cdf + the search's pad | particle planes)
    return planes + sizeof(double) * (2 + 2 * (4 + 2 * D)) * PFC_MAXW + 16;  // scan records + the state's records (x 2)
}
// (measured, profiles/r03_column_route.txt: 1024 x 2048 runs 21 us per step here against 29 on the per-step route, 1024 x
// 4096 56 against 42 - sixteen waves of one workgroup issue-bound on one CU)
#define PF_COLUMN_MAX_N 2048
// Which runs take it: self-contained runs (finalize: the last state's row is flushed by the same call), no state history,
// a column that fits one workgroup.  pf_run_hints.route = PF_ROUTE_PER_STEP keeps everything on the per-step route (tests compare the two).
static inline bool column_eligible(const pf_filter_args* A, const Geom& g, int64_t n_steps, int finalize) {
    if (!finalize || n_steps < 1 || A->ring >= 3) return false;
    if (A->hints.route == PF_ROUTE_PER_STEP) return false;
    const int64_t max_n = A->hints.column_max_n > 0 ? A->hints.column_max_n : PF_COLUMN_MAX_N;
    if (A->N > max_n || column_threads(A->N, column_vec(A, g)) > 1024) return false;
    return column_lds_bytes(A->N, A->model.dim, A->dtype == PF_F64 ? 8 : 4, column_vec(A, g)) <= 64 * 1024;  // (the default dynamic-LDS limit)
}

template <typename T, int D, int VEC>
static int column_run_impl(const pf_filter_args* A, const Geom& g, const WsLayout& wl, int64_t t0, int64_t n_steps,
                           hipStream_t st, float* kernel_ms) {
    FusedArgs<T> a = make_fused_args<T>(A, g, wl, t0);
    const int nt = column_threads(A->N, VEC);
    const size_t lds = column_lds_bytes(A->N, D, sizeof(T), VEC);
    // observed flags: the host's (baked into the launch arguments), the caller's device array, or derived from y here
    const bool auto_flags = !A->observed && !A->observed_dev;
    const bool inline_y = auto_flags && n_steps == 1 && A->y_rows == 1;  // (the online move: the kernel looks at y itself)
    a.obs_dev = A->observed_dev;
    if (auto_flags && !inline_y) {
        uint8_t* fl = (uint8_t*)A->ws + wl.off_ctr + 64;
        const int64_t row = A->y_rows * (int64_t)A->model.obs_dim;
        hipLaunchKernelGGL((k_observed_flags<T>), dim3((unsigned)n_steps), dim3(PF_WAVE), 0, st, (const T*)A->y + t0 * row, row, fl);
        a.obs_dev = fl - t0;
    }
    hipEvent_t ev[2] = {nullptr, nullptr};
    if (kernel_ms) {
        for (auto& e : ev)
            if (hipEventCreate(&e) != hipSuccess) return (int)hipGetLastError();
        (void)hipEventRecord(ev[0], st);
    }
    for (int64_t done = 0; done < n_steps;) {
        ColumnRun r;
        r.t0 = (int)(t0 + done);
        r.n_steps = (int)((n_steps - done < 32 * PFC_OBS_WORDS) ? n_steps - done : 32 * PFC_OBS_WORDS);
        r.use_bits = (a.obs_dev == nullptr && !inline_y) ? 1 : 0;
        r.inline_y = inline_y ? 1 : 0;
        for (int w = 0; w < PFC_OBS_WORDS; ++w) r.obs_bits[w] = 0u;
        if (r.use_bits)
            for (int q = 0; q < r.n_steps; ++q)
                if (A->observed[r.t0 + q]) r.obs_bits[q >> 5] |= 1u << (q & 31);
        a.step = r.t0;
        bool spec_ok = false;  // a specialised instantiation exists for this run (see below)
        if constexpr (sizeof(T) == 4 && D == 1 && VEC == 4) {
            const int hk = A->model.hid_kind;
            const bool generic_only = A->hints.route == PF_ROUTE_COLUMN_GENERIC;
            const bool closed = A->model.obs_kind == PF_OBS_LINEAR && (hk == PF_HID_LINEAR || hk == PF_HID_SINE_EM || hk == PF_HID_OU) &&
                                (A->proposal == PF_PROP_BOOTSTRAP || A->proposal == PF_PROP_LGO);
            const bool sv = A->model.obs_kind == PF_OBS_SV && hk == PF_HID_VERHULST_EM && A->proposal == PF_PROP_BOOTSTRAP;
            spec_ok = !A->z_tape && !generic_only && (closed || sv);  // (any workgroup size: the 256- or the 1024-thread bound)
        }
        if constexpr (sizeof(T) == 4 && D == 3 && VEC == 4) {  // Lorenz-63
            spec_ok = A->N % VEC == 0 && !A->z_tape && A->hints.route != PF_ROUTE_COLUMN_GENERIC && A->model.obs_kind == PF_OBS_LINEAR &&
                      A->model.hid_kind == PF_HID_LORENZ63_EM && (A->proposal == PF_PROP_BOOTSTRAP || A->proposal == PF_PROP_LGO);
        }
        trace_launch(r.t0, (int)sizeof(T), D, VEC, A->resampler == PF_RESAMPLE_MULTINOMIAL ? 1 : 0, A->proposal, spec_ok ? 1 : 0,
                     /*SPEC*/ 9, spec_ok ? A->model.hid_kind : 0, 0);
        const bool user = A->model.hid_kind == PF_HID_USER_AFFINE;
        // columns of N % 4 != 0 particles (scalar states): the RAGGED instantiations
        auto with_rag = [&](auto&& f) {
            if constexpr (VEC == 4) {
                if (A->N % VEC != 0) return f(std::true_type{});
            }
            f(std::false_type{});
        };
        auto launch = [&](auto tpb_c) {
            constexpr int TPB = decltype(tpb_c)::value;
            with_rag([&](auto rag_c) {
                constexpr bool RAG = decltype(rag_c)::value;
                if (user) hipLaunchKernelGGL((k_fused_column<T, D, VEC, TPB, true, -1, -1, -1, RAG>), dim3(g.B), dim3(nt), lds, st, a, r);
                else hipLaunchKernelGGL((k_fused_column<T, D, VEC, TPB, false, -1, -1, -1, RAG>), dim3(g.B), dim3(nt), lds, st, a, r);
            });
        };
        // (a 512-thread bound would lift the scratch of the D > 1 kernels - but at > 128 VGPRs only ONE 8-wave workgroup fits
        // a CU instead of two: 1024 x 2048 measured 33 us per step against 21)
        // specialised instantiations (pf_column.hpp: KIND / FILT / PROP): float, built-in models, four particles per lane,
        // <= 256 threads, Philox normals; PF_COLUMN_GENERIC=1 keeps the run-time kernel (tests compare the two)
        bool specialised = false;
        if constexpr (sizeof(T) == 4 && D == 1 && VEC == 4) {
            const int hk = A->model.hid_kind;
            if (spec_ok) {
                specialised = true;
                auto go = [&](auto kind_c, auto filt_c, auto prop_c) {
                    with_rag([&](auto rag_c) {
                        if (nt <= 256)
                            hipLaunchKernelGGL((k_fused_column<T, D, VEC, 256, false, decltype(kind_c)::value, decltype(filt_c)::value,
                                                               decltype(prop_c)::value, decltype(rag_c)::value>), dim3(g.B), dim3(nt), lds, st, a, r);
                        else
                            hipLaunchKernelGGL((k_fused_column<T, D, VEC, 1024, false, decltype(kind_c)::value, decltype(filt_c)::value,
                                                               decltype(prop_c)::value, decltype(rag_c)::value>), dim3(g.B), dim3(nt), lds, st, a, r);
                    });
                };
                auto with_prop = [&](auto kind_c, auto filt_c) {
                    if (A->proposal == PF_PROP_LGO) go(kind_c, filt_c, std::integral_constant<int, PF_PROP_LGO>{});
                    else go(kind_c, filt_c, std::integral_constant<int, PF_PROP_BOOTSTRAP>{});
                };
                auto with_filt = [&](auto kind_c) {
                    if (A->filter == PF_FILTER_APF) with_prop(kind_c, std::integral_constant<int, PF_FILTER_APF>{});
                    else with_prop(kind_c, std::integral_constant<int, PF_FILTER_SISR>{});
                };
                if (hk == PF_HID_LINEAR) with_filt(std::integral_constant<int, PF_HID_LINEAR>{});
                else if (hk == PF_HID_SINE_EM) with_filt(std::integral_constant<int, PF_HID_SINE_EM>{});
                else if (hk == PF_HID_OU) with_filt(std::integral_constant<int, PF_HID_OU>{});
                else if (A->filter == PF_FILTER_APF)  // Verhulst + stochastic volatility: Bootstrap only
                    go(std::integral_constant<int, PF_HID_VERHULST_EM>{}, std::integral_constant<int, PF_FILTER_APF>{}, std::integral_constant<int, PF_PROP_BOOTSTRAP>{});
                else
                    go(std::integral_constant<int, PF_HID_VERHULST_EM>{}, std::integral_constant<int, PF_FILTER_SISR>{}, std::integral_constant<int, PF_PROP_BOOTSTRAP>{});
            }
        }
        if constexpr (sizeof(T) == 4 && D == 3 && VEC == 4) {
            if (spec_ok) {
                specialised = true;
                auto go3 = [&](auto filt_c, auto prop_c) {  // (the 256- or - 1 024 < N <= 2 048 - the 1024-thread bound)
                    if (nt <= 256)
                        hipLaunchKernelGGL((k_fused_column<T, D, VEC, 256, false, PF_HID_LORENZ63_EM, decltype(filt_c)::value,
                                                           decltype(prop_c)::value>), dim3(g.B), dim3(nt), lds, st, a, r);
                    else
                        hipLaunchKernelGGL((k_fused_column<T, D, VEC, 1024, false, PF_HID_LORENZ63_EM, decltype(filt_c)::value,
                                                           decltype(prop_c)::value>), dim3(g.B), dim3(nt), lds, st, a, r);
                };
                auto with_prop3 = [&](auto filt_c) {
                    if (A->proposal == PF_PROP_LGO) go3(filt_c, std::integral_constant<int, PF_PROP_LGO>{});
                    else go3(filt_c, std::integral_constant<int, PF_PROP_BOOTSTRAP>{});
                };
                if (A->filter == PF_FILTER_APF) with_prop3(std::integral_constant<int, PF_FILTER_APF>{});
                else with_prop3(std::integral_constant<int, PF_FILTER_SISR>{});
            }
        }
        if (specialised) {
        } else if (nt <= 256) launch(std::integral_constant<int, 256>{});
        else launch(std::integral_constant<int, 1024>{});
        done += r.n_steps;
    }
    if (kernel_ms) {
        (void)hipEventRecord(ev[1], st);
        hipError_t se = hipStreamSynchronize(st);
        if (se != hipSuccess) return (int)se;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[0], ev[1]);
        for (auto& e : ev) (void)hipEventDestroy(e);
        kernel_ms[0] = kernel_ms[2] = ms / (float)n_steps;  // the run's one kernel, per time step
        kernel_ms[1] = 0.f;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PF_OK : (int)e;
}
#define PF_COL_ARGS const pf_filter_args* A, const Geom& g, const WsLayout& wl, int64_t t0, int64_t n_steps, hipStream_t st, float* kernel_ms
int pf_run_column_f32(PF_COL_ARGS);
int pf_run_column_f64(PF_COL_ARGS);
#define PF_DEFINE_COLUMN(NAME, T)                                                                     \
    int NAME(PF_COL_ARGS) {                                                                           \
        const int D = A->model.dim;                                                                   \
        /* four particles per lane whatever N: columns of N % 4 != 0 take the RAGGED instantiations */  \
        if (D == 1) return column_run_impl<T, 1, 4>(A, g, wl, t0, n_steps, st, kernel_ms);            \
        if (D == 2) return column_run_impl<T, 2, 4>(A, g, wl, t0, n_steps, st, kernel_ms);            \
        return column_run_impl<T, 3, 4>(A, g, wl, t0, n_steps, st, kernel_ms);                        \
    }
#if defined(PF_TU_COLUMN_F32) || !defined(PF_TU_SPLIT)
PF_DEFINE_COLUMN(pf_run_column_f32, float)
#endif
#if defined(PF_TU_COLUMN_F64) || !defined(PF_TU_SPLIT)
PF_DEFINE_COLUMN(pf_run_column_f64, double)
#endif

// ---- the column-cluster route (pf_cluster.hpp): filters of 2 049 .. 16 384 particles, c workgroups per filter, one launch per
// run and group of columns -------------------------------------------------------------------------------------------------------
#define PFK_HOST_VEC 4  // particles per lane of the cluster kernels 
#define PF_CLUSTER_INFEASIBLE (-1000)  // internal: the cluster kernel cannot be launched here (no launch was issued)
static inline size_t cluster_lds_bytes(int D, size_t tsize) {
    return (size_t)(PFK_WIN_P2 + D * PFK_WIN) * tsize + 2 * PFK_FOLD * sizeof(double);  // window planes | the folds of two states
}
// Opt-in (pf_run_hints.route == PF_ROUTE_CLUSTER): the members of a column wait for each other - a launch that cannot make
// progress reports it through pf_filter_args.status instead of a result, and the caller re-issues the piece on the per-step
// route (include/pf_amd.h: PF_ROUTE_CLUSTER); the all-zero hints of the C ABI never take it.
static inline bool cluster_eligible(const pf_filter_args* A, const Geom& g, int64_t n_steps, int finalize) {
    if (A->hints.route != PF_ROUTE_CLUSTER && A->hints.route != PF_ROUTE_CLUSTER_ALWAYS && A->hints.route != PF_ROUTE_CLUSTER_SPREAD) return false;
    if (!finalize || n_steps < 1 || A->ring >= 3) return false;
    if (A->N <= PF_CLUSTER_MIN_N || A->N > PF_CLUSTER_MAX_N || A->N % PFK_HOST_VEC != 0) return false;
    if (A->resampler != PF_RESAMPLE_SYSTEMATIC || A->model.hid_kind == PF_HID_USER_AFFINE) return false;
    // Where it pays (same-box A/Bs, profiles/r05_cluster_route.txt): a launch holds ~1 024 resident member workgroups (2^20
    // particles) and larger batches run as consecutive launches of ~8.5 us per step each, while a per-step launch of 2^21+
    // particles costs 33 us and grows by 3 us per 2^20 more - two launches' worth is the break-even
    const int64_t members = ((A->N + PFK_TPB * PFK_HOST_VEC - 1) / (PFK_TPB * PFK_HOST_VEC)) * A->B;
    if (A->hints.route == PF_ROUTE_CLUSTER && members > 2 * 1024) return false;
    (void)g;
    return true;
}
// resident workgroups of `kernel` on the current device: CUs x min(occupancy query, 6) - the query can be one block per CU high
// near the SGPR-limited edges (MI355X_MICROARCH.md, "Residency and cooperative launch"); 6 is below every such edge
// (asked once per kernel, LDS size and device: an online move is one such run per observation, and the three queries cost as much
// host time as a launch)
template <typename K> static inline int cluster_slots(K kernel, size_t lds) {
    struct Seen { const void* k; size_t lds; int dev, slots; };
    static Seen seen[32];
    static std::atomic<int> n_seen{0};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const int have = n_seen.load(std::memory_order_acquire);
    for (int i = 0; i < have; ++i)
        if (seen[i].k == (const void*)kernel && seen[i].lds == lds && seen[i].dev == dev) return seen[i].slots;
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, PFK_TPB, lds) != hipSuccess) return 0;
    if (per_cu > 6) per_cu = 6;
    std::lock_guard<std::mutex> lock(mu);
    const int at = n_seen.load(std::memory_order_relaxed);
    if (at < 32) {
        seen[at] = Seen{(const void*)kernel, lds, dev, cus * per_cu};
        n_seen.store(at + 1, std::memory_order_release);
    }
    return cus * per_cu;
}
template <typename T, int D>
static int cluster_run_impl(const pf_filter_args* A, const Geom& g, const WsLayout& wl, int64_t t0, int64_t n_steps,
                            hipStream_t st, float* kernel_ms) {
    constexpr int VEC = PFK_HOST_VEC;
    FusedArgs<T> a = make_fused_args<T>(A, g, wl, t0);
    const size_t lds = cluster_lds_bytes(D, sizeof(T));
    const bool auto_flags = !A->observed && !A->observed_dev;
    a.obs_dev = A->observed_dev;
    uint8_t* const auto_fl = (uint8_t*)A->ws + wl.off_ctr + 64;
    const int64_t auto_row = A->y_rows * (int64_t)A->model.obs_dim;
    // one-step runs on a shared observation row: the kernel reads the flag off y itself (ColumnRun::inline_y)
    const bool inline_y = auto_flags && n_steps == 1 && A->y_rows == 1;
    bool flags_pending = auto_flags && !inline_y;  // (derived by the launch that clears the first piece's records: k_zero_and_flags)
    if (auto_flags && !inline_y) a.obs_dev = auto_fl - t0;
    // the caller numbers its launches (pf_run_hints.cluster_generation): tagged records, nothing to clear
    bool numbered = A->hints.cluster_generation != 0 && A->status != nullptr && n_steps <= 32 * PFC_OBS_WORDS;
    if (numbered) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) numbered = false;  // (a replay repeats the number)
        (void)hipGetLastError();
    }
    const int c = (int)((A->N + PFK_TPB * VEC - 1) / (PFK_TPB * VEC));
    const int nchunks = (int)((A->N + 64 * VEC - 1) / (64 * VEC));
    // which instantiation: float runs of the built-in scalar closed-form models on Philox normals take KIND / FILT / PROP folded
    // (as on the column route), everything else the run-time kernel
    bool spec_ok = false;
    if constexpr (sizeof(T) == 4 && D == 1) {
        const int hk = A->model.hid_kind;
        spec_ok = !A->z_tape && A->model.obs_kind == PF_OBS_LINEAR &&
                  (hk == PF_HID_LINEAR || hk == PF_HID_SINE_EM || hk == PF_HID_OU) &&
                  (A->proposal == PF_PROP_BOOTSTRAP || A->proposal == PF_PROP_LGO);
    }
    auto with_kernel = [&](auto&& f) {
        if constexpr (sizeof(T) == 4 && D == 1) {
            if (spec_ok) {
                auto with_prop = [&](auto kind_c, auto filt_c) {
                    if (A->proposal == PF_PROP_LGO)
                        f(k_fused_cluster<T, D, VEC, decltype(kind_c)::value, decltype(filt_c)::value, PF_PROP_LGO>);
                    else
                        f(k_fused_cluster<T, D, VEC, decltype(kind_c)::value, decltype(filt_c)::value, PF_PROP_BOOTSTRAP>);
                };
                auto with_filt = [&](auto kind_c) {
                    if (A->filter == PF_FILTER_APF) with_prop(kind_c, std::integral_constant<int, PF_FILTER_APF>{});
                    else with_prop(kind_c, std::integral_constant<int, PF_FILTER_SISR>{});
                };
                const int hk = A->model.hid_kind;
                if (hk == PF_HID_LINEAR) with_filt(std::integral_constant<int, PF_HID_LINEAR>{});
                else if (hk == PF_HID_SINE_EM) with_filt(std::integral_constant<int, PF_HID_SINE_EM>{});
                else with_filt(std::integral_constant<int, PF_HID_OU>{});
                return;
            }
        }
        f(k_fused_cluster<T, D, VEC, -1, -1, -1>);
    };
    int rc = PF_OK;
    hipEvent_t ev[2] = {nullptr, nullptr};
    if (kernel_ms) {
        for (auto& e : ev)
            if (hipEventCreate(&e) != hipSuccess) return (int)hipGetLastError();
        (void)hipEventRecord(ev[0], st);
    }
    with_kernel([&](auto kernel) {
        // (nothing has been launched yet: PF_CLUSTER_INFEASIBLE sends the caller - filter_run_checked - to the per-step route)
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            rc = PF_CLUSTER_INFEASIBLE;
            return;
        }
        const int slots = cluster_slots(kernel, lds);
        int per_launch = slots / c;  // columns whose members are all resident at once
        if (per_launch >= 8) per_launch &= ~7;
        if (per_launch < 1) {
            rc = PF_CLUSTER_INFEASIBLE;
            return;
        }
        unsigned char* clu = (unsigned char*)A->ws + wl.off_clu;
        for (int64_t done = 0; done < n_steps;) {
            ColumnRun r;
            r.t0 = (int)(t0 + done);
            r.n_steps = (int)((n_steps - done < 32 * PFC_OBS_WORDS) ? n_steps - done : 32 * PFC_OBS_WORDS);
            r.use_bits = (a.obs_dev == nullptr && !inline_y) ? 1 : 0;
            r.inline_y = inline_y ? 1 : 0;
            for (int w = 0; w < PFC_OBS_WORDS; ++w) r.obs_bits[w] = 0u;
            if (r.use_bits)
                for (int q = 0; q < r.n_steps; ++q)
                    if (A->observed[r.t0 + q]) r.obs_bits[q >> 5] |= 1u << (q & 31);
            a.step = r.t0;
            // fresh tags for this piece: error word + every record of the batch (a kernel, not a memset node - see k_zero_words)
            // (only what this instantiation's records occupy: NG granule rows of 1 KB per column and parity, after the error word)
            const size_t ng = ((size_t)(5 + 2 * D) * (sizeof(T) / 4) + 2 + 2) / 3;
            size_t words = (256 + (size_t)2 * g.B * PF_CLUSTER_NG * 64 * 16) / sizeof(uint32_t);
            if (g.B <= per_launch) words = (256 + (size_t)2 * g.B * ng * 64 * 16) / sizeof(uint32_t);  // (one group: its block is compact)
            const unsigned zb = (unsigned)((words + PF_BLOCK - 1) / PF_BLOCK);
            if (flags_pending)
                hipLaunchKernelGGL((k_zero_and_flags<T>), dim3(zb + (unsigned)n_steps), dim3(PF_BLOCK), 0, st, (uint32_t*)clu, words, zb,
                                   (const T*)A->y + t0 * auto_row, auto_row, auto_fl);
            else if (!numbered)
                hipLaunchKernelGGL((k_zero_words<uint32_t>), dim3(zb), dim3(PF_BLOCK), 0, st, (uint32_t*)clu, words);
            flags_pending = false;
            trace_launch(r.t0, (int)sizeof(T), D, VEC, 0, A->proposal, spec_ok ? 1 : 0, /*SPEC*/ 10, spec_ok ? A->model.hid_kind : 0, c);
            for (int b0 = 0; b0 < g.B; b0 += per_launch) {
                ClusterRun cr;
                cr.b0 = b0;
                cr.nb = (g.B - b0 < per_launch) ? g.B - b0 : per_launch;
                cr.nbp = (cr.nb + 7) & ~7;
                cr.c = c;
                cr.nchunks = nchunks;
                // (numbered launches: the workspace's error word is never cleared - the caller's status word, which it clears itself, is both)
                cr.err = numbered ? A->status : (int*)clu;
                cr.status = numbered ? nullptr : A->status;
                cr.tag_base = numbered ? (unsigned)(A->hints.cluster_generation & 0xFFFFF) * 4096u : 0u;
                cr.patience = A->hints.cluster_patience != 0 ? A->hints.cluster_patience : PFK_SPIN_LIMIT;
                cr.spread = A->hints.route == PF_ROUTE_CLUSTER_SPREAD ? 1 : 0;
                cr.th = ClusterTheta{};
                if (tls_theta_fold != nullptr && g.B <= per_launch && done + r.n_steps == n_steps && done == 0) {
                    // (one launch carries the whole run and every column: its last column to finish does the theta update)
                    ThetaFold* tf = tls_theta_fold;
                    cr.th.enabled = 1;
                    cr.th.w = tf->w;
                    cr.th.ll = tf->ll;
                    cr.th.stats = tf->stats;
                    cr.th.slot = (double*)tf->slot;
                    cr.th.seq = (unsigned long long)tf->seq;
                    cr.th.acc = tf->acc;
                    cr.th.arrive = (unsigned*)clu + 16;
                    tf->folded = 1;
                }
                cr.rec = clu + 256 + (size_t)b0 * 2 * PF_CLUSTER_NG * 64 * 16;  // (this group's [2][nb][NG][64] block)
                hipLaunchKernelGGL(kernel, dim3((unsigned)(cr.nbp * c)), dim3(PFK_TPB), lds, st, a, r, cr);
            }
            done += r.n_steps;
        }
    });
    if (rc != PF_OK) {
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
        return rc;
    }
    if (kernel_ms) {
        (void)hipEventRecord(ev[1], st);
        hipError_t se = hipStreamSynchronize(st);
        if (se != hipSuccess) return (int)se;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[0], ev[1]);
        for (auto& e : ev) (void)hipEventDestroy(e);
        kernel_ms[0] = kernel_ms[2] = ms / (float)n_steps;
        kernel_ms[1] = 0.f;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PF_OK : (int)e;
}
int pf_run_cluster_f32(PF_COL_ARGS);
int pf_run_cluster_f64(PF_COL_ARGS);
#define PF_DEFINE_CLUSTER(NAME, T)                                                        \
    int NAME(PF_COL_ARGS) {                                                               \
        const int D = A->model.dim;                                                       \
        if (D == 1) return cluster_run_impl<T, 1>(A, g, wl, t0, n_steps, st, kernel_ms);  \
        if (D == 2) return cluster_run_impl<T, 2>(A, g, wl, t0, n_steps, st, kernel_ms);  \
        return cluster_run_impl<T, 3>(A, g, wl, t0, n_steps, st, kernel_ms);              \
    }
#if defined(PF_TU_CLUSTER_F32) || !defined(PF_TU_SPLIT)
PF_DEFINE_CLUSTER(pf_run_cluster_f32, float)
#endif
#if defined(PF_TU_CLUSTER_F64) || !defined(PF_TU_SPLIT)
PF_DEFINE_CLUSTER(pf_run_cluster_f64, double)
#endif

// one entry per arithmetic type / state dimension / vector width / tile geometry (see the translation-unit note above)
#define PF_RUN_ARGS const pf_filter_args* A, const Geom& g, const WsLayout& wl, int64_t t0, int64_t n_steps, int finalize, \
                    hipStream_t st, float* kernel_ms
#define PF_RUN_PASS A, g, wl, t0, n_steps, finalize, st, kernel_ms
int pf_run_f32(PF_RUN_ARGS);
int pf_run_f64(PF_RUN_ARGS);
#define PF_DECLARE_LEAVES(SFX)            \
    int pf_run_f32_d1_v4##SFX(PF_RUN_ARGS); \
    int pf_run_f32_d1_v1##SFX(PF_RUN_ARGS); \
    int pf_run_f32_dn##SFX(PF_RUN_ARGS);    \
    int pf_run_f64##SFX(PF_RUN_ARGS);
PF_DECLARE_LEAVES(_m0)
PF_DECLARE_LEAVES(_m1)
#ifndef PF_TU_NO_API
int pf_run_f32(PF_RUN_ARGS) {
    const bool multi = g.rounds_per_tile > 1;
    if (A->model.dim != 1) return multi ? pf_run_f32_dn_m1(PF_RUN_PASS) : pf_run_f32_dn_m0(PF_RUN_PASS);
    if (g.vec == 4) return multi ? pf_run_f32_d1_v4_m1(PF_RUN_PASS) : pf_run_f32_d1_v4_m0(PF_RUN_PASS);
    return multi ? pf_run_f32_d1_v1_m1(PF_RUN_PASS) : pf_run_f32_d1_v1_m0(PF_RUN_PASS);
}
int pf_run_f64(PF_RUN_ARGS) {
    return g.rounds_per_tile > 1 ? pf_run_f64_m1(PF_RUN_PASS) : pf_run_f64_m0(PF_RUN_PASS);
}
#endif
#define RUN(T, DD, V, MULTI) return filter_run_impl<T, DD, V, MULTI>(PF_RUN_PASS);
#define RUN_D(T, V, MULTI) \
    if (D == 1) { RUN(T, 1, V, MULTI) } else if (D == 2) { RUN(T, 2, V, MULTI) } else { RUN(T, 3, V, MULTI) }
#define PF_DEFINE_D1_V4(SFX, MULTI) int pf_run_f32_d1_v4##SFX(PF_RUN_ARGS) { RUN(float, 1, 4, MULTI) }
#define PF_DEFINE_D1_V1(SFX, MULTI) int pf_run_f32_d1_v1##SFX(PF_RUN_ARGS) { RUN(float, 1, 1, MULTI) }
#define PF_DEFINE_DN(SFX, MULTI)                                              \
    int pf_run_f32_dn##SFX(PF_RUN_ARGS) {                                     \
        const int D = A->model.dim;                                           \
        if (g.vec == 4) {                                                     \
            if (D == 2) { RUN(float, 2, 4, MULTI) } else { RUN(float, 3, 4, MULTI) } \
        } else {                                                              \
            if (D == 2) { RUN(float, 2, 1, MULTI) } else { RUN(float, 3, 1, MULTI) } \
        }                                                                     \
    }
#define PF_DEFINE_F64(SFX, MULTI)                                    \
    int pf_run_f64##SFX(PF_RUN_ARGS) {                               \
        const int D = A->model.dim;                                  \
        if (g.vec == 4) { RUN_D(double, 4, MULTI) } else { RUN_D(double, 1, MULTI) } \
    }
#if !defined(PF_TU_MULTI) || PF_TU_MULTI == 0
#define PF_FOR_M0(X) X(_m0, false)
#else
#define PF_FOR_M0(X)
#endif
#if !defined(PF_TU_MULTI) || PF_TU_MULTI == 1
#define PF_FOR_M1(X) X(_m1, true)
#else
#define PF_FOR_M1(X)
#endif
#if !defined(PF_TU_NO_F32D1) && !defined(PF_TU_F64_ONLY) && !defined(PF_TU_F32DN_ONLY)
#if !defined(PF_TU_VEC) || PF_TU_VEC == 4
PF_FOR_M0(PF_DEFINE_D1_V4)
PF_FOR_M1(PF_DEFINE_D1_V4)
#endif
#if !defined(PF_TU_VEC) || PF_TU_VEC == 1
PF_FOR_M0(PF_DEFINE_D1_V1)
PF_FOR_M1(PF_DEFINE_D1_V1)
#endif
#endif
#if !defined(PF_TU_NO_F32DN) && !defined(PF_TU_F64_ONLY) && !defined(PF_TU_F32D1_ONLY)
PF_FOR_M0(PF_DEFINE_DN)
PF_FOR_M1(PF_DEFINE_DN)
#endif
#if !defined(PF_TU_NO_F64) && !defined(PF_TU_F32DN_ONLY) && !defined(PF_TU_F32D1_ONLY)
PF_FOR_M0(PF_DEFINE_F64)
PF_FOR_M1(PF_DEFINE_F64)
#endif
#undef RUN_D
#undef RUN

#ifndef PF_TU_NO_API
static int filter_run_checked(const pf_filter_args* A, int64_t t0, int64_t n_steps, int finalize, void* stream,
                              float* kernel_ms);

extern "C" int pf_filter_run(const pf_filter_args* A, int64_t t0, int64_t n_steps, int finalize, void* stream) {
    return filter_run_checked(A, t0, n_steps, finalize, stream, nullptr);
}

extern "C" int pf_filter_observe(const pf_filter_args* A, int64_t t0, int64_t n_steps, int finalize, void* w, const void* ll, void* stats,
                                 void* host_slot, uint64_t seq, void* acc, void* stream) {
    if (!A || !w || !ll || !stats || ((uintptr_t)host_slot & 7) != 0) return PF_EINVAL;
    ThetaFold tf{w, ll, stats, host_slot, seq, acc, 0};
    tls_theta_fold = &tf;
    const int rc = filter_run_checked(A, t0, n_steps, finalize, stream, nullptr);
    tls_theta_fold = nullptr;
    if (rc != PF_OK) return rc;
    if (tf.folded) return PF_OK;  // (the column-cluster launch did the update itself)
    return pf_theta_step(w, ll, A->B, A->dtype, stats, host_slot, seq, acc, A->status, stream);
}

extern "C" int pf_filter_run_timed(const pf_filter_args* A, int64_t t0, int64_t n_steps, int finalize, void* stream,
                                   float* kernel_ms) {
    if (!kernel_ms) return PF_EINVAL;
    return filter_run_checked(A, t0, n_steps, finalize, stream, kernel_ms);
}

// ---- hipGraph variant: the whole launch sequence of a run captured once, replayed with one host call -----------------
struct PfGraph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

extern "C" int pf_filter_graph_create(const pf_filter_args* A, int64_t t0, int64_t n_steps, int finalize, void* stream,
                                      void** handle) {
    if (!handle) return PF_EINVAL;
    *handle = nullptr;
    (void)stream;
    // capture on a private stream (the caller's may be the legacy default stream, which cannot capture); the graph is
    // replayed on whatever stream pf_filter_graph_launch is given
    hipStream_t st = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) return (int)e;
    e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(st);
        return (int)e;
    }
    const int rc = filter_run_checked(A, t0, n_steps, finalize, (void*)st, nullptr);
    hipGraph_t graph = nullptr;
    e = hipStreamEndCapture(st, &graph);
    (void)hipStreamDestroy(st);
    if (rc != PF_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
    }
    if (e != hipSuccess) return (int)e;
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(graph);
        return (int)e;
    }
    PfGraph* g = new PfGraph{graph, exec};
    *handle = g;
    return PF_OK;
}

extern "C" int pf_filter_graph_launch(void* handle, void* stream) {
    if (!handle) return PF_EINVAL;
    const hipError_t e = hipGraphLaunch(((PfGraph*)handle)->exec, (hipStream_t)stream);
    return e == hipSuccess ? PF_OK : (int)e;
}

extern "C" int pf_filter_graph_destroy(void* handle) {
    if (!handle) return PF_OK;
    PfGraph* g = (PfGraph*)handle;
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
    return PF_OK;
}

static int filter_run_checked(const pf_filter_args* A, int64_t t0, int64_t n_steps, int finalize, void* stream,
                              float* kernel_ms) {
    if (!A || A->struct_size != sizeof(pf_filter_args)) return PF_EINVAL;  // (another ABI version: include/pf_amd.h)
    if (A->hints.route < 0 || A->hints.route > PF_ROUTE_CLUSTER_SPREAD || A->hints.column_max_n < 0 || A->hints.tile_target < 0 ||
        A->hints.cluster_patience < -1 || A->hints.cluster_patience > (1 << 30) || A->hints.cluster_generation < 0)
        return PF_EINVAL;
    int rc = check_model(&A->model, true);
    if (rc) return rc;
    if (bad_shape(A->N, A->B) || t0 < 0 || n_steps < 0) return PF_EINVAL;
    if (A->ring < 0 || A->ring == 1) return PF_EINVAL;
    if (A->hints.prepare_next != 0) {  // (see pf_run_hints: what the last step would have to evaluate must be in the kernels' reach)
        if (A->filter != PF_FILTER_APF || finalize || n_steps < 1) return PF_EINVAL;
        if (A->model.hid_kind == PF_HID_USER_AFFINE && (A->proposal != PF_PROP_LGO || !A->user_scale_per_column)) return PF_EINVAL;
    }
    if (!A->x[0] || !A->logw[0] || (A->ring < 3 && (!A->x[1] || !A->logw[1])) || !A->anc || !A->cdf || !A->means || !A->vars ||
        !A->ll_steps || !A->ll_total || !A->ws)
        return PF_EINVAL;
    if (n_steps > 0 && (!A->y || (!A->observed && !A->observed_dev && n_steps > PF_AUTO_FLAGS))) return PF_EINVAL;
    if (A->y_rows != 1 && A->y_rows != A->B) return PF_EINVAL;
    if (A->proposal == PF_PROP_LGO && A->model.obs_kind != PF_OBS_LINEAR) return PF_EUNSUPPORTED;
    if (A->model.hid_kind == PF_HID_USER_AFFINE) {  // the planes describe ONE incoming state: one step per call, no history
        if (!A->user_loc || !A->user_scale) return PF_EINVAL;
        if (n_steps > 1 || A->ring >= 3) return PF_EUNSUPPORTED;
    }
    if (A->filter != PF_FILTER_SISR && A->filter != PF_FILTER_APF) return PF_EUNSUPPORTED;
    if (!A->pos) return PF_EINVAL;
    const Geom g = make_geom(A->N, A->B, A->hints.tile_target);
    const WsLayout wl = make_ws(g, PF_MAXD);
    if (A->ws_bytes < wl.total) return PF_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (!column_eligible(A, g, n_steps, finalize) && cluster_eligible(A, g, n_steps, finalize) && wl.clu_bytes != 0) {
        if (A->dtype != PF_F32 && A->dtype != PF_F64) return PF_EINVAL;
        rc = A->dtype == PF_F32 ? pf_run_cluster_f32(A, g, wl, t0, n_steps, st, kernel_ms)
                                : pf_run_cluster_f64(A, g, wl, t0, n_steps, st, kernel_ms);
        if (rc != PF_CLUSTER_INFEASIBLE) return rc;  // (else: no slot for one filter's workgroups / LDS refused - the per-step route)
    }
    if (column_eligible(A, g, n_steps, finalize)) {
        if (A->dtype == PF_F32) return pf_run_column_f32(A, g, wl, t0, n_steps, st, kernel_ms);
        if (A->dtype == PF_F64) return pf_run_column_f64(A, g, wl, t0, n_steps, st, kernel_ms);
        return PF_EINVAL;
    }
    if (A->dtype == PF_F32) return pf_run_f32(A, g, wl, t0, n_steps, finalize, st, kernel_ms);
    if (A->dtype == PF_F64) return pf_run_f64(A, g, wl, t0, n_steps, finalize, st, kernel_ms);
    return PF_EINVAL;
}
#endif  // !PF_TU_NO_API
