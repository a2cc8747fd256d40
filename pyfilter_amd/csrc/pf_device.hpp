// pf_device.hpp - workgroup-level building blocks for the particle-filter kernels (gfx950, wave64).
//
// Geometry used by every column kernel:
//   * storage is column-major per filter: weights (B, N), state SoA (D, B, N); a "column" is one filter's N particles;
//   * a workgroup is PF_BLOCK = 256 threads = 4 wave64s; it owns one *tile* of one column;
//   * a tile is R "rounds"; in a round thread t owns VEC consecutive elements  base + r*256*VEC + t*VEC + j,
//     so a wave's load is one fully coalesced 64 x (VEC*sizeof(T)) byte request (dwordx4 for VEC=4, f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef PF_BLOCK
#define PF_BLOCK 256
#endif
#define PF_WAVE 64
#define PF_NWAVES (PF_BLOCK / PF_WAVE)

namespace pf {

template <typename T> struct Lim;
template <> struct Lim<float> {
    __host__ __device__ static constexpr float inf() { return __builtin_huge_valf(); }
    __host__ __device__ static constexpr float lowest() { return -3.40282346638528859812e+38f; }
};
template <> struct Lim<double> {
    __host__ __device__ static constexpr double inf() { return __builtin_huge_val(); }
    __host__ __device__ static constexpr double lowest() { return -1.79769313486231570815e+308; }
};

__device__ __forceinline__ bool is_nan_or_posinf(float v) { return __builtin_amdgcn_classf(v, 0x203); }
__device__ __forceinline__ bool is_nan_or_posinf(double v) { return __builtin_amdgcn_class(v, 0x203); }

// torch.nan_to_num_(w, nan=-inf, posinf=-inf) as pyfilter/utils.py:57 calls it: NaN -> -inf, +inf -> -inf and -
// because ``neginf`` is left at its default - -inf -> the lowest finite value of the dtype.
template <typename T> __device__ __forceinline__ T sanitize_logw(T v) {
    // v_cmp_class: one test for {sNaN, qNaN, +inf} (bits 0, 1, 9), one compare for -inf
    const bool bad = is_nan_or_posinf(v);
    T r = bad ? -Lim<T>::inf() : v;
    return (v == -Lim<T>::inf()) ? Lim<T>::lowest() : r;
}

__device__ __forceinline__ float pf_exp(float x) { return expf(x); }
__device__ __forceinline__ double pf_exp(double x) { return exp(x); }
__device__ __forceinline__ float pf_log(float x) { return logf(x); }
__device__ __forceinline__ double pf_log(double x) { return log(x); }
__device__ __forceinline__ float pf_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double pf_sqrt(double x) { return sqrt(x); }
// float sine: Cody-Waite reduction by pi (three-constant split, exact products for |k| < 2^16) + odd degree-11
// polynomial on [-pi/2, pi/2]; ~16 VALU instructions, error < 1.5 ulp for |x| < 1e5 (libm's sinf carries a
// Payne-Hanek large-argument path that costs ~5x as much in straight-line code).  Larger arguments fall back to sinf.
__device__ __forceinline__ float pf_sin(float x) {
    if (!(fabsf(x) < 1.0e5f)) return sinf(x);
    const float k = rintf(x * 0.318309886183790671538f);
    float r = fmaf(-k, 3.140625f, x);
    r = fmaf(-k, 9.67502593994140625e-4f, r);
    r = fmaf(-k, 1.509957990978376432e-7f, r);
    const float r2 = r * r;
    float p = fmaf(r2, -2.50521083854417187751e-8f, 2.75573192239858906526e-6f);
    p = fmaf(r2, p, -1.98412698412698412698e-4f);
    p = fmaf(r2, p, 8.33333333333333333333e-3f);
    p = fmaf(r2, p, -1.66666666666666666667e-1f);
    float sres = fmaf(r * r2, p, r);
    return ((int)k & 1) ? -sres : sres;
}
__device__ __forceinline__ double pf_sin(double x) { return sin(x); }
// float sine of the closed-form (FAST) production kernels: the hardware's v_sin_f32 on x / 2 pi (two instructions; absolute
// error <= 2^-21.4 = 3.6e-7 on [-pi, pi] plus the reduction's |x| * 6e-8 - the one-step mean x + sin(x - gamma) dt carries a
// tenth of that, two orders below the 1e-5 bar of the teacher-forced tests; valid for |x / 2 pi| <= 256).  Larger arguments -
// a diverged particle - take the Cody-Waite form.  The float64 parity path never comes here.
__device__ __forceinline__ float pf_sin_fast(float x) {
    if (!(fabsf(x) < 1.0e3f)) return pf_sin(x);
    return __builtin_amdgcn_sinf(x * 0.159154943091895335769f);
}
__device__ __forceinline__ double pf_sin_fast(double x) { return sin(x); }
// exp for importance weights: float -> the bare v_exp_f32 (2^x) on x * log2(e): two instructions (clang's __expf
// expands to 13 with its range handling).  Relative error ~|x| * 6e-8, irrelevant next to the fp32 rounding of the
// weights themselves; results below 2^-126 flush to zero, exp(-inf) = 0.  double -> libm
__device__ __forceinline__ float pf_exp_w(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ double pf_exp_w(double x) { return exp(x); }
// per-column constants (evaluated once per thread, on the critical path of every workgroup): hardware log / rcp /
// sqrt for float (1 ulp class), libm for double
__device__ __forceinline__ float pf_log_c(float x) {  // bare v_log_f32 (log2); arguments here are normal numbers
    return __builtin_amdgcn_logf(x) * 0.693147180559945309417f;
}
__device__ __forceinline__ double pf_log_c(double x) { return log(x); }
__device__ __forceinline__ float pf_rcp_c(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double pf_rcp_c(double x) { return 1.0 / x; }
__device__ __forceinline__ float pf_sqrt_c(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double pf_sqrt_c(double x) { return sqrt(x); }
// a / b, sqrt, log in the generic per-particle arithmetic: float -> hardware reciprocal / sqrt / log (1 ulp class; the
// compiler's exact division alone is ~10 instructions), double -> the exact forms (the parity path keeps its rounding)
__device__ __forceinline__ float pf_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ double pf_div(double a, double b) { return a / b; }
__device__ __forceinline__ float pf_sqrt_g(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double pf_sqrt_g(double x) { return sqrt(x); }
__device__ __forceinline__ float pf_log_g(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309417f; }
__device__ __forceinline__ double pf_log_g(double x) { return log(x); }
__device__ __forceinline__ float pf_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double pf_abs(double x) { return fabs(x); }

// exp(a - b) with the convention exp(-inf - anything) = 0 (so empty / all -inf partials merge cleanly)
__device__ __forceinline__ double exp_diff(double a, double b) {
    return (a == -__builtin_huge_val()) ? 0.0 : exp(a - b);
}
// the same evaluated in the filter's arithmetic type T (a, b are tile maxima, i.e. T values carried as doubles):
// float filters use the fast float exp here - these factors are evaluated per tile per workgroup
template <typename T> __device__ __forceinline__ double exp_diff_t(double a, double b) {
    return (a == -__builtin_huge_val()) ? 0.0 : (double)pf_exp_w((T)(a - b));
}

// ---------------------------------------------------------------------------------------------------------------
// wave64 cross-lane primitives on DPP (data-parallel primitives: the lane permutation rides on a VALU move, ~10x
// cheaper than the LDS-crossbar ds_bpermute that __shfl_* lowers to).  Within a row of 16 lanes:
//   quad_perm [1,0,3,2] (xor 1), quad_perm [2,3,0,1] (xor 2), row_half_mirror, row_mirror  -> all-reduce of the row;
//   row_shr:1,2,4,8                                                                       -> inclusive scan of the row;
// across the four rows: v_readlane (reductions) / row_bcast15 + row_bcast31 (scan) - the gfx9 wave64 idiom.
// ---------------------------------------------------------------------------------------------------------------
#define PF_DPP_QUAD_XOR1 0xB1
#define PF_DPP_QUAD_XOR2 0x4E
#define PF_DPP_ROW_HALF_MIRROR 0x141
#define PF_DPP_ROW_MIRROR 0x140
#define PF_DPP_ROW_SHR(n) (0x110 + (n))
#define PF_DPP_ROW_BCAST15 0x142
#define PF_DPP_ROW_BCAST31 0x143

// source lane's value where the DPP pattern has a valid, enabled source; `ident` elsewhere
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ float dpp_get(float v, float ident) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ double dpp_get(double v, double ident) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// the same with the identity ZERO (sums, scans): lanes without a valid source read 0 through `bound_ctrl` when every row takes
// part (ROW_MASK = 0xF) - the `old` operand is then dead and the compiler does not have to zero the destination before every
// exchange (two v_mov per fp64 exchange); rows masked out by ROW_MASK keep `old`, which must then really be zero
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ float dpp_get0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF));
}
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ double dpp_get0(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_get(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double lane_get(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
    v += dpp_get0<PF_DPP_QUAD_XOR1>(v);
    v += dpp_get0<PF_DPP_QUAD_XOR2>(v);
    v += dpp_get0<PF_DPP_ROW_HALF_MIRROR>(v);
    v += dpp_get0<PF_DPP_ROW_MIRROR>(v);
    return (lane_get(v, 0) + lane_get(v, 16)) + (lane_get(v, 32) + lane_get(v, 48));
}
// float: v_max_f32 takes the DPP operand directly - ONE instruction per exchange.  Written as inline asm since round 5: from
// `fmaxf(v, dpp(v))` the compiler emits v_mov_b32 + v_mov_b32_dpp + a canonicalising v_max_f32 v, v, v + the v_max_f32 (IEEE
// mode: the DPP move's result is not known to be quiet), i.e. four VALU instructions per exchange - 30 such canonicalisations
// in the headline step kernel alone.  `s_nop 1`: a DPP operand must not be read within two wait states of the VALU write that
// produced it (the compiler pads its own DPP instructions; inside asm nobody does).  Inputs are NaN-free maxima of sanitised
// log-weights; -inf is an ordinary operand of v_max.  PRECONDITION: all 64 lanes active (every caller reduces in uniform control
// flow) - a lane whose DPP source lane is inactive is not written (no bound_ctrl) and would keep whatever the register held.
#define PF_DPP_MAX_F32(r, v, CTRL) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v))
__device__ __forceinline__ float wave_max_f32(float v) {
    float a, b, c, d;
    PF_DPP_MAX_F32(a, v, "quad_perm:[1,0,3,2]");
    PF_DPP_MAX_F32(b, a, "quad_perm:[2,3,0,1]");
    PF_DPP_MAX_F32(c, b, "row_half_mirror");
    PF_DPP_MAX_F32(d, c, "row_mirror");
    return __builtin_fmaxf(__builtin_fmaxf(lane_get(d, 0), lane_get(d, 16)), __builtin_fmaxf(lane_get(d, 32), lane_get(d, 48)));
}
template <typename T> __device__ __forceinline__ T wave_max(T v) {
    if constexpr (sizeof(T) == 4) return wave_max_f32(v);
    // NaN-free inputs (maxima of sanitised log-weights); the comparison form keeps -inf working
    T o = dpp_get<PF_DPP_QUAD_XOR1>(v, v);
    v = (o > v) ? o : v;
    o = dpp_get<PF_DPP_QUAD_XOR2>(v, v);
    v = (o > v) ? o : v;
    o = dpp_get<PF_DPP_ROW_HALF_MIRROR>(v, v);
    v = (o > v) ? o : v;
    o = dpp_get<PF_DPP_ROW_MIRROR>(v, v);
    v = (o > v) ? o : v;
    const T a = lane_get(v, 0), b = lane_get(v, 16), c = lane_get(v, 32), d = lane_get(v, 48);
    const T ab = (a > b) ? a : b, cd = (c > d) ? c : d;
    return (ab > cd) ? ab : cd;
}
// inclusive scan across the 64 lanes of a wave
template <typename T> __device__ __forceinline__ T wave_scan_incl(T v, int /*lane*/) {
    v += dpp_get0<PF_DPP_ROW_SHR(1)>(v);
    v += dpp_get0<PF_DPP_ROW_SHR(2)>(v);
    v += dpp_get0<PF_DPP_ROW_SHR(4)>(v);
    v += dpp_get0<PF_DPP_ROW_SHR(8)>(v);
    v += dpp_get0<PF_DPP_ROW_BCAST15, 0xA>(v);  // lane 15 -> row 1, lane 47 -> row 3
    v += dpp_get0<PF_DPP_ROW_BCAST31, 0xC>(v);  // lane 31 -> rows 2, 3
    return v;
}

// previous lane's value (lane 0 gets `ident`): the gfx9 whole-wave DPP shift
#define PF_DPP_WAVE_SHR1 0x138
__device__ __forceinline__ int wave_prev(int v, int ident) {
    return __builtin_amdgcn_update_dpp(ident, v, PF_DPP_WAVE_SHR1, 0xF, 0xF, false);
}
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ int dpp_get_i(int v, int ident) {
    return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
// inclusive running maximum of non-negative ints across the 64 lanes
__device__ __forceinline__ int wave_scan_max(int v) {
    v = imax(v, dpp_get_i<PF_DPP_ROW_SHR(1)>(v, 0));
    v = imax(v, dpp_get_i<PF_DPP_ROW_SHR(2)>(v, 0));
    v = imax(v, dpp_get_i<PF_DPP_ROW_SHR(4)>(v, 0));
    v = imax(v, dpp_get_i<PF_DPP_ROW_SHR(8)>(v, 0));
    v = imax(v, dpp_get_i<PF_DPP_ROW_BCAST15, 0xA>(v, 0));
    v = imax(v, dpp_get_i<PF_DPP_ROW_BCAST31, 0xC>(v, 0));
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// workgroup reductions through LDS (result broadcast to every thread).  `red` must hold >= K*PF_NWAVES Ts.
// ---------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T block_max(T v, T* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    v = wave_max(v);
    __syncthreads();  // protect `red` against the previous user
    if (lane == 0) red[wid] = v;
    __syncthreads();
    T r = red[0];
#pragma unroll
    for (int w = 1; w < PF_NWAVES; ++w) r = (red[w] > r) ? red[w] : r;
    return r;
}

template <int K> __device__ __forceinline__ void block_sum(double (&v)[K], double* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) red[k * PF_NWAVES + wid] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double r = red[k * PF_NWAVES];
#pragma unroll
        for (int w = 1; w < PF_NWAVES; ++w) r += red[k * PF_NWAVES + w];
        v[k] = r;
    }
}

// Exclusive scan of one double per thread across the workgroup; returns the exclusive prefix for this thread and
// the workgroup total in `total`.  `red` must hold >= PF_NWAVES doubles.
__device__ __forceinline__ double block_scan_excl(double v, double* red, double& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double incl = wave_scan_incl(v, lane);
    __syncthreads();
    if (lane == 63) red[wid] = incl;
    __syncthreads();
    double wave_off = 0.0, tot = 0.0;
#pragma unroll
    for (int w = 0; w < PF_NWAVES; ++w) {
        double s = red[w];
        if (w < wid) wave_off += s;
        tot += s;
    }
    total = tot;
    return wave_off + incl - v;
}

// ---------------------------------------------------------------------------------------------------------------
// vector load / store of VEC consecutive elements
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int VEC> struct alignas(sizeof(T) * VEC) Pack { T v[VEC]; };

template <typename T, int VEC> __device__ __forceinline__ void load_vec(const T* __restrict__ p, T (&out)[VEC]) {
    Pack<T, VEC> q = *reinterpret_cast<const Pack<T, VEC>*>(p);
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[j] = q.v[j];
}
template <typename T, int VEC> __device__ __forceinline__ void store_vec(T* __restrict__ p, const T (&in)[VEC]) {
    Pack<T, VEC> q;
#pragma unroll
    for (int j = 0; j < VEC; ++j) q.v[j] = in[j];
    *reinterpret_cast<Pack<T, VEC>*>(p) = q;
}

// Stores of the step kernel's OUTPUT planes (x', logw', ancestors, the next local scans: 16 + 8 D bytes per particle, never
// read again by the launch that writes them).  Plain stores leave these lines dirty in the write-back L2 until the
// end-of-kernel release writes them back, and the next launch waits for that (MI355X_MICROARCH.md "boundary": + B / 6 TB/s
// behind B dirty bytes - 16.8 MB per step at 2^20 particles).  `sc1` stores write through to the memory side while the
// workgroup is still computing: measured (profiles/r03_out_store.txt, tools/out_store_variants.sh) 16.3 -> 14.9 us per step
// at 2^20 x 1, 37.5 -> 36.9 at 2^22 x 1, 39.5 -> 38.3 at 64 x 65 536; non-temporal stores (policy 1) gain the same at
// 2^20 but lose 4-9 % at the larger shapes; `sc0 sc1` equals `sc1`.  PF_OUT_STORE: 0 plain, 1 non-temporal, 2 write-through.
// Policy 2 goes through a buffer descriptor - `buffer_store_dwordx4 ... offen sc1` from
// __builtin_amdgcn_raw_buffer_store_b128(..., aux = 16) - which, unlike an inline-asm store, the compiler keeps tracking
// (s_waitcnt before a dependent access, the > 64-bit store-data hazard); a volatile store would carry the same cache bits
// but also an `s_waitcnt vmcnt(0)` per access (measured: slower than plain stores on the multi-round shapes).
// `base` must be wave-uniform (it becomes the descriptor in SGPRs), `elem` is the lane's element offset from it.
#ifndef PF_OUT_STORE
#define PF_OUT_STORE 2
#endif
// WT = false: a plain store.  The step kernels of multi-round tiles take those: written through, their shapes measured
// within +-3 % of plain stores either way (profiles/r03_out_store_multi_round_ab.txt) and the descriptors cost them
// registers (24 - 150 B / lane of additional scratch in the 128-VGPR instantiations).
template <typename T, int VEC, bool WT = true>
__device__ __forceinline__ void store_out(T* __restrict__ base, int elem, const T (&in)[VEC]) {
    constexpr int BYTES = (int)sizeof(T) * VEC;
    if constexpr (WT && PF_OUT_STORE != 0 && BYTES % 16 == 0) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        Pack<T, VEC> q;
#pragma unroll
        for (int j = 0; j < VEC; ++j) q.v[j] = in[j];
#pragma unroll
        for (int k = 0; k < BYTES / 16; ++k) {
            u4 v;
            __builtin_memcpy(&v, reinterpret_cast<const char*>(&q) + 16 * k, 16);
            if constexpr (PF_OUT_STORE == 1) {
                __builtin_nontemporal_store(v, reinterpret_cast<u4*>(base + elem) + k);
            } else {
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, elem * (int)sizeof(T) + 16 * k, 0, /*sc1*/ 16);
            }
        }
    } else {
        store_vec<T, VEC>(base + elem, in);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Agent-scope accesses for data handed between workgroups INSIDE a launch (MI355X_MICROARCH.md, "inter-workgroup visibility": a CU's
// L1 is never refreshed by another CU's stores, the per-XCD L2s are not coherent with each other): `sc1` stores write through to
// the memory side, `sc1` loads bypass this CU's L1.  Through a buffer descriptor (`base` wave-uniform, byte offset per lane) so
// the compiler keeps tracking them.  16-byte {3 words, tag} granules written by ONE such store need no flag and no fence.
// ---------------------------------------------------------------------------------------------------------------
typedef unsigned pf_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pf_u4 ld16_sc1(const void* base, int byte_off) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, /*sc1*/ 16);
}
__device__ __forceinline__ void st16_sc1(void* base, int byte_off, pf_u4 v) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, byte_off, 0, /*sc1*/ 16);
}
// one element written through to the memory side (a scalar `sc1` store is one fabric write: for single values only)
template <typename T> __device__ __forceinline__ void st1_sc1(T* base, int elem, T v) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000);
    if constexpr (sizeof(T) == 4) {
        unsigned w;
        __builtin_memcpy(&w, &v, 4);
        __builtin_amdgcn_raw_buffer_store_b32(w, rsrc, elem * 4, 0, /*sc1*/ 16);
    } else {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        u2 w;
        __builtin_memcpy(&w, &v, 8);
        __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, elem * 8, 0, /*sc1*/ 16);
    }
}
// one element of a plane another workgroup wrote with write-through stores (L1 bypassed); `base` wave-uniform
template <typename T> __device__ __forceinline__ T ld1_sc1(const T* base, int elem) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, 0x7FFFFFFF, 0x00020000);
    if constexpr (sizeof(T) == 4) {
        const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(rsrc, elem * 4, 0, /*sc1*/ 16);
        T r;
        __builtin_memcpy(&r, &w, 4);
        return r;
    } else {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const u2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, elem * 8, 0, /*sc1*/ 16);
        T r;
        __builtin_memcpy(&r, &w, 8);
        return r;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// lower_bound of a lane's VEC NON-DECREASING targets in a sorted LDS array (the persistent kernels' ancestor search: a lane's
// positions are consecutive points of the systematic grid).
//   lb_search     the branch-free binary search, rounds unrolled with BYTE offsets (a probe is one ds_read with an immediate
//                 offset; compare + select + add per target and round), rounds above the power of two `np2` skipped (uniform).
//   sorted_lower_bound   all targets by lb_search, side by side (one chain of dependent LDS round trips).  (Two restructurings were
//                 built and measured in round 5 and are not in the source any more - a probe of the eight entries behind the first
//                 target's answer for the lane's other targets, and two binary levels per LDS round trip: identical answers, 0 .. 7 %
//                 slower, profiles/r05_search_probe_ab.txt.)  Entries [0, np2 + PF_LB_PAD) must be readable, +inf behind the data:
//                 the last round reads the entry AT the answer, which may be entry np2.
// ---------------------------------------------------------------------------------------------------------------
#define PF_LB_PAD 8
template <typename T, int NP, int MAXP2>
__device__ __forceinline__ void lb_search(const unsigned char* cb, int np2, const T* __restrict__ p, int* __restrict__ qb) {
    constexpr int SZ = (int)sizeof(T);
#pragma unroll
    for (int j = 0; j < NP; ++j) qb[j] = 0;
#pragma unroll
    for (int st = MAXP2 / 2; st >= 1; st >>= 1) {
        if (st < np2) {
            T v[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) v[j] = *reinterpret_cast<const T*>(cb + qb[j] + (st - 1) * SZ);
#pragma unroll
            for (int j = 0; j < NP; ++j) qb[j] += (v[j] < p[j]) ? st * SZ : 0;
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) qb[j] += (*reinterpret_cast<const T*>(cb + qb[j]) < p[j]) ? SZ : 0;
}
template <typename T, int VEC, int MAXP2>
__device__ __forceinline__ void sorted_lower_bound(const T* win, int np2, const T (&p)[VEC], int (&out)[VEC]) {
    constexpr int SZ = (int)sizeof(T);
    const unsigned char* const cb = reinterpret_cast<const unsigned char*>(win);
    // every target by the search, side by side (ONE chain of dependent LDS round trips)
    int qb[VEC];
    lb_search<T, VEC, MAXP2>(cb, np2, &p[0], qb);
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[j] = qb[j] / SZ;
}

// Online (max, sum-exp) accumulator; sums are carried in double, the exponentials are evaluated in T.
template <typename T> struct OnlineLse {
    T m;
    double s;
    __device__ __forceinline__ void init() {
        m = -Lim<T>::inf();
        s = 0.0;
    }
    // returns the factor by which companion sums must be rescaled (1 when the max did not move) and e = exp(v - m)
    __device__ __forceinline__ void push(T v, double& rescale, double& e) {
        rescale = 1.0;
        if (v > m) {
            rescale = (m == -Lim<T>::inf()) ? 0.0 : (double)pf_exp_w(m - v);
            s *= rescale;
            m = v;
        }
        e = (v == -Lim<T>::inf()) ? 0.0 : (double)pf_exp_w(v - m);
        if (v != v) e = v;  // NaN propagates (an un-sanitised input); sanitised inputs never hit this
        s += e;
    }
};

}  // namespace pf
