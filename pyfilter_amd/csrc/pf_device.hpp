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

#define PF_BLOCK 256
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

// torch.nan_to_num_(w, nan=-inf, posinf=-inf) as pyfilter/utils.py:57 calls it: NaN -> -inf, +inf -> -inf and -
// because ``neginf`` is left at its default - -inf -> the lowest finite value of the dtype.
template <typename T> __device__ __forceinline__ T sanitize_logw(T v) {
    if (v != v) return -Lim<T>::inf();
    if (v == Lim<T>::inf()) return -Lim<T>::inf();
    if (v == -Lim<T>::inf()) return Lim<T>::lowest();
    return v;
}

__device__ __forceinline__ float pf_exp(float x) { return expf(x); }
__device__ __forceinline__ double pf_exp(double x) { return exp(x); }
__device__ __forceinline__ float pf_log(float x) { return logf(x); }
__device__ __forceinline__ double pf_log(double x) { return log(x); }
__device__ __forceinline__ float pf_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double pf_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float pf_sin(float x) { return sinf(x); }
__device__ __forceinline__ double pf_sin(double x) { return sin(x); }
__device__ __forceinline__ float pf_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double pf_abs(double x) { return fabs(x); }

// exp(a - b) with the convention exp(-inf - anything) = 0 (so empty / all -inf partials merge cleanly)
__device__ __forceinline__ double exp_diff(double a, double b) {
    return (a == -__builtin_huge_val()) ? 0.0 : exp(a - b);
}

// ---------------------------------------------------------------------------------------------------------------
// wave64 shuffles
// ---------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T other = __shfl_xor(v, o, PF_WAVE);
        v = (other > v) ? other : v;
    }
    return v;
}
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, PF_WAVE);
    return v;
}
// inclusive scan across the 64 lanes of a wave
template <typename T> __device__ __forceinline__ T wave_scan_incl(T v, int lane) {
#pragma unroll
    for (int o = 1; o < PF_WAVE; o <<= 1) {
        T up = __shfl_up(v, o, PF_WAVE);
        if (lane >= o) v += up;
    }
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
            rescale = (m == -Lim<T>::inf()) ? 0.0 : (double)pf_exp(m - v);
            s *= rescale;
            m = v;
        }
        e = (v == -Lim<T>::inf()) ? 0.0 : (double)pf_exp(v - m);
        if (v != v) e = v;  // NaN propagates (an un-sanitised input); sanitised inputs never hit this
        s += e;
    }
};

}  // namespace pf
