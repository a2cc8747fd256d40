// pf_philox.hpp - counter-based RNG for in-kernel draws (Philox4x32-10, Salmon et al. SC'11), written for this
// library.  One call = 4 x 32 random bits addressed by (seed, stream, step, element), so every kernel that needs
// "the draw of particle i of column b at step t" regenerates the same bits without any state in HBM.
//
// The reference draws from torch's CPU mt19937 stream; bit-matching that stream on the GPU
// is not possible, so parity runs inject the draws as a tape and performance runs use this generator.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pf {

struct Philox4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32), not a mul_hi / mul_lo pair
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += W0;
        k1 += W1;
    }
    return Philox4{c0, c1, c2, c3};
}
// The same with the (wave-uniform) key carried in vector registers: the compiler would otherwise precompute the 20 round
// keys in scalar registers - in the step kernel, whose scalar file is oversubscribed, each of them costs a spill and a
// reload (v_writelane / v_readlane) on top of its s_add; two v_add per round are cheaper.
__device__ __forceinline__ Philox4 philox4x32_10_vkey(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                      uint32_t k1) {
    uint32_t v0, v1;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v0) : "s"(k0));
    asm volatile("v_mov_b32 %0, %1" : "=v"(v1) : "s"(k1));
    return philox4x32_10(c0, c1, c2, c3, v0, v1);
}

// streams
enum { PF_STREAM_NORMAL = 0, PF_STREAM_UNIFORM = 1, PF_STREAM_INIT = 2, PF_STREAM_MULTINOMIAL = 3 };

// uniform in (0, 1]: never 0 so log() is safe
__device__ __forceinline__ float u01_open0(uint32_t a) { return ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f); }
// uniform in [0, 1)
__device__ __forceinline__ float u01(uint32_t a) { return (float)(a >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ double u01_open0_d(uint32_t a, uint32_t b) {
    const uint64_t v = (((uint64_t)a << 32) | b) >> 11;  // 53 bits
    return ((double)v + 1.0) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double u01_d(uint32_t a, uint32_t b) {
    const uint64_t v = (((uint64_t)a << 32) | b) >> 11;
    return (double)v * (1.0 / 9007199254740992.0);
}

// float Box-Muller on the hardware transcendentals: v_log_f32, v_sqrt_f32 and v_sin_f32 / v_cos_f32, whose argument is
// in revolutions - exactly the 2*pi*u2 of Box-Muller.  ~10 VALU instructions per pair (the libm versions cost ~150).
__device__ __forceinline__ void box_muller(float u1, float u2, float& z0, float& z1) {
    // -2 ln u = (-2 ln 2) log2 u, u in [2^-24, 1]: the bare v_log_f32
    const float r = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1));
    z0 = r * __builtin_amdgcn_cosf(u2);
    z1 = r * __builtin_amdgcn_sinf(u2);
}
__device__ __forceinline__ void box_muller(double u1, double u2, double& z0, double& z1) {
    const double r = sqrt(-2.0 * log(u1));
    const double a = 6.28318530717958647692 * u2;
    z0 = r * cos(a);
    z1 = r * sin(a);
}

// Standard normals are addressed by a flat index n = elem * D + d.  One Philox call yields NPC of them (4 for float,
// 2 for double), so the VEC consecutive particles a thread owns share calls: D = 1, float, VEC = 4 -> one call per
// thread per round instead of four.
template <typename T> struct NormalCall;
template <> struct NormalCall<float> {
    static constexpr int NPC = 4;
    __device__ __forceinline__ static void call(uint64_t seed, uint32_t stream, uint32_t step, uint64_t c, float (&z)[4]) {
        const Philox4 r = philox4x32_10_vkey((uint32_t)c, (uint32_t)(c >> 32), step, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
        box_muller(u01_open0(r.x), u01(r.y), z[0], z[1]);
        box_muller(u01_open0(r.z), u01(r.w), z[2], z[3]);
    }
};
template <> struct NormalCall<double> {
    static constexpr int NPC = 2;
    __device__ __forceinline__ static void call(uint64_t seed, uint32_t stream, uint32_t step, uint64_t c, double (&z)[2]) {
        const Philox4 r = philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), step, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
        box_muller(u01_open0_d(r.x, r.y), u01_d(r.z, r.w), z[0], z[1]);
    }
};

// z[j][d] for the VEC consecutive particles elem0 .. elem0 + VEC - 1
// (VEC > 1 implies N % VEC == 0 and elem0 % VEC == 0 - the geometry's rule - so the aligned branch is known statically)
template <typename T, int D, int VEC>
__device__ __forceinline__ void draw_normals(uint64_t seed, uint32_t stream, uint32_t step, uint64_t elem0, T (&z)[VEC][D]) {
    constexpr int NPC = NormalCall<T>::NPC;
    const uint64_t n0 = elem0 * D;
    if constexpr (VEC > 1) __builtin_assume(elem0 % VEC == 0);
    if ((VEC * D) % NPC == 0 && ((VEC > 1 && VEC % NPC == 0) || (n0 % NPC) == 0)) {
        // aligned: exactly VEC * D / NPC calls, every normal used
#pragma unroll
        for (int c = 0; c < (VEC * D) / NPC; ++c) {
            T zz[NPC];
            NormalCall<T>::call(seed, stream, step, n0 / NPC + c, zz);
#pragma unroll
            for (int q = 0; q < NPC; ++q) {
                const int n = c * NPC + q;
                z[n / D][n % D] = zz[q];
            }
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < VEC * D; ++n) {
        T zz[NPC];
        NormalCall<T>::call(seed, stream, step, (n0 + n) / NPC, zz);
        const int q = (int)((n0 + n) % NPC);
        T v = zz[0];
#pragma unroll
        for (int k = 1; k < NPC; ++k) v = (q == k) ? zz[k] : v;
        z[n / D][n % D] = v;
    }
}

// The same numbers for VEC consecutive particles starting at ANY element (no vector boundary: a column of N % VEC != 0
// particles in the column-persistent kernel): the calls covering flat indices [elem0 D, (elem0 + VEC) D) - one more than
// the aligned case needs - and a select by the start's offset inside its call.
template <typename T, int D, int VEC>
__device__ __forceinline__ void draw_normals_ragged(uint64_t seed, uint32_t stream, uint32_t step, uint64_t elem0, T (&z)[VEC][D]) {
    constexpr int NPC = NormalCall<T>::NPC;
    constexpr int TOT = VEC * D;
    constexpr int K = (TOT + NPC - 1) / NPC + 1;
    const uint64_t n0 = elem0 * D;
    const uint64_t c0 = n0 / NPC;
    const int r = (int)(n0 % NPC);
    T buf[K * NPC];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        T zz[NPC];
        NormalCall<T>::call(seed, stream, step, c0 + k, zz);
#pragma unroll
        for (int q = 0; q < NPC; ++q) buf[k * NPC + q] = zz[q];
    }
#pragma unroll
    for (int n = 0; n < TOT; ++n) {
        T v = buf[n];
#pragma unroll
        for (int q = 1; q < NPC; ++q) v = (r == q) ? buf[n + q] : v;
        z[n / D][n % D] = v;
    }
}

// D (<= 4) standard normals for the single element `elem`
template <typename T, int D> struct NormalDraw {
    __device__ __forceinline__ static void draw(uint64_t seed, uint32_t stream, uint32_t step, uint64_t elem, T (&z)[D]) {
        T zz[1][D];
        draw_normals<T, D, 1>(seed, stream, step, elem, zz);
#pragma unroll
        for (int d = 0; d < D; ++d) z[d] = zz[0][d];
    }
};

__device__ __forceinline__ float neg_log_u(uint32_t a) {  // -ln u, u in (0, 1]
    return -0.693147180559945309417f * __builtin_amdgcn_logf(u01_open0(a));
}
// Exp(1) draws for the sorted-uniform (exponential spacings) multinomial resampler, VEC consecutive elements.
template <typename T, int VEC>
__device__ __forceinline__ void draw_exponentials(uint64_t seed, uint32_t stream, uint32_t step, uint64_t elem0, T (&e)[VEC]);
template <> __device__ __forceinline__ void draw_exponentials<float, 4>(uint64_t seed, uint32_t stream, uint32_t step,
                                                                        uint64_t elem0, float (&e)[4]) {
    const uint64_t c = elem0 >> 2;  // elem0 % 4 == 0: one call serves the four elements
    const Philox4 r = philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), step, stream | 0x200u, (uint32_t)seed, (uint32_t)(seed >> 32));
    e[0] = neg_log_u(r.x);
    e[1] = neg_log_u(r.y);
    e[2] = neg_log_u(r.z);
    e[3] = neg_log_u(r.w);
}
template <> __device__ __forceinline__ void draw_exponentials<float, 1>(uint64_t seed, uint32_t stream, uint32_t step,
                                                                        uint64_t elem0, float (&e)[1]) {
    const uint64_t c = elem0 >> 2;
    const Philox4 r = philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), step, stream | 0x200u, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t q = (uint32_t)(elem0 & 3);
    const uint32_t w = q == 0 ? r.x : (q == 1 ? r.y : (q == 2 ? r.z : r.w));
    e[0] = neg_log_u(w);
}
template <> __device__ __forceinline__ void draw_exponentials<double, 4>(uint64_t seed, uint32_t stream, uint32_t step,
                                                                         uint64_t elem0, double (&e)[4]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint64_t c = (elem0 >> 1) + h;
        const Philox4 r = philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), step, stream | 0x200u, (uint32_t)seed, (uint32_t)(seed >> 32));
        e[2 * h] = -log(u01_open0_d(r.x, r.y));
        e[2 * h + 1] = -log(u01_open0_d(r.z, r.w));
    }
}
template <> __device__ __forceinline__ void draw_exponentials<double, 1>(uint64_t seed, uint32_t stream, uint32_t step,
                                                                         uint64_t elem0, double (&e)[1]) {
    const uint64_t c = elem0 >> 1;
    const Philox4 r = philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), step, stream | 0x200u, (uint32_t)seed, (uint32_t)(seed >> 32));
    e[0] = (elem0 & 1) ? -log(u01_open0_d(r.z, r.w)) : -log(u01_open0_d(r.x, r.y));
}

template <typename T> __device__ __forceinline__ T uniform_draw(uint64_t seed, uint32_t stream, uint32_t step,
                                                               uint64_t elem);
template <> __device__ __forceinline__ float uniform_draw<float>(uint64_t seed, uint32_t stream, uint32_t step,
                                                                 uint64_t elem) {
    const Philox4 r = philox4x32_10((uint32_t)elem, (uint32_t)(elem >> 32), step, stream, (uint32_t)seed,
                                    (uint32_t)(seed >> 32));
    return u01(r.x);
}
template <> __device__ __forceinline__ double uniform_draw<double>(uint64_t seed, uint32_t stream, uint32_t step,
                                                                   uint64_t elem) {
    const Philox4 r = philox4x32_10((uint32_t)elem, (uint32_t)(elem >> 32), step, stream, (uint32_t)seed,
                                    (uint32_t)(seed >> 32));
    return u01_d(r.x, r.y);
}

}  // namespace pf
