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
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
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

__device__ __forceinline__ void box_muller(float u1, float u2, float& z0, float& z1) {
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.28318530717958647692f * u2, &s, &c);
    z0 = r * c;
    z1 = r * s;
}
__device__ __forceinline__ void box_muller(double u1, double u2, double& z0, double& z1) {
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincos(6.28318530717958647692 * u2, &s, &c);
    z0 = r * c;
    z1 = r * s;
}

// D (<= 4) standard normals for element `elem` of step `step`.
template <typename T, int D> struct NormalDraw;

template <int D> struct NormalDraw<float, D> {
    __device__ __forceinline__ static void draw(uint64_t seed, uint32_t stream, uint32_t step, uint64_t elem,
                                                float (&z)[D]) {
        const Philox4 r = philox4x32_10((uint32_t)elem, (uint32_t)(elem >> 32), step, stream, (uint32_t)seed,
                                        (uint32_t)(seed >> 32));
        float a, b;
        box_muller(u01_open0(r.x), u01(r.y), a, b);
        z[0] = a;
        if (D > 1) z[1] = b;
        if (D > 2) {
            box_muller(u01_open0(r.z), u01(r.w), a, b);
            z[2] = a;
            if (D > 3) z[3] = b;
        }
    }
};

template <int D> struct NormalDraw<double, D> {
    __device__ __forceinline__ static void draw(uint64_t seed, uint32_t stream, uint32_t step, uint64_t elem,
                                                double (&z)[D]) {
        Philox4 r = philox4x32_10((uint32_t)elem, (uint32_t)(elem >> 32), step, stream, (uint32_t)seed,
                                  (uint32_t)(seed >> 32));
        double a, b;
        box_muller(u01_open0_d(r.x, r.y), u01_d(r.z, r.w), a, b);
        z[0] = a;
        if (D > 1) z[1] = b;
        if (D > 2) {
            r = philox4x32_10((uint32_t)elem, (uint32_t)(elem >> 32), step, stream | 0x100u, (uint32_t)seed,
                              (uint32_t)(seed >> 32));
            box_muller(u01_open0_d(r.x, r.y), u01_d(r.z, r.w), a, b);
            z[2] = a;
            if (D > 3) z[3] = b;
        }
    }
};

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
