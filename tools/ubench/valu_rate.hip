// DEVELOPMENT TOOL - VALU issue rates on gfx950, measured: how many cycles a wave64 instruction occupies its SIMD for
// v_fma_f32, v_pk_fma_f32, v_fma_f64, v_exp_f32, v_mov_dpp.  Answers "does packed fp32 buy issue slots here?" before the
// step kernel is rewritten around it.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHAINS 8
template <int KIND> __global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[CHAINS];
    f2 p[CHAINS];
    double d[CHAINS];
    for (int i = 0; i < CHAINS; ++i) {
        x[i] = threadIdx.x * 1e-3f + i;
        p[i] = f2{x[i], x[i] + 1.f};
        d[i] = x[i];
    }
    const f2 a2{a, a}, b2{b, b};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (KIND == 0) x[i] = __builtin_fmaf(x[i], a, b);
            if (KIND == 1) p[i] = __builtin_elementwise_fma(p[i], a2, b2);
            if (KIND == 2) d[i] = __builtin_fma(d[i], (double)a, (double)b);
            if (KIND == 3) x[i] = __builtin_amdgcn_exp2f(x[i]);
            if (KIND == 4) x[i] = x[i] + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[i]), 0x111, 0xf, 0xf, false));
            if (KIND == 5) x[i] = __builtin_amdgcn_logf(x[i]);
            if (KIND == 6) x[i] = __builtin_amdgcn_sinf(x[i]);
            if (KIND == 7) x[i] = __builtin_amdgcn_rcpf(x[i]);
            if (KIND == 8) x[i] = __builtin_amdgcn_sqrtf(x[i]);
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < CHAINS; ++i) s += x[i] + p[i].x + p[i].y + (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;
}

template <int KIND> void run(const char* name, int wgs_per_cu, float* out) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<KIND><<<256 * wgs_per_cu, 256>>>(out, 16, 1.0001f, 1e-7f);
    hipEventRecord(e0);
    k<KIND><<<256 * wgs_per_cu, 256>>>(out, iters, 1.0001f, 1e-7f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long cyc;
    hipMemcpy(&cyc, (char*)out + (size_t)(1 << 20) * 8, 8, hipMemcpyDeviceToHost);
    // wave-instructions per SIMD: wgs_per_cu waves on each SIMD (256 threads = 4 waves = one per SIMD), CHAINS * iters each
    const double insts = (double)wgs_per_cu * CHAINS * iters;
    printf("%-14s %d waves/SIMD: %8.3f ms  -> %6.2f ns per wave-instruction per SIMD  (clock64 delta %lld for one wave = %.2f ticks/inst)\n",
           name, wgs_per_cu, ms, ms * 1e6 / insts, cyc, (double)cyc / (CHAINS * iters));
}

int main() {
    float* out;
    hipMalloc(&out, ((size_t)1 << 20) * 8 + 64);
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("v_fma_f32", 1, out); run<1>("v_pk_fma_f32", 1, out); run<2>("v_fma_f64", 1, out); run<3>("v_exp_f32", 1, out); run<4>("v_add+dpp", 1, out); run<5>("v_log_f32", 1, out); run<6>("v_sin_f32", 1, out); run<7>("v_rcp_f32", 1, out); run<8>("v_sqrt_f32", 1, out); }
        if (w == 2) { run<0>("v_fma_f32", 2, out); run<1>("v_pk_fma_f32", 2, out); run<2>("v_fma_f64", 2, out); run<3>("v_exp_f32", 2, out); }
        if (w == 4) { run<0>("v_fma_f32", 4, out); run<1>("v_pk_fma_f32", 4, out); run<2>("v_fma_f64", 4, out); run<3>("v_exp_f32", 4, out); run<4>("v_add+dpp", 4, out); }
    }
    return 0;
}
