// TEST PROGRAM - libpfamd.so driven through include/pf_amd.h alone: no Python, no PyTorch, device memory straight from
// the HIP runtime.  It is what a binding from any other host language does: describe the model (pf_model), size the
// workspace, hand over raw device pointers, run T steps (pf_filter_run), read the results back.  AR(1) + linear Gaussian
// observation (tests/filters/models.py:13-15 of the reference), SISR + Bootstrap + systematic and APF + the optimal
// proposal - and the same APF as 16 filters of 8 192 particles on the opt-in column-cluster route -, checked against the exact
// Kalman filter computed here on the host.
//   g++ -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I<repo>/include standalone.cpp -o standalone \
//       -L<repo>/pyfilter_amd -lpfamd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,<repo>/pyfilter_amd
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pf_amd.h"

#define HIP_OK(call)                                                                    \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            std::fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));      \
            return 2;                                                                   \
        }                                                                               \
    } while (0)
#define PF_CALL(call)                                                                   \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != PF_OK) {                                                             \
            std::fprintf(stderr, "%s failed: %s (%d)\n", #call, pf_error_string(rc_), rc_); \
            return 3;                                                                   \
        }                                                                               \
    } while (0)

static uint64_t lcg_state = 0x2545F4914F6CDD1Dull;
static double uniform01() {
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return ((lcg_state >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
static double gauss() { return std::sqrt(-2.0 * std::log(uniform01())) * std::cos(6.283185307179586 * uniform01()); }

int main() {
    const int64_t N = 1 << 16, B = 2, D = 1, T = 64;
    const double alpha = 0.0, beta = 0.99, sigma = 0.05, a = 1.0, b = 0.0, s = 0.15, m0 = 0.0, s0 = 0.05;

    // data + exact Kalman filter (host, double)
    std::vector<float> y(T);
    double x = m0 + s0 * gauss();
    for (int t = 0; t < T; ++t) {
        x = alpha + beta * x + sigma * gauss();
        y[t] = (float)(a * x + b + s * gauss());
    }
    double km = m0, kp = s0 * s0, kll = 0.0;
    for (int t = 0; t < T; ++t) {
        const double mp = alpha + beta * km, pp = beta * beta * kp + sigma * sigma;
        const double S = a * a * pp + s * s, r = (double)y[t] - (a * mp + b);
        kll += -0.5 * (std::log(2.0 * M_PI * S) + r * r / S);
        const double K = pp * a / S;
        km = mp + K * r;
        kp = (1.0 - K * a) * pp;
    }

    // device buffers
    float *x0, *x1, *w0, *w1, *cdf, *pos, *yd, *means, *vars, *ll_steps, *ll_total, *params;
    int32_t* anc;
    void* ws;
    size_t ws_bytes = 0;
    PF_CALL(pf_workspace_bytes(N, B, D, &ws_bytes));
    const size_t plane = sizeof(float) * B * N;
    HIP_OK(hipMalloc((void**)&x0, plane * D));
    HIP_OK(hipMalloc((void**)&x1, plane * D));
    HIP_OK(hipMalloc((void**)&w0, plane));
    HIP_OK(hipMalloc((void**)&w1, plane));
    HIP_OK(hipMalloc((void**)&cdf, plane));
    HIP_OK(hipMalloc((void**)&pos, plane));
    HIP_OK(hipMalloc((void**)&anc, sizeof(int32_t) * B * N));
    HIP_OK(hipMalloc((void**)&yd, sizeof(float) * T));
    HIP_OK(hipMalloc((void**)&means, sizeof(float) * (T + 1) * B * D));
    HIP_OK(hipMalloc((void**)&vars, sizeof(float) * (T + 1) * B * D));
    HIP_OK(hipMalloc((void**)&ll_steps, sizeof(float) * T * B));
    HIP_OK(hipMalloc((void**)&ll_total, sizeof(float) * B));
    HIP_OK(hipMalloc(&ws, ws_bytes));
    const int NP = 4 * (int)D + 1 * (int)D + 2 * 1;  // [hp0 hp1 hp2 hp3 | A | b | s]
    std::vector<float> prow(B * NP);
    for (int c = 0; c < B; ++c) {
        const float row[7] = {(float)alpha, (float)beta, (float)sigma, 0.f, (float)a, (float)b, (float)s};
        for (int k = 0; k < NP; ++k) prow[c * NP + k] = row[k];
    }
    HIP_OK(hipMalloc((void**)&params, sizeof(float) * prow.size()));
    HIP_OK(hipMemcpy(params, prow.data(), sizeof(float) * prow.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(yd, y.data(), sizeof(float) * T, hipMemcpyHostToDevice));
    std::vector<int32_t> iota(B * N);
    for (int64_t i = 0; i < B * N; ++i) iota[i] = (int32_t)(i % N);
    std::vector<uint8_t> observed(T, 1);

    int failures = 0;
    for (int variant = 0; variant < 2; ++variant) {
        pf_filter_args A = {};
        A.struct_size = sizeof(A);  // ABI 2: the block names the header it was built against (hints: all zero = the library's choices)
        A.model.hid_kind = PF_HID_LINEAR;
        A.model.obs_kind = PF_OBS_LINEAR;
        A.model.dim = (int32_t)D;
        A.model.obs_dim = 1;
        A.model.dt = 1.0;
        A.model.inc_scale = 1.0;
        A.model.params = params;
        A.filter = variant == 0 ? PF_FILTER_SISR : PF_FILTER_APF;
        A.proposal = variant == 0 ? PF_PROP_BOOTSTRAP : PF_PROP_LGO;
        A.resampler = PF_RESAMPLE_SYSTEMATIC;
        A.dtype = PF_F32;
        A.N = N;
        A.B = B;
        A.ess_threshold = 0.9;
        A.seed = 1234 + variant;
        A.x[0] = x0; A.x[1] = x1;
        A.logw[0] = w0; A.logw[1] = w1;
        A.anc = anc; A.cdf = cdf; A.pos = pos;
        A.y = yd; A.y_rows = 1; A.observed = observed.data();
        A.means = means; A.vars = vars; A.ll_steps = ll_steps; A.ll_total = ll_total;
        A.ws = ws; A.ws_bytes = ws_bytes;
        const double m0v[1] = {m0}, s0v[1] = {s0};
        PF_CALL(pf_initial_sample(m0v, s0v, nullptr, A.seed ^ 0x9E3779B97F4A7C15ull, x0, N, B, D, PF_F32, nullptr));
        HIP_OK(hipMemset(w0, 0, plane));
        HIP_OK(hipMemset(ll_total, 0, sizeof(float) * B));
        HIP_OK(hipMemcpy(anc, iota.data(), sizeof(int32_t) * B * N, hipMemcpyHostToDevice));
        PF_CALL(pf_filter_run(&A, 0, T, 1, nullptr));
        HIP_OK(hipDeviceSynchronize());
        std::vector<float> h_means((T + 1) * B * D), h_ll(B);
        HIP_OK(hipMemcpy(h_means.data(), means, sizeof(float) * h_means.size(), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(h_ll.data(), ll_total, sizeof(float) * B, hipMemcpyDeviceToHost));
        for (int c = 0; c < B; ++c) {
            const double last = h_means[(T * B + c) * D];
            std::printf("%s column %d: loglikelihood %.4f (Kalman %.4f)  final mean %.5f (Kalman %.5f)\n",
                        variant == 0 ? "SISR+Bootstrap" : "APF+LGO", c, h_ll[c], kll, last, km);
            if (!(std::fabs(h_ll[c] - kll) < 0.25) || !(std::fabs(last - km) < 0.01)) ++failures;
        }
    }
    // ---- the column-cluster route (include/pf_amd.h: PF_ROUTE_CLUSTER): 16 filters of 8 192 particles - the same number of
    // particles, so the state buffers above serve - held in registers by 8 workgroups each for the whole run, ONE launch.  Opt-in
    // through pf_run_hints, with the contract of the route: the caller passes a status word, looks at it where it waits for the
    // device anyway, and re-issues the piece with PF_ROUTE_PER_STEP when a launch reported that it gave up - driven below both
    // ways: the default patience (nothing gives up) and cluster_patience = -1 (every launch gives up; the fallback's result counts).
    {
        const int64_t N2 = 8192, B2 = 16;
        static_assert(8192 * 16 == (1 << 16) * 2, "the cluster block reuses the state buffers");
        float *params2, *means2, *vars2, *ll_steps2, *ll_total2;
        void* ws2;
        size_t ws2_bytes = 0;
        PF_CALL(pf_workspace_bytes(N2, B2, D, &ws2_bytes));
        HIP_OK(hipMalloc(&ws2, ws2_bytes));
        HIP_OK(hipMalloc((void**)&means2, sizeof(float) * (T + 1) * B2 * D));
        HIP_OK(hipMalloc((void**)&vars2, sizeof(float) * (T + 1) * B2 * D));
        HIP_OK(hipMalloc((void**)&ll_steps2, sizeof(float) * T * B2));
        HIP_OK(hipMalloc((void**)&ll_total2, sizeof(float) * B2));
        std::vector<float> prow2(B2 * NP);
        for (int c = 0; c < B2; ++c) {
            const float row[7] = {(float)alpha, (float)beta, (float)sigma, 0.f, (float)a, (float)b, (float)s};
            for (int k = 0; k < NP; ++k) prow2[c * NP + k] = row[k];
        }
        HIP_OK(hipMalloc((void**)&params2, sizeof(float) * prow2.size()));
        HIP_OK(hipMemcpy(params2, prow2.data(), sizeof(float) * prow2.size(), hipMemcpyHostToDevice));
        pf_filter_args A = {};
        A.struct_size = sizeof(A);
        A.hints.route = PF_ROUTE_CLUSTER;
        A.model.hid_kind = PF_HID_LINEAR;
        A.model.obs_kind = PF_OBS_LINEAR;
        A.model.dim = (int32_t)D;
        A.model.obs_dim = 1;
        A.model.dt = 1.0;
        A.model.inc_scale = 1.0;
        A.model.params = params2;
        A.filter = PF_FILTER_APF;
        A.proposal = PF_PROP_LGO;
        A.resampler = PF_RESAMPLE_SYSTEMATIC;
        A.dtype = PF_F32;
        A.N = N2;
        A.B = B2;
        A.ess_threshold = 0.9;
        A.seed = 4321;
        A.x[0] = x0; A.x[1] = x1;
        A.logw[0] = w0; A.logw[1] = w1;
        A.anc = anc; A.cdf = cdf; A.pos = pos;
        A.y = yd; A.y_rows = 1; A.observed = observed.data();
        A.means = means2; A.vars = vars2; A.ll_steps = ll_steps2; A.ll_total = ll_total2;
        A.ws = ws2; A.ws_bytes = ws2_bytes;
        int32_t* status_dev = nullptr;
        HIP_OK(hipMalloc((void**)&status_dev, sizeof(int32_t)));
        HIP_OK(hipMemset(status_dev, 0, sizeof(int32_t)));
        A.status = status_dev;
        const double m0v[1] = {m0}, s0v[1] = {s0};
        for (int64_t i = 0; i < B2 * N2; ++i) iota[i] = (int32_t)(i % N2);
        auto load_state = [&]() -> int {  // the incoming state of the run: what a re-issue starts from again
            PF_CALL(pf_initial_sample(m0v, s0v, nullptr, A.seed ^ 0x9E3779B97F4A7C15ull, x0, N2, B2, D, PF_F32, nullptr));
            HIP_OK(hipMemset(w0, 0, plane));
            HIP_OK(hipMemset(ll_total2, 0, sizeof(float) * B2));
            HIP_OK(hipMemcpy(anc, iota.data(), sizeof(int32_t) * B2 * N2, hipMemcpyHostToDevice));
            return 0;
        };
        int32_t trace[10] = {0};
        int got = 0;
        // 1. cluster_patience = -1: the launch gives up, says so, and the piece is issued again on the per-step route
        {
            std::vector<float> h_forced(B2), h_per_step(B2);
            if (load_state()) return 1;
            A.hints.cluster_patience = -1;
            PF_CALL(pf_filter_run(&A, 0, T, 1, nullptr));
            HIP_OK(hipDeviceSynchronize());
            int32_t st = 0;
            HIP_OK(hipMemcpy(&st, status_dev, sizeof(st), hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(h_forced.data(), ll_total2, sizeof(float) * B2, hipMemcpyDeviceToHost));
            int nan_ll = 0;
            for (int c = 0; c < B2; ++c) nan_ll += std::isnan(h_forced[c]) ? 1 : 0;
            if (!(st & 1) || nan_ll == 0) {
                std::printf("FAIL: a cluster launch with cluster_patience = -1 did not report (status %d, %d NaN log-likelihoods)\n", st, nan_ll);
                ++failures;
            }
            HIP_OK(hipMemset(status_dev, 0, sizeof(int32_t)));  // (the caller's word: the library only ever sets bits)
            if (load_state()) return 1;
            A.hints.route = PF_ROUTE_PER_STEP;
            PF_CALL(pf_filter_run(&A, 0, T, 1, nullptr));
            HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipMemcpy(&st, status_dev, sizeof(st), hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(h_per_step.data(), ll_total2, sizeof(float) * B2, hipMemcpyDeviceToHost));
            int bad = st != 0;
            for (int c = 0; c < B2; ++c) bad += !(std::fabs(h_per_step[c] - kll) < 0.6);
            std::printf("cluster route, cluster_patience = -1: status %d -> re-issued with PF_ROUTE_PER_STEP: loglikelihood %.4f (Kalman %.4f)\n",
                        nan_ll ? 1 : 0, h_per_step[0], kll);
            if (bad) ++failures;
            A.hints.route = PF_ROUTE_CLUSTER;
            A.hints.cluster_patience = 0;
        }
        // 2. the default patience
        if (load_state()) return 1;
        PF_CALL(pf_filter_run(&A, 0, T, 1, nullptr));
        HIP_OK(hipDeviceSynchronize());
        {
            int32_t st = -1;
            HIP_OK(hipMemcpy(&st, status_dev, sizeof(st), hipMemcpyDeviceToHost));
            if (st != 0) {
                std::printf("FAIL: the cluster run reported status %d\n", st);
                ++failures;
            }
        }
        got = pf_debug_launch_trace(trace, 1);  // which kernel the library took: field 7 = 10 for the cluster kernel
        std::vector<float> h_means((T + 1) * B2 * D), h_ll(B2);
        HIP_OK(hipMemcpy(h_means.data(), means2, sizeof(float) * h_means.size(), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(h_ll.data(), ll_total2, sizeof(float) * B2, hipMemcpyDeviceToHost));
        double ll_mean = 0.0;
        for (int c = 0; c < B2; ++c) {
            const double last = h_means[(T * B2 + c) * D];
            ll_mean += h_ll[c] / B2;
            if (c < 2)
                std::printf("APF+LGO, cluster route (launch trace %d) column %d: loglikelihood %.4f (Kalman %.4f)  final mean %.5f (Kalman %.5f)\n",
                            got == 1 ? trace[7] : -1, c, h_ll[c], kll, last, km);
            if (!(std::fabs(h_ll[c] - kll) < 0.6) || !(std::fabs(last - km) < 0.03)) ++failures;  // (8 192 particles per filter)
        }
        if (got != 1 || trace[7] != 10 || !(std::fabs(ll_mean - kll) < 0.25)) ++failures;
    }
    // ---- the theta-level entry points (what an SMC^2 driver in any host language does with the filters' log-likelihoods):
    // ESS of B_t log-weights, their systematic resampling, the Gaussian fit of the proposal - against the same arithmetic on
    // the host in double
    {
        const int64_t BT = 1000;
        const int P = 2;
        std::vector<float> lw(BT), vals(BT * P);
        for (int64_t i = 0; i < BT; ++i) {
            lw[i] = (float)(2.0 * gauss());
            vals[i * P] = (float)gauss();
            vals[i * P + 1] = (float)(0.5 * vals[i * P] + 0.3 * gauss());
        }
        double mx = -1e300, sw = 0.0, sw2 = 0.0, m[2] = {0.0, 0.0}, c[3] = {0.0, 0.0, 0.0};
        for (float v : lw) mx = v > mx ? v : mx;
        for (int64_t i = 0; i < BT; ++i) { const double e = std::exp(lw[i] - mx); sw += e; sw2 += e * e; }
        for (int64_t i = 0; i < BT; ++i) { const double w = std::exp(lw[i] - mx) / sw; m[0] += w * vals[i * P]; m[1] += w * vals[i * P + 1]; }
        for (int64_t i = 0; i < BT; ++i) {
            const double w = std::exp(lw[i] - mx) / sw, d0 = vals[i * P] - m[0], d1 = vals[i * P + 1] - m[1];
            c[0] += w * d0 * d0; c[1] += w * d0 * d1; c[2] += w * d1 * d1;
        }
        const double l00 = std::sqrt(c[0]), l10 = c[1] / l00, l11 = std::sqrt(c[2] - l10 * l10);
        float *d_lw, *d_vals, *d_stats, *d_scratch, *d_mean, *d_chol;
        int64_t* d_idx;
        HIP_OK(hipMalloc((void**)&d_lw, sizeof(float) * BT));
        HIP_OK(hipMalloc((void**)&d_vals, sizeof(float) * BT * P));
        HIP_OK(hipMalloc((void**)&d_stats, sizeof(float) * 2));
        HIP_OK(hipMalloc((void**)&d_scratch, sizeof(float) * BT));
        HIP_OK(hipMalloc((void**)&d_mean, sizeof(float) * P));
        HIP_OK(hipMalloc((void**)&d_chol, sizeof(float) * P * P));
        HIP_OK(hipMalloc((void**)&d_idx, sizeof(int64_t) * BT));
        HIP_OK(hipMemcpy(d_lw, lw.data(), sizeof(float) * BT, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_vals, vals.data(), sizeof(float) * BT * P, hipMemcpyHostToDevice));
        PF_CALL(pf_theta_ess(d_lw, 1, BT, PF_F32, d_stats, nullptr));
        PF_CALL(pf_theta_resample(d_lw, BT, 0.37, PF_F32, d_idx, d_scratch, nullptr));
        PF_CALL(pf_theta_fit(d_vals, d_lw, BT, P, 1.1, PF_F32, d_mean, d_chol, nullptr));
        HIP_OK(hipDeviceSynchronize());
        float stats[2], mean[2], chol[4];
        std::vector<int64_t> idx(BT);
        HIP_OK(hipMemcpy(stats, d_stats, sizeof(stats), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(mean, d_mean, sizeof(mean), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(chol, d_chol, sizeof(chol), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(idx.data(), d_idx, sizeof(int64_t) * BT, hipMemcpyDeviceToHost));
        // the host's systematic resampling of the same weights (resampling.py:24-52)
        int64_t wrong = 0, j = 0;
        double run = std::exp(lw[0] - mx) / sw;
        for (int64_t i = 0; i < BT; ++i) {
            const double p = ((double)i + 0.37) / (double)BT;
            while (j < BT - 1 && run < p) { ++j; run += std::exp(lw[j] - mx) / sw; }
            if (std::llabs(idx[i] - j) > 1) ++wrong;  // (float cdf on the device: a position within rounding of a step may take the neighbour)
        }
        const double ess = sw * sw / sw2;
        std::printf("theta level: ESS %.3f (host %.3f)  mean (%.5f, %.5f) (host %.5f, %.5f)  1.1 L = [%.5f; %.5f %.5f] (host [%.5f; %.5f %.5f])  ancestors off by more than one: %lld\n",
                    stats[0], ess, mean[0], mean[1], m[0], m[1], chol[0], chol[2], chol[3], 1.1 * l00, 1.1 * l10, 1.1 * l11, (long long)wrong);
        if (!(std::fabs(stats[0] - ess) < 1e-3 * ess) || stats[1] != 1.0f || wrong != 0 || !(std::fabs(mean[0] - m[0]) < 1e-5) ||
            !(std::fabs(mean[1] - m[1]) < 1e-5) || !(std::fabs(chol[0] - 1.1 * l00) < 1e-5) || !(std::fabs(chol[2] - 1.1 * l10) < 1e-5) ||
            !(std::fabs(chol[3] - 1.1 * l11) < 1e-5) || chol[1] != 0.0f)
            ++failures;
        // one observation of SMC2.step (smc2.py:53-65): w += ll in place, the (ESS, all finite) pair on the device and - polled, no
        // copy command - in pf_host_alloc memory; three observations through one slot, the sequence number tells them apart
        void* slot = nullptr;
        PF_CALL(pf_host_alloc(64, &slot));
        volatile double* hv = (volatile double*)slot;
        volatile unsigned long long* hseq = (volatile unsigned long long*)((char*)slot + 16);
        std::vector<float> w(lw), inc(BT);
        float* d_inc;
        HIP_OK(hipMalloc((void**)&d_inc, sizeof(float) * BT));
        int64_t slot_bad = 0;
        for (unsigned long long seq = 1; seq <= 3; ++seq) {
            for (int64_t i = 0; i < BT; ++i) inc[i] = 0.25f * std::sin(0.37f * (float)(i + 11 * seq));
            HIP_OK(hipMemcpy(d_inc, inc.data(), sizeof(float) * BT, hipMemcpyHostToDevice));
            PF_CALL(pf_theta_step(d_lw, d_inc, BT, PF_F32, d_stats, slot, seq, nullptr, nullptr, nullptr));
            long long spins = 0;
            while (*hseq != seq && ++spins < (1ll << 31)) {}
            const double ess_slot = hv[0], fin_slot = hv[1];
            double mxs = -1e300, s1 = 0.0, s2 = 0.0;
            for (int64_t i = 0; i < BT; ++i) { w[i] = w[i] + inc[i]; mxs = std::max(mxs, (double)w[i]); }
            for (int64_t i = 0; i < BT; ++i) { const double e = std::exp((double)w[i] - mxs); s1 += e; s2 += e * e; }
            HIP_OK(hipDeviceSynchronize());
            float st2[2];
            std::vector<float> wdev(BT);
            HIP_OK(hipMemcpy(st2, d_stats, sizeof(st2), hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(wdev.data(), d_lw, sizeof(float) * BT, hipMemcpyDeviceToHost));
            int64_t wdiff = 0;
            for (int64_t i = 0; i < BT; ++i) wdiff += wdev[i] != w[i];
            if (*hseq != seq || ess_slot != (double)st2[0] || fin_slot != (double)st2[1] || wdiff != 0 ||
                !(std::fabs(ess_slot - s1 * s1 / s2) < 1e-3 * s1 * s1 / s2) || fin_slot != 1.0)
                ++slot_bad;
            if (seq == 3)
                std::printf("theta step: observation %llu polled from host memory after %lld spins: ESS %.3f (device copy %.3f, host %.3f), weights differing %lld\n",
                            seq, spins, ess_slot, (double)st2[0], s1 * s1 / s2, (long long)wdiff);
        }
        PF_CALL(pf_host_free(slot));
        if (slot_bad) ++failures;
    }
    std::printf("%s (%s)\n", failures ? "c-abi FAILED" : "c-abi ok", pf_version());
    return failures ? 1 : 0;
}
