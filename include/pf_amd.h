/*
 * pf_amd.h - C ABI of libpfamd.so: the MI355X (gfx950) particle-filter inner loop that replaces, behind pyfilter's
 * own Python API, the per-time-step propagate -> log-weight -> normalise -> resample -> gather cycle of
 * pyfilter.filters.particle.{SISR, APF} (reference tree: /root/reference/pyfilter, v0.29.0).
 *
 * The reference has no FFI of its own (it is pure Python on torch), so each entry point below
 * names the reference *function* it replaces.  INTEGRATION.md shows the ctypes stub a pyfilter maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (PF_OK) or a negative PF_E* code / positive hipError_t; nothing throws or aborts;
 *   - all pointers are BORROWED raw device pointers (torch.Tensor.data_ptr()); the caller owns and keeps them alive;
 *   - the library allocates nothing: scratch is a caller-provided workspace sized by pf_workspace_bytes();
 *   - every launch goes to the caller's hipStream_t (passed as void*) and is asynchronous; no host sync inside;
 *   - no global mutable state: re-entrant per (device, stream);
 *   - dtype: PF_F32 or PF_F64 - the arithmetic type of state and weights (reference default fp32, constants.py:6);
 *   - layout ("column" = one filter of the reference's batch dim):
 *         weights / log-weights / cdf / ancestors : (B, N)    contiguous, particle index fastest
 *         state                                   : (D, B, N) contiguous SoA (D = 1 for a scalar state)
 *     i.e. the reference's (N, [B], [D]) tensors (filters/particle/base.py:51-62) are *views* of these buffers
 *     (torch: buf.permute(2, 1, 0)).  Ancestors are int32 here, int64 in the reference.
 */
#ifndef PF_AMD_H
#define PF_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PF_ABI_VERSION 4 /* 4: pf_filter_args.status + pf_run_hints.cluster_patience (a column-cluster launch that cannot make
                          * progress reports it; the caller re-issues the piece on the per-step route), pf_theta_step takes the
                          * running total to accumulate into and that status word.
                          * 3: pf_filter_args.user_dt (an Euler-Maruyama user process hands over its DRIFT; the kernels form x + f dt).
                          * 2: pf_filter_args starts with its own size and ends with pf_run_hints; the library has no
                          * environment variables and no process-wide switches - every choice a caller can override is an
                          * argument */

#define PF_OK 0
#define PF_EINVAL (-1)     /* bad argument (shape, dtype, null pointer) */
#define PF_EWORKSPACE (-2) /* workspace too small */
#define PF_EUNSUPPORTED (-3)

#define PF_F32 0
#define PF_F64 1

/* hidden-process kinds: x' = loc(x) + scale(x) * eps, eps ~ N(0, inc_scale)    */
#define PF_HID_LINEAR 0      /* loc = alpha + beta x,                       scale = sigma     hp = (alpha, beta, sigma) */
#define PF_HID_SINE_EM 1     /* loc = x + sin(x - gamma) dt,                scale = sigma     hp = (gamma, sigma)        */
#define PF_HID_VERHULST_EM 2 /* loc = x + kappa (gamma - x) x dt,           scale = sigma x   hp = (kappa, gamma, sigma) */
#define PF_HID_LORENZ63_EM 3 /* Lorenz-63 drift, Euler-Maruyama (D = 3),    scale = sigma     hp = (s, r, b, sigma)      */
#define PF_HID_OU 4          /* exact OU step                                                 hp = (kappa, gamma, sigma) */
#define PF_HID_USER_AFFINE 5 /* loc, scale EVALUATED BY THE CALLER for the incoming particles (the reference's plug-in seam: a user
                              * lambda `mean_scale(x, *parameters)`, README.md:44-67, stochproc AffineProcess): the fused kernels
                              * read them from pf_filter_args.user_loc / user_scale at the PARENT's index - mean_scale acts per
                              * particle, so loc(x[anc]) = loc(x)[anc] - and do everything else (ancestors, gather, draws, Bootstrap
                              * or the optimal linear-Gaussian proposal, weights, moments, log-likelihood) as for a built-in
                              * kind.  One step per pf_filter_run call (the next step's planes need the new particles). */
/* observation kinds */
#define PF_OBS_LINEAR 0 /* y ~ N(b + A x, s)   (LinearStateSpaceModel; proposals/linear.py:48) */
#define PF_OBS_SV 1     /* y ~ N(mu, scale = x)                                                 */
/* proposals (pyfilter/filters/particle/proposals) */
#define PF_PROP_BOOTSTRAP 0
#define PF_PROP_LGO 1 /* LinearGaussianObservations */
/* filters */
#define PF_FILTER_SISR 0
#define PF_FILTER_APF 1
/* resamplers (pyfilter/resampling.py) */
#define PF_RESAMPLE_SYSTEMATIC 0
#define PF_RESAMPLE_MULTINOMIAL 1 /* in pf_filter_run: N iid draws realised as their order statistics (sorted uniforms from
                                   * normalised Exp(1) spacings, scanned by the same kernels) -> the ancestors come out
                                   * SORTED; torch.multinomial returns the same multiset in iid order */

/* Closed description of a state-space model (the "kernel_id" view of a stochproc StateSpaceModel). */
typedef struct pf_model {
    int32_t hid_kind;
    int32_t obs_kind;
    int32_t dim;     /* D >= 1 (a scalar state is D = 1) */
    int32_t obs_dim; /* O >= 1 (a scalar observation is O = 1); O = 1 required when D = 1 */
    double dt;
    double inc_scale; /* 1 for the discrete kinds, sqrt(dt) for Euler-Maruyama */
    /* (B, NP) parameter rows, dtype = the call's dtype, NP = 4*D + O*D + 2*O, one row per column:
     *   [ hp0[D] hp1[D] hp2[D] hp3[D] | A[O x D] row-major | b[O] | s[O] ]   (PF_OBS_SV: b[0] = mu) */
    const void* params;
} pf_model;

/* "pfamd <version> (gfx950) abi <PF_ABI_VERSION> src:<sha256 of the concatenated sources the binary was built from>" -
 * __graft_entry__.build() passes the digest; a log line with it shows which tree a shipped binary belongs to. */
const char* pf_version(void);
int pf_abi_version(void);
const char* pf_error_string(int code);

/* Scratch bytes any call below needs for an (N, B, D) problem - an upper bound over every tile geometry a call may be
 * given (pf_run_hints.tile_target), so a workspace of this size serves all of them. */
int pf_workspace_bytes(int64_t N, int64_t B, int64_t D, size_t* bytes);

/* ------------------------------------------------------------------------------------------------------------ *
 * L1 primitives
 * ------------------------------------------------------------------------------------------------------------ */

/* pyfilter.utils.normalize (utils.py:49-64) + get_ess (utils.py:8-20).
 * logw (B,N) is sanitised IN PLACE exactly as the reference does (NaN,+inf -> -inf; -inf -> lowest finite);
 * W (B,N) <- softmax over the particle axis (may be NULL); lse (B) <- log sum exp (may be NULL);
 * ess (B) <- 1 / sum W^2 (may be NULL). */
int pf_normalize(void* logw, void* W, void* lse, void* ess, int64_t N, int64_t B, int dtype, void* ws,
                 size_t ws_bytes, void* stream);

/* pyfilter.resampling.systematic (resampling.py:24-52), normalized=True path:
 * cdf = cumsum(W) with an fp64 carry rounded per element to dtype, cdf[N-1] = 1, idx = searchsorted_left(cdf, (i+u)/N).
 * W (B,N) normalised weights; u: (B) uniforms, one per column - or, with u_per_element != 0, (B,N) one per grid
 * position, the arrangement of the reference's own known-answer test (tests/test_resampling.py:39-47);
 * colmask (B) uint8 or NULL: only columns with colmask != 0 are
 * resampled, the others' idx are left untouched (SISR's masked resampling, sisr.py:29-31);
 * cdf (B,N) scratch/out; idx (B,N) int32 out.
 * cdf == NULL (where pf_systematic_cdf_free says yes, PF_EINVAL elsewhere): the ancestors only, in TWO launches instead of three
 * on columns of several tiles - the cdf values are rebuilt from the weights where they are used (same definition, same rounding per
 * element; the fp64 sum under it is associated per 256-particle chunk, which float weights do not see). */
int pf_systematic(const void* W, const void* u, int u_per_element, const uint8_t* colmask, void* cdf, int32_t* idx,
                  int64_t N, int64_t B, int dtype, void* ws, size_t ws_bytes, void* stream);

/* *yes <- 1 when pf_systematic / pf_systematic_logw (N, B, dtype, u_per_element) take cdf == NULL (columns of more than one tile, N % 4 == 0, one u per
 * column, float: N <= 2^22), else 0. */
int pf_systematic_cdf_free(int64_t N, int64_t B, int dtype, int u_per_element, int* yes);

/* systematic with normalized=False (resampling.py:8-21 then :24-52): logw is sanitised in place, the softmax is
 * never materialised (cdf is built from exp(logw - tile max) with an fp64 carry).  cdf == NULL: as pf_systematic. */
int pf_systematic_logw(void* logw, const void* u, int u_per_element, const uint8_t* colmask, void* cdf, int32_t* idx,
                       int64_t N, int64_t B, int dtype, void* ws, size_t ws_bytes, void* stream);

/* pyfilter.resampling.multinomial (resampling.py:55-65): N iid inverse-CDF draws per column.
 * v (B,N) uniforms or NULL (then Philox(seed, step)); output order is iid (unsorted) like torch.multinomial. */
int pf_multinomial(const void* W, const void* v, uint64_t seed, uint32_t step, const uint8_t* colmask, void* cdf,
                   int32_t* idx, int64_t N, int64_t B, int dtype, void* ws, size_t ws_bytes, void* stream);

/* pyfilter.filters.utils.batched_gather (filters/utils.py:4-21) along the particle axis, for all D components;
 * columns with colmask == 0 are copied through unchanged (SISR's masked_scatter, sisr.py:37-44). */
int pf_gather(const void* x, const int32_t* idx, const uint8_t* colmask, void* out, int64_t N, int64_t B,
              int64_t D, int dtype, void* stream);

/* Whole-filter moves along the batch dim - the step either side of the hot path inside SMC^2 / PMMH:
 *   pf_columns_gather   : ParticleFilterCorrection.resample (particle/state.py:150-158), FilterResult.resample
 *                         (result.py:97-117):   dst[p][b][:] = src[p][idx[b]][:]          (out of place, idx (B) int64)
 *   pf_columns_exchange : ParticleFilterCorrection.exchange (particle/state.py:160-168), FilterResult.exchange
 *                         (result.py:76-95):    dst[p][b][:] = src[p][b][:] where mask[b] (in place in dst)
 * src / dst are `planes` stacked (B, N) arrays of `elem_bytes`-sized elements (4 or 8: float, double, int32, int64):
 * weights / ancestors have planes = 1, a (D, B, N) state has planes = D.  Pure byte moves, dtype-agnostic. */
int pf_columns_gather(const void* src, const int64_t* idx, void* dst, int64_t N, int64_t B, int64_t planes,
                      int elem_bytes, void* stream);
int pf_columns_exchange(void* dst, const void* src, const uint8_t* mask, int64_t N, int64_t B, int64_t planes,
                        int elem_bytes, void* stream);

/* pyfilter.filters.particle.utils.log_likelihood (particle/utils.py:7-22): max v + log sum W exp(v - max);
 * W == NULL means 1/N.  v (B,N) is NOT sanitised (a NaN poisons the column, as in the reference). out (B). */
int pf_loglik(const void* v, const void* W, void* out, int64_t N, int64_t B, int dtype, void* ws, size_t ws_bytes,
              void* stream);

/* get_filter_mean_and_variance (particle/utils.py:26-65), covariance=False: mean (B,D), var (B,D). */
int pf_moments(const void* x, const void* W, void* mean, void* var, int64_t N, int64_t B, int64_t D, int dtype,
               void* ws, size_t ws_bytes, void* stream);

/* Proposal.pre_weight (proposals/base.py:69-85 | linear.py:57-86) for a built-in model: out (B,N).
 * y: (By, O) with By in {1, B}. */
int pf_pre_weight(const pf_model* model, int proposal, const void* x, const void* y, int64_t y_rows, void* out,
                  int64_t N, int64_t B, int dtype, void* stream);

/* Proposal.sample_and_weight (bootstrap.py:10-14 | linear.py:38-55) for a built-in model:
 * x_out (D,B,N) new particles, w_out (B,N) importance log-weights (NOT sanitised).
 * z (D,B,N) standard normals, or NULL -> Philox(seed, step). weigh = 0: propagate only (w_out may be NULL)
 * (ParticleFilterPrediction.create_state_from_prediction, particle/state.py:38-42). */
int pf_sample_and_weight(const pf_model* model, int proposal, int weigh, const void* x, const void* y,
                         int64_t y_rows, const void* z, uint64_t seed, uint32_t step, void* x_out, void* w_out,
                         int64_t N, int64_t B, int dtype, void* stream);

/* hidden.initial_sample: x (D,B,N) <- m0[d] + s0[d] * z, z from `z` or Philox(seed). m0, s0: (D) host doubles. */
int pf_initial_sample(const double* m0, const double* s0, const void* z, uint64_t seed, void* x, int64_t N,
                      int64_t B, int64_t D, int dtype, void* stream);

/* ... with one initial mean / scale per filter (theta-particles on the batch dimension: the stationary law of an
 * Ornstein-Uhlenbeck process depends on theta): element (b, d) of m0 / s0 at [b * stride_b + d * stride_d], strides in
 * elements, 0 = broadcast; device arrays of `dtype`.  x <- m0 + s0 * z. */
int pf_initial_sample_cols(const void* m0, int64_t m0_stride_b, int64_t m0_stride_d, const void* s0, int64_t s0_stride_b,
                           int64_t s0_stride_d, const void* z, uint64_t seed, void* x, int64_t N, int64_t B, int64_t D,
                           int dtype, void* stream);

/* "Observation k carries information": out[k] = 1 unless every element of y[k] ((steps, row_elems), row_elems =
 * y_rows * O) is NaN - the reference's host test `y.isnan().all()` that turns a move into propagate-only
 * (filters/base.py:212), for all steps of a series in one launch.  `out`: device bytes (steps). */
int pf_observed_flags(const void* y, int64_t steps, int64_t row_elems, int dtype, uint8_t* out, void* stream);

/* theta-level bookkeeping of SMC^2 (inference/sequential/state.py:35-44 `get_ess(normalize(w))`, smc2.py:59-62
 * `(~isfinite(w)).any()`) in one launch, for `rows` weight vectors (rows, B) at once (the observations of a block):
 * out[r][0] = effective sample size of row r's B log-weights under pyfilter.utils.normalize (utils.py:49-64: NaN /
 * +inf count as -inf, all -inf -> uniform), out[r][1] = 1 if every weight of the row is finite else 0.
 * `out`: (rows, 2) device values of `dtype`. */
int pf_theta_ess(const void* logw, int64_t rows, int64_t B, int dtype, void* out, void* stream);

/* theta-level arithmetic of one SMC^2 / PMMH move on B theta-particles of P <= PF_THETA_MAXP scalar parameters, in
 * unconstrained space (pyfilter/inference/utils.py:42-76 construct_mvn; batch/mcmc/utils.py:48-70 run_pmmh; prior.py:47-123
 * the priors' bijections) - three small kernels instead of ~130 torch launches per move.  All arithmetic in double; arrays
 * are of `dtype`.  Prior families: parameters (a, b) as torch.distributions names them. */
#define PF_THETA_MAXP 8
#define PF_PRIOR_NORMAL 0      /* Normal(loc a, scale b)               real support:        x = u                       */
#define PF_PRIOR_LOGNORMAL 1   /* LogNormal(loc a, scale b)            positive:            x = exp(u)                  */
#define PF_PRIOR_EXPONENTIAL 2 /* Exponential(rate a)                                                                   */
#define PF_PRIOR_GAMMA 3       /* Gamma(concentration a, rate b)                                                        */
#define PF_PRIOR_HALFNORMAL 4  /* HalfNormal(scale a)                                                                   */
#define PF_PRIOR_BETA 5        /* Beta(concentration1 a, concentration0 b)   unit interval: x = sigmoid(u)              */
#define PF_PRIOR_UNIFORM 6     /* Uniform(low a, high b)               interval:            x = a + (b - a) sigmoid(u)  */
typedef struct pf_theta_priors {
    int32_t P;
    int32_t kind[PF_THETA_MAXP];
    double a[PF_THETA_MAXP], b[PF_THETA_MAXP];
} pf_theta_priors;

/* calc_mean_chol / construct_mvn (inference/utils.py:42-76): normalised weights of `logw` (B) as pyfilter.utils.normalize
 * defines them (NULL: equal weights), mean (P) <- weighted mean of values (B, P), chol (P, P) <- scale * lower Cholesky factor
 * of the weighted covariance - its diagonal's square root when the covariance is not positive definite. */
int pf_theta_fit(const void* values, const void* logw, int64_t B, int32_t P, double scale, int dtype, void* mean, void* chol,
                 void* stream);

/* theta* ~ N(mean, chol chol^T) from the standard normals eps (B, P) (mcmc/utils.py:48): u_out (B, P) <- mean + chol eps;
 * x_out[p] (B) <- the constrained value of parameter p (its prior's bijection; parameter.py:79-107); prior_out (B) <- the
 * summed log priors of u (TransformedDistribution(prior, bijection^-1).log_prob, prior.py:98-123). */
int pf_theta_propose(const pf_theta_priors* priors, const void* mean, const void* chol, const void* eps, int64_t B, int dtype,
                     void* u_out, void* const* x_out, void* prior_out, void* stream);

/* The acceptance step (mcmc/utils.py:57-70): log_acc (B) <- [log q_r(u_cur) - log q_f(u_star)] + [prior_star - prior_cur] +
 * [ll_star - ll_cur] with q_f = N(mean_f, chol_f chol_f^T) the kernel theta* was drawn from and q_r the one fitted to
 * theta*; accepted (B, uint8) <- log(unif) < log_acc (NaN: rejected); rate (1) <- the share accepted. */
int pf_theta_accept(const void* u_cur, const void* u_star, const void* mean_f, const void* chol_f, const void* mean_r,
                    const void* chol_r, const void* prior_cur, const void* prior_star, const void* ll_cur, const void* ll_star,
                    const void* unif, int64_t B, int32_t P, int dtype, void* log_acc, uint8_t* accepted, void* rate, void* stream);

/* The theta-weights along a block of n observations (sequential/state.py:35-44, n times): w_path (n, B) <- w0 (B) + the
 * running sum of the log-likelihood increments ll (n, B); stats (n, 2) <- (ESS, 1 if every weight is finite) per row, as
 * pf_theta_ess reports them.  n = 1: w_path may be w0 itself (the weights updated in place - one observation of the
 * reference's step(), smc2.py:53-65).
 * host_rows != NULL: n x 32 bytes of pf_host_alloc memory - row q's two values as doubles, then the 64-bit `seq`, then the status
 * word, written by row q's own workgroup as pf_theta_step writes its one slot: a host that decides per observation (smc2.py:59-62)
 * polls the rows in order and has row q when it is done - no device -> host copy command behind the block, no event.
 * status (or NULL): pf_filter_args.status of the run that produced ll - non-zero: that run gave up, the rows report NaN / 0 and
 * the word (the caller re-issues the block, see PF_ROUTE_CLUSTER), w_path is not written. */
int pf_theta_path(const void* w0, const void* ll, int64_t n, int64_t B, int dtype, void* w_path, void* stats, void* host_rows,
                  uint64_t seq, const int32_t* status, void* stream);

/* ONE observation of the reference's SMC2.step() (smc2.py:53-65; sequential/state.py:35-44): w (B) += ll (B) in place and stats
 * (2) <- (ESS, 1 if every weight is finite) as pf_theta_ess reports them - pf_theta_path with n = 1 - and, host_slot != NULL,
 * the same two values as doubles, the 64-bit `seq` and (fourth word) the run status into 32 bytes of pf_host_alloc memory:
 * the reference tests the ESS on the host after every observation (smc2.py:59-62), and a host thread that polls the third
 * word for `seq` has the values without a device -> host copy command and its synchronisation (~12 us per observation).
 * acc (B) or NULL: the filters' running log-likelihood, acc += ll as well (filters/result.py:130 - one elementwise launch of
 * the caller's less).  status or NULL: the pf_filter_args.status word of the move that produced ll - when it is non-zero
 * NOTHING is updated (w and acc keep their values, stats <- (NaN, 0)) and the slot's fourth word carries it: the caller
 * re-issues the move on the per-step route and calls again. */
int pf_theta_step(void* w, const void* ll, int64_t B, int dtype, void* stats, void* host_slot, uint64_t seq, void* acc,
                  const int32_t* status, void* stream);

/* Host memory the device writes and the host polls (hipHostMalloc, coherent + mapped, zero-filled); pf_host_free releases it. */
int pf_host_alloc(size_t bytes, void** out);
int pf_host_free(void* p);

/* Systematic resampling of B theta-particles from their log-weights (kernels/mh.py:52-56: pyfilter.utils.normalize, then
 * resampling.py:24-52 with the uniform u in [0, 1]): ancestors (B, int64).  cdf_scratch: B values of `dtype`. */
int pf_theta_resample(const void* logw, int64_t B, double u, int dtype, int64_t* ancestors, void* cdf_scratch, void* stream);

/* ------------------------------------------------------------------------------------------------------------ *
 * fused filter loop: BaseFilter.batch_filter / filter (filters/base.py:140-221) for SISR (sisr.py:14-56) and
 * APF (apf.py:16-46) with Bootstrap / LinearGaussianObservations on a built-in model; one kernel per step.
 * ------------------------------------------------------------------------------------------------------------ */
/* Choices pf_filter_run normally takes itself, overridable PER CALL (tests pin both kernel routes against the reference,
 * tools measure them).  All zero = the library's own choices.  Nothing in the library reads the environment. */
#define PF_ROUTE_AUTO 0           /* filters of <= column_max_n particles: the column-persistent kernel, else one kernel per step */
#define PF_ROUTE_PER_STEP 1       /* always one k_fused_step launch per time step */
#define PF_ROUTE_COLUMN_GENERIC 2 /* as AUTO, but the column kernel's run-time instantiation (no model kind folded in) */
#define PF_ROUTE_CLUSTER 3        /* as AUTO, and self-contained runs of filters of 2 049 .. 16 384 particles (N % 4 == 0,
                                   * systematic resampling, built-in model) take the column-CLUSTER kernel: ceil(N / 1024)
                                   * workgroups per filter hold it in registers for the whole run and exchange one record per
                                   * wave and step (pf_cluster.hpp).  Those workgroups wait for each other.  Their ids are
                                   * grouped so that the resident workgroups of a launch are whole filters (plus one partly
                                   * dispatched filter per XCD) whatever else runs on the device: any number of such runs may be
                                   * in flight on different streams, threads or processes - they share the slots, none starves
                                   * (tests/test_cluster_route_gpu.py drives two streams from two threads).  HIP promises no
                                   * dispatch order, so every wait is bounded all the same: a launch that cannot make progress
                                   * (a foreign kernel holding the device for seconds) gives up, returns NaN log-likelihoods and
                                   * sets bit 0 of pf_filter_args.status.  Opt-in because of that contract: a caller that takes
                                   * this route passes `status`, looks at it where it next waits for the device, and on a
                                   * non-zero word re-issues the piece from the same incoming state with PF_ROUTE_PER_STEP - the
                                   * draws are keyed by (seed, step, particle), so the numbers are the one-piece run's.  Batches
                                   * of more than two launches' worth of member workgroups (B ceil(N / 1024) > 2 048) stay on
                                   * the per-step route, which is the faster one there; so do runs the kernel cannot be launched
                                   * for (no resident slot for one filter's workgroups) */
#define PF_ROUTE_CLUSTER_ALWAYS 4 /* as CLUSTER for a batch of any size (consecutive launches; tests and measurements) */
#define PF_ROUTE_CLUSTER_SPREAD 5 /* as CLUSTER_ALWAYS with the members of a filter on DIFFERENT XCDs (consecutive workgroup ids) and
                                   * the exchange in its placement-independent form - agent-scope write-through stores and
                                   * L1-bypassing loads, never the same-XCD fast path: what a run falls back to whenever its
                                   * members do not share an XCD, pinned by the tests on every box */
typedef struct pf_run_hints {
    int32_t route;           /* PF_ROUTE_* */
    int32_t column_max_n;    /* largest filter the column-persistent kernel takes; 0 = the default (2048) */
    int32_t tile_target;     /* workgroups per launch the tile size aims at; 0 = the default (1024 = 4 per CU) */
    int32_t ancestor_search; /* != 0: systematic ancestors by searching the staged window at any size (default: only float
                              * grids beyond 2^22 positions, where the inverted grid's closed form is not exact) */
    int32_t resume;          /* != 0 (t0 > 0): the incoming state of step t0 is the one the PREVIOUS pf_filter_run call on
                              * this argument block wrote - its per-tile partials and local scans are still in the workspace
                              * (a SISR step leaves them for its successor; an APF step does when that call ran with
                              * prepare_next), so the pass that re-reduces the incoming state is skipped.  A run issued
                              * move by move (user-defined models: one call per move) pays one launch per move less.
                              * Ignored on the column route (which has no such pass) */
    int32_t prepare_next;    /* != 0 (APF, finalize == 0): the run's LAST step also prepares the first-stage weights of the
                              * step after it, as every earlier step of a run does for its successor - the caller promises
                              * that y holds row t0 + n_steps, that it carries information, and that the next call on this
                              * block starts there with `resume`.  Built-in models; PF_HID_USER_AFFINE only with PF_PROP_LGO
                              * and user_scale_per_column: the optimal proposal's first-stage weight (linear.py:57-86) needs
                              * the particle and its transition scale - not the caller's one-step mean, which does not exist
                              * yet for the new particles */
    int32_t cluster_generation; /* 0: every column-cluster launch first clears its records (one small launch more).  != 0 (needs
                               * pf_filter_args.status, runs of <= 2 048 steps, not under stream capture): the caller NUMBERS its
                               * cluster launches on this workspace - any value in [1, 2^20) that differs from those of the
                               * previous 2^20 - 1 launches on it, e.g. a counter; the records then carry it in their tags, stale
                               * ones never match and nothing is cleared: an online move of 2 049 .. 16 384 particles is ONE launch.
                               * The workspace must have been zero-filled once before its first such launch */
    int32_t cluster_patience; /* polls a workgroup of the column-cluster kernel spends on one wait for its siblings before it
                               * gives up (see PF_ROUTE_CLUSTER); 0 = the default, 2^21 (seconds).  -1 (tests): every workgroup
                               * gives up at its first wait whatever it finds - the give-up path, deterministically */
} pf_run_hints;

typedef struct pf_filter_args {
    uint64_t struct_size; /* sizeof(pf_filter_args) of the header the caller was built against: a caller of another ABI
                           * version is refused (PF_EINVAL) instead of being read past its end */
    pf_model model;
    int32_t filter;    /* PF_FILTER_* */
    int32_t proposal;  /* PF_PROP_* */
    int32_t resampler; /* PF_RESAMPLE_* */
    int32_t dtype;
    int64_t N, B;
    double ess_threshold; /* relative: resample when ess < ess_threshold * N (particle/base.py:42) */
    uint64_t seed;
    /* state, double buffered: slot (step & 1) is read, the other written */
    void* x[2];     /* (D,B,N) */
    void* logw[2];  /* (B,N)   */
    int32_t* anc;   /* (B,N) ancestors of the latest step (SISR keeps them when no resampling happened) */
    void* cdf;      /* (B,N) scratch: tile-local scans of the resampling weights, even steps */
    void* pos;      /* (B,N) scratch: the same for odd steps (double buffered like the state; required) */
    /* observations */
    const void* y;            /* (T, y_rows, O) */
    int64_t y_rows;           /* 1 or B */
    const uint8_t* observed;  /* OPTIONAL HOST array (T) - the one exception to "every pointer is a device pointer", kept for
                               * callers that already know the flags on the host (a hipGraph capture bakes them into the
                               * launches).  The DEFAULT is `observed_dev` below (or neither: the library derives the flags
                               * from y on the device) - no host-resident argument, no synchronisation.  Meaning:
                               * 0 = all-NaN observation or unobserved sub-step: propagate only,
                               * ll = 0.  The only host-resident argument: the launch loop reads it to set each
                               * kernel's flags, so no kernel needs a dependent flag load before its first data load.
                               * NULL together with `observed_dev` (runs of <= 128 steps): the library derives the
                               * flags from y itself, on the device (pf_observed_flags into the workspace). */
    /* optional tapes (parity mode); NULL -> Philox */
    const void* z_tape; /* (T, D, B, N) */
    const void* u_tape; /* (T, B) */
    /* results */
    void* means;     /* (T+1, B, D) filter_means incl. the initial state (filters/result.py:119-131) */
    void* vars;      /* (T+1, B, D) */
    void* ll_steps;  /* (T, B) per-step log-likelihood increments */
    void* ll_total;  /* (B) running sum, updated in place */
    int32_t* step_counter; /* optional device uint64 "epoch" added to `seed` by every kernel (NULL = 0): lets a captured
                            * graph draw fresh Philox numbers on every replay */
    void* ws;
    size_t ws_bytes;
    const uint8_t* observed_dev; /* the DEFAULT way to pass the flags: DEVICE array (T) with the meaning of `observed` (e.g.
                                  * written by pf_observed_flags); when non-NULL it is used
                                  * instead (every kernel reads its step's flag with one scalar load), so the caller
                                  * needs no host-side knowledge of NaN observations - `filter()` runs without a sync */
    int64_t ring; /* state history (FilterResult's recorded states, particle/base.py:105-157 smoothing): 0 or 2 = none - x[0] /
                   * x[1], logw[0] / logw[1] are two buffers used alternately and `anc` holds the latest ancestors.
                   * ring >= 3: x[0] is the base of a (ring, D, B, N) array, logw[0] of a (ring, B, N) array, anc of a
                   * (ring, B, N) array (x[1] / logw[1] ignored); the state after step s - and the ancestors that lead to
                   * it - live in slot (s + 1) % ring, the incoming state of step t0 in slot t0 % ring.  ring = n_steps + 1
                   * keeps every state of a run.  A SISR step that does not resample copies its ancestors forward, as the
                   * reference carries prev_inds (sisr.py:25-26). */
    const void* user_loc;   /* PF_HID_USER_AFFINE only: (D, B, N) one-step mean of every particle of the INCOMING state ... */
    const void* user_scale; /* ... and its transition scale, both in the state's layout and dtype (else NULL) */
    int64_t user_scale_per_column; /* 0: user_scale is a (D, B, N) plane like user_loc; 1: a (D, B) array - ONE transition scale
                                    * per filter and state component (a diffusion that does not depend on the state: the common
                                    * case, and no plane to fill per move) */
    double user_dt;         /* PF_HID_USER_AFFINE: 0 = user_loc is the one-step mean itself; != 0 = user_loc is the DRIFT f(x) of an
                             * Euler-Maruyama discretisation (README.md:44-62: AffineEulerMaruyama) and the kernels form the mean
                             * x + f(x) dt at the parent themselves - the addition is one elementwise launch of the caller's less
                             * per move (~5 us at 2^20 particles) */
    int32_t* status;        /* optional DEVICE word owned by the caller: the run ORs bits into it and never clears it (the caller
                             * zeroes it when it starts watching).  bit 0: a column-cluster launch gave up waiting for a
                             * sibling workgroup; bit 1: an ancestor fell outside the chunks a cluster member staged.  Non-zero =
                             * the log-likelihoods of that run are NaN and its final state is not to be used: re-issue the piece
                             * with PF_ROUTE_PER_STEP (see PF_ROUTE_CLUSTER).  The other routes never touch it */
    pf_run_hints hints;     /* all zero = the library's own choices */
} pf_filter_args;

/* Runs steps [t0, t0 + n_steps) - indices into y / observed / the tapes / the result rows; ONE kernel launch per step
 * (prologue: the column's tile-prefix table + window start from the previous launch's per-tile partials; body: ancestors
 * + gather + propagate + weight + the next state's partials and tile-local scans; one workgroup per column keeps the
 * books: moments row, log-likelihood increment) plus one reduce launch for the incoming state.
 * finalize != 0 additionally flushes the moments / log-likelihood of the last state (row t0 + n_steps).
 * A run may be issued in pieces on ONE argument block - e.g. move by move, (t0 = s, n_steps = 1, finalize = 1) for
 * s = 0, 1, ...: the workspace carries the per-filter bookkeeping (log-likelihood bases, what has been flushed) from call
 * to call, ll_total keeps accumulating, and the result rows land where the one-piece run writes them; t0 = 0 starts a
 * fresh filter.  (PF_HID_USER_AFFINE runs are issued this way: the planes of move s + 1 need the particles move s wrote.) */
int pf_filter_run(const pf_filter_args* args, int64_t t0, int64_t n_steps, int finalize, void* stream);

/* ONE observation of an online SMC^2 loop (smc2.py:53-65) in one call: pf_filter_run(args, t0, n_steps, finalize, stream) followed -
 * when it returned PF_OK - by pf_theta_step(w, ll, args->B, args->dtype, stats, host_slot, seq, acc, args->status, stream): the
 * filters' move, then theta-weights += its log-likelihood increments `ll` (a row of args->ll_steps), their (ESS, all finite) pair
 * into `stats` and the polled host slot, the running total `acc`.  `ll` must be the row of args->ll_steps the run's LAST step
 * writes.  The same results as the two calls; one crossing of the host language's FFI per observation instead of two - and on the
 * column-cluster route ONE launch: the last filter to finish does the theta update inside the run's kernel. */
int pf_filter_observe(const pf_filter_args* args, int64_t t0, int64_t n_steps, int finalize, void* w, const void* ll, void* stats,
                      void* host_slot, uint64_t seq, void* acc, void* stream);

/* hipGraph variant: captures the launch sequence pf_filter_run would issue (every pointer, the step flags and the
 * observation offsets are baked into the kernel nodes) and returns an opaque handle OWNED BY THE CALLER; replaying it
 * costs one host call instead of T launches (an eager launch costs the host ~5 us, a graph kernel node ~1.5 us).
 * The buffers named in `args` must stay alive and in place for the handle's lifetime; `y`'s contents may change. */
int pf_filter_graph_create(const pf_filter_args* args, int64_t t0, int64_t n_steps, int finalize, void* stream,
                           void** handle);
int pf_filter_graph_launch(void* handle, void* stream);
int pf_filter_graph_destroy(void* handle);

/* Measurement variant of pf_filter_run (synchronises the stream).  kernel_ms[0] = kernel_ms[2] = in-sequence duration of
 * one step = of its one kernel (HIP events on `stream` around the whole step loop / n_steps); kernel_ms[1] = 0 (the
 * planning kernel of earlier versions).  Same results as pf_filter_run; not for throughput numbers. */
int pf_filter_run_timed(const pf_filter_args* args, int64_t t0, int64_t n_steps, int finalize, void* stream,
                        float* kernel_ms);

/* ------------------------------------------------------------------------------------------------------------ *
 * smoothing over recorded states: ParticleFilter.smooth (particle/base.py:105-157).  S states, time-major histories:
 * x_hist (S, D, B, N), logw_hist (S, B, N), anc_hist (S, B, N) with anc_hist[t] = ParticleFilterCorrection.previous_indices
 * of state t (row 0 is never read); out (S, D, B, N).
 * ------------------------------------------------------------------------------------------------------------ */

/* method "fl" (_do_sample_fl, :136-152): out[S-1] = x[S-1]; walking back, out[t][i] = x[t][a_t(i)] with
 * a_{S-1}(i) = i, a_t(i) = anc[t+1][a_{t+1}(i)]. */
int pf_smooth_fixed_lag(const void* x_hist, const int32_t* anc_hist, void* out, int64_t S, int64_t N, int64_t B,
                        int64_t D, int dtype, void* stream);

/* method "ffbs" (_do_sample_ffbs, :105-134) for a built-in model: out[S-1] = x_last (D, B, N) - the last state resampled
 * by the filter's resampler, done by the caller -, then for t = S-2 .. 0 trajectory j draws i from
 * Categorical(logits_i = logw[t][i] + log p(out[t+1][j] | x[t][i])) and out[t][j] = x[t][i].
 * u (S-1, B, N) uniforms in [0, 1) for the inverse-CDF draws (row t serves step t), or NULL -> Philox(seed). */
int pf_smooth_ffbs(const pf_model* model, const void* x_hist, const void* logw_hist, const void* x_last, const void* u,
                   uint64_t seed, void* out, int64_t S, int64_t N, int64_t B, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------ *
 * test support (no reference counterpart): lets a parity test feed the oracle the very draws a *production* run used
 * ------------------------------------------------------------------------------------------------------------ */

/* out (n_steps, D, B, N) <- the standard normals the fused step kernel draws from Philox(seed) at local steps
 * step0 .. step0 + n_steps - 1 (seed = pf_filter_args.seed + *step_counter of the run to reproduce). */
int pf_debug_draw_normals(uint64_t seed, uint32_t step0, int64_t n_steps, void* out, int64_t N, int64_t B, int64_t D,
                          int dtype, void* stream);

/* The step-kernel instantiations the calling thread's most recent pf_filter_run launches selected, oldest first:
 * out[i] = { step, sizeof(T), D, VEC, MODE, PROP, FAST, SPEC, MK, MULTI } (10 int32 per record, the last 2048 are kept).
 * Returns the number of records written (>= 0) or a negative PF_E* code. */
int pf_debug_launch_trace(int32_t* out, int max_records);

#ifdef __cplusplus
}
#endif
#endif /* PF_AMD_H */
