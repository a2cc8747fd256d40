"""
TEST INFRASTRUCTURE — the oracle.  A torch-CPU restatement of pyfilter's SISR/APF propagate -> log-weight ->
resample cycle, issuing the same aten-op sequence as the reference functions it cites, so that on CPU its fp32
results carry the reference's own rounding and its fp64 results are the "exact" side of the 1e-5 parity target.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product (``pyfilter_amd``) never does and never falls back to it.

Pinned by: ``tests/golden/*.npz`` - outputs of the *unmodified* reference (``/root/reference/pyfilter`` imported
behind ``oracle/ref_shim``) dumped by ``oracle/make_golden.py``; see ``tests/test_oracle_golden.py``.
The model arithmetic itself (``oracle/models.py``) is "parity unpinned" vs the absent third-party ``stochproc``.

RNG: the reference draws from torch's CPU mt19937 generator; here every draw is an explicit input ("tape"):
``z`` standard normals of the state's shape and ``u`` uniforms of shape ``(B,)`` per step - the draw order in the
reference is ``u`` then ``z``.
"""
import math
from typing import Optional

import torch

from . import models as M


# --------------------------------------------------------------------------------------------------------------
# L1 primitives
# --------------------------------------------------------------------------------------------------------------
def normalize(weights: torch.Tensor) -> torch.Tensor:
    """pyfilter/utils.py:49-64 - NaN/+inf -> -inf **in place**, max-shifted softmax over dim 0, all -inf column
    -> 1/N."""
    weights = weights.nan_to_num_(-math.inf, posinf=-math.inf)
    normalized = (weights - weights.max(dim=0)[0]).softmax(dim=0)
    ax_sum = normalized.sum(dim=0)
    normalized.masked_fill_(ax_sum == 0.0, 1.0 / normalized.shape[0])
    return normalized


def get_ess(weights: torch.Tensor, normalized: bool = False) -> torch.Tensor:
    """pyfilter/utils.py:8-20."""
    if not normalized:
        weights = normalize(weights)
    return weights.pow(2.0).sum(dim=0).reciprocal()


def systematic(w: torch.Tensor, normalized: bool = False, u: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pyfilter/resampling.py:8-52.  ``w`` is ``(N,)`` or ``(N,B)``; ``u`` is ``(B,1)`` (or ``(B,N)`` in the
    reference's own known-answer test).  NB the reference drops ``u`` for 1-D input (``f(w, *kwargs)``,
    resampling.py:14); here ``u`` is honoured for 1-D too (it must then have shape ``(1,1)``)."""
    if not normalized:
        w = normalize(w)

    is_1d = w.dim() == 1
    wt = w.unsqueeze(0) if is_1d else w.moveaxis(0, 1)

    if u is None:
        u = torch.empty((wt.shape[0], 1), dtype=wt.dtype).uniform_()

    n = wt.shape[1]
    index_range = torch.arange(n, dtype=u.dtype).unsqueeze(0)
    probs = (index_range + u) / n
    cumsum = wt.cumsum(-1)
    cumsum[..., -1] = 1.0
    res = torch.searchsorted(cumsum, probs)

    return res.squeeze(0) if is_1d else res.moveaxis(0, 1)


def multinomial(w: torch.Tensor, normalized: bool = False) -> torch.Tensor:
    """pyfilter/resampling.py:55-65 (iid inverse-CDF draws; statistical parity only)."""
    if not normalized:
        w = normalize(w)
    if w.dim() == 1:
        return torch.multinomial(w, w.shape[-1], replacement=True)
    return torch.multinomial(w.moveaxis(0, 1), w.shape[0], replacement=True).moveaxis(0, 1)


def batched_gather(x: torch.Tensor, indices: torch.Tensor, dim: int = 0) -> torch.Tensor:
    """pyfilter/filters/utils.py:4-21."""
    if x.dim() > indices.dim():
        indices = indices.unsqueeze(-1).expand_as(x)
    return x.gather(dim, indices)


def log_likelihood(importance_weights: torch.Tensor, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pyfilter/filters/particle/utils.py:7-22."""
    max_w, _ = importance_weights.max(dim=0)
    temp = (importance_weights - max_w).exp()
    if weights is None:
        weights = 1.0 / importance_weights.shape[0]
    return max_w + (weights * temp).sum(dim=0).log()


def get_filter_mean_and_variance(values: torch.Tensor, weights: torch.Tensor, has_event_dim: bool):
    """pyfilter/filters/particle/utils.py:26-65 (``covariance=False, keep_dim=True`` branch): always returns a
    trailing dim."""
    if not has_event_dim:
        values = values.unsqueeze(-1)
    weights = weights.unsqueeze(-1)
    mean = (weights * values).sum(dim=0)
    centered = values - mean
    var = (weights * centered.pow(2.0)).sum(dim=0)
    return mean, var


# --------------------------------------------------------------------------------------------------------------
# proposals
# --------------------------------------------------------------------------------------------------------------
def bootstrap_sample_and_weight(spec, y, x, z):
    """proposals/bootstrap.py:10-14."""
    new_x = M.propagate(spec, x, z)
    return new_x, M.obs_log_prob(spec, y, new_x)


def default_pre_weight(spec, y, x):
    """proposals/base.py:69-85 with pre_weight_funcs.py:9-11: ``log p(y | x = mean(x_{t-1}))``."""
    loc, _ = M.mean_scale(spec, x)
    return M.obs_log_prob(spec, y, loc)


def _diag(v, n):
    """utils.py:23-46 ``construct_diag_from_flat``."""
    eye = torch.eye(max(n, 1), dtype=v.dtype)
    return eye * (v.view(*v.shape, 1, 1) if n == 0 else v.unsqueeze(-1))


def lgo_sample_and_weight(spec, y, x, z):
    """proposals/linear.py:38-55 + proposals/utils.py:219-267 (``find_optimal_density``) + proposals/base.py:45-50."""
    assert spec.obs == M.OBS_LINEAR
    mean, scale = M.mean_scale(spec, x)
    h_var_inv = scale.pow(-2.0)
    a, b, s = [M._t(q, x) for q in spec.obs_params]
    o_var_inv = s.pow(-2.0)
    yy = y - b

    hidden_is_1d = spec.dim == 0
    obs_is_1d = spec.obs_dim == 0

    c = a.unsqueeze(-1) if hidden_is_1d else a
    c_unsq = c if not obs_is_1d else c.unsqueeze(-2)
    c_t = c_unsq.transpose(-2, -1)
    o_inv_cov = _diag(o_var_inv, spec.obs_dim)
    t_2 = c_t.matmul(o_inv_cov).matmul(c_unsq)
    cov = (_diag(h_var_inv, spec.dim) + t_2).inverse()
    t_1 = h_var_inv * mean
    if hidden_is_1d:
        t_1 = t_1.unsqueeze(-1)
    t_2 = o_inv_cov.squeeze(-1) * yy.unsqueeze(-1) if obs_is_1d else o_inv_cov.matmul(yy.unsqueeze(-1)).squeeze(-1)
    t_3 = c_t.matmul(t_2.unsqueeze(-1))
    kmean = cov.matmul(t_1.unsqueeze(-1) + t_3).squeeze(-1)

    if hidden_is_1d:
        kloc, kscale = kmean.squeeze(-1), cov[..., 0, 0].sqrt()
        x_new = kloc + kscale * z  # torch.normal(mu, sigma) == z*sigma + mu
        k_lp = M.normal_log_prob(x_new, kloc, kscale)
    else:
        L = torch.linalg.cholesky_ex(cov)[0]
        x_new = kmean + (L @ z.unsqueeze(-1)).squeeze(-1)
        k_lp = torch.distributions.MultivariateNormal(kmean, scale_tril=L, validate_args=False).log_prob(x_new)

    y_lp = M.obs_log_prob(spec, y, x_new)
    x_lp = M.transition_log_prob(spec, x_new, mean, scale)
    return x_new, y_lp + x_lp - k_lp


def lgo_pre_weight(spec, y, x):
    """proposals/linear.py:57-86 - NB evaluated at ``x.value`` (the *un-propagated* state), as the reference does."""
    _, h_scale = M.mean_scale(spec, x)
    h_var = h_scale.pow(2.0)
    a, b, s = [M._t(q, x) for q in spec.obs_params]
    o_var = s.pow(2.0)

    if spec.dim == 0:
        a = a.unsqueeze(-1)
    obs_is_1d = spec.obs_dim == 0
    a_unsq = a if not obs_is_1d else a.unsqueeze(-2)
    a_t = a_unsq.transpose(-2, -1)
    cov = _diag(o_var, spec.obs_dim) + a_unsq.matmul(_diag(h_var, spec.dim)).matmul(a_t)

    if obs_is_1d:
        o_loc = b + a.squeeze(-1) * x
        return M.normal_log_prob(y, o_loc, cov[..., 0, 0].sqrt())

    o_loc = b + (a_unsq @ x.unsqueeze(-1)).squeeze(-1)
    L = torch.linalg.cholesky_ex(cov)[0]
    return torch.distributions.MultivariateNormal(o_loc, scale_tril=L, validate_args=False).log_prob(y)


PROPOSALS = {
    "bootstrap": (bootstrap_sample_and_weight, default_pre_weight),
    "lgo": (lgo_sample_and_weight, lgo_pre_weight),
}


# --------------------------------------------------------------------------------------------------------------
# filters: one step, then the batch_filter driver
# --------------------------------------------------------------------------------------------------------------
def sisr_predict(spec, x, w, prev_inds, u, resample_threshold, resampler="systematic"):
    """SISR.predict: particle/sisr.py:14-48.  Returns (x, w, W, indices, resampled-mask)."""
    W = normalize(w)
    ess = get_ess(W, normalized=True)
    mask = ess < resample_threshold

    batched = w.dim() > 1
    if not bool(mask.any()):
        return x, w, W, prev_inds, mask

    if batched:
        Wm = W[..., mask]
        um = None if u is None else u[mask].reshape(-1, 1)
        sub = systematic(Wm, normalized=True, u=um) if resampler == "systematic" else multinomial(Wm, True)
        um_ = mask.unsqueeze(0)
        indices = prev_inds.masked_scatter(um_, sub)
        w = w.masked_fill(um_, 0.0)
        W = W.masked_fill(um_, 1.0 / w.shape[0])
        temp = x[sub, mask]
        if spec.dim > 0:
            um_ = um_.unsqueeze(-1)
        x = x.masked_scatter(um_, temp)
    else:
        uu = None if u is None else u.reshape(1, 1)
        sub = systematic(W, normalized=True, u=uu) if resampler == "systematic" else multinomial(W, True)
        indices = sub
        w = torch.zeros_like(w)
        W = torch.full_like(W, 1.0 / w.shape[0])
        x = x[sub]
    return x, w, W, indices, mask


def sisr_step(spec, proposal, y, x, w, prev_inds, z, u, resample_threshold, resampler="systematic"):
    """SISR.predict + SISR.correct: particle/sisr.py:14-56.  Returns (x', w', ll, indices, resampled-mask)."""
    sample_and_weight, _ = PROPOSALS[proposal]
    x, w, W, indices, mask = sisr_predict(spec, x, w, prev_inds, u, resample_threshold, resampler)
    x_new, wi = sample_and_weight(spec, y, x, z)
    new_w = wi + w
    ll = log_likelihood(wi, W)
    return x_new, new_w, ll, indices, mask


def apf_step(spec, proposal, y, x, w, z, u, resampler="systematic"):
    """APF.predict + APF.correct: particle/apf.py:16-46."""
    sample_and_weight, pre_weight = PROPOSALS[proposal]
    W_prev = normalize(w)
    pre = pre_weight(spec, y, x)
    rw = pre + w
    if resampler == "systematic":
        uu = None if u is None else u.reshape(-1, 1)
        indices = systematic(rw, u=uu)
    else:
        indices = multinomial(rw)
    x_r = batched_gather(x, indices, 0)
    x_new, ws = sample_and_weight(spec, y, x_r, z)
    w_new = ws - pre.gather(0, indices)
    ll = log_likelihood(w_new) + (W_prev * pre.exp()).sum(dim=0).log()
    return x_new, w_new, ll, indices


def propagate_only_step(spec, x, w, z):
    """ParticleFilterPrediction.create_state_from_prediction (particle/state.py:38-42): NaN observation or an
    unobserved sub-step: propagate, keep weights, ll = 0."""
    return M.propagate(spec, x, z), w, torch.zeros(w.shape[1:], dtype=w.dtype)


def batch_filter(
    spec: M.ModelSpec,
    filt: str,
    proposal: str,
    y: torch.Tensor,
    x0: torch.Tensor,
    z_tape: torch.Tensor,
    u_tape: Optional[torch.Tensor],
    ess_threshold: float = 0.9,
    resampler: str = "systematic",
    record_steps: bool = False,
    observe_every_step: Optional[int] = None,
    time_index: int = 0,
):
    """BaseFilter.batch_filter / filter (filters/base.py:140-221) + FilterResult.append (filters/result.py:119-133);
    ``observe_every_step`` defaults to the spec's, ``time_index`` is the incoming state's.

    Args:
        y: ``(T,[O])`` or - for B independent scalar series - ``(T,B)``.
        x0: ``(N,[B],[D])`` initial particles.
        z_tape: ``(T,N,[B],[D])`` standard normals;  u_tape: ``(T,B)`` uniforms (``B=1`` when unbatched) or None - one
            row per MOVE (= per observation when ``observe_every_step == 1``).
            Either may be None: the draws then come from torch's global CPU generator, as in the reference (this is
            the mode ``bench.py`` times as the CPU baseline).

    Returns dict with ``filter_means (T+1,[B],max(D,1))``, ``filter_variance``, ``loglikelihood ([B])``, final
    ``x, w, prev_inds`` and - with ``record_steps`` - per-step ``x, w, ll, idx``.
    """
    n = x0.shape[0]
    has_d = spec.dim > 0
    x = x0
    w = torch.zeros(x0.shape[: x0.dim() - (1 if has_d else 0)], dtype=x0.dtype)
    prev = torch.arange(n)
    if w.dim() > 1:
        prev = prev.unsqueeze(-1).expand(w.shape)
    ll_total = torch.zeros(w.shape[1:], dtype=x0.dtype)

    mean, var = get_filter_mean_and_variance(x, normalize(w), has_d)
    means, variances = [mean], [var]
    steps = {"x": [], "w": [], "ll": [], "idx": []}
    thr = ess_threshold * n

    # observe_every_step > 1 (filters/base.py:204-210): before the weighted move of an observation the filter makes
    # ``(-time_index) % observe_every_step`` propagate-only moves, each preceded by its own ``predict`` (a SISR may resample
    # there).  The tapes are per MOVE; only the state after an observation's weighted move is reported (the reference's
    # default ``record_intermediary_states=False``), its log-likelihood increment being that move's (the sub-steps add 0).
    oes = int(getattr(spec, "observe_every_step", 1) if observe_every_step is None else observe_every_step)
    time_index, move = int(time_index), 0

    def sub_step(x, w, prev, u, z):
        if filt == "sisr":  # predict still runs (and may resample) before the propagate-only move
            x, w, _, idx, _ = sisr_predict(spec, x, w, prev, u, thr, resampler)
        else:  # APF.predict hands out identity ancestors (apf.py:18-23)
            idx = torch.arange(n)
            if w.dim() > 1:
                idx = idx.unsqueeze(-1).expand(w.shape)
        x, w, ll = propagate_only_step(spec, x, w, z)
        return x, w, ll, idx

    for t in range(y.shape[0]):
        y_t = y[t]
        for _ in range((-time_index) % oes):
            u = None if u_tape is None else u_tape[move]
            z = z_tape[move] if z_tape is not None else torch.randn(x.shape, dtype=x.dtype)
            x, w, _, prev = sub_step(x, w, prev, u, z)
            move, time_index = move + 1, time_index + 1
        # no tape: draw like the reference does (torch's global CPU generator), u first then z (Appendix A)
        u = None if u_tape is None else u_tape[move]
        z = z_tape[move] if z_tape is not None else torch.randn(x.shape, dtype=x.dtype)
        move, time_index = move + 1, time_index + 1
        if bool(y_t.isnan().all()):
            x, w, ll, idx = sub_step(x, w, prev, u, z)
        elif filt == "sisr":
            x, w, ll, idx, _ = sisr_step(spec, proposal, y_t, x, w, prev, z, u, thr, resampler)
        elif filt == "apf":
            x, w, ll, idx = apf_step(spec, proposal, y_t, x, w, z, u, resampler)
        else:
            raise NotImplementedError(filt)
        prev = idx

        mean, var = get_filter_mean_and_variance(x, normalize(w), has_d)
        means.append(mean)
        variances.append(var)
        ll_total = ll_total + ll
        if record_steps:
            for k, v in zip(("x", "w", "ll", "idx"), (x, w, ll, idx)):
                steps[k].append(v.clone())

    out = {
        "filter_means": torch.stack(means, 0),
        "filter_variance": torch.stack(variances, 0),
        "loglikelihood": ll_total,
        "x": x,
        "w": w,
        "prev_inds": prev,
    }
    if record_steps:
        out.update({f"step_{k}": torch.stack(v, 0) for k, v in steps.items()})
    return out


# --------------------------------------------------------------------------------------------------------------
# smoothing over recorded states (pyfilter/filters/particle/base.py:105-157)
# --------------------------------------------------------------------------------------------------------------
def smooth_fl(xs, prev_inds):
    """``_do_sample_fl`` (particle/base.py:136-152).  ``xs``: list of S state tensors ``(N,[B],[D])``; ``prev_inds``: list
    of S ``previous_indices`` tensors ``(N,[B])`` (entry 0 unused).  Returns ``(S, N, [B], [D])``."""
    n = xs[-1].shape[0]
    result = (xs[-1],)
    idx = torch.arange(n)
    if prev_inds[-1].dim() > 1:
        idx = idx.unsqueeze(-1).expand(prev_inds[-1].shape)
    latest = len(xs) - 1
    for s in range(len(xs) - 2, -1, -1):
        idx = batched_gather(prev_inds[latest], idx, 0)
        result += (batched_gather(xs[s], idx, 0),)
        latest = s
    return torch.stack(result[::-1], dim=0)


def ffbs_logits(spec, x_t, w_t, x_next):
    """The logits of one backward step, ``weights = state.weights.unsqueeze(0) + density.log_prob(res[-1].unsqueeze(1))``
    (particle/base.py:112-117): ``(N_j, N_i, [B])`` for trajectories j at ``x_next`` and candidates i of state t."""
    loc, scale = M.mean_scale(spec, x_t)  # (N_i, [B], [D]) each
    lp = M.transition_log_prob(spec, x_next.unsqueeze(1), loc.unsqueeze(0), scale.unsqueeze(0))
    return w_t.unsqueeze(0) + lp


def smooth_ffbs(spec, xs, ws, start, u):
    """``_do_sample_ffbs`` (particle/base.py:105-134) with the ``Categorical(logits).sample()`` draws (not injectable in
    the reference) realised by inverse CDF from the SAME logits: trajectory j takes the first candidate whose running
    probability exceeds ``u[t, j]``.  ``start``: the resampled last state ``(N,[B],[D])``; ``u (S-1, N, [B])``."""
    res = [start]
    for t in range(len(xs) - 2, -1, -1):
        logits = ffbs_logits(spec, xs[t], ws[t], res[-1])  # (N_j, N_i, [B])
        if logits.dim() == 3:
            logits = logits.moveaxis(1, 2)  # (N_j, B, N_i)
        m = logits.max(dim=-1, keepdim=True)[0]
        e = (logits - m).exp().double()
        cdf = e.cumsum(dim=-1)
        target = u[t].double().unsqueeze(-1) * cdf[..., -1:]
        idx = (cdf > target).to(torch.int64).argmax(dim=-1)  # first candidate beyond the target
        x_t = xs[t]
        if spec.dim > 0:
            res.append(x_t.gather(0, idx.unsqueeze(-1).expand(idx.shape + (spec.dim,))))
        else:
            res.append(x_t.gather(0, idx))
    return torch.stack(res[::-1], dim=0)


# --------------------------------------------------------------------------------------------------------------
# exact Kalman filter for the linear-Gaussian kinds (replaces pykalman in the reference's statistical tests,
# tests/filters/test_particle.py:63-111)
# --------------------------------------------------------------------------------------------------------------
def kalman_filter_1d(y, alpha, beta, sigma, a, b, s, m0, p0):
    """Scalar Kalman filter: returns (filtered means (T,), total log-likelihood)."""
    m, p = float(m0), float(p0)
    means, ll = [], 0.0
    for yt in y.tolist():
        m, p = alpha + beta * m, beta * beta * p + sigma * sigma
        if not math.isnan(yt):
            sv = a * a * p + s * s
            k = p * a / sv
            r = yt - (b + a * m)
            ll += -0.5 * (math.log(2 * math.pi * sv) + r * r / sv)
            m, p = m + k * r, (1 - k * a) * p
        means.append(m)
    return torch.tensor(means, dtype=torch.float64), ll


def kalman_filter(y, F, Q, H, R, m0, P0, c=None, d=None):
    """Exact Kalman filter of ``x' = c + F x + N(0, Q)``, ``y = d + H x + N(0, R)`` in NumPy float64 - the role pykalman's
    ``KalmanFilter`` plays in the reference's acceptance test of its 2-D model (tests/filters/models.py:40-47,
    tests/filters/test_particle.py:63-111).  ``y`` is ``(T, O)``; a row with any NaN is a missing observation (predict
    only).  ``(m0, P0)`` is the law of the state BEFORE the first transition.  Returns (filtered means ``(T, D)``, total
    log-likelihood)."""
    import numpy as np

    y = np.asarray(y, dtype=np.float64)
    F, Q, H, R = (np.atleast_2d(np.asarray(a, dtype=np.float64)) for a in (F, Q, H, R))
    m, P = np.atleast_1d(np.asarray(m0, dtype=np.float64)), np.atleast_2d(np.asarray(P0, dtype=np.float64))
    c = np.zeros(F.shape[0]) if c is None else np.asarray(c, dtype=np.float64)
    d = np.zeros(H.shape[0]) if d is None else np.asarray(d, dtype=np.float64)
    means, ll = [], 0.0
    for yt in y.reshape(y.shape[0], -1):
        m, P = c + F @ m, F @ P @ F.T + Q
        if not np.isnan(yt).any():
            S = H @ P @ H.T + R
            K = np.linalg.solve(S, H @ P).T
            r = yt - (d + H @ m)
            ll += -0.5 * (len(yt) * math.log(2 * math.pi) + np.linalg.slogdet(S)[1] + r @ np.linalg.solve(S, r))
            m, P = m + K @ r, (np.eye(len(m)) - K @ H) @ P
        means.append(m.copy())
    return torch.from_numpy(np.stack(means)), float(ll)
