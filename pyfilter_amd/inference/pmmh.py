"""One particle marginal Metropolis-Hastings move for B parallel chains / theta-particles
(``pyfilter/inference/batch/mcmc/utils.py:14-77``, ``proposals/symmetric_mh.py``): propose theta*, re-filter the whole
data set with it (ONE fused ``batch_filter`` call for all B filters - a replayed hipGraph), accept per filter, swap the
accepted filters in with the column-move kernels.  Everything stays on the device: there is no host branch per move."""
import torch
from torch.distributions import Distribution

from .utils import construct_mvn, theta_normalize


class SymmetricMH:
    """The proposal of the SMC^2 paper (``proposals/symmetric_mh.py``): a Gaussian fitted to the weighted theta-particles
    (unconstrained space), Cholesky factor scaled by 1.1.  Sharded runs fit it to ALL theta-particles (all-gather of
    ``(B, P)`` values and weights), so every rank holds the same kernel."""

    def build(self, theta, state, filter_, y) -> Distribution:
        values = theta.stack_parameters(constrained=False)
        weights_log = state.w
        shard = getattr(theta, "shard", None)
        if shard is not None and shard.world > 1:
            values, weights_log = shard.all_gather(values), shard.all_gather(weights_log)
        return construct_mvn(values, theta_normalize(weights_log), scale=1.1)

    def exchange(self, latest, candidate, mask) -> None:
        return


def _draw(kernel: Distribution, size, shard, generator):
    """theta* ~ kernel.  With a (CPU) generator the draws of ALL theta-particles come from that one stream - every rank
    advances it identically and keeps its block, so a run's numbers do not depend on how many GPUs share it."""
    if generator is None:
        return kernel.sample(size)
    total = shard.total if shard is not None else (size[0] if len(size) else 1)
    eps = torch.randn((total,) + tuple(kernel.event_shape), generator=generator, dtype=torch.float64)
    if shard is not None:
        eps = shard.slice(eps)
    eps = eps.to(device=kernel.loc.device, dtype=kernel.loc.dtype)
    rvs = kernel.loc + (kernel.scale_tril @ eps.unsqueeze(-1)).squeeze(-1)
    return rvs if len(size) else rvs[0]


def _uniforms(like: torch.Tensor, shard, generator):
    if generator is None:
        return torch.rand(like.shape, device=like.device, dtype=like.dtype)
    total = shard.total if shard is not None else like.shape[0]
    u = torch.rand(total, generator=generator, dtype=torch.float64)
    if shard is not None:
        u = shard.slice(u)
    return u.to(device=like.device, dtype=like.dtype)


def run_pmmh(theta, state, proposal, proposal_kernel: Distribution, proposal_filter, proposal_theta, y: torch.Tensor,
             size=torch.Size([]), mutate_kernel: bool = False, generator=None) -> torch.Tensor:
    """One PMMH iteration (``mcmc/utils.py:14-77``).  ``theta`` / ``state``: the chains' parameters and algorithm state
    (``state.filter_state`` a ``FilterResult``); ``proposal_filter`` reads ``proposal_theta``.  Returns the ``(B,)``
    boolean mask of accepted proposals; ``state`` and ``theta`` are updated in place."""
    shard = getattr(theta, "shard", None)
    rvs = _draw(proposal_kernel, size, shard, generator)
    proposal_theta.unstack_parameters(rvs, constrained=False)
    new_res = proposal_filter.batch_filter(y, bar=False)

    diff_logl = new_res.loglikelihood - state.filter_state.loglikelihood
    diff_prior = proposal_theta.eval_priors(constrained=False) - theta.eval_priors(constrained=False)
    new_kernel = proposal.build(proposal_theta, state.replicate(new_res), proposal_filter, y)
    current = theta.stack_parameters(constrained=False)
    diff_prop = new_kernel.log_prob(current) - proposal_kernel.log_prob(rvs)

    log_acc = diff_prop + diff_prior + diff_logl
    accepted = _uniforms(log_acc, shard, generator).log() < log_acc  # (NaN compares False: a failed proposal is rejected)

    state.filter_state.exchange(new_res, accepted)
    theta.exchange(proposal_theta, accepted)
    if mutate_kernel:
        proposal.exchange(proposal_kernel, new_kernel, accepted)
    return accepted
