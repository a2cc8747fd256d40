"""The callers either side of the hot path: theta-particles on the filter's batch dimension,
particle marginal Metropolis-Hastings moves that re-run ``batch_filter`` and the SMC^2 loop around ``filter()`` - written
against this library's filters and, for more than one GPU, its ``Shard`` collectives.  Interfaces follow the reference's
``pyfilter.inference`` where it has one (``run_pmmh``, ``PMMH``, ``construct_mvn``, ``ParticleMetropolisHastings``, ``SMC2``); the
reference's prior / context machinery is replaced by the small ``ThetaParticles`` class."""
from .parameters import Prior, ThetaParticles
from .pmmh import PMMH, PMMHState, RandomWalk, SymmetricMH, run_pmmh
from .smc2 import SMC2, ParticleMetropolisHastings, SMC2State, TooManyIncreases
from .utils import calc_mean_chol, construct_mvn

__all__ = ["Prior", "ThetaParticles", "SymmetricMH", "RandomWalk", "PMMH", "PMMHState", "run_pmmh", "SMC2", "SMC2State", "ParticleMetropolisHastings",
           "TooManyIncreases", "calc_mean_chol", "construct_mvn"]
