"""f2 pinned against the reference, CPU leg: the PRODUCT's SMC^2 / PMMH host code
(``pyfilter_amd.inference``: ``SMC2.step``, ``SMC2State.append``, ``ParticleMetropolisHastings.update`` incl.
``_increase_states``, ``run_pmmh``, ``construct_mvn``, ``SymmetricMH`` / ``RandomWalk``) replays event logs recorded from
the unmodified reference (``oracle/make_golden_inference.py``) - same draws at the same points, every intermediate
quantity compared.  The particle filter underneath is the oracle here; ``tests/test_inference_reference_gpu.py`` runs
the same replay on the HIP filters."""
import pytest
import torch

from tests.replay import Cursor, ReplayDraws, close, compare_update, load_events, taped

SMC2_CASES = {"inference_smc2_ou": dict(B=12, N=96, threshold=0.5, kwargs=dict(num_steps=2)),
              "inference_smc2_ou_adaptive": dict(B=10, N=64, threshold=0.6, kwargs=dict(num_steps=6, distance_threshold=0.5)),
              "inference_smc2_ou_increase": dict(B=12, N=16, threshold=0.6, kwargs=dict(num_steps=2, acceptance_threshold=0.6))}


def priors():
    from torch.distributions import Exponential, LogNormal, Normal

    return {"kappa": Exponential(10.0), "gamma": Normal(0.0, 1.0), "sigma": LogNormal(-2.0, 1.0)}  # tests/inference/models.py:29-31


def replay_smc2(name, make_filter, device, rtol=1e-8):
    """Drives ``SMC2.step`` over the fixture's observations; returns the number of rejuvenations compared."""
    from pyfilter_amd.inference import SMC2

    case = SMC2_CASES[name]
    events = load_events(name)
    cur = Cursor(events)
    head = cur.take("theta0")
    y = head["y"].to(device)
    filt = make_filter(cur, case["N"])
    alg = SMC2(filt, case["B"], priors(), threshold=case["threshold"], device=device, dtype=torch.float64, **case["kwargs"])
    alg._gen = ReplayDraws(cur)
    alg._kernel.trace = []
    state = alg.initialize(theta0=head["theta"])
    close(alg.theta.stack_parameters(True), head["theta"], "theta0")
    updates = 0
    for t in range(y.shape[0]):
        at = cur.at
        ev = events[at][1]
        assert events[at][0] == "move"
        state = alg.step(y[t], state)
        close(filt.last_move_ll, ev["ll"], f"t={t}: log-likelihood increments", rtol=rtol)
        close(state.ess[t + 1], ev["ess_after"], f"t={t}: ESS of the theta-weights", rtol=rtol)
        if cur.peek() == "rejuvenated" or events[at + 1][0] == "rejuvenate":
            done = cur.take("rejuvenated")
            compare_update(alg._kernel.trace, events, at + 1, cur.at, f"{name} t={t}")
            alg._kernel.trace.clear()
            close(alg.theta.stack_parameters(True), done["theta"], f"t={t}: theta after the update", rtol=rtol)
            close(state.w, done["w"], f"t={t}: theta-weights after the update", rtol=1e-7, atol=1e-7)
            close(state.filter_state.loglikelihood, done["ll"], f"t={t}: log-likelihoods after the update", rtol=rtol)
            assert int(done["n"]) == filt._base_particles[0], "state particles after the update"
            updates += 1
    fin = cur.take("final")
    assert cur.peek() is None
    close(state.filter_state.filter_means, fin["filter_means"], "filter means of the whole run", rtol=1e-7, atol=1e-9)
    close(state.filter_state.filter_variance, fin["filter_variance"], "filter variances", rtol=1e-6, atol=1e-10)
    close(state.filter_state.loglikelihood, fin["ll"], "log-likelihoods", rtol=rtol)
    close(state.w, fin["w"], "theta-weights", rtol=1e-7, atol=1e-7)
    close(torch.stack(state.ess), fin["ess"], "ESS history", rtol=rtol)
    close(alg.theta.stack_parameters(True), fin["theta"], "theta")
    return updates, [str(f["outcome"]) for k, f in events if k == "rejuvenated"]


def replay_pmmh(name, make_filter, device, rtol=1e-8):
    """``run_pmmh`` with the random-walk kernel (``mutate_kernel=True``), the way ``PMMH.fit`` calls it (pmmh.py:84-101)."""
    from pyfilter_amd.inference import RandomWalk, ThetaParticles
    from pyfilter_amd.inference.pmmh import PMMHState, run_pmmh

    events = load_events(name)
    cur = Cursor(events)
    head = cur.take("theta0")
    y = head["y"].to(device)
    b = head["theta"].shape[0]
    theta = ThetaParticles(priors(), b, device=device, dtype=torch.float64).initialize_parameters(torch.Generator().manual_seed(0))
    theta.unstack_parameters(head["theta"].to(device), constrained=True)
    filt = make_filter(cur, 64)
    filt.set_batch_shape(torch.Size([b]))
    filt.initialize_model(theta)
    first = filt.batch_filter(y, bar=False)
    close(first.loglikelihood, filt.last_run_ll[1], "the chains' first filter run", rtol=rtol)
    state = PMMHState(first, theta.stack_parameters(True), 5)
    proposal = RandomWalk(0.05)
    kernel = proposal.build(theta, state, filt, y)
    k0 = cur.take("kernel0")
    close(kernel.mean, k0["loc"], "random-walk kernel: loc")
    close(kernel.stddev, k0["scale"], "random-walk kernel: scale")
    proposal_theta, proposal_filter = theta.like(), filt.copy()
    proposal_filter.initialize_model(proposal_theta)
    draws = ReplayDraws(cur)
    moves = 0
    while cur.peek() == "pmmh_draw":
        at, trace = cur.at, []
        run_pmmh(theta, state, proposal, kernel, proposal_filter, proposal_theta, y, torch.Size([]), mutate_kernel=True,
                 generator=draws, trace=trace)
        dr, ac = events[at][1], events[cur.at - 1][1]
        close(trace[0]["rvs"], dr["rvs"], f"move {moves}: theta*")
        close(trace[0]["proposed_ll"], proposal_filter.last_run_ll[1], f"move {moves}: proposal filter log-likelihood", rtol=rtol)
        close(trace[0]["log_acc"], ac["log_acc"], f"move {moves}: log acceptance probability", rtol=1e-7, atol=1e-7)
        assert torch.equal(trace[0]["accepted"].cpu(), ac["accepted"]), f"move {moves}: accepted mask"
        close(theta.stack_parameters(True), ac["theta"], f"move {moves}: theta after the exchange")
        close(state.filter_state.loglikelihood, ac["ll"], f"move {moves}: log-likelihoods after the exchange", rtol=rtol)
        close(kernel.mean, ac["kernel_loc_after"], f"move {moves}: re-centred kernel")
        moves += 1
    fin = cur.take("final")
    close(state.filter_state.filter_means, fin["filter_means"], "filter means", rtol=1e-7, atol=1e-9)
    return moves


def _oracle_filter(cursor, n):
    from tests.oracle_filter import OracleAPF

    class OracleLGO(OracleAPF):
        PROPOSAL, STATIONARY_INIT = "lgo", True

    cls = taped(OracleLGO)
    cls.cursor = cursor
    return cls(None, n)


@pytest.mark.parametrize("name", sorted(SMC2_CASES))
def test_smc2_replays_the_reference_event_log(name):
    updates, outcomes = replay_smc2(name, _oracle_filter, "cpu")
    assert updates == len(outcomes) >= 2
    if name.endswith("increase"):
        assert "increase" in outcomes and "done" in outcomes


def test_run_pmmh_random_walk_replays_the_reference_event_log():
    assert replay_pmmh("inference_pmmh_ou_rw", _oracle_filter, "cpu") == 5
