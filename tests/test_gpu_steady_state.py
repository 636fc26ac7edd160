"""Steady state with the inner value-function fixed point on the device (hank_vfi; SURVEY.md §8f rank 1,
SteadyState.jl:132-141): the EGM step kernels of the hot path iterated on HBM, price Newton and invariant_dist on
the host."""
import time

import numpy as np
import pytest

from conftest import ROOT, ks_setup

pytestmark = pytest.mark.gpu


def test_device_vfi_matches_the_host_iteration(hank):
    """same stopping rule as the reference (max|Δvalue| < tol after every step): the device loop stops after the same
    number of steps as the numpy loop and lands on the same value and policy (the two differ in the rounding of the
    interpolation only)."""
    m, ss, _ = ks_setup(50, 2, 100)
    hb = hank.household_block(m)
    xv = dict(ss.vars)
    vf = m.value_fn
    value = np.ones((50, 2))
    tol = 1e-11
    res = vf.host_steady_state_step(value, xv, m)
    steps = 0
    for _ in range(10_000):
        vn = res["Value"]; d = np.max(np.abs(vn - value)); value = vn; steps += 1
        if d < tol:
            break
        res = vf.host_steady_state_step(value, xv, m)
    v, pol, it, nrm = hb.vfi(np.ones((50, 2)), [xv["r"], xv["w"]], tol)
    assert abs(it - steps) <= 1 and nrm < tol
    assert np.max(np.abs(v - value)) < 1e-9 * np.abs(value).max()
    assert np.max(np.abs(pol - res["KD"])) < 1e-9 * np.abs(res["KD"]).max()
    # a step from the converged value is a fixed point of the granular step too
    v2, pol2 = hb.backward_step(v, [xv["r"], xv["w"]])
    assert np.max(np.abs(v2 - v)) < 10 * tol
    assert hb.stats()["vfi_iterations"] >= it


def test_headline_steady_state_from_a_cold_start(hank):
    """2000x11 Krusell-Smith steady state from the YAML guesses (no fixture, VFI from ones) with the device VFI:
    the committed fixture's prices, value and distribution to 1e-10 (relative), in seconds instead of a minute."""
    ov = {"T": 300, "dimensions": {"wealth": {"n": 2000}, "productivity": {"n": 11}}}
    m = hank.build_model_from_yaml(str(ROOT / "examples" / "krusell_smith.yaml"), overrides=ov)
    t0 = time.perf_counter()
    ss, _ = hank.get_SteadyStates(m, vfi="device")
    el = time.perf_counter() - t0
    g = np.load(ROOT / "examples" / "fixtures" / "ks_ss_2000x11.npz")
    for k in m.variables:
        assert abs(ss.vars[k] - float(g[f"var_{k}"])) < 1e-8 * max(1.0, abs(float(g[f"var_{k}"]))), k
    assert np.max(np.abs(ss.value - g["value"])) < 1e-8 * np.abs(g["value"]).max()
    assert np.max(np.abs(ss.D - g["D"])) < 1e-8
    assert np.max(np.abs(ss.policies["KD"] - g["policy"])) < 1e-8 * np.abs(g["policy"]).max()
    print(f"2000x11 steady state, device VFI: {el:.2f} s")
    assert el < 10.0


def test_device_stationary_distribution_matches_the_host(hank):
    """hank_stationary_dist (power method with the forward step kernel) against the host's invariant_dist on the lottery
    matrix of the same policy: same fixed point (both stop when iterates 25 steps apart agree to 1e-15)."""
    import scipy.sparse as sp
    m, ss, _ = ks_setup(500, 4, 300)
    hb = hank.household_block(m)
    pol = ss.policies["KD"]
    D_dev, steps = hb.stationary_dist(pol)
    wd, pdm = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    Λ_exog = sp.kron(sp.csc_matrix(pdm.transition.T), sp.identity(wd.n, format="csc"), format="csc")
    Λ = (Λ_exog @ hank.make_endogenous_transition(pol, wd, pdm.n)).tocsc()
    assert np.max(np.abs(Λ @ D_dev - D_dev)) < 1e-14            # a fixed point of the reference's transition matrix
    assert abs(D_dev.sum() - 1.0) < 1e-14 and D_dev.min() >= 0.0
    assert np.max(np.abs(D_dev - ss.D)) < 1e-10
    assert steps > 25


def test_persistent_and_launched_value_iteration_agree(hank, monkeypatch):
    """hank_vfi as ONE persistent launch (k_xvfi: the vote on convergence rides on the group barrier) against the
    per-step launches: the same number of steps, the same value and policy (same expressions; both stop on the Float64
    comparison max|Δvalue| < tol), at a small grid and at the headline grid."""
    for n_a, n_e, T in ((50, 2, 100), (40, 16, 8), (2000, 11, 300)):      # (40x16: the 1024-thread variants)
        m, ss, _ = ks_setup(n_a, n_e, T)
        xv = dict(ss.vars)
        out = {}
        for sched in ("launch", "xcd"):
            monkeypatch.setenv("HANK_SCHEDULE", sched)
            hb = hank.HouseholdBlock(m.heterogeneity["wealth"].grid, m.heterogeneity["productivity"].grid,
                                     m.heterogeneity["productivity"].transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
            out[sched] = hb.vfi(np.ones((n_a, n_e)), [xv["r"], xv["w"]], 1e-11)
            assert hb.stats()["fallbacks"] == 0
            hb.close()
        (v0, p0, it0, n0), (v1, p1, it1, n1) = out["launch"], out["xcd"]
        assert it0 == it1 and n1 < 1e-11 and n0 < 1e-11
        assert np.max(np.abs(v1 - v0)) <= 1e-13 * np.abs(v0).max()
        assert np.max(np.abs(p1 - p0)) <= 1e-13 * np.abs(p0).max()
        # not converged within the cap: both report the cap and the last iterate
        monkeypatch.setenv("HANK_SCHEDULE", "xcd")
        hb = hank.HouseholdBlock(m.heterogeneity["wealth"].grid, m.heterogeneity["productivity"].grid,
                                 m.heterogeneity["productivity"].transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
        v2, p2, it2, n2 = hb.vfi(np.ones((n_a, n_e)), [xv["r"], xv["w"]], 1e-11, 7)
        assert it2 == 7 and n2 > 1e-11
        hb.close()


def test_persistent_value_iteration_reports_the_reference_errors(hank, monkeypatch):
    """a wage so negative that consumption turns negative: the DomainError of the reference's power, raised by the
    persistent loop in the step it happens (it leaves the loop through the vote), not after max_iter steps."""
    monkeypatch.setenv("HANK_SCHEDULE", "xcd")
    m, ss, _ = ks_setup(50, 2, 100)
    hb = hank.HouseholdBlock(m.heterogeneity["wealth"].grid, m.heterogeneity["productivity"].grid,
                             m.heterogeneity["productivity"].transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
    with pytest.raises((hank.DomainError, hank.KnotsNotSortedError)):
        hb.vfi(np.ones((50, 2)), [0.01, -50.0], 1e-11)
    # the context is usable afterwards
    xv = dict(ss.vars)
    v, pol, it, nrm = hb.vfi(np.ones((50, 2)), [xv["r"], xv["w"]], 1e-11)
    assert nrm < 1e-11
    hb.close()


def test_persistent_and_launched_power_method_agree(hank, monkeypatch):
    """hank_stationary_dist as ONE persistent launch (k_xstat) against one launch per iteration: the same fixed point of the
    reference's transition matrix to 1e-14, from a uniform start and from a warm start; the cap on the iterations is
    honoured."""
    import scipy.sparse as sp
    for n_a, n_e, T in ((500, 4, 300), (40, 16, 8), (2000, 11, 300)):
        m, ss, _ = ks_setup(n_a, n_e, T)
        pol = ss.policies["KD"]
        wd, pdm = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
        Λ_exog = sp.kron(sp.csc_matrix(pdm.transition.T), sp.identity(wd.n, format="csc"), format="csc")
        Λ = (Λ_exog @ hank.make_endogenous_transition(pol, wd, pdm.n)).tocsc()
        out = {}
        for sched in ("launch", "xcd"):
            monkeypatch.setenv("HANK_SCHEDULE", sched)
            hb = hank.HouseholdBlock(wd.grid, pdm.grid, pdm.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
            D, steps = hb.stationary_dist(pol)
            assert np.max(np.abs(Λ @ D - D)) < 1e-14 and abs(D.sum() - 1.0) < 1e-13 and D.min() >= 0.0
            Dw, steps_w = hb.stationary_dist(pol, D0=D)          # warm start at the fixed point: the first check passes (the
            # launched loop reports the steps it had enqueued when the host saw the flag: a chunk of 16 checks)
            assert steps_w <= (50 if sched == "xcd" else 400) and np.max(np.abs(Dw - D)) < 1e-14
            Dc, steps_c = hb.stationary_dist(pol, max_iter=60)
            assert steps_c <= 75
            assert hb.stats()["fallbacks"] == 0
            out[sched] = (D, steps)
            hb.close()
        assert np.max(np.abs(out["xcd"][0] - out["launch"][0])) < 1e-13
        assert np.max(np.abs(out["xcd"][0] - ss.D)) < 1e-10


def test_power_method_stops_only_when_every_member_has_converged(hank, monkeypatch):
    """far from the solution (r = 4 %: every household saves, the mass ends on the top grid point) the bottom of the grid
    settles hundreds of iterations before the top. The persistent launch must keep going until EVERY member's rows pass the
    rule — the first version of its vote looked at member 0 only and stopped at 500 of 800 iterations with |ΛD − D| = 6e-9,
    which the steady state's price Newton then paid for with twice as many inner solves."""
    import scipy.sparse as sp
    m, ss, _ = ks_setup(2000, 11, 300)
    wd, pdm = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    Λ_exog = sp.kron(sp.csc_matrix(pdm.transition.T), sp.identity(wd.n, format="csc"), format="csc")
    out = {}
    for sched in ("launch", "xcd"):
        monkeypatch.setenv("HANK_SCHEDULE", sched)
        hb = hank.HouseholdBlock(wd.grid, pdm.grid, pdm.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
        v, pol, it, nrm = hb.vfi(np.ones((2000, 11)), [0.04, 1.0], 1e-11)
        Λ = (Λ_exog @ hank.make_endogenous_transition(pol, wd, pdm.n)).tocsc()
        D, steps = hb.stationary_dist(pol)
        assert np.max(np.abs(Λ @ D - D)) < 1e-14, (sched, steps)
        assert D.reshape(2000, 11, order="F")[-1].sum() > 0.999          # the mass sits on the top grid point
        out[sched] = (D, steps)
        hb.close()
    assert out["xcd"][1] == 800 and out["launch"][1] == 800               # the rule first holds at the 32nd check
    assert np.max(np.abs(out["xcd"][0] - out["launch"][0])) < 1e-13


def _oracle_vfi(orc, shape, r, w, tol, cap=20_000):
    """the reference's inner fixed point (SteadyState.jl:132-141) on the CPU oracle's ValueFunction: value <- value_fn(value).Value
    from ones until max|new - old| < tol; returns (value, policy, steps, per-step sup-norms)."""
    value = np.ones(shape)
    norms = []
    for k in range(1, cap + 1):
        st, V, KD = orc.value_function(value, r, w, 1)
        assert st == 0
        vn, pol = V[..., 0], KD[..., 0]
        norms.append(float(np.max(np.abs(vn - value))))
        value = vn
        if norms[-1] < tol:
            return value, pol, k, norms
    raise AssertionError("oracle VFI did not converge")


@pytest.mark.parametrize("n_a,n_e,T", [(50, 2, 100), (37, 3, 9)])
def test_device_vfi_matches_the_oracle_iteration(hank, monkeypatch, n_a, n_e, T):
    """hank_vfi (both schedules) against oracle/ — the C restatement of KrusellSmith.jl:43-83 iterated with the reference's
    stopping rule: the same number of steps, value and policy to rel 1e-10 (SURVEY.md 8c tolerance)."""
    m, ss, orc = ks_setup(n_a, n_e, T)
    r, w, tol = ss.vars["r"], ss.vars["w"], 1e-11
    v_o, p_o, steps_o, _ = _oracle_vfi(orc, (n_a, n_e), r, w, tol)
    wd, pdm = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    for sched in ("launch", "xcd"):
        monkeypatch.setenv("HANK_SCHEDULE", sched)
        hb = hank.HouseholdBlock(wd.grid, pdm.grid, pdm.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
        v, pol, it, nrm = hb.vfi(np.ones((n_a, n_e)), [r, w], tol)
        assert abs(it - steps_o) <= 1 and nrm < tol, (sched, it, steps_o)      # (a sup-norm within rounding of tol may tip one step)
        assert np.max(np.abs(v - v_o)) < 1e-10 * np.abs(v_o).max()
        assert np.max(np.abs(pol - p_o)) < 1e-10 * max(1.0, np.abs(p_o).max())
        hb.close()


def test_value_iteration_stops_only_when_every_member_has_converged(hank, monkeypatch):
    """the twin of the power-method case above for k_xvfi's vote: at r = 1.5 %, w = 1 the marginal value of the rich keeps
    moving long after the bottom of the grid has settled — member 0's 63 rows pass max|Δ| < tol at step 662, the top member at
    818 (on the oracle). A vote that looked at member 0 only would stop 156 steps early. The persistent loop, the launched
    loop and the oracle iteration must all stop in the step in which the LAST rows pass, and agree on value and policy."""
    m, ss, orc = ks_setup(500, 4, 300)
    wd, pdm = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    r, w, tol = 0.015, 1.0, 1e-10
    v_o, p_o, steps_o, norms = _oracle_vfi(orc, (500, 4), r, w, tol)
    # the members really do converge at different times: the first 63 rows pass the rule long before the whole grid does
    value = np.ones((500, 4)); first = None
    for k in range(1, steps_o + 1):
        st, V, _ = orc.value_function(value, r, w, 1)
        d = np.abs(V[..., 0] - value); value = V[..., 0]
        if first is None and np.max(d[:63]) < tol:
            first = k
    assert first is not None and first < steps_o - 100, (first, steps_o)
    out = {}
    for sched in ("launch", "xcd"):
        monkeypatch.setenv("HANK_SCHEDULE", sched)
        hb = hank.HouseholdBlock(wd.grid, pdm.grid, pdm.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
        out[sched] = hb.vfi(np.ones((500, 4)), [r, w], tol)
        assert hb.stats()["fallbacks"] == 0
        hb.close()
    assert out["xcd"][2] == out["launch"][2]
    assert abs(out["xcd"][2] - steps_o) <= 1
    for sched in out:
        assert np.max(np.abs(out[sched][0] - v_o)) < 1e-10 * np.abs(v_o).max()
        assert np.max(np.abs(out[sched][1] - p_o)) < 1e-10 * max(1.0, np.abs(p_o).max())
