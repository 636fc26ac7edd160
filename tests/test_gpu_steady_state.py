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
