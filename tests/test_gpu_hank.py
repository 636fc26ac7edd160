"""The second native value-function family (one-asset HANK: EGM step with a lump-sum transfer, three household
inputs r, om, Tr). NOT in the reference: parity is unpinned by construction — the GPU path is
checked against the oracle's restatement of the same family, against central differences, and by solving the
perfect-foresight response to a monetary shock. Tolerance as for the KS family: rel 1e-10 + abs 1e-12."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def close(a, b, rel=1e-10, abs_=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(np.max(np.abs(b)), 1e-300)
    err = np.max(np.abs(a - b))
    assert err <= abs_ + rel * scale, f"max err {err:.3e} vs scale {scale:.3e}"


@pytest.fixture(scope="module")
def hank_model():
    from examples.solve_hank import build
    return build(80, 3, 40)


def _paths(m, ss, P, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(P)
    x = np.stack([ss.vars["r"] + 0.002 * 0.8 ** t, ss.vars["om"] * (1 + 0.01 * 0.7 ** t), ss.vars["Tr"] * (1 - 0.02 * 0.9 ** t)])
    return x, rng


@pytest.mark.parametrize("N", [1, 4, 32])
def test_household_block_matches_oracle(hank, hank_model, N):
    from oracle.oracle import Oracle, pad_N
    m, ss = hank_model
    P = m.compspec.T - 1
    x, rng = _paths(m, ss, P)
    y = rng.standard_normal((3, P, N))
    hb = hank.household_block(m)
    assert hb.n_hh == 3
    hb.set_boundary(ss.value, ss.D)
    agg, dagg = hb.primal_jvp(x, y)
    wd, pdm = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    orc = Oracle(wd.grid, pdm.grid, pdm.transition, m.params.β, m.params.γ, m.params.borrow_cons)
    Nc = pad_N(N)
    xd = np.zeros((3, P, 1 + Nc))
    xd[..., 0] = x
    xd[..., 1:1 + N] = y
    st, oagg, opol = orc.household_block(xd[0], xd[1], ss.value, ss.D, Nc, xt=xd[2])
    assert st == 0
    close(agg, oagg[:, 0])
    close(dagg, oagg[:, 1:1 + N])
    close(hb.policy_seq().transpose(2, 0, 1), opol[..., 0])
    close(hb.dpolicy_seq(N).transpose(2, 0, 1, 3), opol[..., 1:1 + N])
    # hank_primal, then hank_jvp: the same numbers (to the rounding of the aggregate sums: the default schedule runs these
    # two as XCD-local persistent sweeps and hank_primal_jvp as per-period launches)
    close(hb.primal(x), agg, rel=1e-13)
    close(hb.jvp(y), dagg, rel=1e-12)


def test_transfer_tangent_against_central_differences(hank, hank_model):
    m, ss = hank_model
    P = m.compspec.T - 1
    x, rng = _paths(m, ss, P, 1)
    y = np.zeros((3, P, 1))
    y[2, :, 0] = rng.standard_normal(P)          # a pure transfer direction
    hb = hank.household_block(m)
    hb.set_boundary(ss.value, ss.D)
    _, dagg = hb.primal_jvp(x, y)
    h_ = 1e-6
    fd = (hb.primal(x + h_ * y[..., 0]) - hb.primal(x - h_ * y[..., 0])) / (2 * h_)
    close(dagg[:, 0], fd, rel=2e-6, abs_=1e-8)


def test_granular_step_with_transfer(hank, hank_model):
    from oracle.oracle import Oracle
    m, ss = hank_model
    hb = hank.household_block(m)
    wd, pdm = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    orc = Oracle(wd.grid, pdm.grid, pdm.transition, m.params.β, m.params.γ, m.params.borrow_cons)
    rng = np.random.default_rng(2)
    N = 3
    xt = np.array([ss.vars["r"], ss.vars["om"], ss.vars["Tr"]])
    dxt = rng.standard_normal((3, N))
    dV = 1e-2 * rng.standard_normal(ss.value.shape + (N,))
    V, dVo, pol, dpol = hb.backward_step_dual(ss.value, dV, xt, dxt)
    vin = np.concatenate([ss.value[..., None], dV], axis=-1)
    st, oV, oK = orc.value_function(vin, np.r_[xt[0], dxt[0]], np.r_[xt[1], dxt[1]], N, tr=np.r_[xt[2], dxt[2]])
    assert st == 0
    close(V, oV[..., 0]); close(pol, oK[..., 0]); close(dVo, oV[..., 1:]); close(dpol, oK[..., 1:])


def test_monetary_shock_converges(hank):
    from examples.solve_hank import solve
    out, x, m, ss = solve(80, 3, 60, shock=0.0025)
    assert out["residual_norm"] < 1e-8
    # a contractionary shock: the real rate rises, output and inflation fall on impact
    assert out["impact"]["r"] > 0 or out["impact"]["i"] > 0
    assert out["impact"]["Y"] < 0 and out["impact"]["infl"] < 0


def test_solve_above_2000_unknowns_keeps_the_persistent_schedule(hank):
    """find_ss + NewtonRaphsonHANK with >= 2 000 unknowns (the size from which the device linear solver and its library warm-up
    are used): no device work of the host layer may overlap a persistent sweep — the context must end on the schedule it started
    on, with no fallback (the warm-up used to run in a background thread beside the steady state's sweeps)."""
    from examples.solve_hank import solve
    import hank_amd as h
    out, x, m, ss = solve(130, 3, 300, shock=0.0025)           # 7 unknowns x 299 periods
    assert out["residual_norm"] < 1e-8
    st = h.household_block(m).stats()
    assert st["fallbacks"] == 0 and st["schedule"] != 0, st


def test_device_and_host_linear_solvers_reach_the_same_path(hank):
    """`linear_solver="device"` (J̅⁻¹ formed once on the model's GPU, one GEMV per inner iteration) against `"lu"` (host LU) on a
    model large enough for the device branch (7 x 299 unknowns): the same converged path, and the factorisations are released when
    NewtonRaphsonHANK returns (they used to live for the life of the process, on torch's current device)."""
    from examples.solve_hank import build
    import hank_amd as h
    from hank_amd import NewtonRaphson as nr
    m, ss = build(130, 3, 300)
    P = m.compspec.T - 1
    ei = {"ei": 0.0025 * 0.6 ** np.arange(P)}
    keys = h.vars_of_type(m, "endogenous")
    x0 = np.tile(np.array([ss.vars[k] for k in keys]), P)
    J = h.getSteadyStateJacobian(ss, m)
    xs = {ls: h.NewtonRaphsonHANK(x0, J, ei, m, ss, ss, linear_solver=ls) for ls in ("lu", "device")}
    assert not nr._INV_CACHE and not nr._LU_CACHE
    close(xs["device"], xs["lu"], rel=1e-8)
    assert np.linalg.norm(h.LinearizedFunction(xs["device"], ei, m, ss, ss).Fx) < 1e-8


def test_residual_layer_linearised_once_equals_the_dual_evaluation(hank):
    """LinearizedFunction.jvp through the sparse maps dR/dx, dR/dagg built once per x (one Dual evaluation of the compiled
    equations, colours = padded column mod (1 + max_lag + max_lead)) against the reference's way — the equations re-evaluated
    under the Dual at every call (Aggregation.jl:20-22) — for a model with leads AND lags around the block (one-asset HANK) and
    for Krusell-Smith, single tangents and a batch, away from the steady state."""
    from conftest import ks_paths, ks_setup
    from examples.solve_hank import build
    m, ss = build(80, 3, 40)
    P = 39
    keys = hank.vars_of_type(m, "endogenous")
    rng = np.random.default_rng(11)
    x = np.tile(np.array([ss.vars[k] for k in keys]), P) * (1.0 + 1e-3 * rng.standard_normal(len(keys) * P))
    ei = 0.0025 * 0.6 ** np.arange(P)
    cases = [(m, ss, x, {"ei": ei})]
    mk, ssk, _ = ks_setup(50, 2, 100)
    xk, Zk = ks_paths(mk, ssk, "x1", 0.05)
    cases.append((mk, ssk, xk.reshape(-1, order="F"), {"Z": Zk}))
    for mod, s_, x_, exog in cases:
        lin = hank.LinearizedFunction(x_, exog, mod, s_, s_)
        Y = rng.standard_normal((len(x_), 5))
        fast1, fastN = lin.jvp(Y[:, 0]), lin.jvp(Y)
        lin.exact_residual_layer = True
        ref1, refN = lin.jvp(Y[:, 0]), lin.jvp(Y)
        scale = np.abs(refN).max()
        assert np.max(np.abs(fast1 - ref1)) <= 1e-12 * scale and np.max(np.abs(fastN - refN)) <= 1e-12 * scale
