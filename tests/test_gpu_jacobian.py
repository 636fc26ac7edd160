"""The steady-state Jacobian from its Toeplitz structure (hank_fake_news + the reference's recursion,
SteadyStateJacobian.jl:187-256, :293-323, :358-387) against the same matrix assembled column by column from unit-tangent
JVPs of the full pipeline (method="columns"), and — directly — against the CPU oracle: the household block's Jacobian columns
against the oracle's unit-tangent household block, the assembled J̅ against the oracle's full-pipeline JVP on the reference's
seven columns (test_SteadyState.jl:194-231). Tolerance: 1e-8 of the largest entry (observed 5e-10 .. 1.3e-9) — the Toeplitz
form assumes the recorded steady state is exactly stationary (VFI tolerance 1e-11), the oracle differentiates the path as it
is."""
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import ks_setup

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


@pytest.mark.parametrize("n_a,n_e,T", [(50, 2, 100), (500, 4, 300), (130, 3, 20)])
def test_toeplitz_jacobian_equals_the_unit_tangent_jacobian(hank, n_a, n_e, T):
    m, ss, _ = ks_setup(n_a, n_e, T)
    Jt = hank.getSteadyStateJacobian(ss, m, method="toeplitz").toarray()
    Jc = hank.getSteadyStateJacobian(ss, m, method="columns").toarray()
    assert Jt.shape == Jc.shape == (4 * (T - 1), 4 * (T - 1))
    assert np.max(np.abs(Jc)) > 0.5
    assert np.max(np.abs(Jt - Jc)) < 1e-8 * np.max(np.abs(Jc))
    # the household block's own Jacobian, entry by entry: d KD_t / d r_s and d KD_t / d w_s from unit tangents
    from hank_amd.SteadyStateJacobian import household_jacobian
    from hank_amd.BackwardIteration import household_block
    hb = household_block(m)
    P = T - 1
    x = np.tile(np.array([[ss.vars["r"]], [ss.vars["w"]]]), (1, P))
    hb.set_boundary(ss.value, ss.D)
    hb.primal(x)
    Jhh = household_jacobian(*hb.fake_news())
    assert Jhh.shape == (2, P, P)
    cols = [0, 1, P // 2, P - 2, P - 1]
    y = np.zeros((2, P, 2 * len(cols)))
    for q, s_ in enumerate(cols):
        y[0, s_, 2 * q] = 1.0
        y[1, s_, 2 * q + 1] = 1.0
    d = hb.jvp(y)
    for q, s_ in enumerate(cols):
        assert np.max(np.abs(d[:, 2 * q] - Jhh[0][:, s_])) < 1e-8 * max(1.0, np.max(np.abs(d)))
        assert np.max(np.abs(d[:, 2 * q + 1] - Jhh[1][:, s_])) < 1e-8 * max(1.0, np.max(np.abs(d)))


def test_toeplitz_jacobian_one_asset_hank(hank):
    """three household inputs, leads and lags in the equations around the block"""
    from examples.solve_hank import build
    m, ss = build(80, 3, 40)
    Jt = hank.getSteadyStateJacobian(ss, m, method="toeplitz").toarray()
    Jc = hank.getSteadyStateJacobian(ss, m, method="columns").toarray()
    assert np.max(np.abs(Jt - Jc)) < 1e-8 * max(1.0, np.max(np.abs(Jc)))


def test_fake_news_needs_a_recorded_primal(hank):
    m, ss, _ = ks_setup(50, 2, 100)
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    hb = hank.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
    hb.set_boundary(ss.value, ss.D)
    with pytest.raises(hank.HankHIPError):
        hb.fake_news()
    hb.close()


def test_newton_with_the_toeplitz_preconditioner(hank):
    """the converged path does not depend on which branch built J̅"""
    m, ss, _ = ks_setup(50, 2, 60)
    P = 59
    Z = 1.0 + 0.01 * 0.8 ** np.arange(1, P + 1)
    x0 = np.tile(np.array([ss.vars[k] for k in ("Y", "KS", "r", "w")]), P)
    xs = [hank.NewtonRaphsonHANK(x0, hank.getSteadyStateJacobian(ss, m, method=meth), {"Z": Z}, m, ss, ss, ε=1e-9) for meth in ("toeplitz", "columns")]
    assert np.max(np.abs(xs[0] - xs[1])) < 1e-8


def test_krylov_inner_loop_reaches_the_same_path_with_fewer_jvps(hank):
    """opt-in inner="krylov" (GMRES on J(x), right-preconditioned by the steady-state Jacobian) against the reference's damped
    fixed point (α = 0.5 hard-coded, NewtonRaphson.jl:102): same converged path to 1e-8, a fraction of the JVPs; and a passed
    α is honoured (α = 1: the undamped iteration, fewer inner steps than α = 0.5)."""
    m, ss, _ = ks_setup(50, 2, 60)
    P = 59
    Z = 1.0 + 0.01 * 0.8 ** np.arange(1, P + 1)
    x0 = np.tile(np.array([ss.vars[k] for k in ("Y", "KS", "r", "w")]), P)
    J = hank.getSteadyStateJacobian(ss, m)
    out = {}
    for name, kw in (("reference", {}), ("krylov", {"inner": "krylov"}), ("undamped", {"α": 1.0})):
        hank.y_Iteration.total_jvps = 0
        x = hank.NewtonRaphsonHANK(x0, J, {"Z": Z}, m, ss, ss, ε=1e-9, **kw)
        out[name] = (x, hank.y_Iteration.total_jvps, hank.NewtonRaphsonHANK.iterations)
        lin = hank.LinearizedFunction(x, {"Z": Z}, m, ss, ss)
        assert np.linalg.norm(lin.Fx) < 1e-8
    assert np.max(np.abs(out["krylov"][0] - out["reference"][0])) < 1e-8
    assert np.max(np.abs(out["undamped"][0] - out["reference"][0])) < 1e-8
    assert out["krylov"][1] < out["reference"][1] / 2
    assert out["undamped"][1] < out["reference"][1]


@pytest.mark.parametrize("n_a,n_e,T", [(50, 2, 100), (130, 3, 20)])
def test_toeplitz_household_jacobian_against_the_oracle(hank, n_a, n_e, T):
    """household_jacobian(*hb.fake_news()) columns [0, 1, P//2, P-2, P-1] for every household input against the ORACLE's household
    block under unit tangents (SteadyStateJacobian.jl:300-305, :363-371): 1e-8 of the largest entry."""
    from oracle.oracle import pad_N
    from hank_amd.SteadyStateJacobian import household_jacobian
    from hank_amd.BackwardIteration import household_block
    m, ss, orc = ks_setup(n_a, n_e, T)
    P = T - 1
    hb = household_block(m)
    x = np.tile(np.array([[ss.vars["r"]], [ss.vars["w"]]]), (1, P))
    hb.set_boundary(ss.value, ss.D)
    hb.primal(x)
    Jhh = household_jacobian(*hb.fake_news())
    cols = sorted(set([0, 1, P // 2, P - 2, P - 1]))
    N = 2 * len(cols)
    Nc = pad_N(N)
    xr = np.zeros((P, 1 + Nc)); xw = np.zeros((P, 1 + Nc))
    xr[:, 0], xw[:, 0] = x[0], x[1]
    for q, s_ in enumerate(cols):
        xr[s_, 1 + 2 * q] = 1.0                # d / d r_s
        xw[s_, 1 + 2 * q + 1] = 1.0            # d / d w_s
    st, oagg, _ = orc.household_block(xr, xw, ss.value, ss.D, Nc)
    assert st == 0
    scale = np.max(np.abs(oagg[:, 1:1 + N]))
    assert scale > 1e-3
    for q, s_ in enumerate(cols):
        assert np.max(np.abs(Jhh[0][:, s_] - oagg[:, 1 + 2 * q])) < 1e-8 * scale, ("r", s_)
        assert np.max(np.abs(Jhh[1][:, s_] - oagg[:, 1 + 2 * q + 1])) < 1e-8 * scale, ("w", s_)


def test_toeplitz_jacobian_on_the_reference_seven_columns_against_the_oracle(hank):
    """test_SteadyState.jl:194-231's check — columns [1, 2, three seeded interior, n-1, n] of getSteadyStateJacobian against
    JVP(fullPipelineFunc, x_ss, e_i) — with the ORACLE as the pipeline and 1e-8 of the largest entry instead of the
    reference's abs 1e-5."""
    m, ss, orc = ks_setup(50, 2, 100)
    P = 99
    n = 4 * P
    J = hank.getSteadyStateJacobian(ss, m, method="toeplitz").toarray()
    x0 = np.tile(np.array([ss.vars[k] for k in ("Y", "KS", "r", "w")])[:, None], (1, P))
    Z = np.ones(P)
    rng = np.random.default_rng(42)
    cols = [0, 1, *rng.integers(2, n - 2, 3).tolist(), n - 2, n - 1]
    Y = np.zeros((n, 8))
    for q, c_ in enumerate(cols):
        Y[c_, q] = 1.0
    F_o, J_o = orc.ks_jvp(x0, Y.reshape(4, P, 8, order="F"), Z, m.params.α, m.params.δ, ss.vars["KS"], ss.value, ss.D)
    assert np.max(np.abs(F_o)) < 1e-6                     # F(x_ss) = 0 through the full pipeline (SteadyState.jl:272-286), to the price Newton's tolerance
    # 1e-8 of the largest response of the household block (d KD_t / d r_s, the entries the Toeplitz form computes; the equations'
    # own entries are O(1)): the Toeplitz form takes the recorded steady state as exactly stationary, the oracle differentiates
    # the path as it is (stationary to the value iteration's tolerance)
    from hank_amd.SteadyStateJacobian import household_jacobian
    from hank_amd.BackwardIteration import household_block
    hb = household_block(m)
    hb.set_boundary(ss.value, ss.D)
    hb.primal(np.tile(np.array([[ss.vars["r"]], [ss.vars["w"]]]), (1, P)))
    scale = max(np.max(np.abs(J)), np.max(np.abs(household_jacobian(*hb.fake_news()))))
    for q, c_ in enumerate(cols):
        assert np.max(np.abs(J[:, c_] - J_o[:, q])) < 1e-8 * scale, c_
