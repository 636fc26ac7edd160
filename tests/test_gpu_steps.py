"""Step-level parity on the MI355X, through the C ABI, against the CPU oracle:
hank_backward_step[_dual] == ValueFunction (KrusellSmith.jl:43-83),
hank_forward_step[_dual]  == transition_step + dot (ForwardIteration.jl:37-99, :305-307).
Tolerance: rel 1e-10 of the array scale + abs 1e-12 — summation-order and libm
`pow` differences only."""
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def close(a, b, rel=1e-10, abs_=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(np.max(np.abs(b)), 1e-300)
    err = np.max(np.abs(a - b))
    assert err <= abs_ + rel * scale, f"max err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("N", [1, 3, 8])
def test_backward_step_matches_value_function(ks_small, hank, N):
    m, ss, orc = ks_small
    hb = hank.household_block(m)
    rng = np.random.default_rng(N)
    r, w = ss.vars["r"] * 1.3, ss.vars["w"] * 0.97
    V = ss.value * rng.uniform(0.98, 1.02, ss.value.shape)
    V = np.sort(V, axis=0)[::-1]                        # keep the marginal value decreasing in wealth
    dV = rng.standard_normal(V.shape + (N,))
    dx = rng.standard_normal((2, N))
    st, oV, oKD = orc.value_function(np.concatenate([V[..., None], dV], -1), np.r_[r, dx[0]], np.r_[w, dx[1]], N)
    assert st == 0
    Vo, Po = hb.backward_step(V, [r, w])
    close(Vo, oV[..., 0]); close(Po, oKD[..., 0])
    Vo2, dVo, Po2, dPo = hb.backward_step_dual(V, dV, [r, w], dx)
    assert np.array_equal(Vo, Vo2) and np.array_equal(Po, Po2)
    close(dVo, oV[..., 1:]); close(dPo, oKD[..., 1:])
    # constrained households: policy exactly at the borrowing limit with zero partials
    con = Po == m.params.borrow_cons
    assert con.any() and np.all(dPo[con] == 0.0)


def test_backward_step_flat_extrapolation_both_sides(hank, oracle_mod):
    """queries below the first and above the last knot take the end values with zero partials
    (Flat(), KrusellSmith.jl:71-72)."""
    a = np.linspace(0.0, 10.0, 40)
    z = np.array([0.5, 1.5]); Pi = np.array([[0.9, 0.1], [0.2, 0.8]])
    hb = hank.HouseholdBlock(a, z, Pi, 0.95, 2.0, 0.0, 5)
    orc = oracle_mod.Oracle(a, z, Pi, 0.95, 2.0, 0.0)
    V = np.outer((2.0 + a) ** -2.0, [1.2, 0.8])
    for r, w in [(0.03, 1.0), (-0.5, 0.1), (1.0, 3.0)]:   # the last two push the grid off both ends
        dV = np.random.default_rng(1).standard_normal(V.shape + (2,))
        dx = np.array([[1.0, 0.0], [0.0, 1.0]])
        st, oV, oKD = orc.value_function(np.concatenate([V[..., None], dV], -1), np.r_[r, dx[0]], np.r_[w, dx[1]], 2)
        assert st == 0
        Vo, dVo, Po, dPo = hb.backward_step_dual(V, dV, [r, w], dx)
        close(Vo, oV[..., 0]); close(Po, oKD[..., 0]); close(dVo, oV[..., 1:]); close(dPo, oKD[..., 1:])
    hb.close()


def test_forward_step_edge_cases_golden(hank):
    """clamps, exact grid hits (searchsortedfirst ties), non-monotone policy — against the golden."""
    g = np.load(G / "forward_step_edge_30x3_N3.npz")
    k = np.load(G / "ks_30x3_T25_N3.npz")
    hb = hank.HouseholdBlock(k["a_grid"], k["z_grid"], k["Pi"], float(k["beta"]), float(k["gamma"]), float(k["borrow_cons"]), 5)
    Do, agg = hb.forward_step(g["policy"], g["D_prev"])
    close(Do, g["D_new"][..., 0])
    assert abs(agg - np.sum(g["policy"] * g["D_new"][..., 0])) < 1e-11
    Do2, dDo, agg2, dagg = hb.forward_step_dual(g["policy"], g["dpolicy"], g["D_prev"], g["dD_prev"])
    close(Do2, g["D_new"][..., 0]); close(dDo, g["D_new"][..., 1:])
    exp_dagg = np.einsum("aen,ae->n", g["dpolicy"], g["D_new"][..., 0]) + np.einsum("ae,aen->n", g["policy"], g["D_new"][..., 1:])
    close(dagg, exp_dagg)
    hb.close()


def test_forward_step_matches_oracle_random(ks_small, hank):
    m, ss, orc = ks_small
    hb = hank.household_block(m)
    rng = np.random.default_rng(0)
    N = 4
    pol = ss.policies["KD"] * rng.uniform(0.9, 1.1, ss.value.shape)
    dpol = rng.standard_normal(pol.shape + (N,))
    D = ss.D.reshape(pol.shape, order="F")
    dD = rng.standard_normal(pol.shape + (N,)) * 1e-3
    oD = orc.transition_step(np.concatenate([pol[..., None], dpol], -1), np.concatenate([D[..., None], dD], -1), N)
    Do, dDo, agg, dagg = hb.forward_step_dual(pol, dpol, D, dD)
    close(Do, oD[..., 0]); close(dDo, oD[..., 1:])
    assert abs(Do.sum() - 1.0) < 1e-13


def test_errors_surface_like_julia_exceptions(ks_small, hank):
    m, ss, _ = ks_small
    hb = hank.household_block(m)
    bad = np.array(ss.value, copy=True)
    bad[10, :] *= 1e-4
    with pytest.raises(hank.KnotsNotSortedError, match="sorted"):
        hb.backward_step(bad, [ss.vars["r"], ss.vars["w"]])
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    hb2 = hank.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, 2.5, 0.0, 10)
    with pytest.raises((hank.DomainError, hank.KnotsNotSortedError)):
        hb2.backward_step(-np.abs(ss.value), [ss.vars["r"], ss.vars["w"]])
    with pytest.raises(hank.HankHIPError):          # hank_jvp before hank_primal
        hb2.jvp(np.zeros((2, 9, 1)))
    with pytest.raises(hank.HankHIPError):          # bad shapes are rejected at hank_create
        hank.HouseholdBlock(np.array([0.0, 0.0, 1.0]), np.ones(2), np.full((2, 2), .5), .9, 2., 0., 5)
    hb2.close()
