"""The oracle against the self-consistency checks the reference itself relies on:
 1 steady-state residual; 2 F(x_ss)≈0 through the full pipeline; 4 AD vs finite differences;
 5 residual length; 6 column-stochastic lottery / mass conservation; plus dual-specific checks
 (chunk independence, linearity in the tangent)."""
import numpy as np
import pytest

from conftest import ks_paths


def test_ss_residual_and_full_pipeline_at_ss(ks_small):
    m, ss, orc = ks_small
    P = m.compspec.T - 1
    # item 1: residuals_fn on the tiled SS column (test_SteadyState.jl:61-84)
    xMat = np.tile(np.array([ss.vars[k] for k in m.variables])[:, None], (1, 2))
    assert np.linalg.norm(m.residuals_fn(xMat, m.params)) < 10 * m.compspec.ε
    # hand-written equilibrium identities (test_SteadyState.jl:46-59)
    α, δ = m.params.α, m.params.δ
    v = ss.vars
    assert abs(v["Y"] - v["Z"] * v["KS"] ** α) < 1e-5
    assert abs(v["r"] + δ - α * v["Z"] * v["KS"] ** (α - 1)) < 1e-5
    assert abs(v["w"] - (1 - α) * v["Z"] * v["KS"] ** α) < 1e-5
    assert abs(v["KS"] - v["KD"]) < 1e-5
    # item 2: F(x_ss) ~ 0 through Backward -> Forward -> Residuals with a constant SS path
    x0, _ = ks_paths(m, ss, "x0")
    F, _ = orc.ks_jvp(x0, np.zeros((4, P, 1)), np.ones(P), α, δ, v["KS"], ss.value, ss.D)
    assert F.shape == (4 * P,)          # item 5 (test_Model.jl:84-92)
    assert np.max(np.abs(F)) < 10 * m.compspec.ε


def test_mass_conservation_and_stochastic_lottery(ks_small):
    m, ss, orc = ks_small
    P = m.compspec.T - 1
    x, _ = ks_paths(m, ss, "x1", 0.05)
    N = 2
    xr = np.zeros((P, 1 + N)); xw = np.zeros((P, 1 + N))
    xr[:, 0], xw[:, 0] = x[2], x[3]
    rng = np.random.default_rng(1)
    xr[:, 1:], xw[:, 1:] = rng.standard_normal((P, N)), rng.standard_normal((P, N))
    st, pol = orc.backward_iteration(xr, xw, ss.value, N)
    assert st == 0
    agg, Dseq = orc.forward_iteration(pol, ss.D, N, return_D=True)
    # item 6 (ForwardIteration.jl:28-35): D_t stays a probability vector, its partials sum to zero
    np.testing.assert_allclose(Dseq[..., 0].sum(axis=(1, 2)), 1.0, atol=1e-12)
    assert np.max(np.abs(Dseq[..., 1:].sum(axis=(1, 2)))) < 1e-10
    assert Dseq[..., 0].min() >= 0.0
    # policies are monotone in wealth and respect the borrowing constraint
    assert np.all(np.diff(pol[..., 0], axis=1) >= 0)
    assert pol[..., 0].min() >= m.params.borrow_cons


@pytest.mark.parametrize("kind,shock", [("x0", 0.0), ("x1", 0.05)])
def test_ad_vs_central_differences(ks_small, kind, shock):
    """item 4 (SteadyState.jl:296-356): dual-number JVP vs finite differences of the Float64 path."""
    m, ss, orc = ks_small
    P = m.compspec.T - 1
    α, δ = m.params.α, m.params.δ
    x, Z = ks_paths(m, ss, kind, shock)
    rng = np.random.default_rng(2)
    y = rng.standard_normal((4, P, 2))
    _, J = orc.ks_jvp(x, y, Z, α, δ, ss.vars["KS"], ss.value, ss.D)
    f = lambda xx: orc.ks_jvp(xx, np.zeros((4, P, 1)), Z, α, δ, ss.vars["KS"], ss.value, ss.D)[0]
    for k in range(2):
        h = 1e-6
        fd = (f(x + h * y[:, :, k]) - f(x - h * y[:, :, k])) / (2 * h)
        # the map is piecewise smooth (kinks at bracket changes): FD agrees to O(h) relative
        assert np.max(np.abs(fd - J[:, k])) < 2e-4 * np.max(np.abs(J[:, k]))


def test_chunk_independence_and_linearity(ks_small):
    """partials are independent of the chunk they travel in (N=1 vs N=3 vs padded N=4) and linear."""
    m, ss, orc = ks_small
    P = m.compspec.T - 1
    α, δ = m.params.α, m.params.δ
    x, Z = ks_paths(m, ss, "x1", 0.05)
    rng = np.random.default_rng(3)
    y = rng.standard_normal((4, P, 3))
    F3, J3 = orc.ks_jvp(x, y, Z, α, δ, ss.vars["KS"], ss.value, ss.D)
    for k in range(3):
        F1, J1 = orc.ks_jvp(x, y[:, :, k], Z, α, δ, ss.vars["KS"], ss.value, ss.D)
        assert np.array_equal(F1, F3)
        assert np.array_equal(J1[:, 0], J3[:, k])
    comb = 0.3 * y[:, :, 0] - 1.7 * y[:, :, 1]
    _, Jc = orc.ks_jvp(x, comb, Z, α, δ, ss.vars["KS"], ss.value, ss.D)
    ref = 0.3 * J3[:, 0] - 1.7 * J3[:, 1]
    np.testing.assert_allclose(Jc[:, 0], ref, rtol=0, atol=1e-10 * np.abs(ref).max())


def test_oracle_reports_julia_errors(ks_small):
    """unsorted knots -> Interpolations error; negative base under a fractional power -> DomainError."""
    m, ss, orc = ks_small
    bad = np.array(ss.value, copy=True)
    bad[10, :] *= 1e-4                            # a spike in consumption: knots no longer sorted
    st, _, _ = orc.value_function(bad, [ss.vars["r"]], [ss.vars["w"]], 1)
    assert st == 3
    from oracle.oracle import Oracle
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    o2 = Oracle(wd.grid, pd_.grid, pd_.transition, m.params.β, 2.5, m.params.borrow_cons)
    st, _, _ = o2.value_function(-np.abs(ss.value), [ss.vars["r"]], [ss.vars["w"]], 1)
    assert st in (3, 4)
