"""The reference-shaped API on the MI355X: BackwardIteration / ForwardIteration / JVP closures,
the steady-state-Jacobian column test of test_SteadyState.jl:162-231 and Newton to convergence."""
import numpy as np
import pytest

from conftest import ks_paths, ks_setup

pytestmark = pytest.mark.gpu


def test_fullFunction_closure_like_y_Iteration(hank):
    """NewtonRaphson.jl:77-83 written with this package's functions; JVP (GeneralStructures.jl:542-550)
    with one direction and with a batch; equals the oracle's dual pipeline."""
    m, ss, orc = ks_setup(50, 2, 100)
    P = 99
    x, Z = ks_paths(m, ss, "x1", 0.05)
    exog = {"Z": Z}

    def fullFunction(x_Vec):
        policy_seqs = hank.BackwardIteration(x_Vec, exog, m, ss)
        agg_seqs = hank.ForwardIteration(policy_seqs, m, ss)
        padded = hank.assemble_full_xMat(x_Vec, agg_seqs, exog, m, ss, ss)
        return hank.Residuals(padded, m)

    xv = x.reshape(-1, order="F")
    Y = np.random.default_rng(0).standard_normal((4 * P, 3))
    Fx = fullFunction(xv)
    F_o, J_o = orc.ks_jvp(x, Y.reshape(4, P, 3, order="F"), Z, m.params.α, m.params.δ, ss.vars["KS"], ss.value, ss.D)
    assert np.max(np.abs(Fx - F_o)) < 1e-11
    one = hank.JVP(fullFunction, xv, Y[:, 0])
    assert one.shape == (4 * P,) and np.max(np.abs(one - J_o[:, 0])) < 1e-10 * np.abs(J_o).max()
    batch = hank.JVP(fullFunction, xv, Y)
    assert np.max(np.abs(batch - J_o)) < 1e-10 * np.abs(J_o).max()
    # BackwardIteration's return value: NamedTuple-like, T-1 matrices per heterogeneous variable
    seqs = hank.BackwardIteration(xv, exog, m, ss)
    assert list(seqs.keys()) == ["KD"] and len(seqs["KD"]) == P and seqs["KD"][0].shape == (50, 2)


def test_forward_iteration_with_user_supplied_policies(hank):
    """ForwardIteration on explicit matrices (not tagged device sequences) takes the granular path."""
    m, ss, orc = ks_setup(30, 3, 25)
    x, Z = ks_paths(m, ss, "x1", 0.05)
    seqs = hank.BackwardIteration(x.reshape(-1, order="F"), {"Z": Z}, m, ss)
    plain = {"KD": [np.array(p) for p in seqs["KD"]]}
    agg_generic = hank.ForwardIteration(plain, m, ss)["KD"]
    agg_fused = hank.ForwardIteration(seqs, m, ss)["KD"]
    assert np.max(np.abs(agg_generic - agg_fused)) < 1e-11


def test_ss_jacobian_columns_vs_full_pipeline_jvp(hank):
    """test_SteadyState.jl:162-231: 7 columns (1, 2, three seeded interior, n-1, n) of
    getSteadyStateJacobian vs JVP(fullPipelineFunc, x_ss, e_i), abs tol 1e-5 on the column norm."""
    m, ss, orc = ks_setup(50, 2, 100)
    P = 99
    n = 4 * P
    J = hank.getSteadyStateJacobian(ss, m).toarray()
    assert J.shape == (n, n)
    x0, _ = ks_paths(m, ss, "x0")
    rng = np.random.default_rng(42)
    cols = [0, 1, *rng.integers(2, n - 2, 3).tolist(), n - 2, n - 1]
    E = np.zeros((4, P, len(cols)))
    for k, c in enumerate(cols):
        E.reshape(n, len(cols), order="F")[c, k] = 1.0
    E = np.zeros((n, len(cols))); E[cols, range(len(cols))] = 1.0
    _, J_o = orc.ks_jvp(x0, E.reshape(4, P, len(cols), order="F"), np.ones(P), m.params.α, m.params.δ, ss.vars["KS"], ss.value, ss.D)
    for k, c in enumerate(cols):
        assert np.linalg.norm(J[:, c] - J_o[:, k]) < 1e-5


def test_newton_converges_on_a_small_shock(hank):
    """NewtonRaphsonHANK (NewtonRaphson.jl:27-46) with J̅ = the SS Jacobian: the converged path
    zeroes the full-pipeline residual and matches the oracle's residual there."""
    m, ss, orc = ks_setup(50, 2, 60)
    P = 59
    Z = 1.0 + 0.01 * 0.8 ** np.arange(1, P + 1)
    J = hank.getSteadyStateJacobian(ss, m)
    x0, _ = ks_paths(m, ss, "x0")
    lin0 = hank.LinearizedFunction(x0.reshape(-1, order="F"), {"Z": Z}, m, ss, ss)
    # start from the linear solution so that the reference's undamped y-iteration converges fast
    xs = hank.NewtonRaphsonHANK(x0.reshape(-1, order="F"), J, {"Z": Z}, m, ss, ss, ε=1e-9)
    lin = hank.LinearizedFunction(xs, {"Z": Z}, m, ss, ss)
    assert np.linalg.norm(lin.Fx) < 1e-8 < np.linalg.norm(lin0.Fx)
    F_o, _ = orc.ks_jvp(xs.reshape(4, P, order="F"), np.zeros((4, P, 1)), Z, m.params.α, m.params.δ, ss.vars["KS"], ss.value, ss.D)
    assert np.max(np.abs(F_o - lin.Fx)) < 1e-10


def test_permanent_shock_between_two_steady_states(hank):
    """the reference YAML's two-steady-state scenario (`ending:` block, KrusellSmith.yaml:109-116): terminal value from
    the ENDING steady state, initial distribution and KS_0 from the INITIAL one (BackwardIteration.jl:85,
    ForwardIteration.jl:293, GeneralStructures.jl:329-377); Newton converges and the path joins the two."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from examples.solve_transition import solve_permanent
    out, x, ss_i, ss_e = solve_permanent(n_a=60, n_e=2, T=120, Z_end=1.02)
    assert out["residual_norm"] < 1e-8
    P = 119
    KS = x.reshape(4, P, order="F")[1]
    assert ss_e.vars["KS"] > ss_i.vars["KS"]
    # capital starts near the old steady state, rises monotonically, ends at the new one
    assert abs(KS[0] - ss_i.vars["KS"]) < 0.2 * (ss_e.vars["KS"] - ss_i.vars["KS"])
    assert np.all(np.diff(KS) > -1e-9)
    assert abs(KS[-1] - ss_e.vars["KS"]) < 2e-3 * ss_e.vars["KS"]
