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


def test_four_argument_backward_iteration_issues_one_sweep(hank):
    """the reference's own call pair (NewtonRaphson.jl:78-79): BackwardIteration with FOUR positionals, then
    ForwardIteration — ONE fused device sweep in all (the device call is deferred until ss_initial is known), for a
    Float64 pass and for a Dual pass; reading a policy matrix first still works (placeholder D_0, then one redo)."""
    m, ss, orc = ks_setup(50, 2, 100)
    P = 99
    x, Z = ks_paths(m, ss, "x1", 0.05)
    xv = x.reshape(-1, order="F")
    hb = hank.household_block(m)
    before = dict(hb.calls)
    seqs = hank.BackwardIteration(xv, {"Z": Z}, m, ss)
    assert hb.calls == before                                   # nothing ran yet
    agg = hank.ForwardIteration(seqs, m, ss)["KD"]
    assert hb.calls["primal"] == before["primal"] + 1 and hb.calls["jvp"] == before["jvp"] and hb.calls["primal_jvp"] == before["primal_jvp"]
    assert len(seqs["KD"]) == P                                 # the record is still this call's: no new sweep
    assert hb.calls["primal"] == before["primal"] + 1
    y = np.random.default_rng(3).standard_normal((4 * P, 2))
    xd = hank.Dual.seed(xv, y)
    before = dict(hb.calls)
    dseqs = hank.BackwardIteration(xd, {"Z": Z}, m, ss)
    dagg = hank.ForwardIteration(dseqs, m, ss)["KD"]
    assert hb.calls == {"primal": before["primal"], "jvp": before["jvp"], "primal_jvp": before["primal_jvp"] + 1}
    assert np.max(np.abs(dagg.v - agg)) <= 1e-13 * np.abs(agg).max()     # Dual pass (launches) vs Float64 pass (persistent sweeps): rounding
    # the five-argument form gives the same numbers
    ref = hank.ForwardIteration(hank.BackwardIteration(xd, {"Z": Z}, m, ss, ss_initial=ss), m, ss)["KD"]
    assert np.array_equal(ref.v, dagg.v) and np.array_equal(ref.p, dagg.p)
    # policies read BEFORE ForwardIteration: sweep with a placeholder D_0, redone once with the right one
    s2 = hank.BackwardIteration(xv, {"Z": Z}, m, ss)
    pol_first = np.array(s2["KD"][0])
    agg2 = hank.ForwardIteration(s2, m, ss)["KD"]
    assert np.array_equal(agg2, agg) and np.array_equal(pol_first, seqs["KD"][0])


def test_two_live_linearisations_do_not_mix(hank):
    """two LinearizedFunction objects at different x share the model's device context: a jvp on the older one must
    not silently use the newer one's record (it restores its own primal first)."""
    m, ss, orc = ks_setup(50, 2, 100)
    P = 99
    xa, Z = ks_paths(m, ss, "x1", 0.05)
    xb, _ = ks_paths(m, ss, "x1", 0.3)
    y = np.random.default_rng(4).standard_normal(4 * P)
    la = hank.LinearizedFunction(xa.reshape(-1, order="F"), {"Z": Z}, m, ss, ss)
    ja = la.jvp(y)
    lb = hank.LinearizedFunction(xb.reshape(-1, order="F"), {"Z": Z}, m, ss, ss)
    jb = lb.jvp(y)
    assert np.max(np.abs(ja - jb)) > 1e-6                       # genuinely different linearisations
    assert np.array_equal(la.jvp(y), ja)                        # la re-records its own primal
    assert np.array_equal(lb.jvp(y), jb)
    _, J_o = orc.ks_jvp(xa, y.reshape(4, P, 1, order="F"), Z, m.params.α, m.params.δ, ss.vars["KS"], ss.value, ss.D)
    assert np.max(np.abs(la.jvp(y) - J_o[:, 0])) < 1e-10 * np.abs(J_o).max()


def test_newton_with_the_reference_gmres_inner_solves(hank):
    """the faithful inner loop (NewtonRaphson.jl:97-98: two warm-started restarted GMRES solves per inner iteration,
    IterativeSolvers defaults restart = min(20, n), reltol = sqrt(eps)) reaches the same converged path as the
    LU branch this package defaults to, within 1e-8."""
    m, ss, orc = ks_setup(50, 2, 60)
    P = 59
    Z = 1.0 + 0.01 * 0.8 ** np.arange(1, P + 1)
    J = hank.getSteadyStateJacobian(ss, m, method="columns")
    x0, _ = ks_paths(m, ss, "x0")
    x_lu = hank.NewtonRaphsonHANK(x0.reshape(-1, order="F"), J, {"Z": Z}, m, ss, ss, ε=1e-9)
    it_lu = hank.NewtonRaphsonHANK.iterations
    x_gm = hank.NewtonRaphsonHANK(x0.reshape(-1, order="F"), J, {"Z": Z}, m, ss, ss, ε=1e-9, linear_solver="gmres")
    assert np.max(np.abs(x_gm - x_lu)) < 1e-8
    lin = hank.LinearizedFunction(x_gm, {"Z": Z}, m, ss, ss)
    assert np.linalg.norm(lin.Fx) < 1e-8
    # (the loosely converged, warm-started GMRES solves make the count of outer steps sensitive to the last digits of J̅:
    # 5-7 steps with J̅ from the launched sweeps, 12 with the persistent ones forced everywhere, 19 with the Toeplitz J̅ — equal to
    # the unit-tangent one to 1e-9; the LU branch takes 4 with all of them)
    assert hank.NewtonRaphsonHANK.iterations <= it_lu + 20


def test_reference_shaped_y_iteration_reuses_the_primal(hank, monkeypatch):
    """NewtonRaphson.jl:91-95 calls JVP(fullFunction, x, y) for ~21 different y at ONE x and the reference's Dual pass recomputes
    the primal every time. Through the reference's own signatures (the closure of :77-83, nothing about a primal in it) the
    library runs ONE primal sweep and 20 tangent sweeps: the host-pointer hank_primal_jvp recognises the x on record. Results
    equal the un-memoised ones (HANK_PRIMAL_MEMO=0) to the rounding of the aggregate sums in the default schedule, and bit for
    bit where one implementation serves both (HANK_SCHEDULE=launch)."""
    from hank_amd.BackwardIteration import household_block
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
    Ys = np.random.default_rng(7).standard_normal((20, 4 * P))
    out = {}
    for sched, memo in ((None, "1"), (None, "0"), ("launch", "1"), ("launch", "0")):
        monkeypatch.setenv("HANK_PRIMAL_MEMO", memo)
        if sched:
            monkeypatch.setenv("HANK_SCHEDULE", sched)
        else:
            monkeypatch.delenv("HANK_SCHEDULE", raising=False)
        if m._hip_block is not None:          # (the knobs are read at hank_create: a fresh context per round)
            m._hip_block.close()
            m._hip_block = None
        hb = household_block(m)
        st0 = hb.stats()
        res = [hank.JVP(fullFunction, xv, y) for y in Ys]
        st1 = hb.stats()
        primal = st1["primal_sweeps"] - st0["primal_sweeps"]
        hits = st1["primal_memo_hits"] - st0["primal_memo_hits"]
        if memo == "1":
            assert primal <= 1 and hits >= 19, (primal, hits)          # (<= 1: the x may already be on record from the previous round)
        else:
            assert primal == 20 and hits == 0, (primal, hits)
        out[(sched, memo)] = np.array(res)
    monkeypatch.delenv("HANK_PRIMAL_MEMO", raising=False)
    monkeypatch.delenv("HANK_SCHEDULE", raising=False)
    m._hip_block.close()
    m._hip_block = None                       # (later tests get a context with the default knobs)
    assert np.array_equal(out[("launch", "1")], out[("launch", "0")])
    scale = np.abs(out[(None, "0")]).max()
    assert np.max(np.abs(out[(None, "1")] - out[(None, "0")])) <= 1e-12 * scale
    F_o, J_o = orc.ks_jvp(x, Ys.T.reshape(4, P, 20, order="F")[:, :, :3], Z, m.params.α, m.params.δ, ss.vars["KS"], ss.value, ss.D)
    assert np.max(np.abs(out[(None, "1")][:3].T - J_o)) < 1e-10 * np.abs(J_o).max()
