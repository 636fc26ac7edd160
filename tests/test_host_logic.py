"""Host-side mirror of the reference's API: parser, equation compiler, grids, shifts, duals."""
from pathlib import Path

import numpy as np
import pytest

REF_YAML = Path("/root/reference/KrusellSmith.yaml")


def test_reference_yaml_is_accepted_unchanged(hank):
    if not REF_YAML.exists():
        pytest.skip("reference checkout not present on this machine")
    m = hank.build_model_from_yaml(str(REF_YAML))
    cs = m.compspec
    # KrusellSmith.yaml:27-37, :44-60, :67-94 ; variable order endogenous -> heterogeneous -> exogenous
    assert (cs.T, cs.ε, cs.dx, cs.n_v, cs.n_endog, cs.max_lag, cs.max_lead) == (150, 1e-6, 0.001, 6, 4, 1, 0)
    assert tuple(m.variables) == ("Y", "KS", "r", "w", "KD", "Z")
    assert m.heterogeneity["wealth"].n == 200 and m.heterogeneity["productivity"].n == 7
    assert m.heterogeneity["wealth"].policy_var == "KD"
    assert m.ss_initial.fixed == {"Z": 1.0} and m.ss_ending.fixed == {"Z": 2.0}
    assert m.ss_initial is not m.ss_ending
    assert m.value_fn.name == "ValueFunction"
    # test_Model.jl:84-92: residual vector length on an all-ones padded matrix
    T_pad = (cs.T - 1) + cs.max_lag + cs.max_lead
    assert m.residuals_fn(np.ones((cs.n_v, T_pad)), m.params).shape == (4 * (cs.T - 1),)


def test_single_ss_block_means_transitory_shock(hank):
    m = hank.build_model_from_yaml(str(Path(__file__).parent.parent / "examples" / "krusell_smith.yaml"))
    assert m.ss_initial is m.ss_ending          # ModelParser.jl:372-373


def test_detect_lag_lead_and_compile(hank):
    eqs = ["C(+2) = r(-3) * Y", "Y = Z * KS(-1)^α", "r = log(Y) - exp(KS(0)) + sqrt(Z)"]
    names = ("C", "r", "Y", "KS", "Z")
    assert hank.detect_max_lag_lead(eqs, names) == (3, 2)
    fn = hank.compile_residuals(eqs, names, {"α"})
    from types import SimpleNamespace
    rng = np.random.default_rng(0)
    P = 7
    X = rng.uniform(0.5, 2.0, (5, P + 5))
    out = fn(X, SimpleNamespace(α=0.36)).reshape(3, P, order="F")
    C, r, Y, KS, Z = X
    lo, hi = 3, P + 3
    sl = hank.shift_lag; sd = hank.shift_lead
    np.testing.assert_allclose(out[0], (sd(C, 2) - sl(r, 3) * Y)[lo:hi])
    np.testing.assert_allclose(out[1], (Y - Z * sl(KS, 1) ** 0.36)[lo:hi])
    np.testing.assert_allclose(out[2], (r - (np.log(Y) - np.exp(KS) + np.sqrt(Z)))[lo:hi])
    with pytest.raises(ValueError):
        hank.compile_residuals(["Y + Z"], names, set())


def test_shift_operators(hank):
    x = np.arange(1.0, 6.0)
    assert np.array_equal(hank.shift_lag(x, 2), [1, 1, 1, 2, 3])      # GeneralStructures.jl:441-443
    assert np.array_equal(hank.shift_lead(x, 2), [3, 4, 5, 5, 5])     # :453-455
    d = hank.Dual(x, np.arange(10.0).reshape(5, 2))
    assert np.array_equal(hank.shift_lag(d, 1).p[:, 1], [1, 1, 3, 5, 7])


def test_grids(hank):
    g = hank.make_DoubleExponentialGrid(0.0, 200.0, 200)
    assert g[0] == 0.0 and abs(g[-1] - 200.0) < 1e-9 and np.all(np.diff(g) > 0)
    u = np.log(1 + np.log(1 + g))                                      # GeneralStructures.jl:474-483
    np.testing.assert_allclose(np.diff(u), u[1] - u[0], rtol=1e-9)
    Π, D, z = hank.get_RouwenhorstDiscretization(7, 0.966, 0.283)
    np.testing.assert_allclose(Π.sum(axis=1), 1.0, atol=1e-14)
    np.testing.assert_allclose(D @ Π, D, atol=1e-13)
    assert abs(np.sum(z * D) - 1.0) < 1e-14                            # E[z] = 1 (:520-522)
    assert abs(np.log(z[1] / z[0]) - 2 * 0.283 / np.sqrt(6)) < 1e-12
    # n = 2 base case and persistence p = (1+ρ)/2
    Π2, _, _ = hank.get_RouwenhorstDiscretization(2, 0.5, 0.1)
    np.testing.assert_allclose(Π2, [[0.75, 0.25], [0.25, 0.75]])


def test_invariant_dist_both_methods(hank):
    rng = np.random.default_rng(0)
    A = rng.uniform(0.1, 1, (30, 30)); A /= A.sum(axis=1, keepdims=True)
    d1 = hank.invariant_dist(A)
    d2 = hank.invariant_dist(A, direct_max=0)
    np.testing.assert_allclose(d1 @ A, d1, atol=1e-13)
    np.testing.assert_allclose(d1, d2, atol=1e-12)


def test_young_lottery_host(hank):
    """make_endogenous_transition: column-stochastic, clamps, searchsortedfirst ties (ForwardIteration.jl:37-78)."""
    from types import SimpleNamespace
    grid = np.array([0.0, 1.0, 2.0, 4.0])
    dim = SimpleNamespace(n=4, grid=grid)
    pol = np.array([[-1.0, 0.0], [0.5, 1.0], [3.0, 4.0], [5.0, 2.0]])
    L = hank.make_endogenous_transition(pol, dim, 2).toarray()
    np.testing.assert_allclose(L.sum(axis=0), 1.0)
    assert L[0, 0] == 1.0 and L[4, 4] == 1.0                 # below / at grid[1] -> first point
    assert L[0, 1] == 0.5 and L[1, 1] == 0.5                 # interior
    assert L[4, 5] == 0.0 and L[5, 5] == 1.0                 # p == grid[2]: m=2, w = 1 on the upper node
    assert L[2, 2] == 0.5 and L[3, 2] == 0.5
    assert L[3, 3] == 1.0 and L[7, 6] == 1.0                 # above / at the top
    assert np.all(L[:4, 4:] == 0) and np.all(L[4:, :4] == 0)  # block diagonal in e


def test_host_dual_matches_oracle_rules(hank, oracle_mod):
    rng = np.random.default_rng(5)
    D = hank.Dual
    for _ in range(50):
        x = np.concatenate([[rng.uniform(0.2, 3)], rng.standard_normal(3)])
        y = np.concatenate([[rng.uniform(0.2, 3)], rng.standard_normal(3)])
        dx, dy = D(np.array(x[0]), x[1:]), D(np.array(y[0]), y[1:])
        cases = [(0, dx + dy), (1, dx - dy), (2, dx * dy), (3, dx / dy), (4, dx * y[0]), (5, x[0] - dy), (6, x[0] / dy),
                 (7, dx ** y[0])]
        for op, got in cases:
            exp = oracle_mod.dual_binop(op, x, y, 3)
            np.testing.assert_allclose(np.concatenate([[got.v], got.p]), exp, rtol=1e-15, atol=0)


def test_residuals_under_duals_match_oracle_ks(ks_small, hank):
    """compiled YAML equations on Dual matrices == the oracle's hand-written KS residual duals."""
    m, ss, orc = ks_small
    P = m.compspec.T - 1
    rng = np.random.default_rng(4)
    N = 3
    x = np.tile(np.array([ss.vars[k] for k in ("Y", "KS", "r", "w")])[:, None], (1, P)) * rng.uniform(0.9, 1.1, (4, P))
    y = rng.standard_normal((4, P, N))
    Z = 1 + 0.1 * rng.standard_normal(P)
    xd = np.zeros((4, P, 1 + 4)); xd[..., 0] = x; xd[..., 1:1 + N] = y
    st, F, agg = orc.ks_full_function(xd, Z, m.params.α, m.params.δ, ss.vars["KS"], ss.value, ss.D, 4)
    assert st == 0
    xdual = hank.Dual(x.reshape(-1, order="F"), y.reshape(4 * P, N, order="F"))
    aggd = {"KD": hank.Dual(agg[:, 0], agg[:, 1:1 + N])}
    xMat = hank.assemble_full_xMat(xdual, aggd, {"Z": Z}, m, ss, ss)
    assert xMat.shape == (6, P + 1)
    res = hank.Residuals(xMat, m)
    np.testing.assert_allclose(res.v, F[..., 0].reshape(-1, order="F"), rtol=1e-14, atol=1e-14)
    np.testing.assert_allclose(res.p, F[..., 1:1 + N].reshape(4 * P, N, order="F"), rtol=1e-13, atol=1e-13)


def test_flatten_unflatten_roundtrip():
    """test_SteadyState.jl:93-141: policy sequences flatten/unflatten are exact inverses."""
    rng = np.random.default_rng(0)
    P, shape = 9, (5, 3)
    seqs = {"KD": [rng.standard_normal(shape) for _ in range(P)]}
    flat = np.concatenate([np.concatenate([m.reshape(-1, order="F") for m in s]) for s in seqs.values()])
    Tv = shape[0] * shape[1]
    back = [flat[i * Tv:(i + 1) * Tv].reshape(shape, order="F") for i in range(P)]
    assert len(flat) == Tv * P and all(np.array_equal(a, b) for a, b in zip(seqs["KD"], back))


def test_household_jacobian_recursion_is_the_closed_form():
    """J[t, s] = [t <= s] Dv[s - t] + sum_{tau <= min(t, s)} F[t - tau, s - tau]  (SteadyStateJacobian.jl:363-371)"""
    from hank_amd.SteadyStateJacobian import household_jacobian
    rng = np.random.default_rng(5)
    P, K = 9, 2
    F, Dv = rng.standard_normal((P, P, K)), rng.standard_normal((P, K))
    J = household_jacobian(F, Dv)
    for k in range(K):
        for t in range(P):
            for s in range(P):
                ref = (Dv[s - t, k] if t <= s else 0.0) + sum(F[t - tau, s - tau, k] for tau in range(min(t, s) + 1))
                assert abs(J[k, t, s] - ref) < 1e-12


def test_direct_blocks_reproduce_the_residual_layer():
    """the equations' own Jacobian blocks (SteadyStateJacobian.jl:124-145) placed on the block diagonals equal the JVP of
    assemble_full_xMat + Residuals with the aggregates held fixed, column by column"""
    import hank_amd as h
    from hank_amd.Aggregation import Residuals
    from hank_amd.dual import Dual
    from hank_amd.GeneralStructures import assemble_full_xMat, var_names, vars_of_type
    from hank_amd.SteadyStateJacobian import direct_blocks
    from tests.conftest import ks_setup  # noqa: F401
    m, ss, _ = ks_setup(50, 2, 12)
    cs = m.compspec
    P, n_endog, n_eq = cs.T - 1, cs.n_endog, len(m.equations)
    keys, ek = var_names(m), vars_of_type(m, "endogenous")
    B = direct_blocks(m, ss)
    J4 = np.zeros((P, n_eq, P, n_endog))
    tt = np.arange(P)
    for o, Bo in B.items():
        ok = (tt + o >= 0) & (tt + o < P)
        for j, k in enumerate(ek):
            J4[tt[ok], :, tt[ok] + o, j] += Bo[:, keys.index(k)][None, :]
    J = J4.reshape(P * n_eq, P * n_endog)
    n = n_endog * P
    x = np.tile(np.array([ss.vars[k] for k in ek]), P)
    xd = Dual.seed(x, np.eye(n))
    exog = {k: np.full(P, float(ss.vars[k])) for k in vars_of_type(m, "exogenous")}
    agg = {k: np.full(P, float(ss.vars[k])) for k in vars_of_type(m, "heterogeneous")}
    res = Residuals(assemble_full_xMat(xd, agg, exog, m, ss, ss), m)
    assert np.max(np.abs(res.p - J)) < 1e-12


def test_device_group_shards_columns_contiguously():
    """one process, one context per GPU (parallel.DeviceGroup over hank_create_on): the column blocks of shard_bounds, in
    order, assembled on the host — with stand-in contexts (no GPU here)."""
    from hank_amd.parallel import DeviceGroup, shard_bounds

    class Fake:
        def __init__(self, device=0):
            self.device, self.seen = device, []

        def clone(self, device=None):
            return Fake(device)

        def set_boundary(self, v, D):
            self.b = (v, D)

        def primal(self, x):
            return np.asarray(x).sum(axis=0)

        def jvp(self, y):
            self.seen.append(y.shape[2])
            return y[0] * (1 + self.device)            # (P, k), tagged by the device that computed it

        def close(self):
            pass

    g = DeviceGroup(Fake(0), [0, 1, 2])
    assert [b.device for b in g.blocks] == [0, 1, 2]
    g.set_boundary(1.0, 2.0)
    y = np.arange(2 * 5 * 8, dtype=float).reshape(2, 5, 8)
    out = g.jvp(y)
    assert out.shape == (5, 8)
    for r in range(3):
        lo, hi = shard_bounds(8, 3, r)
        assert np.array_equal(out[:, lo:hi], y[0][:, lo:hi] * (1 + r))
    assert [b.seen for b in g.blocks] == [[3], [3], [2]]
    g.close()
