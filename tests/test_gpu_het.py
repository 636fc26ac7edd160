"""More than one heterogeneous variable through the fused sweeps: the reference keeps one policy sequence per heterogeneous
variable (BackwardIteration.jl:99-112) and aggregates every one of them with the same D_t (ForwardIteration.jl:303-307). The
device reduces a second, grid-weighted dot product next to the policy-weighted one in every forward kernel family and
hank_get_het_outputs assembles (savings, consumption) from them. The reference ships no two-output plugin (parity unpinned by
construction): the checks are the CPU oracle's two-variable household block (orc_consumption_policy +
orc_forward_iteration_het: each variable's OWN policy dotted with D_t under the dual arithmetic), at rel 1e-10 + abs 1e-12 like
every sweep test, the distribution path itself, and the Newton solve of the goods-market-clearing model."""
import os

import numpy as np
import pytest

from conftest import ks_paths, ks_setup

pytestmark = pytest.mark.gpu


def _block(hank, m, schedule):
    old = os.environ.get("HANK_SCHEDULE")
    if schedule:
        os.environ["HANK_SCHEDULE"] = schedule
    else:
        os.environ.pop("HANK_SCHEDULE", None)
    try:
        wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
        return hank.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T,
                                   m.value_fn.value_fn_id)
    finally:
        if old is None:
            os.environ.pop("HANK_SCHEDULE", None)
        else:
            os.environ["HANK_SCHEDULE"] = old


def _close(a, b, rel=1e-10, ab=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    err = np.max(np.abs(a - b))
    assert err <= ab + rel * np.abs(b).max(), f"max err {err:.3e} vs scale {np.abs(b).max():.3e}"


def _oracle_two(orc, ss, x, y):
    """x (n_hh, P), y (n_hh, P, N) -> agg (2, P), dagg (2, P, N) of (savings, consumption)."""
    from oracle.oracle import pad_N
    n_hh, P, N = y.shape
    Nc = pad_N(N)
    xd = np.zeros((n_hh, P, 1 + Nc))
    xd[..., 0] = x
    xd[..., 1:1 + N] = y
    st, agg, _, _ = orc.household_block_het(xd[0], xd[1], ss.value, ss.D, Nc, xt=xd[2] if n_hh > 2 else None)
    assert st == 0
    return agg[..., 0], agg[..., 1:1 + N]


def _check(hb, x, y, oagg, odagg, family):
    N = y.shape[2]
    agg, dagg = hb.primal_jvp(x, y)
    assert hb.info()["last_tangent_family_name"] == family
    aggs, daggs = hb.het_outputs(2, y)
    assert np.array_equal(aggs[:, 0], agg) and np.array_equal(daggs[:, 0, :], dagg)        # output 0 IS the call's own result
    _close(aggs[:, 0], oagg[0]); _close(aggs[:, 1], oagg[1])
    _close(daggs[:, 0, :], odagg[0]); _close(daggs[:, 1, :], odagg[1])
    # hank_primal, then hank_jvp (a recorded primal): the same two columns
    hb.primal(x); hb.jvp(y)
    a2, d2 = hb.het_outputs(2, y)
    _close(a2, aggs, 1e-12); _close(d2, daggs, 1e-11)
    # the raw second reduction against the distribution path the device holds
    ad, dad = hb.grid_aggregates(N)
    Dseq = hb.dist_seq()                                        # (n_a, n_e, P)
    _close(ad, np.einsum("i,iet->t", hb_grid(hb), Dseq), 1e-12)
    # one output asked for: the policy variable alone; values only: no tangent input needed
    a1, d1 = hb.het_outputs(1, y)
    assert a1.shape == (hb.P, 1) and np.array_equal(a1[:, 0], a2[:, 0]) and np.array_equal(d1[:, 0, :], d2[:, 0, :])
    assert np.array_equal(hb.het_outputs(2)[0], a2)


def hb_grid(hb):
    return hb.a_grid


@pytest.mark.parametrize("schedule,family,N", [("launch", "launch-per-period", 5), ("xcd", "xcd-persistent", 5), ("xcd", "xcd-persistent", 40),
                                               ("wide", "on-chip-wide", 5), ("launch", "launch-per-period", 33)])
def test_two_outputs_krusell_smith_130x3(hank, schedule, family, N):
    m, ss, orc = ks_setup(130, 3, 40)
    P = 39
    x, _ = ks_paths(m, ss, "x1", 0.05)
    y = np.random.default_rng(3).standard_normal((2, P, N))
    k = min(N, 32)
    oagg, odagg = _oracle_two(orc, ss, x[2:4], y[:, :, :k])
    hb = _block(hank, m, schedule)
    hb.set_boundary(ss.value, ss.D)
    if N <= 32:
        _check(hb, x[2:4], y, oagg, odagg, family)
    else:                   # a batch of several passes / chunks: the first 32 columns against the oracle, linearity for the rest
        hb.primal_jvp(x[2:4], y)
        aggs, daggs = hb.het_outputs(2, y)
        _close(aggs[:, 1], oagg[1]); _close(daggs[:, 1, :k], odagg[1])
        hb.primal_jvp(x[2:4], y[:, :, k:])
        _close(hb.het_outputs(2, y[:, :, k:])[1], daggs[:, :, k:], 1e-11)
    hb.close()


@pytest.mark.parametrize("schedule,family", [("launch", "launch-per-period"), ("xcd", "xcd-persistent"), ("wide", "on-chip-wide")])
def test_two_outputs_one_asset_hank_1000x7_T500(hank, schedule, family):
    """BASELINE.json configs[4]'s shape: the HANK family (three household inputs: the transfer enters consumption directly)."""
    from examples.solve_hank import build
    from oracle.oracle import Oracle
    m, ss = _hank_1000x7()
    P = m.compspec.T - 1
    t = np.arange(P)
    x = np.stack([ss.vars["r"] + 0.002 * 0.8 ** t, ss.vars["om"] * (1 + 0.01 * 0.7 ** t), ss.vars["Tr"] * (1 - 0.02 * 0.9 ** t)])
    y = np.random.default_rng(5).standard_normal((3, P, 2))
    wd, pdm = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    orc = Oracle(wd.grid, pdm.grid, pdm.transition, m.params.β, m.params.γ, m.params.borrow_cons)
    key = "oracle"
    if key not in _CACHE:
        _CACHE[key] = _oracle_two(orc, ss, x, y)
    oagg, odagg = _CACHE[key]
    hb = _block(hank, m, schedule)
    hb.set_boundary(ss.value, ss.D)
    _check(hb, x, y, oagg, odagg, family)
    hb.close()


_CACHE = {}


def _hank_1000x7():
    if "model" not in _CACHE:
        from examples.solve_hank import build
        _CACHE["model"] = build(1000, 7, 500, "one_asset_hank_goods.yaml")
    return _CACHE["model"]


def test_device_pointer_form_equals_the_host_form(hank):
    import torch
    m, ss, _ = ks_setup(130, 3, 40)
    P, N = 39, 6
    x, _ = ks_paths(m, ss, "x1", 0.05)
    y = np.random.default_rng(4).standard_normal((2, P, N))
    hb = _block(hank, m, None)
    hb.set_boundary(ss.value, ss.D)
    hb.primal_jvp(x[2:4], y)
    aggs, daggs = hb.het_outputs(2, y)
    dev = torch.device("cuda", 0)
    d_y = torch.from_numpy(np.asfortranarray(y).reshape(-1, order="F").copy()).to(dev)
    d_a = torch.empty(2 * P, dtype=torch.float64, device=dev); d_d = torch.empty(2 * P * N, dtype=torch.float64, device=dev)
    hb.het_outputs_dev(2, d_y.data_ptr(), N, d_a.data_ptr(), d_d.data_ptr())
    hb.sync()
    assert np.array_equal(d_a.cpu().numpy().reshape(P, 2, order="F"), aggs)
    assert np.array_equal(d_d.cpu().numpy().reshape(P, 2, N, order="F"), daggs)
    # argument errors are statuses, not crashes
    with pytest.raises(hank.HankHIPError):
        hb.het_outputs(3, y)
    with pytest.raises(hank.HankHIPError):
        hb.het_outputs(2, y[:, :, :3])           # no tangent sweep of that width is current
    hb.close()


def test_goods_market_clearing_model_solves(hank):
    """examples/one_asset_hank_goods.yaml: savings AND consumption are heterogeneous variables, the last equation is goods-market
    clearing. Steady state, J̅ (unit-tangent columns: two outputs per JVP), Newton on a monetary shock; the fused path of
    ForwardIteration returns both aggregates and agrees with the generic one (explicit policy matrices of BOTH variables, one
    granular device step per period)."""
    from examples.solve_hank import solve
    import hank_amd as h
    out, x, m, ss = solve(80, 3, 60, shock=0.0025, spec="one_asset_hank_goods.yaml")
    assert out["residual_norm"] < 1e-8, out
    assert out["impact"]["Y"] < 0 and out["impact"]["infl"] < 0
    assert abs(ss.vars["C"] - ss.vars["Y"]) < 1e-8 and abs(ss.vars["A"] - m.params.B) < 1e-5
    P = m.compspec.T - 1
    ei = {"ei": 0.0025 * 0.6 ** np.arange(P)}
    seqs = h.BackwardIteration(x, ei, m, ss)
    fused = h.ForwardIteration(seqs, m, ss)
    assert set(fused) == {"A", "C"}
    generic = h.ForwardIteration({k: list(seqs[k]) for k in ("A", "C")}, m, ss)
    for k in ("A", "C"):
        _close(fused[k], generic[k], 1e-11)
    # goods-market clearing holds along the converged path
    keys = h.vars_of_type(m, "endogenous")
    X = x.reshape(len(keys), P, order="F")
    Y, infl = X[keys.index("Y")], X[keys.index("infl")]
    μ, κ = m.params.μ, m.params.κ
    _close(fused["C"], Y - (μ / (μ - 1) / (2 * κ)) * np.log(1 + infl) ** 2 * Y, 1e-8)


def test_dual_pass_of_the_two_variable_model_equals_the_linearised_one(hank):
    """JVP(fullFunction, x, y) under a Dual (BackwardIteration -> ForwardIteration -> Residuals, the reference's way) against
    LinearizedFunction.jvp (recorded primal + sparse residual layer) for the two-variable model."""
    from examples.solve_hank import build
    import hank_amd as h
    m, ss = build(80, 3, 40, "one_asset_hank_goods.yaml")
    P = 39
    keys = h.vars_of_type(m, "endogenous")
    rng = np.random.default_rng(12)
    x = np.tile(np.array([ss.vars[k] for k in keys]), P) * (1.0 + 1e-3 * rng.standard_normal(len(keys) * P))
    ei = {"ei": 0.0025 * 0.6 ** np.arange(P)}
    y = rng.standard_normal((len(x), 3))
    f = h.make_fullFunction(ei, m, ss, ss)
    Fx, Jy = f(x), h.JVP(f, x, y)
    lin = h.LinearizedFunction(x, ei, m, ss, ss)
    _close(lin.Fx, Fx, 1e-12)
    _close(lin.jvp(y), Jy, 1e-9, 1e-11)
