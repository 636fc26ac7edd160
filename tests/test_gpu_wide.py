"""The on-chip wide sweeps (csrc/hank_wide.h: one workgroup per tangent direction, the loop-carried state in registers) through the
C ABI: the partials of BackwardIteration.jl:90-113 / ForwardIteration.jl:297-308 against the CPU oracle (rel 1e-10 + abs 1e-12, the
tolerance of every sweep test) and against the per-period launches, at ragged shapes, both value-function families, odd grids,
batches wider than the chip, and the schedule's own choice."""
import os

import numpy as np
import pytest

from conftest import ks_paths, ks_setup

pytestmark = pytest.mark.gpu


def _block(hank, m, schedule, **env):
    keys = ("HANK_SCHEDULE", *env)
    old = {k: os.environ.get(k) for k in keys}
    if schedule:
        os.environ["HANK_SCHEDULE"] = schedule
    else:
        os.environ.pop("HANK_SCHEDULE", None)
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
        return hank.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T,
                                   m.value_fn.value_fn_id)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _oracle(orc, ss, x, y, k):
    from oracle.oracle import pad_N
    P, Nc = y.shape[1], pad_N(k)
    xr = np.zeros((P, 1 + Nc)); xw = np.zeros((P, 1 + Nc))
    xr[:, 0], xw[:, 0] = x[0], x[1]
    xr[:, 1:1 + k], xw[:, 1:1 + k] = y[0][:, :k], y[1][:, :k]
    st, oagg, opol = orc.household_block(xr, xw, ss.value, ss.D, Nc)
    assert st == 0
    return oagg, opol


def _close(a, b, rel=1e-10, ab=1e-12):
    return np.max(np.abs(a - b)) <= ab + rel * np.abs(b).max()


@pytest.mark.parametrize("n_a,n_e,T,N", [(50, 2, 20, 4), (130, 3, 20, 5), (30, 3, 25, 1), (50, 2, 60, 33), (500, 4, 300, 8)])
def test_wide_sweeps_against_the_oracle_and_the_launches(hank, n_a, n_e, T, N):
    m, ss, orc = ks_setup(n_a, n_e, T)
    P = T - 1
    x, _ = ks_paths(m, ss, "x1", 0.05)
    y = np.random.default_rng(0).standard_normal((2, P, N))
    k = min(N, 32)
    oagg, opol = _oracle(orc, ss, x[2:4], y, k)
    hb = _block(hank, m, "wide")
    hb.set_boundary(ss.value, ss.D)
    agg, dagg = hb.primal_jvp(x[2:4], y)
    assert hb.info()["last_tangent_family_name"] == "on-chip-wide"
    dpol = hb.dpolicy_seq(N)
    assert _close(agg, oagg[:, 0]) and _close(dagg[:, :k], oagg[:, 1:1 + k])
    assert _close(dpol.transpose(2, 0, 1, 3)[..., :k], opol[..., 1:1 + k])
    assert np.array_equal(hb.jvp(y), dagg)                      # fixed summation order: bit for bit
    hl = _block(hank, m, "launch")
    hl.set_boundary(ss.value, ss.D)
    aggl, daggl = hl.primal_jvp(x[2:4], y)
    assert _close(dagg, daggl, 1e-12) and _close(dpol, hl.dpolicy_seq(N), 1e-12)
    hb.close(); hl.close()


def test_wide_sweeps_odd_grid_and_a_batch_wider_than_the_chip(hank):
    """n_a odd (a half-filled row pair at the top of the grid) and more directions than CUs (workgroups are independent: the
    hardware runs them in rounds); linearity in the tangent."""
    m, ss, orc = ks_setup(51, 3, 16)
    P, N = 15, 300
    x, _ = ks_paths(m, ss, "x1", 0.05)
    y = np.random.default_rng(3).standard_normal((2, P, N))
    hb = _block(hank, m, "wide")
    hb.set_boundary(ss.value, ss.D)
    agg, dagg = hb.primal_jvp(x[2:4], y)
    oagg, opol = _oracle(orc, ss, x[2:4], y, 8)
    assert _close(agg, oagg[:, 0]) and _close(dagg[:, :8], oagg[:, 1:9])
    assert _close(hb.dpolicy_seq(N).transpose(2, 0, 1, 3)[..., :8], opol[..., 1:9])
    hl = _block(hank, m, "launch")
    hl.set_boundary(ss.value, ss.D)
    assert _close(dagg, hl.primal_jvp(x[2:4], y)[1], 1e-12)
    c = np.random.default_rng(4).standard_normal(N)
    comb = hb.jvp(np.tensordot(y, c, axes=([2], [0]))[:, :, None])[:, 0]
    assert _close(comb, dagg @ c, 1e-9)
    hb.close(); hl.close()


def test_wide_sweeps_one_asset_hank_family(hank):
    """three household inputs (r, w, transfer): the wide sweeps against the launches and the oracle's restatement of the family."""
    from examples.solve_hank import build
    from oracle.oracle import Oracle
    m, ss = build(130, 3, 30)
    P, N = 29, 6
    t = np.arange(P)
    x = np.stack([ss.vars["r"] + 0.002 * 0.8 ** t, ss.vars["om"] * (1 + 0.01 * 0.7 ** t), ss.vars["Tr"] * (1 - 0.02 * 0.9 ** t)])
    y = np.random.default_rng(5).standard_normal((3, P, N))
    hb = _block(hank, m, "wide")
    hb.set_boundary(ss.value, ss.D)
    agg, dagg = hb.primal_jvp(x, y)
    assert hb.info()["last_tangent_family_name"] == "on-chip-wide"
    hl = _block(hank, m, "launch")
    hl.set_boundary(ss.value, ss.D)
    aggl, daggl = hl.primal_jvp(x, y)
    assert _close(agg, aggl, 1e-12) and _close(dagg, daggl, 1e-12)
    wd, pdm = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    orc = Oracle(wd.grid, pdm.grid, pdm.transition, m.params.β, m.params.γ, m.params.borrow_cons)
    xr = np.zeros((P, 1 + N)); xw = np.zeros((P, 1 + N)); xt = np.zeros((P, 1 + N))
    xr[:, 0], xw[:, 0], xt[:, 0] = x
    xr[:, 1:], xw[:, 1:], xt[:, 1:] = y
    from oracle.oracle import pad_N
    assert pad_N(N) >= N
    Nc = pad_N(N)
    pad = lambda a: np.concatenate([a, np.zeros((P, 1 + Nc - a.shape[1]))], axis=1)      # noqa: E731
    st, oagg, _ = orc.household_block(pad(xr), pad(xw), ss.value, ss.D, Nc, xt=pad(xt))
    assert st == 0
    assert _close(dagg, oagg[:, 1:1 + N])
    hb.close(); hl.close()


def test_default_schedule_sends_full_rounds_to_the_wide_sweeps(hank):
    """auto: a batch of at least 80 directions goes to the on-chip wide sweeps (a round of them costs what 80 directions cost the
    per-period launches since the L2 warming of round 5: DESIGN.md section 4), a narrower one does not; a forced schedule keeps its
    one implementation."""
    m, ss, _ = ks_setup(50, 2, 20)
    P = 19
    x, _ = ks_paths(m, ss, "x1", 0.05)
    hb = _block(hank, m, None)
    hb.set_boundary(ss.value, ss.D)
    info = hb.info()
    assert info["wide_supported"] == 1 and info["wide_mode"] == 1
    for N, fam in ((256, "on-chip-wide"), (32, "xcd-persistent"), (72, "launch-per-period"), (120, "on-chip-wide"), (300, "on-chip-wide"), (512, "on-chip-wide")):
        y = np.random.default_rng(N).standard_normal((2, P, N))
        agg, dagg = hb.primal_jvp(x[2:4], y)
        assert hb.info()["last_tangent_family_name"] == fam, (N, hb.info())
        if N == 256:
            keep = (y, dagg)
    # the same batch on the launches: equal to rounding
    hl = _block(hank, m, "launch")
    hl.set_boundary(ss.value, ss.D)
    assert hl.info()["wide_mode"] == 0
    assert _close(keep[1], hl.primal_jvp(x[2:4], keep[0])[1], 1e-12)
    hb.close(); hl.close()


def test_wide_sweeps_refuse_what_they_cannot_hold(hank):
    """a horizon whose per-period inputs do not fit LDS, or a productivity grid that is not instantiated: auto keeps the other
    families, a forced `wide` fails at hank_create with the reason."""
    import hank_amd as h
    g = np.linspace(0.0, 50.0, 40)
    Pi6 = np.full((6, 6), 1.0 / 6.0)
    old = os.environ.get("HANK_SCHEDULE")
    try:
        os.environ.pop("HANK_SCHEDULE", None)
        hb = h.HouseholdBlock(g, np.linspace(0.5, 1.5, 6), Pi6, 0.98, 2.0, 0.0, 12)
        assert hb.info()["wide_supported"] == 0 and hb.info()["wide_mode"] == 0
        hb.close()
        os.environ["HANK_SCHEDULE"] = "wide"
        with pytest.raises(h.HankHIPError, match="wide"):
            h.HouseholdBlock(g, np.linspace(0.5, 1.5, 6), Pi6, 0.98, 2.0, 0.0, 12)
    finally:
        if old is None:
            os.environ.pop("HANK_SCHEDULE", None)
        else:
            os.environ["HANK_SCHEDULE"] = old
