"""The XCD-local persistent sweeps (csrc/hank_xsweep.h) forced on EVERY entry point (HANK_SCHEDULE=xcd) — in the default
schedule they only serve hank_primal and narrow hank_jvp batches — against the CPU oracle: ragged shapes (n_a not a
multiple of the 63-row slabs, one slab only, the 16-column block, batch widths that are not a multiple of the 8 groups or
need several passes), both value-function families, the error surface, and bit-reproducibility.
Tolerance: rel 1e-10 of the output scale + abs 1e-12."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import ks_paths, ks_setup

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def close(a, b, rel=1e-10, abs_=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(np.max(np.abs(b)), 1e-300)
    err = np.max(np.abs(a - b))
    assert err <= abs_ + rel * scale, f"max err {err:.3e} vs scale {scale:.3e}"


def forced_block(hank, m, sched="xcd"):
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    os.environ["HANK_SCHEDULE"] = sched
    try:
        return hank.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons,
                                   m.compspec.T, m.value_fn.value_fn_id)
    finally:
        os.environ.pop("HANK_SCHEDULE")


@pytest.mark.parametrize("n_a,n_e,T,N,shock", [(50, 2, 100, 3, 0.8), (37, 3, 9, 5, 0.05), (130, 3, 20, 9, 0.05), (500, 4, 300, 1, 0.01),
                                               (40, 16, 8, 6, 0.05), (37, 3, 9, 70, 0.05), (200, 7, 40, 32, 0.05)])
def test_forced_persistent_sweeps_match_the_oracle(hank, n_a, n_e, T, N, shock):
    from oracle.oracle import pad_N, SUPPORTED_N
    m, ss, orc = ks_setup(n_a, n_e, T)
    P = T - 1
    x, Z = ks_paths(m, ss, "x1", shock)
    y = np.random.default_rng(0).standard_normal((2, P, N))
    hb = forced_block(hank, m)
    hb.set_boundary(ss.value, ss.D)
    agg, dagg = hb.primal_jvp(x[2:4], y)
    st = hb.stats()
    assert st["schedule"] == 1 and st["fallbacks"] == 0 and st["sweep_launches"] >= 2      # (a one-pass Dual pass is two persistent launches)
    cols, pols = [], []
    for c0 in range(0, N, SUPPORTED_N[-1]):
        c1 = min(N, c0 + SUPPORTED_N[-1])
        Nc = pad_N(c1 - c0)
        xr = np.zeros((P, 1 + Nc)); xw = np.zeros((P, 1 + Nc))
        xr[:, 0], xw[:, 0] = x[2], x[3]
        xr[:, 1:1 + c1 - c0], xw[:, 1:1 + c1 - c0] = y[0][:, c0:c1], y[1][:, c0:c1]
        s_, oa, op = orc.household_block(xr, xw, ss.value, ss.D, Nc)
        assert s_ == 0
        o0, p0 = oa[:, 0], op[..., 0]
        cols.append(oa[:, 1:1 + c1 - c0]); pols.append(op[..., 1:1 + c1 - c0])
    close(agg, o0); close(dagg, np.concatenate(cols, axis=1))
    close(hb.policy_seq().transpose(2, 0, 1), p0)
    close(hb.dpolicy_seq(N).transpose(2, 0, 1, 3), np.concatenate(pols, axis=-1))
    D = hb.dist_seq()
    np.testing.assert_allclose(D.sum(axis=(0, 1)), 1.0, atol=1e-12)
    # one primal, then JVPs at its record — and again: bit-reproducible (fixed summation order; the LDS adds of one wave execute
    # in program order). The dual pass takes the term dpol_t . D_t of the aggregate at the target rows, the sweep at a recorded
    # primal at the source rows: the same sum in another order
    assert np.array_equal(hb.primal(x[2:4]), agg)
    again = hb.jvp(y)
    close(again, dagg, rel=1e-13)
    assert np.array_equal(hb.jvp(y), again)
    hb.close()


def test_forced_persistent_sweeps_one_asset_hank(hank):
    """the second value-function family (three household inputs, lump-sum transfer) through the persistent sweeps."""
    from examples.solve_hank import build
    from oracle.oracle import Oracle, pad_N
    m, ss = build(80, 3, 40)
    P, N = 39, 6
    t = np.arange(P)
    x = np.stack([ss.vars["r"] + 0.002 * 0.8 ** t, ss.vars["om"] * (1 + 0.01 * 0.7 ** t), ss.vars["Tr"] * (1 - 0.02 * 0.9 ** t)])
    y = np.random.default_rng(3).standard_normal((3, P, N))
    hb = forced_block(hank, m)
    assert hb.n_hh == 3
    hb.set_boundary(ss.value, ss.D)
    agg, dagg = hb.primal_jvp(x, y)
    wd, pdm = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    orc = Oracle(wd.grid, pdm.grid, pdm.transition, m.params.β, m.params.γ, m.params.borrow_cons)
    Nc = pad_N(N)
    xd = np.zeros((3, P, 1 + Nc)); xd[..., 0] = x; xd[..., 1:1 + N] = y
    st, oagg, opol = orc.household_block(xd[0], xd[1], ss.value, ss.D, Nc, xt=xd[2])
    assert st == 0
    close(agg, oagg[:, 0]); close(dagg, oagg[:, 1:1 + N])
    close(hb.dpolicy_seq(N).transpose(2, 0, 1, 3), opol[..., 1:1 + N])
    hb.close()


def test_forced_persistent_sweeps_error_surface(hank):
    """Interpolations' knot error comes out of the persistent primal sweep with the reference's meaning (period 1-based),
    a JVP without a valid primal is refused, and the context recovers."""
    m, ss, _ = ks_setup(50, 2, 100)
    hb = forced_block(hank, m)
    bad = np.array(ss.value, copy=True)
    bad[7, :] *= 1e-4
    hb.set_boundary(bad, ss.D)
    x, _ = ks_paths(m, ss, "x0")
    with pytest.raises(hank.KnotsNotSortedError, match="period 99"):
        hb.primal(x[2:4])
    with pytest.raises(hank.HankHIPError):
        hb.jvp(np.zeros((2, 99, 1)))
    hb.set_boundary(ss.value, ss.D)
    hb.primal(x[2:4])
    assert hb.jvp(np.ones((2, 99, 2))).shape == (99, 2)
    hb.close()


def test_persistent_sweeps_are_race_free_at_full_size(hank):
    """2000x11, T=300 — every CU of every XCD busy, uneven work per member (the lottery segments near the borrowing
    constraint are long): 25 repetitions of the tangent sweeps at one recorded primal, 32 directions and then a single one,
    must reproduce the first result bit for bit (a lost hand-off between workgroups shows up as a difference), and two
    columns agree with the oracle."""
    m, ss, orc = ks_setup(2000, 11, 300)
    P = 299
    x, Z = ks_paths(m, ss, "x1", 0.01)
    hb = forced_block(hank, m)
    hb.set_boundary(ss.value, ss.D)
    agg = hb.primal(x[2:4])
    y = np.random.default_rng(21).standard_normal((2, P, 32))
    first = hb.jvp(y)
    one = hb.jvp(y[:, :, 7:8])
    # a direction's result does not depend on its batch — to rounding: the forward sweep of a batch with four partials per group adds
    # neighbouring sources' parts for one tile row in ONE LDS add (pre-combine, round 5), the single-direction kernel keeps one add per part
    assert np.max(np.abs(one[:, 0] - first[:, 7])) <= 1e-13 * np.max(np.abs(first[:, 7]))
    for _ in range(25):
        assert np.array_equal(hb.jvp(y), first)
        assert np.array_equal(hb.jvp(y[:, :, 7:8]), one)
    for _ in range(5):
        assert np.array_equal(hb.primal(x[2:4]), agg)
    xr = np.zeros((P, 3)); xw = np.zeros((P, 3))
    xr[:, 0], xw[:, 0] = x[2], x[3]
    xr[:, 1:], xw[:, 1:] = y[0][:, [3, 30]], y[1][:, [3, 30]]
    st, oagg, _ = orc.household_block(xr, xw, ss.value, ss.D, 2)
    assert st == 0
    close(agg, oagg[:, 0]); close(first[:, [3, 30]], oagg[:, 1:])
    hb.close()


def test_persistent_dual_pass_is_race_free_at_full_size(hank, monkeypatch):
    """The default benched entry at its benched size (2000x11, T=300, N=32: k_xdual_back<4> + k_xfwd<4, true>, every group
    repeating the Float64 step and sharing the record arrays out for writing): 20 repetitions reproduce value, partials, policy
    and policy partials bit for bit, the record they leave serves a later hank_jvp, and everything equals the per-period launches
    (policies and partials bit for bit, aggregates to the order of their sums) and, for two columns, the oracle."""
    m, ss, orc = ks_setup(2000, 11, 300)
    P = 299
    x, Z = ks_paths(m, ss, "x1", 0.01)
    y = np.random.default_rng(22).standard_normal((2, P, 32))
    monkeypatch.setenv("HANK_PRIMAL_MEMO", "0")            # every call runs its Float64 sweep
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    hb = hank.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T,
                             m.value_fn.value_fn_id)       # a context of its own, default schedule
    monkeypatch.delenv("HANK_PRIMAL_MEMO", raising=False)
    hb.set_boundary(ss.value, ss.D)
    agg, dagg = hb.primal_jvp(x[2:4], y)
    assert hb.last_timings()["dual_backward"]["ms"] <= 0.0 and hb.stats()["schedule"] == 2      # the persistent Dual pass ran
    pol, dpol = hb.policy_seq(), hb.dpolicy_seq(32)
    for k in range(20):
        if k % 5 == 4:
            hb.primal_jvp(x[2:4] * 1.01, y)                 # another record in between
        a2, d2 = hb.primal_jvp(x[2:4], y)
        assert np.array_equal(a2, agg) and np.array_equal(d2, dagg)
    assert np.array_equal(hb.policy_seq(), pol) and np.array_equal(hb.dpolicy_seq(32), dpol)
    later = hb.jvp(y[:, :, 5:6])                            # the record serves the persistent tangent sweeps
    close(later[:, 0], dagg[:, 5], rel=1e-11)
    ref = forced_block(hank, m, "launch")
    ref.set_boundary(ss.value, ss.D)
    a0, d0 = ref.primal_jvp(x[2:4], y)
    assert np.array_equal(ref.policy_seq(), pol) and np.array_equal(ref.dpolicy_seq(32), dpol)
    close(agg, a0, rel=1e-12); close(dagg, d0, rel=1e-11)
    ref.close()
    xr = np.zeros((P, 3)); xw = np.zeros((P, 3))
    xr[:, 0], xw[:, 0] = x[2], x[3]
    xr[:, 1:], xw[:, 1:] = y[0][:, [0, 31]], y[1][:, [0, 31]]
    st, oagg, _ = orc.household_block(xr, xw, ss.value, ss.D, 2)
    assert st == 0
    close(agg, oagg[:, 0]); close(dagg[:, [0, 31]], oagg[:, 1:])
    assert hb.stats()["fallbacks"] == 0 and hb.stats()["primal_memo_hits"] == 0
    hb.close()
