"""Sweep-level parity on the MI355X (hank_primal / hank_jvp through the C ABI) vs the CPU oracle,
on BASELINE.json configs 1 and 2 (+ a batched variant), the committed golden vectors, and the
Julia-exception surface. Tolerance: rel 1e-10 of the output scale + abs 1e-12."""
from pathlib import Path

import numpy as np
import pytest

from conftest import ks_paths, ks_setup

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def close(a, b, rel=1e-10, abs_=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(np.max(np.abs(b)), 1e-300)
    err = np.max(np.abs(a - b))
    assert err <= abs_ + rel * scale, f"max err {err:.3e} vs scale {scale:.3e}"


def run_case(hank, n_a, n_e, T, N, kind, shock, check_policies=True):
    m, ss, orc = ks_setup(n_a, n_e, T)
    P = T - 1
    x, Z = ks_paths(m, ss, kind, shock)
    rng = np.random.default_rng(0)
    y = rng.standard_normal((4, P, N))
    hb = hank.household_block(m)
    hb.set_boundary(ss.value, ss.D)
    agg = hb.primal(x[2:4])
    dagg = hb.jvp(y[2:4])
    from oracle.oracle import pad_N, SUPPORTED_N
    oagg_cols, opol_cols = [], []
    for c0 in range(0, N, SUPPORTED_N[-1]):            # the oracle carries at most 32 partials per pass
        c1 = min(N, c0 + SUPPORTED_N[-1])
        Nc = pad_N(c1 - c0)
        xr = np.zeros((P, 1 + Nc)); xw = np.zeros((P, 1 + Nc))
        xr[:, 0], xw[:, 0] = x[2], x[3]
        xr[:, 1:1 + c1 - c0], xw[:, 1:1 + c1 - c0] = y[2][:, c0:c1], y[3][:, c0:c1]
        st, oa, op = orc.household_block(xr, xw, ss.value, ss.D, Nc)
        assert st == 0
        oagg0, opol0 = oa[:, 0], op[..., 0]
        oagg_cols.append(oa[:, 1:1 + c1 - c0]); opol_cols.append(op[..., 1:1 + c1 - c0])
    oagg = np.concatenate([oagg0[:, None]] + oagg_cols, axis=1)
    opol = np.concatenate([opol0[..., None]] + opol_cols, axis=-1)
    close(agg, oagg[:, 0]); close(dagg, oagg[:, 1:])
    if check_policies:
        close(hb.policy_seq().transpose(2, 0, 1), opol[..., 0])
        close(hb.dpolicy_seq(N).transpose(2, 0, 1, 3), opol[..., 1:])
        D = hb.dist_seq()
        np.testing.assert_allclose(D.sum(axis=(0, 1)), 1.0, atol=1e-12)
    return m, ss, orc, x, Z, y


@pytest.mark.parametrize("kind,shock,N", [("x0", 0.0, 1), ("x1", 0.05, 1), ("x1", 0.8, 3), ("x1", 0.01, 32)])
def test_config1_50x2_T100(hank, kind, shock, N):
    """configs[0]: 50x2, T=100 (incl. the RunMain shock Z_t = 1 + 0.8^t, RunMain.jl:22-23)."""
    run_case(hank, 50, 2, 100, N, kind, shock)


@pytest.mark.parametrize("N", [1, 4])
def test_config2_500x4_T300(hank, N):
    """configs[1]: 500x4, T=300, single tangent (and a small batch)."""
    run_case(hank, 500, 4, 300, N, "x1", 0.01)


def test_odd_sizes_and_nonpow2_batch(hank):
    """ragged shapes: n_a not a multiple of any block size, n_e = 3, N = 5, N = 70 (> one chunk) and
    N = 130 (the wide-batch geometry of the forward kernel: two row groups per wave from N = 128 on)."""
    run_case(hank, 37, 3, 9, 5, "x1", 0.05)
    run_case(hank, 37, 3, 9, 70, "x1", 0.05, check_policies=False)
    run_case(hank, 37, 3, 9, 130, "x1", 0.05, check_policies=False)
    run_case(hank, 50, 2, 100, 130, "x1", 0.8, check_policies=False)
    run_case(hank, 37, 3, 9, 33, "x1", 0.05, check_policies=False)      # odd and > 32: one direction per lane, 64-lane groups


def test_sixteen_productivity_states(hank):
    """n_e = 16 is the largest block (1024 threads, one wavefront per productivity column)."""
    run_case(hank, 40, 16, 8, 6, "x1", 0.05)


def test_golden_full_pipeline(hank):
    """HIP path + host residual layer == golden F and J·y of the full KS pipeline."""
    g = np.load(G / "ks_30x3_T25_N3.npz")
    m, ss, orc = ks_setup(30, 3, 25)
    assert np.array_equal(ss.value, g["ss_value"]), "steady state drifted: regenerate the golden"
    P, N = 24, 3
    lin = hank.LinearizedFunction(g["x"].reshape(-1, order="F"), {"Z": g["Z"]}, m, ss, ss)
    close(lin.Fx, g["F"][..., 0].reshape(-1, order="F"))
    Jy = lin.jvp(g["y"].reshape(4 * P, N, order="F"))
    close(Jy, g["F"][..., 1:].reshape(4 * P, N, order="F"))
    close(hank.household_block(m).policy_seq().transpose(2, 0, 1), g["policy_seq"][..., 0])


def test_jvp_is_reusable_and_batch_invariant(hank):
    """same primal, different batches: column k of a batch == the single-tangent JVP of column k,
    and a second primal invalidates nothing silently."""
    m, ss, orc = ks_setup(50, 2, 100)
    P = 99
    x, Z = ks_paths(m, ss, "x1", 0.05)
    hb = hank.household_block(m)
    hb.set_boundary(ss.value, ss.D)
    hb.primal(x[2:4])
    y = np.random.default_rng(1).standard_normal((2, P, 6))
    full = hb.jvp(y)
    for k in (0, 5):
        one = hb.jvp(y[:, :, k:k + 1])
        close(one[:, 0], full[:, k], rel=1e-13)
    again = hb.jvp(y)
    assert np.array_equal(again, full)          # fixed summation order: bitwise reproducible


def test_knots_error_from_the_sweep(hank):
    m, ss, _ = ks_setup(50, 2, 100)
    hb = hank.household_block(m)
    bad = np.array(ss.value, copy=True)
    bad[7, :] *= 1e-4
    hb.set_boundary(bad, ss.D)
    x, _ = ks_paths(m, ss, "x0")
    with pytest.raises(hank.KnotsNotSortedError, match="period 99"):
        hb.primal(x[2:4])
    with pytest.raises(hank.HankHIPError):
        hb.jvp(np.zeros((2, 99, 1)))          # no valid primal -> refused
    hb.set_boundary(ss.value, ss.D)
    hb.primal(x[2:4])                          # context recovers


def _forced(hank, m, sched):
    """a context of model m with one implementation forced for every entry point (HANK_SCHEDULE)."""
    import os
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    os.environ["HANK_SCHEDULE"] = sched
    try:
        return hank.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
    finally:
        os.environ.pop("HANK_SCHEDULE")


def test_dual_sweep_equals_primal_then_jvp(hank):
    """hank_primal_jvp (value and partials in one pass) == hank_primal followed by hank_jvp, and it leaves the same record
    behind: bit for bit with the per-period launches; to rounding of the aggregate sums with the XCD-local persistent sweeps
    (their dual pass carries D_t along with its partials and takes the term dpol_t . D_t at the target rows, the tangent sweep
    at a recorded primal takes it at the source rows: another order of the same sum) and in the default schedule, which mixes
    the two implementations. Policies and their partials stay bit-identical throughout."""
    for (na, ne, T, N) in [(50, 2, 100, 3), (37, 3, 9, 5), (500, 4, 300, 1)]:
        m, ss, orc = ks_setup(na, ne, T)
        P = T - 1
        x, Z = ks_paths(m, ss, "x1", 0.05)
        y = np.random.default_rng(5).standard_normal((2, P, N))
        for sched in ("launch", "xcd", None):
            hb = _forced(hank, m, sched) if sched else hank.household_block(m)
            same = np.array_equal if sched == "launch" else (lambda a, b: np.max(np.abs(a - b)) <= 1e-13 * max(np.max(np.abs(b)), 1e-300))
            hb.set_boundary(ss.value, ss.D)
            agg0 = hb.primal(x[2:4]); dagg0 = hb.jvp(y); pol0 = hb.policy_seq(); dpol0 = hb.dpolicy_seq(N)
            hb.primal(x[2:4] * 1.01)                       # scramble the record
            agg1, dagg1 = hb.primal_jvp(x[2:4], y)
            assert same(agg0, agg1) and same(dagg0, dagg1)
            assert np.array_equal(pol0, hb.policy_seq()) and np.array_equal(dpol0, hb.dpolicy_seq(N))
            assert same(hb.jvp(y), dagg0)                  # the record it leaves serves later JVPs
            if sched:
                hb.close()


def test_alternating_batch_widths_reuse_workspaces(hank):
    """J̅ assembly (N = 256) and the Newton inner loop (N = 1) alternate on one context: the tangent workspaces come
    from a small most-recently-used cache, so no width is allocated twice (hank_stats) and results do not depend on
    what ran in between."""
    m, ss, orc = ks_setup(50, 2, 100)
    P = 99
    x, Z = ks_paths(m, ss, "x1", 0.05)
    hb = hank.household_block(m)
    hb.set_boundary(ss.value, ss.D)
    hb.primal(x[2:4])
    rng = np.random.default_rng(9)
    y1, y256, y32 = rng.standard_normal((2, P, 1)), rng.standard_normal((2, P, 256)), rng.standard_normal((2, P, 32))
    a, b, c = hb.jvp(y1), hb.jvp(y256), hb.jvp(y32)
    allocated = hb.stats()["tangent_workspaces_allocated"]
    for _ in range(3):
        assert np.array_equal(hb.jvp(y1), a)
        assert np.array_equal(hb.jvp(y256), b)
        assert np.array_equal(hb.jvp(y32), c)
    assert hb.stats()["tangent_workspaces_allocated"] == allocated
    assert np.array_equal(hb.dpolicy_seq(32)[..., 0], hb.dpolicy_seq(32)[..., 0])
    close(b[:, :1] * 0 + a, a)      # shapes (P, 1)
    with pytest.raises(hank.HankHIPError):
        hb.dpolicy_seq(256)          # the last sweep carried 32 directions, not 256


def test_schedules_agree(hank):
    """the XCD-local persistent sweeps and the per-period launches are two implementations of the same arithmetic (the
    default schedule picks per entry point and batch width): policies and their partials bit for bit, aggregates to
    summation order."""
    m, ss, orc = ks_setup(500, 4, 300)
    P, N = 299, 12
    x, Z = ks_paths(m, ss, "x1", 0.01)
    y = np.random.default_rng(2).standard_normal((2, P, N))
    res = {}
    for sched in ("xcd", "launch"):
        hb = _forced(hank, m, sched)
        hb.set_boundary(ss.value, ss.D)
        agg, dagg = hb.primal_jvp(x[2:4], y)
        assert hb.stats()["schedule"] == (1 if sched == "xcd" else 0) and hb.stats()["fallbacks"] == 0
        res[sched] = (agg, dagg, hb.policy_seq(), hb.dpolicy_seq(N), hb.dist_seq())
        hb.close()
    assert np.array_equal(res["xcd"][2], res["launch"][2])
    assert np.array_equal(res["xcd"][3], res["launch"][3])
    close(res["xcd"][4], res["launch"][4], rel=1e-12)
    close(res["xcd"][0], res["launch"][0], rel=1e-12)
    close(res["xcd"][1], res["launch"][1], rel=1e-11)


def test_default_schedule_runs_one_pass_dual_batches_on_the_persistent_dual_pass(hank):
    """hank_primal_jvp in the default schedule: a batch of one pass (N <= 32) is TWO persistent launches that carry value and
    partials together (k_xdual_back, k_xfwd<D, true>), a wider batch the dual-sweep launches; either way the numbers are the
    launches' (policies and partials bit for bit, aggregates to the order of their sums)."""
    m, ss, _ = ks_setup(500, 4, 300)
    P = 299
    x, _ = ks_paths(m, ss, "x1", 0.01)
    rng = np.random.default_rng(11)
    y32, y40 = rng.standard_normal((2, P, 32)), rng.standard_normal((2, P, 40))
    ref = _forced(hank, m, "launch")
    ref.set_boundary(ss.value, ss.D)
    hb = hank.household_block(m)
    hb.set_boundary(ss.value, ss.D)
    n0 = hb.stats()["sweep_launches"]
    agg, dagg = hb.primal_jvp(x[2:4], y32)
    tm = hb.last_timings()
    assert hb.stats()["schedule"] == 2 and hb.stats()["sweep_launches"] - n0 == 2
    assert tm["dual_backward"]["ms"] <= 0.0 and tm["primal_backward"]["launches"] == 1 and tm["tangent_forward"]["launches"] == 1
    agg0, dagg0 = ref.primal_jvp(x[2:4], y32)
    close(agg, agg0, rel=1e-12); close(dagg, dagg0, rel=1e-11)
    assert np.array_equal(hb.policy_seq(), ref.policy_seq()) and np.array_equal(hb.dpolicy_seq(32), ref.dpolicy_seq(32))
    hb.primal_jvp(x[2:4] * 1.01, y32)                                           # (another x on record: the next call is no memo hit)
    agg2, dagg2 = hb.primal_jvp(x[2:4], y32)
    assert np.array_equal(agg2, agg) and np.array_equal(dagg2, dagg)          # reproducible bit for bit
    aggw, daggw = hb.primal_jvp(x[2:4] * 1.02, y40)
    assert hb.last_timings()["dual_backward"]["launches"] > 1                   # the dual-sweep launches
    agg1, dagg1 = ref.primal_jvp(x[2:4] * 1.02, y40)
    assert np.array_equal(aggw, agg1) and np.array_equal(daggw, dagg1)
    assert hb.stats()["fallbacks"] == 0
    ref.close()


def test_segment_records_are_built_on_demand_behind_a_persistent_dual_pass(hank):
    """The persistent Dual pass leaves the per-target segment records of the lottery (R.seg) unwritten — its forward half reads
    work units — and whoever reads them next builds them first (k_seg_build): a launch-family batch (64 < N < the wide family's
    crossover) at the primal a Dual pass recorded returns what it returns at the same primal recorded by hank_primal."""
    m, ss, _ = ks_setup(500, 4, 300)
    P = 299
    x, _ = ks_paths(m, ss, "x1", 0.01)
    rng = np.random.default_rng(5)
    y32, y100 = rng.standard_normal((2, P, 32)), rng.standard_normal((2, P, 72))     # (72: between the persistent sweeps' 64 and the wide family's 80)
    hb = hank.household_block(m)
    hb.set_boundary(ss.value, ss.D)
    hb.primal(x[2:4])                                                           # the lottery writes the records itself
    want = hb.jvp(y100)
    assert hb.info()["last_tangent_family_name"] == "launch-per-period"
    hb.primal_jvp(x[2:4] * 1.03, y32)                                           # other segments on record, written by k_lottery ...
    hb.primal(x[2:4] * 1.03); hb.jvp(y100)                                      # ... and read by the launches
    agg, dagg = hb.primal_jvp(x[2:4], y32)                                      # the persistent Dual pass at x: no records written
    assert hb.last_timings()["tangent_forward"]["launches"] == 1
    got = hb.jvp(y100)                                                          # the launches at that record: built on demand
    assert hb.info()["last_tangent_family_name"] == "launch-per-period"
    close(got, want, rel=1e-11)                                                 # (stale records — the other x's segments — are off by percents)
    assert hb.stats()["fallbacks"] == 0


@pytest.mark.parametrize("n_a,n_e,T,N", [(130, 3, 40, 6), (500, 4, 300, 32), (200, 7, 40, 17)])
def test_device_pointer_dual_pass_equals_the_host_pointer_one(hank, n_a, n_e, T, N):
    """hank_primal_jvp_dev of a one-pass batch wraps its two sweeps in ONE launch in front and ONE behind (k_xdual_prologue /
    k_xdual_epilogue: inputs read where they lie, sync blocks zeroed, aggregates summed in k_reduce_parts' order and written to the
    caller's buffers); the host-pointer form keeps the separate copies and launches. Same bits: the value, the partials, both
    aggregates of the second heterogeneous output, and the record a later hank_jvp reads."""
    import torch
    m, ss, _ = ks_setup(n_a, n_e, T)
    P = T - 1
    x, _ = ks_paths(m, ss, "x1", 0.05)
    y = np.random.default_rng(N).standard_normal((2, P, N))
    hb = hank.household_block(m)
    hb.set_boundary(ss.value, ss.D)
    agg, dagg = hb.primal_jvp(x[2:4], y)
    assert hb.info()["last_tangent_family_name"] == "xcd-persistent" and hb.last_timings()["tangent_forward"]["launches"] == 1
    het = hb.het_outputs(2, y)
    again = hb.jvp(y)
    dev = torch.device("cuda", 0)
    d_x = torch.from_numpy(np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()).to(dev)
    d_y = torch.from_numpy(np.asfortranarray(y).reshape(-1, order="F").copy()).to(dev)
    d_a = torch.full((P,), np.nan, dtype=torch.float64, device=dev); d_d = torch.full((P * N,), np.nan, dtype=torch.float64, device=dev)
    hb.primal(x[2:4] * 1.01)                                                     # (another record in between)
    hb.primal_jvp_dev(d_x.data_ptr(), d_y.data_ptr(), N, d_a.data_ptr(), d_d.data_ptr())
    hb.sync(); hb.check()
    assert np.array_equal(d_a.cpu().numpy(), agg)
    assert np.array_equal(d_d.cpu().numpy().reshape(P, N, order="F"), dagg)
    het2 = hb.het_outputs(2, y)
    assert np.array_equal(het2[0], het[0]) and np.array_equal(het2[1], het[1])
    assert np.array_equal(hb.jvp(y), again)
    assert hb.stats()["fallbacks"] == 0
