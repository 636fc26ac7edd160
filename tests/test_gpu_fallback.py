"""The fallback from the persistent sweeps to the per-period launches, exercised: the dev knob HANK_XFAULT (read at
hank_create) pre-sets the status word of the persistent launches it names — what a sweep that cannot form its groups
("placement") or gives up waiting ("timeout") leaves behind — so that every entry point's recovery path runs on hardware:
the host-pointer entries return the launches' numbers with stats()["fallbacks"] == 1, the asynchronous *_dev entries
surface HANK_ERR_SWEEP at hank_check."""
import numpy as np
import pytest
import torch

from conftest import ks_paths, ks_setup

pytestmark = pytest.mark.gpu


def _block(hank, m, monkeypatch, fault=None, sched=None):
    for k, v in (("HANK_XFAULT", fault), ("HANK_SCHEDULE", sched)):
        if v is None:
            monkeypatch.delenv(k, raising=False)
        else:
            monkeypatch.setenv(k, v)
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    hb = hank.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
    monkeypatch.delenv("HANK_XFAULT", raising=False)
    monkeypatch.delenv("HANK_SCHEDULE", raising=False)
    return hb


@pytest.mark.parametrize("fault", ["placement", "timeout"])
def test_host_entries_fall_back_to_the_launches(hank, monkeypatch, fault):
    m, ss, _ = ks_setup(130, 3, 20)
    P = 19
    x, _ = ks_paths(m, ss, "x1", 0.05)
    y = np.random.default_rng(1).standard_normal((2, P, 5))
    ref = _block(hank, m, monkeypatch, sched="launch")
    ref.set_boundary(ss.value, ss.D)
    agg0 = ref.primal(x[2:4]); d0 = ref.jvp(y); both0 = ref.primal_jvp(x[2:4], y)
    # hank_primal: the faulted Float64 sweeps are redone by the launches, the context stays on them
    hb = _block(hank, m, monkeypatch, fault=fault)
    hb.set_boundary(ss.value, ss.D)
    assert hb.stats()["schedule"] == 2
    assert np.array_equal(hb.primal(x[2:4]), agg0)
    st = hb.stats()
    assert st["fallbacks"] == 1 and st["schedule"] == 0
    assert np.array_equal(hb.jvp(y), d0) and hb.stats()["fallbacks"] == 1
    hb.close()
    # hank_jvp: the primal sweeps run, the tangent sweeps are faulted — the launches re-record the primal and serve the batch
    hb = _block(hank, m, monkeypatch, fault=fault + ":tangent")
    hb.set_boundary(ss.value, ss.D)
    a_x = hb.primal(x[2:4])         # (persistent Float64 sweeps: the aggregate's partial sums combine in another order than the launches')
    assert np.max(np.abs(a_x - agg0)) <= 1e-13 * np.abs(agg0).max() and hb.stats()["fallbacks"] == 0
    assert np.array_equal(hb.jvp(y), d0)
    assert hb.stats()["fallbacks"] == 1 and hb.stats()["schedule"] == 0
    hb.close()
    # hank_primal_jvp with the persistent sweeps forced on it would fail loudly; in the default schedule it runs on launches
    hb = _block(hank, m, monkeypatch, fault=fault)
    hb.set_boundary(ss.value, ss.D)
    a, d = hb.primal_jvp(x[2:4], y)
    assert np.array_equal(a, both0[0]) and np.array_equal(d, both0[1])
    hb.close()
    hb = _block(hank, m, monkeypatch, fault=fault, sched="xcd")
    hb.set_boundary(ss.value, ss.D)
    with pytest.raises(hank.HankHIPError, match="persistent"):
        hb.primal_jvp(x[2:4], y)
    hb.close()
    ref.close()


def test_steady_state_fixed_points_fall_back(hank, monkeypatch):
    m, ss, _ = ks_setup(130, 3, 20)
    xv = dict(ss.vars)
    ref = _block(hank, m, monkeypatch, sched="launch")
    v0, p0, it0, _ = ref.vfi(np.ones((130, 3)), [xv["r"], xv["w"]], 1e-11)
    D0, _ = ref.stationary_dist(ss.policies["KD"])
    hb = _block(hank, m, monkeypatch, fault="placement:fixedpoint")
    v, p, it, nrm = hb.vfi(np.ones((130, 3)), [xv["r"], xv["w"]], 1e-11)
    assert it == it0 and np.array_equal(v, v0) and np.array_equal(p, p0)
    assert hb.stats()["fallbacks"] == 1
    hb.close()
    hb = _block(hank, m, monkeypatch, fault="timeout:fixedpoint")
    D, _ = hb.stationary_dist(ss.policies["KD"])
    assert np.max(np.abs(D - D0)) < 1e-14 and hb.stats()["fallbacks"] == 1
    hb.close()
    ref.close()


def test_async_entries_report_the_sweep_error_at_check(hank, monkeypatch):
    m, ss, _ = ks_setup(130, 3, 20)
    P = 19
    x, _ = ks_paths(m, ss, "x1", 0.05)
    dev = torch.device("cuda", 0)
    d_x = torch.from_numpy(np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()).to(dev)
    d_agg = torch.empty(P, dtype=torch.float64, device=dev)
    hb = _block(hank, m, monkeypatch, fault="placement")
    hb.set_boundary(ss.value, ss.D)
    hb.primal_dev(d_x.data_ptr(), d_agg.data_ptr())           # enqueued: nothing to report yet
    with pytest.raises(hank.HankHIPError, match="persistent") as ei:
        hb.check()
    assert ei.value.code == hank.hip.HANK_ERR_SWEEP
    # reported once; a context whose schedule was not forced continues on the per-period launches: the caller's next call succeeds
    ref = _block(hank, m, monkeypatch, sched="launch")
    ref.set_boundary(ss.value, ss.D)
    agg0 = ref.primal(x[2:4])
    hb.primal_dev(d_x.data_ptr(), d_agg.data_ptr())
    hb.check()
    st = hb.stats()
    assert st["schedule"] == 0 and st["fallbacks"] == 1
    assert np.array_equal(d_agg.cpu().numpy(), agg0)
    ref.close()
    hb.close()


def test_a_wait_that_really_times_out_is_bounded_in_time(hank, monkeypatch):
    """HANK_XFAULT=stall: member 0 of group 0 of the persistent backward tangent sweep stops publishing after three periods, so its
    neighbours' waits run into the deadline (HANK_XWAIT_MS, s_memrealtime — a time, not a spin count), every later wait falls
    through and the grid drains. The host-pointer entry then returns the launches' numbers, well within 0.2 s; a forced schedule
    reports how long the sweep had waited."""
    import time
    m, ss, _ = ks_setup(130, 3, 20)
    P = 19
    x, _ = ks_paths(m, ss, "x1", 0.05)
    y = np.random.default_rng(1).standard_normal((2, P, 5))
    ref = _block(hank, m, monkeypatch, sched="launch")
    ref.set_boundary(ss.value, ss.D)
    ref.primal(x[2:4]); d0 = ref.jvp(y)
    monkeypatch.setenv("HANK_XWAIT_MS", "5")
    hb = _block(hank, m, monkeypatch, fault="stall")
    hb.set_boundary(ss.value, ss.D)
    hb.primal(x[2:4])
    t0 = time.perf_counter()
    d = hb.jvp(y)
    el = time.perf_counter() - t0
    assert np.array_equal(d, d0) and hb.stats()["fallbacks"] == 1
    assert el < 0.2, el
    hb.close()
    hb = _block(hank, m, monkeypatch, fault="stall", sched="xcd")
    hb.set_boundary(ss.value, ss.D)
    hb.primal(x[2:4])
    with pytest.raises(hank.HankHIPError, match=r"a wait timed out after \d+\.\d ms"):
        hb.jvp(y)
    hb.close()
    monkeypatch.delenv("HANK_XWAIT_MS", raising=False)
    ref.close()


def test_work_unit_overflow_is_served_by_the_launches(hank, monkeypatch):
    """A member whose walk over its sources needs more work units than the budget (XUCAP = 64 per member and period; a savings
    policy that is flat in index space) — exercised through the dev knob HANK_XUCAP=3, which lowers the budget k_xunits_fwd
    checks against. The forward sweeps leave BEFORE computing anything from a truncated unit list (k_xfwd reads the flag at entry);
    the host-pointer entries return the launches' numbers and the context stays on them, the asynchronous entries report the
    overflow once at hank_check, a forced schedule fails loudly."""
    m, ss, _ = ks_setup(130, 3, 20)
    P = 19
    x, _ = ks_paths(m, ss, "x1", 0.05)
    y = np.random.default_rng(2).standard_normal((2, P, 5))
    ref = _block(hank, m, monkeypatch, sched="launch")
    ref.set_boundary(ss.value, ss.D)
    agg0 = ref.primal(x[2:4]); d0 = ref.jvp(y); both0 = ref.primal_jvp(x[2:4], y)
    monkeypatch.setenv("HANK_XUCAP", "3")
    hb = _block(hank, m, monkeypatch)
    hb.set_boundary(ss.value, ss.D)
    assert hb.stats()["schedule"] == 2
    assert np.array_equal(hb.primal(x[2:4]), agg0)
    st = hb.stats()
    assert st["fallbacks"] == 1 and st["schedule"] == 0
    assert np.array_equal(hb.jvp(y), d0)
    hb.close()
    hb = _block(hank, m, monkeypatch)
    hb.set_boundary(ss.value, ss.D)
    a, d = hb.primal_jvp(x[2:4], y)
    assert np.array_equal(a, both0[0]) and np.array_equal(d, both0[1]) and hb.stats()["fallbacks"] == 1
    hb.close()
    # asynchronous entry: reported once at hank_check, then the launches serve the context
    dev = torch.device("cuda", 0)
    d_x = torch.from_numpy(np.asfortranarray(x[2:4]).reshape(-1, order="F").copy()).to(dev)
    d_agg = torch.empty(P, dtype=torch.float64, device=dev)
    hb = _block(hank, m, monkeypatch)
    hb.set_boundary(ss.value, ss.D)
    hb.primal_dev(d_x.data_ptr(), d_agg.data_ptr())
    with pytest.raises(hank.HankHIPError, match="work units") as ei:
        hb.check()
    assert ei.value.code == hank.hip.HANK_ERR_SWEEP
    hb.primal_dev(d_x.data_ptr(), d_agg.data_ptr())
    hb.check()
    assert hb.stats()["schedule"] == 0 and np.array_equal(d_agg.cpu().numpy(), agg0)
    hb.close()
    hb = _block(hank, m, monkeypatch, sched="xcd")
    hb.set_boundary(ss.value, ss.D)
    with pytest.raises(hank.HankHIPError, match="work units"):
        hb.primal(x[2:4])
    hb.close()
    monkeypatch.delenv("HANK_XUCAP", raising=False)
    ref.close()
