"""A cloned context is an independent copy of the same household block (used for two batches in flight)."""
import numpy as np
import pytest

from conftest import ks_paths, ks_setup

pytestmark = pytest.mark.gpu


def test_clone_gives_identical_results_and_is_independent(hank):
    m, ss, _ = ks_setup(50, 2, 20)
    P = 19
    x, _ = ks_paths(m, ss, "x1", 0.05)
    y = np.random.default_rng(3).standard_normal((2, P, 6))
    hb = hank.household_block(m)
    hb.set_boundary(ss.value, ss.D)
    hb2 = hb.clone()
    try:
        a1, d1 = hb.primal_jvp(x[2:4], y)
        a2, d2 = hb2.primal_jvp(x[2:4], y)
        np.testing.assert_array_equal(a1, a2)
        np.testing.assert_array_equal(d1, d2)
        # the clone keeps its own primal: moving one context does not disturb the other
        hb2.primal(x[2:4] * 1.01)
        again = hb.jvp(y)      # (default schedule: a narrow hank_jvp runs as persistent sweeps, d1 came from the dual-sweep launches)
        assert np.max(np.abs(again - d1)) <= 1e-13 * np.abs(d1).max()
        np.testing.assert_array_equal(hb.jvp(y), again)
    finally:
        hb2.close()


def test_create_on_a_named_device_and_device_group(hank):
    """hank_create_on: a context on a named HIP device (the one-GPU box has device 0 only) gives the same numbers as the
    context of the current device; a device outside the visible range is refused; DeviceGroup shards the columns of a batch
    over its contexts (here: two contexts on the same GPU) and assembles them on the host."""
    from hank_amd.parallel import DeviceGroup
    m, ss, _ = ks_setup(50, 2, 20)
    P = 19
    x, _ = ks_paths(m, ss, "x1", 0.05)
    y = np.random.default_rng(4).standard_normal((2, P, 7))
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    args = (wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
    hb0 = hank.HouseholdBlock(*args)
    hb1 = hank.HouseholdBlock(*args, device=0)
    with pytest.raises(hank.HankHIPError, match="device"):
        hank.HouseholdBlock(*args, device=63)
    for hb in (hb0, hb1):
        hb.set_boundary(ss.value, ss.D)
    a0 = hb0.primal(x[2:4]); a1 = hb1.primal(x[2:4])
    np.testing.assert_array_equal(a0, a1)
    d0 = hb0.jvp(y)
    np.testing.assert_array_equal(hb1.jvp(y), d0)
    g = DeviceGroup(hb1, [0, 0])
    g.set_boundary(ss.value, ss.D)
    np.testing.assert_array_equal(g.primal(x[2:4]), a0)
    dg = g.jvp(y)               # columns [0, 4) on the first context, [4, 7) on the second
    assert np.max(np.abs(dg - d0)) <= 1e-13 * np.abs(d0).max()
    # the same partition assembled ON THE DEVICE (hank_gather_columns; on this box the "peer" is the same GPU): the host assembly, bit for bit
    import torch
    from hank_amd.parallel import shard_bounds
    bounds = [shard_bounds(7, 2, k) for k in range(2)]
    dev = torch.device("cuda", 0)
    dy = [torch.from_numpy(np.asfortranarray(y[:, :, lo:hi]).reshape(-1, order="F").copy()).to(dev) for lo, hi in bounds]
    full = g.jvp_dev(dy, [hi - lo for lo, hi in bounds])
    np.testing.assert_array_equal(full.cpu().numpy().reshape((P, 7), order="F"), dg)
    with pytest.raises(hank.HankHIPError):          # a block pointer that is missing is refused, nothing is copied
        hank.hip.gather_columns(g.blocks, [0, 0], [4, 3], full.data_ptr())
    g.close(); hb1.close(); hb0.close()


def test_a_failed_workspace_allocation_is_not_cached(hank):
    """a tangent batch too wide for the card's memory (dpol alone: P*G*N*8 bytes) is refused with NOMEM, and again on a retry
    with the same N — the half-built workspace must not stay in the per-width cache (a second call used to find it, return OK
    and launch on null pointers); the context keeps working."""
    m, ss, _ = ks_setup(2000, 11, 300)
    P = 299
    x, _ = ks_paths(m, ss, "x1", 0.01)
    import torch
    dev = torch.device("cuda", 0)
    for sched in ("launch", "auto"):
        import os
        if sched == "launch":
            os.environ["HANK_SCHEDULE"] = "launch"
        try:
            wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
            hb = hank.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
        finally:
            os.environ.pop("HANK_SCHEDULE", None)
        hb.set_boundary(ss.value, ss.D)
        agg = hb.primal(x[2:4])
        N = 8192                    # dpol: 299 * 22000 * 8192 * 8 B = 431 GB > 288 GB
        d_dx = torch.zeros(2 * P * N, dtype=torch.float64, device=dev)
        for _ in range(2):
            with pytest.raises(hank.HankHIPError) as ei:
                hb.jvp_dev(d_dx.data_ptr(), N, 0)
                hb.check()
            assert ei.value.code in (hank.hip.HANK_ERR_NOMEM, hank.hip.HANK_ERR_BAD_ARG), ei.value
        y = np.random.default_rng(2).standard_normal((2, P, 2))
        assert hb.jvp(y).shape == (P, 2)
        np.testing.assert_array_equal(hb.primal(x[2:4]), agg)
        hb.close()


def test_a_long_horizon_goes_to_the_launches(hank):
    """the persistent sweeps keep the per-period inputs of the whole horizon in LDS; a horizon that does not fit must select
    the per-period launches (which take any T) instead of failing at the launch. T = 6000 at 130x3: the Float64 sweeps need
    8*(3*64 + 9 + 130 + 4*5999) + 64 = 194 KB > 160 KB."""
    m, ss, _ = ks_setup(130, 3, 20)
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    T = 6000
    hb = hank.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, T)
    assert hb.stats()["schedule"] == 0
    hb.set_boundary(ss.value, ss.D)
    P = T - 1
    x = np.tile(np.array([[ss.vars["r"]], [ss.vars["w"]]]), (1, P))
    agg = hb.primal(x)
    assert abs(agg[-1] - ss.vars["KD"]) < 1e-6 * ss.vars["KD"]
    y = np.zeros((2, P, 1)); y[0, 100, 0] = 1.0
    assert np.isfinite(hb.jvp(y)).all()
    hb.close()
    with pytest.raises(hank.HankHIPError, match="LDS"):
        import os
        os.environ["HANK_SCHEDULE"] = "xcd"
        try:
            hank.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, T)
        finally:
            os.environ.pop("HANK_SCHEDULE", None)
