"""The host drivers run under a capped BLAS / OpenMP pool (hank_amd/_threads.py): the cap is applied inside them and
lifted afterwards, HANK_HOST_THREADS overrides it, and a missing threadpoolctl changes nothing."""
import numpy as np
import pytest

import hank_amd as h
from hank_amd import _threads


def _blas_threads():
    tpc = pytest.importorskip("threadpoolctl")
    pools = [p for p in tpc.threadpool_info() if p["user_api"] == "blas"]
    if not pools:
        pytest.skip("numpy without a controllable BLAS pool")
    return max(p["num_threads"] for p in pools)


def test_cap_applies_inside_a_driver_and_is_lifted_after(monkeypatch):
    np.ones((8, 8)) @ np.ones((8, 8))        # make sure the pool exists
    before = _blas_threads()
    seen = {}

    @_threads.host_algebra
    def driver():
        seen["inside"] = _blas_threads()
        return 7

    monkeypatch.setenv("HANK_HOST_THREADS", "2")
    assert driver() == 7
    assert seen["inside"] == min(2, before)
    assert _blas_threads() == before
    monkeypatch.setenv("HANK_HOST_THREADS", "0")      # 0 = leave the pools alone
    driver()
    assert seen["inside"] == before


def test_the_drivers_are_wrapped_and_keep_their_counters():
    for fn in (h.find_ss, h.getSteadyStateJacobian, h.y_Iteration, h.NewtonRaphsonHANK):
        assert hasattr(fn, "__wrapped__"), fn.__name__
    # the counters the examples read live on the public (wrapped) names
    h.y_Iteration.total_jvps = 0
    assert h.y_Iteration.total_jvps == 0


def test_without_threadpoolctl_nothing_changes(monkeypatch):
    import builtins
    real = builtins.__import__

    def fake(name, *a, **k):
        if name == "threadpoolctl":
            raise ImportError("no threadpoolctl here")
        return real(name, *a, **k)

    monkeypatch.setattr(builtins, "__import__", fake)
    with _threads.host_threads():
        assert float(np.dot(np.ones(4), np.ones(4))) == 4.0
