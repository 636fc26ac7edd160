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
