"""Pins the oracle's dual arithmetic against the REFERENCE's own C++ dual classes
(ForwardDiff.jl/benchmarks/cpp, compiled from /root/reference into oracle/_ref by oracle/Makefile)
and against the rosenbrock known answers the reference asserts (benchmarks.cpp:39-61)."""
import ctypes as C

import numpy as np
import pytest


def _ref(oracle_mod):
    lib = oracle_mod.ref_lib()
    if lib is None:
        pytest.skip("oracle/_ref not built (reference sources absent at build time)")
    return lib


# (ref op code, oracle op code): add, sub, mul, real*dual, real-dual
OPS = [(0, 0), (1, 1), (2, 2), (3, 4), (4, 5)]


@pytest.mark.parametrize("N,fname", [(1, "ref_dual1_op"), (3, "ref_dual3_op")])
def test_binary_rules_match_reference_cpp(oracle_mod, N, fname):
    lib = _ref(oracle_mod)
    f = getattr(lib, fname)
    rng = np.random.default_rng(7)
    dp = C.POINTER(C.c_double)
    for _ in range(200):
        x, y = rng.standard_normal(1 + N), rng.standard_normal(1 + N)
        for rop, oop in OPS:
            out = np.empty(1 + N)
            f(rop, x.ctypes.data_as(dp), y.ctypes.data_as(dp), out.ctypes.data_as(dp))
            if oop == 4:      # oracle Dual*Real with the roles swapped: (y as dual) * x.value
                mine = oracle_mod.dual_binop(4, y, x, N)
            else:
                mine = oracle_mod.dual_binop(oop, x, y, N)
            assert np.array_equal(out, mine), (rop, x, y, out, mine)


def test_pow_rule_matches_reference_sqrt_exp(oracle_mod):
    """reference sqrt(Dual) = (sqrt v, eps/(2 sqrt v)); oracle Dual^0.5 must agree to rounding."""
    lib = _ref(oracle_mod)
    dp = C.POINTER(C.c_double)
    rng = np.random.default_rng(3)
    for _ in range(100):
        x = np.array([rng.uniform(0.1, 5.0), rng.standard_normal()])
        out = np.empty(2)
        lib.ref_dual1_op(5, x.ctypes.data_as(dp), x.ctypes.data_as(dp), out.ctypes.data_as(dp))
        mine = oracle_mod.dual_binop(7, x, np.array([0.5, 0.0]), 1)
        np.testing.assert_allclose(mine, out, rtol=4e-16, atol=0)


def test_rosenbrock_known_answers(oracle_mod):
    """the integers the reference asserts for its chunk-1 gradient (benchmarks.cpp:39-61), computed
    (a) by the reference's own template through oracle/_ref and (b) with the oracle's dual ops."""
    expect10 = np.array([-2., -200., 1002., 5804., 16606., 35808., 65810., 109012., 167814., -11000.])
    x = np.arange(10, dtype=np.float64)
    lib = oracle_mod.ref_lib()
    if lib is not None:
        out = np.empty(10)
        dp = C.POINTER(C.c_double)
        lib.ref_rosenbrock_grad1(x.ctypes.data_as(dp), 10, out.ctypes.data_as(dp))
        assert np.array_equal(out, expect10)
    B = oracle_mod.dual_binop
    grad = np.empty(10)
    for i in range(10):
        d = [np.array([x[k], 1.0 if k == i else 0.0]) for k in range(10)]
        res = np.zeros(2)
        for k in range(9):
            t1 = B(5, np.array([1.0, 0.0]), d[k], 1)              # b - x[k]
            t2 = B(1, d[k + 1], B(2, d[k], d[k], 1), 1)           # x[k+1] - x[k]*x[k]
            res = B(0, B(0, res, B(2, t1, t1, 1), 1), B(4, B(2, t2, t2, 1), np.array([100.0, 0.0]), 1), 1)
        grad[i] = res[1]
    assert np.array_equal(grad, expect10)


def test_max_and_pow_edge_rules(oracle_mod):
    """DiffRules max: partial passes unless y > x (ties pass); Dual^Real: zero partials when the
    exponent is 0 or the partials are all zero (dual.jl:563-572)."""
    B = oracle_mod.dual_binop
    assert np.array_equal(B(8, np.array([1.0, 2.0]), np.array([0.5, 0.0]), 1), [1.0, 2.0])
    assert np.array_equal(B(8, np.array([0.2, 2.0]), np.array([0.5, 0.0]), 1), [0.5, 0.0])
    assert np.array_equal(B(8, np.array([0.0, 2.0]), np.array([0.0, 0.0]), 1), [0.0, 2.0])
    assert np.array_equal(B(8, np.array([-0.0, 2.0]), np.array([0.0, 0.0]), 1), [0.0, 0.0])
    assert np.array_equal(B(7, np.array([3.0, 5.0]), np.array([0.0, 0.0]), 1), [1.0, 0.0])
    assert np.array_equal(B(7, np.array([3.0, 0.0]), np.array([2.0, 0.0]), 1), [9.0, 0.0])
    np.testing.assert_allclose(B(7, np.array([3.0, 1.0]), np.array([2.0, 0.0]), 1), [9.0, 6.0])
    # Real/Dual and Dual/Dual (dual.jl:528-539)
    np.testing.assert_allclose(B(6, np.array([1.0, 0.0]), np.array([1.25, 2.0]), 1), [0.8, -2.0 / 1.25 ** 2])
    np.testing.assert_allclose(B(3, np.array([1.0, 0.5]), np.array([2.0, 3.0]), 1), [0.5, 0.5 / 2 - 3.0 / 4])
