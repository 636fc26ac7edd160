"""The reference's OWN identities for dual division and exponentiation (ForwardDiff.jl/test/DualTest.jl:428-447), transcribed as
randomized checks of the two dual-number implementations this build carries: the CPU oracle's (oracle/hank_oracle.c, ops 3 / 6 / 7
of orc_dual_binop) and the host's `hank_amd.dual.Dual` (the residual layer). The reference holds no fixture for the household
block (a3-a8 stay "parity unpinned", oracle/hank_oracle.c header), but with these a2 — the Dual / Partials rules, dual.jl:495-581,
partials.jl:80-120 — is covered by reference-held checks: `+ - * sqrt` bit-exact against the reference's C++ classes
(test_oracle_dual_vs_ref.py), `/` and `^` by the identities its own test-suite asserts.

  :431  FDNUM / FDNUM2  ~ Dual(v1 / v2, _div_partials(p1, p2, v1, v2)),  _div_partials = p1 * inv(v2) + p2 * (-(v1 / (v2 * v2)))   (partials.jl:85-87)
  :432  FDNUM / PRIMAL  ~ Dual(v / PRIMAL, p / PRIMAL)
  :433  PRIMAL / FDNUM  ~ Dual(PRIMAL / v, (-(PRIMAL) / v^2) * p)
  :443  FDNUM ^ PRIMAL  ~ exp(PRIMAL * log(FDNUM))
  :451  partials(NaNMath.pow(Dual(-2.0, 1.0), Dual(2.0, 0.0)), 1) == -4.0
`dual_isapprox` is isapprox on value and partials (rtol = sqrt(eps)); the checks here hold to 1e-13."""
import numpy as np
import pytest

DRAWS = 200


def _draws(N, seed):
    rng = np.random.default_rng(seed)
    for _ in range(DRAWS):
        v1, v2, pr = rng.uniform(0.1, 3.0, 3) * rng.choice([-1.0, 1.0], 3)     # "all random numbers nonzero" (DualTest.jl:422)
        yield v1, rng.uniform(0.1, 2.0, N) * rng.choice([-1.0, 1.0], N), v2, rng.uniform(0.1, 2.0, N) * rng.choice([-1.0, 1.0], N), pr


def _close(a, b):
    np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("N", [1, 3])
def test_oracle_division_and_power_meet_the_references_identities(oracle_mod, N):
    B = oracle_mod.dual_binop
    for v1, p1, v2, p2, pr in _draws(N, 10 + N):
        x, y = np.r_[v1, p1], np.r_[v2, p2]
        _close(B(3, x, y, N), np.r_[v1 / v2, p1 * (1.0 / v2) + p2 * (-(v1 / (v2 * v2)))])       # :431
        _close(B(3, x, np.r_[pr, np.zeros(N)], N), np.r_[v1 / pr, p1 / pr])                      # :432 (a Real is a Dual with zero partials)
        _close(B(6, np.r_[pr, np.zeros(N)], x, N), np.r_[pr / v1, (-(pr) / v1 ** 2) * p1])       # :433
        b = abs(v1)                                                                              # :443 needs log(FDNUM): positive base
        lv = np.r_[np.log(b), p1 / b]                                                            # log(FDNUM)
        ev = np.exp(pr * lv[0])
        _close(B(7, np.r_[b, p1], np.r_[pr, np.zeros(N)], N), np.r_[ev, ev * pr * lv[1:]])       # exp(PRIMAL * log(FDNUM))
    _close(B(7, np.r_[-2.0, 1.0, np.zeros(N - 1)], np.r_[2.0, np.zeros(N)], N)[:2], [4.0, -4.0])  # :451


@pytest.mark.parametrize("N", [1, 3])
def test_host_dual_division_and_power_meet_the_references_identities(hank, N):
    from hank_amd.dual import Dual
    for v1, p1, v2, p2, pr in _draws(N, 20 + N):
        x, y = Dual(np.array(v1), p1), Dual(np.array(v2), p2)
        q = x / y
        _close(q.v, v1 / v2); _close(q.p, p1 * (1.0 / v2) + p2 * (-(v1 / (v2 * v2))))
        q = x / pr
        _close(q.v, v1 / pr); _close(q.p, p1 / pr)
        q = pr / x
        _close(q.v, pr / v1); _close(q.p, (-(pr) / v1 ** 2) * p1)
        b = abs(v1)
        q = Dual(np.array(b), p1) ** pr
        ev = np.exp(pr * np.log(b))
        _close(q.v, ev); _close(q.p, ev * pr * p1 / b)
    q = Dual(np.array(-2.0), np.r_[1.0, np.zeros(N - 1)]) ** 2.0
    _close(q.v, 4.0); assert q.p[0] == -4.0


@pytest.mark.gpu
def test_device_step_meets_the_oracle_on_random_two_point_grids(hank, oracle_mod):
    """the device's own division / power partials (the EGM step: (beta E)^(-1/gamma), the interpolation slope, c^(-gamma)) on 100
    random 2-point-grid economies with gamma in {1, 2, 1.5, 3}: hank_backward_step_dual against the oracle whose `/` and `^` rules
    the tests above tie to the reference's identities (rel 1e-10 + abs 1e-12, the tolerance of every step test)."""
    rng = np.random.default_rng(99)
    for k in range(100):
        a = np.sort(rng.uniform(0.0, 5.0, 2)); a[1] += 0.5
        z = np.sort(rng.uniform(0.3, 2.0, 2)); z[1] += 0.1
        p, q = rng.uniform(0.6, 0.95, 2)
        Pi = np.array([[p, 1 - p], [1 - q, q]])
        gamma = [1.0, 2.0, 1.5, 3.0][k % 4]
        beta, N = rng.uniform(0.9, 0.99), 1 + (k % 3)
        hb = hank.HouseholdBlock(a, z, Pi, beta, gamma, 0.0, 4)
        orc = oracle_mod.Oracle(a, z, Pi, beta, gamma, 0.0)
        V = np.sort(rng.uniform(0.2, 2.0, (2, 2)), axis=0)[::-1]
        dV = rng.standard_normal((2, 2, N)); dx = rng.standard_normal((2, N))
        r, w = rng.uniform(-0.02, 0.08), rng.uniform(0.5, 2.0)
        st, oV, oKD = orc.value_function(np.concatenate([V[..., None], dV], -1), np.r_[r, dx[0]], np.r_[w, dx[1]], N)
        if st != 0:          # (a draw whose knots are not sorted: the reference errors too)
            hb.close()
            continue
        Vo, dVo, Po, dPo = hb.backward_step_dual(V, dV, [r, w], dx)
        for got, ref in ((Vo, oV[..., 0]), (Po, oKD[..., 0]), (dVo, oV[..., 1:]), (dPo, oKD[..., 1:])):
            assert np.max(np.abs(got - ref)) <= 1e-12 + 1e-10 * max(np.abs(ref).max(), 1e-300), (k, gamma)
        hb.close()
