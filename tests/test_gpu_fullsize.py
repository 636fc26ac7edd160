"""BASELINE.json's full size (2000x11, T=300, 32-wide tangent batch) on the MI355X.
The oracle needs ~1.5 s per tangent here: the benched entry is checked on ALL 32 of its directions (round 5), the other entries on
two, and everything through size-independent properties: linearity in the tangent, mass conservation of D_t and
of its partials, finite-difference agreement of the Float64 path, reproducibility."""
import numpy as np
import pytest

from conftest import ks_paths, ks_setup

pytestmark = pytest.mark.gpu


def same(a, b, rel=1e-13):
    """equal to the rounding of the aggregate sums: the default schedule runs hank_primal / narrow hank_jvp batches as
    XCD-local persistent sweeps and hank_primal_jvp / wide batches as per-period launches (same arithmetic per grid
    point, different summation order over the grid)."""
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) <= rel * max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def big(hank):
    m, ss, orc = ks_setup(2000, 11, 300)
    hb = hank.household_block(m)
    hb.set_boundary(ss.value, ss.D)
    x, Z = ks_paths(m, ss, "x1", 0.01)
    agg = hb.primal(x[2:4])
    y = np.random.default_rng(0).standard_normal((2, 299, 32))
    dagg = hb.jvp(y)
    return m, ss, orc, hb, x, Z, y, agg, dagg


def test_two_columns_against_the_oracle(big):
    m, ss, orc, hb, x, Z, y, agg, dagg = big
    P, N = 299, 2
    xr = np.zeros((P, 1 + N)); xw = np.zeros((P, 1 + N))
    xr[:, 0], xw[:, 0] = x[2], x[3]
    xr[:, 1:], xw[:, 1:] = y[0][:, [0, 31]], y[1][:, [0, 31]]
    st, oagg, _ = orc.household_block(xr, xw, ss.value, ss.D, N)
    assert st == 0
    assert np.max(np.abs(agg - oagg[:, 0])) < 1e-10 * np.abs(oagg[:, 0]).max()
    ref = oagg[:, 1:]
    assert np.max(np.abs(dagg[:, [0, 31]] - ref)) < 1e-12 + 1e-10 * np.abs(ref).max()


def test_linearity_and_reproducibility(big):
    m, ss, orc, hb, x, Z, y, agg, dagg = big
    c = np.random.default_rng(1).standard_normal(32)
    comb = hb.jvp(np.tensordot(y, c, axes=([2], [0]))[:, :, None])[:, 0]
    ref = dagg @ c
    assert np.max(np.abs(comb - ref)) < 1e-9 * np.abs(ref).max()
    assert np.array_equal(hb.jvp(y), dagg)


def test_mass_conservation(big):
    m, ss, orc, hb, x, Z, y, agg, dagg = big
    D = hb.dist_seq()
    np.testing.assert_allclose(D.sum(axis=(0, 1)), 1.0, atol=1e-11)
    assert D.min() >= 0.0
    pol = hb.policy_seq()
    assert np.all(np.diff(pol, axis=0) >= 0) and pol.min() >= 0.0


def test_finite_difference_of_the_device_primal(big):
    m, ss, orc, hb, x, Z, y, agg, dagg = big
    h = 1e-6
    up = hb.primal(x[2:4] + h * y[:, :, 3])
    dn = hb.primal(x[2:4] - h * y[:, :, 3])
    hb.primal(x[2:4])
    fd = (up - dn) / (2 * h)
    assert np.max(np.abs(fd - dagg[:, 3])) < 5e-4 * np.abs(dagg[:, 3]).max()


def _oracle_cols(orc, x, y, cols, ss, xt=None, yt=None):
    """value + the tangent columns `cols` through the CPU oracle (r, w[, transfer] duals)."""
    P, N = y.shape[1], len(cols)
    xr = np.zeros((P, 1 + N)); xw = np.zeros((P, 1 + N))
    xr[:, 0], xw[:, 0] = x[0], x[1]
    xr[:, 1:], xw[:, 1:] = y[0][:, cols], y[1][:, cols]
    xtd = None
    if xt is not None:
        xtd = np.zeros((P, 1 + N)); xtd[:, 0] = xt; xtd[:, 1:] = yt[:, cols]
    st, oagg, _ = orc.household_block(xr, xw, ss.value, ss.D, N, xt=xtd)
    assert st == 0
    return oagg


def test_benched_entry_primal_jvp_N32(big):
    """the entry point bench.py times (hank_primal_jvp: value and 32 partials in one dual pass) at the benched size:
    same numbers as hank_primal + hank_jvp, and two of its columns against the oracle (NewtonRaphson.jl:95)."""
    m, ss, orc, hb, x, Z, y, agg, dagg = big
    agg2, dagg2 = hb.primal_jvp(x[2:4], y)
    assert same(agg2, agg) and same(dagg2, dagg, 1e-12)
    oagg = _oracle_cols(orc, x[2:4], y, [5, 18], ss)
    assert np.max(np.abs(agg2 - oagg[:, 0])) < 1e-10 * np.abs(oagg[:, 0]).max()
    assert np.max(np.abs(dagg2[:, [5, 18]] - oagg[:, 1:])) < 1e-12 + 1e-10 * np.abs(oagg[:, 1:]).max()
    assert same(hb.jvp(y), dagg, 1e-12)            # the record it leaves serves later JVPs


def test_benched_entry_every_column_against_the_oracle(big):
    """ALL 32 columns of the benched Dual pass against the oracle at the benched size (four oracle passes of eight partials, one
    host thread each — the C calls release the interpreter lock): nothing of the headline result rides on linearity alone."""
    from concurrent.futures import ThreadPoolExecutor
    m, ss, orc, hb, x, Z, y, agg, dagg = big
    agg2, dagg2 = hb.primal_jvp(x[2:4] * 1.001, y)             # (another point than the fixture's: a full Dual pass, no memo)
    assert hb.last_timings()["tangent_forward"]["launches"] == 1 and hb.info()["last_tangent_family_name"] == "xcd-persistent"
    chunks = [list(range(c0, c0 + 8)) for c0 in range(0, 32, 8)]
    with ThreadPoolExecutor(max_workers=4) as ex:
        outs = list(ex.map(lambda cols: _oracle_cols(orc, x[2:4] * 1.001, y, cols, ss), chunks))
    ref = np.concatenate([o[:, 1:] for o in outs], axis=1)
    assert np.max(np.abs(agg2 - outs[0][:, 0])) < 1e-10 * np.abs(outs[0][:, 0]).max()
    assert np.max(np.abs(dagg2 - ref)) < 1e-12 + 1e-10 * np.abs(ref).max()
    hb.primal(x[2:4]); hb.jvp(y)                       # leave the module fixture's state behind


def test_wide_batch_N256_full_size(big):
    """configs[3]'s per-node batch (256 tangents) on one GPU at 2000x11, T=300: 32 columns against the oracle,
    linearity in the tangent, bitwise repeatability, and batch invariance against the N=32 result."""
    m, ss, orc, hb, x, Z, y, agg, dagg = big
    N = 256
    yw = np.random.default_rng(7).standard_normal((2, 299, N))
    yw[:, :, :32] = y                                  # the first 32 columns are the N=32 batch
    aggw, daggw = hb.primal_jvp(x[2:4], yw)
    assert hb.info()["last_tangent_family_name"] == "on-chip-wide"      # a full round: one workgroup per direction (csrc/hank_wide.h)
    assert same(aggw, agg)
    scale = np.abs(dagg).max()
    assert np.max(np.abs(daggw[:, :32] - dagg)) < 1e-12 + 1e-11 * scale      # same directions in another batch geometry
    from concurrent.futures import ThreadPoolExecutor
    cols = list(range(32, 256, 7))                     # 32 of the 224 new columns against the oracle (four passes of eight partials)
    with ThreadPoolExecutor(max_workers=4) as ex:
        outs = list(ex.map(lambda cc: _oracle_cols(orc, x[2:4], yw, cc, ss), [cols[k:k + 8] for k in range(0, 32, 8)]))
    ref = np.concatenate([o[:, 1:] for o in outs], axis=1)
    assert np.max(np.abs(daggw[:, cols] - ref)) < 1e-12 + 1e-10 * np.abs(ref).max()
    c = np.random.default_rng(8).standard_normal(N)
    comb = hb.jvp(np.tensordot(yw, c, axes=([2], [0]))[:, :, None])[:, 0]
    assert np.max(np.abs(comb - daggw @ c)) < 1e-9 * np.abs(daggw @ c).max()
    again = hb.jvp(yw)
    assert np.array_equal(again, daggw)
    hb.primal(x[2:4]); hb.jvp(y)                       # leave the module fixture's state behind


def test_config4_hank_1000x7_T500():
    """BASELINE.json configs[4] at its stated size on one GPU: one-asset HANK 1000x7, T=500 — hank_primal_jvp with 32
    tangents, eight columns against the oracle's restatement of the same family (parity unpinned by construction:
    the family is not in the reference), plus the split schedule."""
    import hank_amd as h
    from examples.solve_hank import build
    from oracle.oracle import Oracle
    m, ss = build(1000, 7, 500)
    P, N = 499, 32
    rng = np.random.default_rng(11)
    t = np.arange(P)
    x = np.stack([ss.vars["r"] + 0.002 * 0.8 ** t, ss.vars["om"] * (1 + 0.01 * 0.7 ** t), ss.vars["Tr"] * (1 - 0.02 * 0.9 ** t)])
    y = rng.standard_normal((3, P, N))
    hb = h.household_block(m)
    hb.set_boundary(ss.value, ss.D)
    agg, dagg = hb.primal_jvp(x, y)
    wd, pdm = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    orc = Oracle(wd.grid, pdm.grid, pdm.transition, m.params.β, m.params.γ, m.params.borrow_cons)
    hcols = [0, 4, 9, 13, 18, 22, 27, 31]             # eight of the 32 columns (one oracle pass of eight partials)
    oagg = _oracle_cols(orc, x[:2], y[:2], hcols, ss, xt=x[2], yt=y[2])
    assert np.max(np.abs(agg - oagg[:, 0])) < 1e-10 * np.abs(oagg[:, 0]).max()
    assert np.max(np.abs(dagg[:, hcols] - oagg[:, 1:])) < 1e-12 + 1e-10 * np.abs(oagg[:, 1:]).max()
    assert same(hb.primal(x), agg) and same(hb.jvp(y), dagg, 1e-12)
    D = hb.dist_seq()
    np.testing.assert_allclose(D.sum(axis=(0, 1)), 1.0, atol=1e-11)
