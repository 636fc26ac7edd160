"""Oracle vs the committed golden vectors (regression anchor; see tests/golden/make_golden.py)."""
from pathlib import Path

import numpy as np

G = Path(__file__).resolve().parent / "golden"


def test_oracle_reproduces_ks_golden(oracle_mod):
    g = np.load(G / "ks_30x3_T25_N3.npz")
    orc = oracle_mod.Oracle(g["a_grid"], g["z_grid"], g["Pi"], float(g["beta"]), float(g["gamma"]), float(g["borrow_cons"]))
    P, N = int(g["T"]) - 1, g["y"].shape[2]
    xd = np.zeros((4, P, 1 + N)); xd[..., 0] = g["x"]; xd[..., 1:] = g["y"]
    st, F, agg = orc.ks_full_function(xd, g["Z"], float(g["alpha"]), float(g["delta"]), float(g["KS_ss"]), g["ss_value"], g["ss_D"], N)
    assert st == 0
    assert np.array_equal(F, g["F"]) and np.array_equal(agg, g["agg"])


def test_oracle_reproduces_forward_edge_golden(oracle_mod):
    g = np.load(G / "forward_step_edge_30x3_N3.npz")
    k = np.load(G / "ks_30x3_T25_N3.npz")
    orc = oracle_mod.Oracle(k["a_grid"], k["z_grid"], k["Pi"], float(k["beta"]), float(k["gamma"]), float(k["borrow_cons"]))
    N = g["dpolicy"].shape[2]
    Dn = orc.transition_step(np.concatenate([g["policy"][..., None], g["dpolicy"]], -1),
                             np.concatenate([g["D_prev"][..., None], g["dD_prev"]], -1), N)
    assert np.array_equal(Dn, g["D_new"])
    # mass is conserved and the partials of the total mass are those of the input
    assert abs(Dn[..., 0].sum() - 1.0) < 1e-14
    np.testing.assert_allclose(Dn[..., 1:].sum(axis=(0, 1)), g["dD_prev"].sum(axis=(0, 1)), atol=1e-13)
