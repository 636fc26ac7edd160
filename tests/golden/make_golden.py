"""Generates tests/golden/*.npz. The reference holds NO golden vectors for this path and cannot be
run here (no Julia), so these vectors are produced by the CPU oracle (oracle/hank_oracle.c) — they
freeze the restated semantics (regression anchor for the oracle, parity target for the HIP path);
they are not reference outputs ("parity unpinned" in that sense).

    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import ks_paths, ks_setup  # noqa: E402


def main():
    out = Path(__file__).resolve().parent
    m, ss, orc = ks_setup(30, 3, 25)
    P, N = m.compspec.T - 1, 3
    x, Z = ks_paths(m, ss, "x1", 0.05)
    rng = np.random.default_rng(2024)
    y = rng.standard_normal((4, P, N))
    xd = np.zeros((4, P, 1 + N)); xd[..., 0] = x; xd[..., 1:] = y
    st, F, agg = orc.ks_full_function(xd, Z, m.params.α, m.params.δ, ss.vars["KS"], ss.value, ss.D, N)
    assert st == 0
    xr = np.ascontiguousarray(xd[2]); xw = np.ascontiguousarray(xd[3])
    st, pol = orc.backward_iteration(xr, xw, ss.value, N)
    _, Dseq = orc.forward_iteration(pol, ss.D, N, return_D=True)
    wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
    np.savez_compressed(out / "ks_30x3_T25_N3.npz", a_grid=wd.grid, z_grid=pd_.grid, Pi=pd_.transition,
                        beta=m.params.β, gamma=m.params.γ, borrow_cons=m.params.borrow_cons, alpha=m.params.α,
                        delta=m.params.δ, T=m.compspec.T, ss_value=ss.value, ss_D=ss.D, KS_ss=ss.vars["KS"],
                        x=x, Z=Z, y=y, F=F, agg=agg, policy_seq=pol, D_last=Dseq[-1])

    # granular forward step: clamps, exact grid hits (searchsortedfirst ties), non-monotone policy
    grid = wd.grid
    pol_e = rng.uniform(-1.0, grid[-1] + 5.0, (30, 3))
    pol_e[0, 0] = grid[0]; pol_e[1, 0] = grid[5]; pol_e[2, 0] = grid[-1]; pol_e[3, 1] = grid[0] - 1.0
    pol_e[4, 2] = grid[-1] + 1.0; pol_e[5, 2] = 0.5 * (grid[3] + grid[4])
    dpol_e = rng.standard_normal((30, 3, N))
    Dp = rng.uniform(0, 1, (30, 3)); Dp /= Dp.sum()
    dDp = rng.standard_normal((30, 3, N)) * 1e-2
    Dn = orc.transition_step(np.concatenate([pol_e[..., None], dpol_e], -1), np.concatenate([Dp[..., None], dDp], -1), N)
    np.savez_compressed(out / "forward_step_edge_30x3_N3.npz", policy=pol_e, dpolicy=dpol_e, D_prev=Dp, dD_prev=dDp, D_new=Dn)
    print("golden written")


if __name__ == "__main__":
    main()
