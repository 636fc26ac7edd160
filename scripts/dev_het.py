#!/usr/bin/env python
"""dev: the grid-weighted aggregate (hank_get_grid_aggregates) of the three forward families against each other, column by column."""
import sys
from pathlib import Path

import numpy as np
import torch

torch.cuda.init()
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "scripts"))
from conftest import ks_paths, ks_setup  # noqa: E402
from dev_wide import block  # noqa: E402

for (n_a, n_e, T, N) in ((130, 3, 40, 5), (130, 3, 40, 4), (50, 2, 20, 7), (500, 4, 60, 3)):
    m, ss, orc = ks_setup(n_a, n_e, T)
    P = T - 1
    x, _ = ks_paths(m, ss, "x1", 0.05)
    y = np.random.default_rng(3).standard_normal((2, P, N))
    res = {}
    for sched in ("launch", "xcd", "wide"):
        hb = block(m, sched)
        hb.set_boundary(ss.value, ss.D)
        hb.primal_jvp(x[2:4], y)
        res[sched] = hb.grid_aggregates(N)
        hb.close()
    for sched in ("xcd", "wide"):
        d = np.abs(res[sched][1] - res["launch"][1])
        print(f"{n_a}x{n_e} T={T} N={N} {sched}: primal diff {np.abs(res[sched][0] - res['launch'][0]).max():.2e} | per column max {d.max(axis=0)} | worst period of the worst column {d[:, d.max(axis=0).argmax()].argmax()}", flush=True)
        if d.max() > 1e-9:
            k = d.max(axis=0).argmax()
            print("   diff over t:", (res[sched][1][:, k] - res["launch"][1][:, k])[:12])
