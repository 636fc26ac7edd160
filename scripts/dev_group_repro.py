#!/usr/bin/env python
"""dev: two contexts on ONE GPU driven from two host threads (parallel.DeviceGroup), many rounds: every round's columns against the
single-context result; prints the contexts' counters when a round differs.   python scripts/dev_group_repro.py [rounds]"""
import sys
from pathlib import Path

import numpy as np
import torch

torch.cuda.init()
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import hank_amd as h  # noqa: E402
from conftest import ks_paths, ks_setup  # noqa: E402
from hank_amd.parallel import DeviceGroup  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
m, ss, _ = ks_setup(50, 2, 20)
P = 19
x, _ = ks_paths(m, ss, "x1", 0.05)
y = np.random.default_rng(4).standard_normal((2, P, 7))
wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
args = (wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
hb0 = h.HouseholdBlock(*args)
hb1 = h.HouseholdBlock(*args, device=0)
for hb in (hb0, hb1):
    hb.set_boundary(ss.value, ss.D)
a0 = hb0.primal(x[2:4]); hb1.primal(x[2:4])
d0 = hb0.jvp(y)
bad = 0
g = None
for k in range(rounds):
    if g is None or k % 2 == 0:          # a FRESH second context every other round: its first call allocates its workspaces beside the other's sweeps
        if g is not None:
            g.close()
        g = DeviceGroup(hb1, [0, 0])
        g.set_boundary(ss.value, ss.D)
    g.primal(x[2:4] * (1.0 + 1e-6 * (k % 3)))
    g.primal(x[2:4])
    dg = g.jvp(y)
    err = np.max(np.abs(dg - d0)) / np.abs(d0).max()
    if err > 1e-12:
        bad += 1
        rows = np.argwhere(np.abs(dg - d0) > 1e-9 * np.abs(d0).max())
        print(f"round {k}: rel err {err:.3e}; differing (row, col): {rows[:8].tolist()} ...; stats {[b.stats() for b in g.blocks]}; families {[b.info()['last_tangent_family_name'] for b in g.blocks]}", flush=True)
print(f"{bad} of {rounds} rounds differ; stats {[b.stats() for b in g.blocks]}")
