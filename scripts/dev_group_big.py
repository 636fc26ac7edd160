#!/usr/bin/env python
"""dev: two contexts on ONE GPU at the headline grid, driven from two host threads (parallel.DeviceGroup): every call of each is a
persistent sweep that wants the whole chip. Counts fallbacks (a sweep whose groups did not form because another context's sweep
was resident beside it) and checks the columns against the single-context result.   python scripts/dev_group_big.py [rounds]"""
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

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m, ss, _ = ks_setup(2000, 11, 300)
P = 299
x, _ = ks_paths(m, ss, "x1", 0.01)
y = np.random.default_rng(4).standard_normal((2, P, 64))
hb0 = h.household_block(m)
hb0.set_boundary(ss.value, ss.D)
hb0.primal(x[2:4])
d0 = np.concatenate([hb0.jvp(y[:, :, :32]), hb0.jvp(y[:, :, 32:])], axis=1)
g = DeviceGroup(hb0, [0, 0])
g.set_boundary(ss.value, ss.D)
bad = 0
for k in range(rounds):
    g.primal(x[2:4] * (1.0 + 1e-6 * (k % 3)))
    g.primal(x[2:4])
    dg = g.jvp(y)
    err = np.max(np.abs(dg - d0)) / np.abs(d0).max()
    if err > 1e-11:
        bad += 1
        print(f"round {k}: rel err {err:.3e}", flush=True)
print(f"{bad} of {rounds} rounds differ; stats {[ {k2: v for k2, v in b.stats().items() if k2 in ('schedule', 'fallbacks', 'sweep_launches')} for b in g.blocks]}")
