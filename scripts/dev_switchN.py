#!/usr/bin/env python
"""dev: what does a change of the batch width cost? (J̅ assembly at N=256 followed by Newton's N=1, and back)
The tangent workspaces are kept in a small most-recently-used cache per context and the hipGraphs of a width are captured
the first time a schedule runs at it: the first call at a width pays for allocation (and capture), a return to it nothing."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

torch.cuda.init()
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import hank_amd as h  # noqa: E402
from conftest import ks_paths, ks_setup  # noqa: E402

m, ss, _ = ks_setup(2000, 11, 300)
P = 299
x, Z = ks_paths(m, ss, "x1", 0.01)
hb = h.household_block(m)
hb.set_boundary(ss.value, ss.D)
hb.primal(x[2:4])
rng = np.random.default_rng(0)
ys = {N: rng.standard_normal((2, P, N)) for N in (1, 32, 256)}
for label, N in (("first", 256), ("first", 1), ("first", 32), ("again", 256), ("again", 1), ("again", 32), ("again", 256), ("again", 1)):
    t0 = time.perf_counter()
    hb.jvp(ys[N])
    t1 = time.perf_counter()
    hb.jvp(ys[N])
    t2 = time.perf_counter()
    st = hb.stats()
    print(f"N={N:4d} {label}: {1e3 * (t1 - t0):8.2f} ms (the next call at the same width {1e3 * (t2 - t1):7.2f} ms) | workspaces allocated so far "
          f"{st['tangent_workspaces_allocated']}, graphs captured {st['graphs_captured']}")
