#!/usr/bin/env python
"""dev: where a period of the on-chip wide sweeps goes — s_memtime stamps of wave 0 of workgroup 0 (`make stamp`).
    python scripts/dev_wstamps.py [N]"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

torch.cuda.init()
ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("HANK_HIP_LIB", str(ROOT / "dev" / "libhank_hip_stamp.so"))
os.environ["HANK_SCHEDULE"] = "wide"
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import hank_amd as h  # noqa: E402
from conftest import ks_paths, ks_setup  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m, ss, _ = ks_setup(2000, 11, 300)
P, NE = 299, 11
x, Z = ks_paths(m, ss, "x1", 0.01)
wd, pd_ = m.heterogeneity["wealth"], m.heterogeneity["productivity"]
hb = h.HouseholdBlock(wd.grid, pd_.grid, pd_.transition, m.params.β, m.params.γ, m.params.borrow_cons, m.compspec.T)
hb.set_boundary(ss.value, ss.D)
y = np.random.default_rng(0).standard_normal((2, P, N))
hb.primal(x[2:4])
for _ in range(2):
    hb.jvp(y)
buf = (C.c_ulonglong * (2 * 4 * 64))()
hb._lib.hank_debug_wstamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
assert hb._lib.hank_debug_wstamps(hb._ctx, buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(2, 4, 64)
tm = hb.last_timings()
print(f"N={N}: backward {tm['tangent_backward']['ms']:.3f} ms, forward {tm['tangent_forward']['ms']:.3f} ms ({1e3 * tm['tangent_backward']['ms'] / P:.1f} / {1e3 * tm['tangent_forward']['ms'] / P:.1f} us per period)")
for sw, name in ((0, "backward"), (1, "forward")):
    s = st[sw].astype(float)
    if not s[:, 0].all():
        continue
    tops = s[:, 0]
    print(f"--- {name}: ticks between stamped period tops: {np.abs(np.diff(tops))}  (s_memtime ticks)")
    for p in range(4):
        row = s[p]
        cols = []
        if sw == 0:       # backward: 0 top, 1 mixed; per column 2+4e written, 3+4e past the barrier, 5+4e gathered, 4+4e stored
            mix = row[1] - row[0]
            for e in range(NE):
                b = 2 + 4 * e
                prev = row[1] if e == 0 else row[4 + 4 * (e - 1)]
                cols.append((row[b] - prev, row[b + 1] - row[b], row[b + 3] - row[b + 1], row[b + 2] - row[b + 3]))
            print(f"   period {p}: mix {mix:.0f} | per column (loads issued + X + LDS write, barrier wait, gather + dg, dV + stores): " + " ".join(f"[{a:.0f} {b_:.0f} {c:.0f} {d:.0f}]" for a, b_, c, d in cols))
        else:             # forward: 0 top, 1 columns done (then the mixing); per column 2+4e pushed, 3+4e past the barrier, 4+4e gathered
            for e in range(NE):
                b = 2 + 4 * e
                prev = row[0] if e == 0 else row[4 + 4 * (e - 1)]
                cols.append((row[b] - prev, row[b + 1] - row[b], row[b + 2] - row[b + 1]))
            nxt = s[p + 1][0] if p + 1 < 4 else np.nan
            print(f"   period {p}: mix {nxt - row[1]:.0f} | per column (loads issued + push, barrier wait, gather): " + " ".join(f"[{a:.0f} {b_:.0f} {c:.0f}]" for a, b_, c in cols))
