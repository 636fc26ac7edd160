#!/usr/bin/env python
"""dev: where a period of the persistent sweeps goes — s_memtime stamps from the instrumented build (`make stamp`).
    HANK_HIP_LIB=dev/libhank_hip_stamp.so python scripts/dev_xstamps.py [N]"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

torch.cuda.init()
ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("HANK_HIP_LIB", str(ROOT / "dev" / "libhank_hip_stamp.so"))
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import hank_amd as h  # noqa: E402
from conftest import ks_paths, ks_setup  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m, ss, _ = ks_setup(2000, 11, 300)
P = 299
x, Z = ks_paths(m, ss, "x1", 0.01)
hb = h.household_block(m)
hb.set_boundary(ss.value, ss.D)
y = np.random.default_rng(0).standard_normal((2, P, N))
for _ in range(3):
    hb.primal_jvp(x[2:4], y)
buf = (C.c_ulonglong * (2 * 2 * 8 * 12))()
hb._lib.hank_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
assert hb._lib.hank_debug_stamps(hb._ctx, buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(2, 2, 8, 12).astype(np.int64)
names = {0: ["top", "egm_Y done", "dpol+LDS issued", "after WG barrier", "X half issued", "wave ne/2 at LDS barrier", "wave ne-1 at LDS barrier", "run-ahead wave at LDS barrier", "stores drained (vmcnt0)", "WG barrier", "polled (tid0)", "barrier exit"],
         1: ["top", "sources done", "mass point done", "after LDS barrier", "mix+stores+agg issued", "", "wave ne-1 at LDS barrier", "wave ne/2 at LDS barrier", "", "arrived (drain + WG barrier)", "", "barrier exit"]}
tick_ns = 10.0      # s_memtime counts at 100 MHz on gfx950 (constant clock)
for sw, sname in ((0, "backward"), (1, "forward")):
    for mem, mname in ((0, "first member"), (1, "member at a third of the grid")):
        s_ = st[sw, mem]
        order = [0, 1, 2, 6, 7, 3, 4, 9, 11] if sw == 1 else [0, 1, 2, 5, 6, 3, 4, 9, 11]
        print(f"--- {sname}, {mname}, N={N}: ns since period top (median over {s_.shape[0]} periods); period length = next top - top")
        rel = (s_[:, order] - s_[:, [0]]) * tick_ns
        med = np.median(rel, axis=0)
        for k, o in enumerate(order):
            print(f"   {names[sw][o]:28s} {med[k]:9.0f} ns")
        tops = s_[:, 0]
        per = np.abs(np.diff(tops)) * tick_ns
        print(f"   period length (top to top)   {np.median(per):9.0f} ns")
