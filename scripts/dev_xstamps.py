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
    hb.primal(x[2:4]); hb.jvp(y)
buf = (C.c_ulonglong * (2 * 2 * 8 * 12 + 2 * 2 * 8 * 16))()
hb._lib.hank_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
assert hb._lib.hank_debug_stamps(hb._ctx, buf) == 0
raw = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
st = raw[:2 * 2 * 8 * 12].reshape(2, 2, 8, 12)
wv = raw[2 * 2 * 8 * 12:].reshape(2, 2, 8, 16)
names = {0: ["top", "source poll + LDS barrier", "Y half: gathers, dV, dpol + tile store issued", "all-member poll + LDS barrier", "X half: mix, ds, state store issued", "arrived (stores drained, WG barrier)"],
         1: ["top", "source poll + LDS barrier", "sources gathered", "mass point + tile store issued", "all-member poll + LDS barrier", "mix, state store, aggregate issued", "arrived + published",
             "  sync wave: source poll done", "  sync wave: all-member poll done", "  wave ne/2 at top (its prefetch landed)", "  wave ne-1 at top", "  sync wave: published"]}
tm = hb.last_timings()
sweep_ms = {0: tm["tangent_backward"]["ms"], 1: tm["tangent_forward"]["ms"]}
for sw, sname in ((0, "backward"), (1, "forward")):
    for mem, mname in ((0, "first member"), (1, "member at a third of the grid")):
        s_ = st[sw, mem]
        order = list(range(len(names[sw])))
        tops = s_[:, 0]
        # s_memtime's unit is calibrated on the sweep itself: the median top-to-top distance is one period of its event time
        tick_ns = 1e6 * sweep_ms[sw] / P / float(np.median(np.abs(np.diff(tops))))
        print(f"--- {sname}, {mname}, N={N}: ns since period top (median over {s_.shape[0]} periods); period length = next top - top")
        rel = (s_[:, order] - s_[:, [0]]) * tick_ns
        med = np.median(rel, axis=0)
        for k, o in enumerate(order):
            print(f"   {names[sw][o]:28s} {med[k]:9.0f} ns")
        per = np.abs(np.diff(tops)) * tick_ns
        print(f"   period length (top to top)   {np.median(per):9.0f} ns   (1 tick = {tick_ns:.3f} ns)")

# the two stamped members on one clock (s_memrealtime: one 100 MHz counter for the chip): who is ahead?
for sw, sname in ((0, "backward"), (1, "forward")):
    d = (st[sw, 0, :, 0] - st[sw, 1, :, 0]).astype(float)
    tick = 1e6 * sweep_ms[sw] / P / float(np.median(np.abs(np.diff(st[sw, 0, :, 0]))))
    print(f"{sname}: first member's top minus the other's top, same period: median {np.median(d) * tick:.0f} ns (min {d.min() * tick:.0f}, max {d.max() * tick:.0f})")

# every wave's arrival at the end of the first half (Y half / gather + mass point + tile store), relative to wave 0
for sw, sname in ((0, "backward"), (1, "forward")):
    for mem, mname in ((0, "first member"), (1, "member at a third of the grid")):
        rel = (wv[sw, mem, :, :11] - wv[sw, mem, :, [0]].reshape(-1, 1)) * 10.0
        print(f"{sname}, {mname}: waves 0..10 at the end of the first half, ns after wave 0 (median): " + " ".join(f"{v:.0f}" for v in np.median(rel, axis=0)))

