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
DUAL = os.environ.get("DUAL") == "1"        # the persistent Dual pass (k_xdual_back + k_xfwd<D, true>)
if DUAL:
    os.environ["HANK_SCHEDULE"] = "xcd"
    os.environ["HANK_PRIMAL_MEMO"] = "0"      # every call runs its Float64 sweep (the host-pointer entry would recognise x)
hb = h.household_block(m)
hb.set_boundary(ss.value, ss.D)
y = np.random.default_rng(0).standard_normal((2, P, N))
for _ in range(3):
    if DUAL:
        hb.primal_jvp(x[2:4], y)
    else:
        hb.primal(x[2:4]); hb.jvp(y)
NM = 32
buf = (C.c_ulonglong * (2 * NM * 8 * 12 + 2 * NM * 8 * 16))()
hb._lib.hank_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
assert hb._lib.hank_debug_stamps(hb._ctx, buf) == 0
raw = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
st = raw[:2 * NM * 8 * 12].reshape(2, NM, 8, 12)
wv = raw[2 * NM * 8 * 12:].reshape(2, NM, 8, 16)
tm = hb.last_timings()
sweep_ms = {0: tm["primal_backward" if DUAL else "tangent_backward"]["ms"], 1: tm["tangent_forward"]["ms"]}
names = {0: ["top", "search done", "Y issued", "tile barrier", "X issued", "arrived"] if DUAL else ["top", "srcpoll+BA", "Y issued", "allpoll+BB", "X issued", "arrived"],
         1: ["top", "srcpoll+BA", "gathered", "tile done", "allpoll+BB", "C issued", "batch issd", "stores ackd", "published"]}
order = {0: None, 1: [0, 1, 2, 3, 4, 5, 7, 8, 6]}        # the forward sweep's stamps in time order (7, 8 were added between 5 and 6)
# s_memrealtime: one 100 MHz counter for the whole chip (10 ns per tick): every member of group 0 on one clock.
# Per member: median over the stamped periods of (stamp - earliest top of any member in that period), in ns
for sw, sname in ((0, "backward"), (1, "forward")):
    ns = len(names[sw])
    s_ = (st[sw][:, :, :ns] if order[sw] is None else st[sw][:, :, order[sw]]).astype(float) * 10.0            # [member][period][stamp]
    live = (st[sw][:, :, 0] != 0).all(axis=1)
    t0 = s_[live][:, :, 0].min(axis=0)                      # earliest top per period
    per = np.median(np.diff(s_[live][:, :, 0], axis=1))
    if DUAL and sw == 0:      # once-per-kernel stamps of k_xdual_back (slot 0: 8 entry, 9 joined, 10 prologue done, 11 loop done), member 0
        q = st[0][0][0].astype(float) * 10.0
        print(f"   [backward] member 0, us: entry -> joined {(q[9]-q[8])/1e3:.1f}, prologue {(q[10]-q[9])/1e3:.1f}, loop {(q[11]-q[10])/1e3:.1f}, kernel entry -> loop done {(q[11]-q[8])/1e3:.1f}")
    if os.environ.get("STRIDE"):      # a build with HANK_XSTAMP_STRIDE: the stamped periods are STRIDE apart
        print(f"   [{sname}] member 0: ns per period between stamped periods (stride {os.environ['STRIDE']}): " + " ".join(f"{v / int(os.environ['STRIDE']):.0f}" for v in np.diff(s_[0][:, 0])))
    print(f"--- {sname} sweep, N={N}: {sweep_ms[sw]:.3f} ms, period {per:.0f} ns; ns after the earliest member's top (median over 8 periods)")
    print("   member " + " ".join(f"{n:>11s}" for n in names[sw]) + "   | phase lengths")
    for mbr in range(NM):
        if not live[mbr]:
            continue
        rel = np.median(s_[mbr] - t0[:, None], axis=0)
        print(f"   {mbr:6d} " + " ".join(f"{v:11.0f}" for v in rel) + "   | " + " ".join(f"{v:5.0f}" for v in np.diff(rel)))

# every wave's arrival at the chosen point of the period (backward: the tile barrier of the Y half), ns after the member's top
for sw, sname in ((0, "backward"), (1, "forward")):
    for mbr in ((0, 8, 16, 24, 31) if sw == 0 else (0, 4, 8, 9, 10, 11, 12, 14, 20, 31)):
        if (st[sw][mbr, :, 0] == 0).any():
            continue
        rel = np.median(wv[sw][mbr].astype(float) * 10.0 - st[sw][mbr, :, 0:1].astype(float) * 10.0, axis=0)
        print(f"   [{sname}] member {mbr:2d}: waves' arrival after the member's top, ns: " + " ".join(f"{v:5.0f}" for v in rel[:12]))
