#!/usr/bin/env python
"""dev: where a period of the slab sweeps goes — s_memrealtime stamps (10 ns) from the instrumented build (`make stamp`).
    HANK_XTAN=slab python scripts/dev_wstamps.py [N]"""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

torch.cuda.init()
ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("HANK_HIP_LIB", str(ROOT / "dev" / "libhank_hip_stamp.so"))
os.environ.setdefault("HANK_XTAN", "slab")
os.environ["HANK_SCHEDULE"] = "xcd"
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import hank_amd as h  # noqa: E402
from conftest import ks_paths, ks_setup  # noqa: E402

m, ss, _ = ks_setup(2000, 11, 300)
P = 299
x, Z = ks_paths(m, ss, "x1", 0.01)
hb = h.household_block(m)
hb.set_boundary(ss.value, ss.D)
hb._lib.hank_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
names = {0: ["top", "source poll done", "Y done (dpol stores issued)", "reader poll done", "X done (state stores issued)", "drained + published", "past the workgroup barrier",
             "  loader: ring slot written", "  loader: fetches + touches issued"],
         1: ["top", "source poll done", "all columns gathered", "reader poll done", "mix + stores issued", "drained + published", "past the workgroup barrier",
             "  first column summed", "  loader: at the barrier"]}
for N in [int(v) for v in sys.argv[1:]] or [32]:
    y = np.random.default_rng(0).standard_normal((2, P, N))
    for _ in range(3):
        hb.primal(x[2:4]); hb.jvp(y)
    buf = (C.c_ulonglong * (2 * 2 * 8 * 12 + 2 * 2 * 8 * 16))()
    assert hb._lib.hank_debug_stamps(hb._ctx, buf) == 0
    raw = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    st = raw[:2 * 2 * 8 * 12].reshape(2, 2, 8, 12)
    tm = hb.last_timings()
    print(f"=== N={N}: backward {tm['tangent_backward']['ms']:.3f} ms, forward {tm['tangent_forward']['ms']:.3f} ms", flush=True)
    for sw, sname in ((0, "backward"), (1, "forward")):
        for mem, mname in ((0, "first member"), (1, "member at a third")):
            s_ = st[sw, mem]
            rel = (s_[:, :9] - s_[:, [0]]) * 10.0
            med = np.median(rel, axis=0)
            per = np.abs(np.diff(s_[:, 0])) * 10.0
            print(f"--- {sname}, {mname}: ns after the period's top (median of 8 periods); period = {np.median(per):.0f} ns")
            for k, nm in enumerate(names[sw]):
                print(f"   {nm:36s} {med[k]:9.0f}")
hb.close()
