"""dev: in-kernel time stamps of the forward tangent kernel, block 100 (library built with -DHANK_STAMPS)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import hank_amd as h
from hank_amd import hip
from conftest import ks_setup, ks_paths
m, ss, _ = ks_setup(2000, 11, 300)
x, Z = ks_paths(m, ss, "x1", 0.01)
hb = h.household_block(m); hb.set_boundary(ss.value, ss.D)
lib = hip.load_library()
N = 32
y = np.random.default_rng(0).standard_normal((2, 299, N))
hb.primal_jvp(x[2:4], y)
out = (C.c_ulonglong * 32)()
lib.hank_debug_stamps(out, 1)
R = 5
for _ in range(R): hb.primal_jvp(x[2:4], y)
lib.hank_debug_stamps(out, 0)
v = np.array(list(out), dtype=float)[:5] / (R * 299)
names = ["start -> segment bounds loaded", "source loop (loads + LDS adds)", "tile barrier", "mix + dD store", "aggregate reduce (2 barriers)"]
print({n: round(t) for n, t in zip(names, v)}, "sum", round(v.sum()), "ticks per launch (x10 ns)")
print(hb.last_timings()["dual_forward"])
