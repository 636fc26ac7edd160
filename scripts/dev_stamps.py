"""diagnostic: per-phase cycle shares of the backward cluster sweep (HANK_STAMPS build)."""
import ctypes as C, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import hank_amd.hip as hip
hip._LIB_PATH = hip._LIB_PATH.with_name('libhank_hip_stamps.so')
import hank_amd as h
from conftest import ks_setup, ks_paths
m, ss, _ = ks_setup(2000, 11, 300)
hb = h.household_block(m); hb.set_boundary(ss.value, ss.D)
x, Z = ks_paths(m, ss, "x1", 0.01)
hb.primal(x[2:4])
y = np.random.default_rng(0).standard_normal((2, 299, 32))
for _ in range(3): hb.jvp(y)
lib = hip.load_library()
out = (C.c_ulonglong * (256 * 8))()
lib.hank_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
n = lib.hank_debug_stamps(hb._ctx, out, 256)
a = np.array(out[:n * 8], dtype=np.float64).reshape(n, 8) / 299.0   # shader cycles per period
names = ["loop top", "coef issue + X compute", "ring check + granule stores", "barrier (X reads done)", "gather poll", "Y compute + dpol store", "end barrier + done flag"]
print("per-period shader cycles (mean / min / max over workgroups):")
for k, nm in enumerate(names): print(f"  {nm:28s} {a[:,k].mean():9.0f} {a[:,k].min():9.0f} {a[:,k].max():9.0f}")
print("  sum", a[:, :7].sum(axis=1).mean(), hb.last_timings())
