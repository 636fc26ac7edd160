import ctypes as C, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import hank_amd.hip as hip
hip._LIB_PATH = hip._LIB_PATH.with_name('libhank_hip_stamps.so')
import hank_amd as h
from conftest import ks_setup, ks_paths
m, ss, _ = ks_setup(2000, 11, 300)
hb = h.household_block(m); hb.set_boundary(ss.value, ss.D)
x, Z = ks_paths(m, ss, "x1", 0.01)
lib = hip.load_library()
NB = 300
hb.primal(x[2:4])
y = np.random.default_rng(0).standard_normal((2, 299, 32))
hb.jvp(y)
assert lib.hank_debug_stamps_alloc(NB) == 0
hb.jvp(y)      # the stamps of the LAST backward launch (t=0) survive
out = (C.c_ulonglong * (NB * 16 * 8))()
lib.hank_debug_stamps_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert lib.hank_debug_stamps_read(out, NB) == 0
a = np.array(out[:], dtype=np.float64).reshape(NB, 16, 8)[:, :11, :6]
a = a[a[:, 0, 0] > 0]
t0 = a[:, :, 0].min()
print("blocks with stamps", a.shape[0])
names = ["start", "loads issued", "coef arrived", "gathers arrived", "computed+stored+LDS", "after barrier"]
for k in range(6):
    print(f"{names[k]:22s} mean {np.mean(a[:,:,k]-t0):9.0f}  (delta prev {np.mean(a[:,:,k]-a[:,:,max(k-1,0)]):8.0f})  min {np.min(a[:,:,k]-t0):9.0f} max {np.max(a[:,:,k]-t0):9.0f}")
print("wave start spread (cycles):", a[:, :, 0].max() - t0)
