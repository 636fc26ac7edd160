import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import ks_setup, ks_paths
import hank_amd as h
m, ss, orc = ks_setup(50, 2, 100)
hb = h.household_block(m)
hb.set_boundary(ss.value, ss.D)
x, Z = ks_paths(m, ss, "x1", 0.05)
P = m.compspec.T - 1
agg = hb.primal(x[2:4])
N = 3
xr = np.zeros((P, 1+N)); xw = np.zeros((P, 1+N)); xr[:,0] = x[2]; xw[:,0] = x[3]
rng = np.random.default_rng(0)
dx = rng.standard_normal((2, P, N))
xr[:,1:] = dx[0]; xw[:,1:] = dx[1]
st, oagg, opol = orc.household_block(xr, xw, ss.value, ss.D, N)
print('status', st, 'agg err', np.abs(agg - oagg[:,0]).max(), np.abs(agg).max())
pol = hb.policy_seq()
print('pol err', np.abs(pol.transpose(2,0,1) - opol[...,0]).max())
dagg = hb.jvp(dx)
print('dagg err', np.abs(dagg - oagg[:,1:]).max(), np.abs(dagg).max())
dpol = hb.dpolicy_seq(N)
print('dpol err', np.abs(dpol.transpose(2,0,1,3) - opol[...,1:]).max(), np.abs(dpol).max())
print(hb.last_timings())
