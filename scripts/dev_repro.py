import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import ks_setup, ks_paths
import hank_amd as h
mA, ssA, _ = ks_setup(50, 2, 100)
mB, ssB, _ = ks_setup(30, 3, 25)
xA, ZA = ks_paths(mA, ssA, "x1", 0.05)
xB, ZB = ks_paths(mB, ssB, "x1", 0.05)
hbA = h.household_block(mA)
hbA.set_boundary(ssA.value, ssA.D); hbA.primal(xA[2:4]); print('A primal ok')
hbB = h.household_block(mB)
print('B created'); hbA.check(); print('A check ok')
hbB.set_boundary(ssB.value, ssB.D); hbB.primal(xB[2:4]); print('B primal ok'); hbA.check(); print('A check ok')
pol = hbB.policy_seq(); hbA.check(); print('A check ok after B policy_seq')
Do, a = hbB.forward_step(pol[:,:,0], ssB.D); hbA.check(); print('A check ok after B forward_step')
hbA.primal(xA[2:4]); print('A primal ok 2')
