#!/usr/bin/env python
"""One-asset HANK (examples/one_asset_hank.yaml; NOT in the reference) through the same
sequence as examples/solve_transition.py: YAML -> calibrate the bond supply -> steady state (host) -> J̅ (batched
unit-tangent JVPs on the GPU, household family HANK_VF_ONE_ASSET_HANK) -> NewtonRaphsonHANK -> the perfect-foresight
response to a monetary-policy shock.

    python examples/solve_hank.py [--n-a 1000 --n-e 7 --T 500 --shock 0.0025]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/solve_hank.py
        one process per GPU (RCCL): the unit-tangent chunks of the Jacobian assembly are shared out over the ranks and
        all-gathered; the Newton iteration itself (one tangent per inner step) runs replicated on every rank."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def build(n_a=1000, n_e=7, T=500, spec="one_asset_hank.yaml"):
    """spec: one_asset_hank.yaml (asset-market clearing, one heterogeneous variable) or one_asset_hank_goods.yaml (goods-market
    clearing: savings AND consumption aggregated by the device sweeps)."""
    import hank_amd as h
    from hank_amd import OneAssetHANK as oa
    ov = {"T": T, "dimensions": {"wealth": {"n": n_a}, "productivity": {"n": n_e}}}
    m = h.build_model_from_yaml(str(ROOT / "examples" / spec), overrides=ov)
    m.params.B = oa.calibrate_bond_supply(m)
    ss, _ = h.get_SteadyStates(m)
    return m, ss


def solve(n_a=1000, n_e=7, T=500, shock=0.0025, rho=0.6, eps=1e-9, verbose=False, inner="fixed_point", jacobian="toeplitz",
          spec="one_asset_hank.yaml"):
    import hank_amd as h
    import hank_amd.parallel  # noqa: F401  (pulls in torch before the clock starts)
    t0 = time.perf_counter()
    m, ss = build(n_a, n_e, T, spec)
    t_ss = time.perf_counter() - t0
    P = T - 1
    ei = shock * rho ** np.arange(P)
    keys = h.vars_of_type(m, "endogenous")
    x0 = np.tile(np.array([ss.vars[k] for k in keys]), P)
    t0 = time.perf_counter()
    J = h.getSteadyStateJacobian(ss, m, method=jacobian)
    t_jac = time.perf_counter() - t0
    h.y_Iteration.total_jvps = 0
    h.y_Iteration.setup_s = 0.0
    t0 = time.perf_counter()
    x = h.NewtonRaphsonHANK(x0, J, {"ei": ei}, m, ss, ss, ε=eps, verbose=verbose, inner=inner)
    t_newton = time.perf_counter() - t0
    lin = h.LinearizedFunction(x, {"ei": ei}, m, ss, ss)
    X = x.reshape(len(keys), P, order="F")
    out = {"model": "one-asset HANK", "spec": spec, "grid": f"{n_a}x{n_e}", "T": T, "shock": f"ei_t = {shock}*{rho}^(t-1)",
           "B": m.params.B, "calibrate_and_steady_state_s": round(t_ss, 3), "ss_jacobian_s": round(t_jac, 3),
           "newton_s": round(t_newton, 3), "preconditioner_setup_s": round(h.y_Iteration.setup_s, 3),      # (inside newton_s: J̅⁻¹ on the device, once per J̅)
           "newton_iterations": h.NewtonRaphsonHANK.iterations,
           "jvps": h.y_Iteration.total_jvps, "residual_norm": float(np.linalg.norm(lin.Fx)),
           "wall_to_converged_path_s": round(t_jac + t_newton, 3), "inner": inner, "jacobian": jacobian,
           "impact": {k: float(X[j, 0] - ss.vars[k]) for j, k in enumerate(keys)}}
    return out, x, m, ss


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-a", type=int, default=1000)
    ap.add_argument("--n-e", type=int, default=7)
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--shock", type=float, default=0.0025)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--inner", default="fixed_point", choices=["fixed_point", "krylov"])
    ap.add_argument("--jacobian", default="toeplitz", choices=["toeplitz", "columns"])
    ap.add_argument("--spec", default="one_asset_hank.yaml", choices=["one_asset_hank.yaml", "one_asset_hank_goods.yaml"])
    a = ap.parse_args()
    import os
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
    out = solve(a.n_a, a.n_e, a.T, a.shock, verbose=a.verbose and rank == 0, inner=a.inner, jacobian=a.jacobian, spec=a.spec)[0]
    out["n_gpus"] = world
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
