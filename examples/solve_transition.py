#!/usr/bin/env python
"""The RunMain.jl sequence (RunMain.jl:35-55, as intended — SURVEY.md §3.1) on the MI355X build:

    YAML -> model -> steady state (host) -> J̅ (its Toeplitz structure from n_hh backward tangent sweeps on the GPU; or n unit tangents)
         -> NewtonRaphsonHANK (Boehl y-iteration; one hank_jvp per inner iteration) -> converged path

    python examples/solve_transition.py [--n-a 500 --n-e 4 --T 300 --shock 0.01]

`--shock 0.8` is RunMain's own Z_t = 1 + 0.8^t (an 80 % TFP jump; may not converge with the
reference's fixed damping α = 0.5 — SURVEY.md App. B item 5)."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def solve(n_a=500, n_e=4, T=300, shock=0.01, eps=1e-9, verbose=False, cold=False, inner="fixed_point", jacobian="toeplitz"):
    """cold=True: the steady state is solved here from the YAML guesses (value iteration and stationary distribution on
    the device where one is present) instead of coming from the test fixtures' cache."""
    import hank_amd as h
    import hank_amd.parallel  # noqa: F401  (pulls in torch before the clocks start)
    from conftest import ks_setup
    t0 = time.perf_counter()
    if cold:
        ov = {"T": T, "dimensions": {"wealth": {"n": n_a}, "productivity": {"n": n_e}}}
        m = h.build_model_from_yaml(str(ROOT / "examples" / "krusell_smith.yaml"), overrides=ov)
        ss, _ = h.get_SteadyStates(m)
    else:
        m, ss, _ = ks_setup(n_a, n_e, T)
    t_ss = time.perf_counter() - t0
    P = T - 1
    Z = 1.0 + shock * 0.8 ** np.arange(1, P + 1)                       # RunMain.jl:22-23
    x0 = np.tile(np.array([ss.vars[k] for k in ("Y", "KS", "r", "w")]), P)   # SteadyState.jl:277-278
    t0 = time.perf_counter()
    J = h.getSteadyStateJacobian(ss, m, method=jacobian)
    t_jac = time.perf_counter() - t0
    h.y_Iteration.total_jvps = 0
    h.y_Iteration.setup_s = 0.0
    t0 = time.perf_counter()
    x = h.NewtonRaphsonHANK(x0, J, {"Z": Z}, m, ss, ss, ε=eps, verbose=verbose, inner=inner)
    t_newton = time.perf_counter() - t0
    lin = h.LinearizedFunction(x, {"Z": Z}, m, ss, ss)
    return {"grid": f"{n_a}x{n_e}", "T": T, "shock": f"Z_t = 1 + {shock}*0.8^t", "steady_state_s": round(t_ss, 3),
            "ss_jacobian_s": round(t_jac, 3), "newton_s": round(t_newton, 3), "preconditioner_setup_s": round(h.y_Iteration.setup_s, 3),
            "newton_iterations": h.NewtonRaphsonHANK.iterations, "jvps": h.y_Iteration.total_jvps, "residual_norm": float(np.linalg.norm(lin.Fx)),
            "wall_to_converged_path_s": round(t_jac + t_newton, 3), "steady_state": "cold start" if cold else "cached",
            "inner": inner, "jacobian": jacobian}, x


def solve_permanent(n_a=200, n_e=3, T=150, Z_end=1.03, eps=1e-9, verbose=False):
    """The two-steady-state scenario of the reference YAML (`ending:` block, KrusellSmith.yaml:109-116): TFP moves
    to Z_end for good in period 1. The path starts from the initial steady state (KS_0, D_0 = ss_initial), the terminal
    value is the ending steady state's (BackwardIteration.jl:85), Newton starts at the ending steady state repeated and
    uses the sequence-space Jacobian there."""
    import hank_amd as h
    import hank_amd.parallel  # noqa: F401
    ov = {"T": T, "dimensions": {"wealth": {"n": n_a}, "productivity": {"n": n_e}},
          "steady_states": {"ending": {"fixed": {"Z": Z_end}, "guesses": {"r": 0.04, "w": 1.0, "Y": 1.5, "KS": 3.5}}}}
    m = h.build_model_from_yaml(str(ROOT / "examples" / "krusell_smith.yaml"), overrides=ov)
    t0 = time.perf_counter()
    ss_i, ss_e = h.get_SteadyStates(m)
    t_ss = time.perf_counter() - t0
    P = T - 1
    Z = np.full(P, float(Z_end))
    x0 = np.tile(np.array([ss_e.vars[k] for k in ("Y", "KS", "r", "w")]), P)
    t0 = time.perf_counter()
    J = h.getSteadyStateJacobian(ss_e, m)
    x = h.NewtonRaphsonHANK(x0, J, {"Z": Z}, m, ss_i, ss_e, ε=eps, verbose=verbose)
    t_solve = time.perf_counter() - t0
    lin = h.LinearizedFunction(x, {"Z": Z}, m, ss_i, ss_e)
    return {"grid": f"{n_a}x{n_e}", "T": T, "shock": f"Z: 1 -> {Z_end} for good", "steady_states_s": round(t_ss, 3),
            "newton_iterations": h.NewtonRaphsonHANK.iterations, "residual_norm": float(np.linalg.norm(lin.Fx)),
            "wall_to_converged_path_s": round(t_solve, 3), "KS_start": ss_i.vars["KS"], "KS_end": ss_e.vars["KS"],
            "KS_path_first_last": [float(x.reshape(4, P, order="F")[1, 0]), float(x.reshape(4, P, order="F")[1, -1])]}, x, ss_i, ss_e


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-a", type=int, default=500)
    ap.add_argument("--n-e", type=int, default=4)
    ap.add_argument("--T", type=int, default=300)
    ap.add_argument("--shock", type=float, default=0.01)
    ap.add_argument("--permanent", type=float, default=None, metavar="Z_END", help="two-steady-state scenario: Z jumps to Z_END for good")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--cold", action="store_true", help="solve the steady state from the YAML guesses (no fixture)")
    ap.add_argument("--inner", default="fixed_point", choices=["fixed_point", "krylov"], help="y-iteration: the reference's damped fixed point or GMRES on J(x) preconditioned by the steady-state Jacobian")
    ap.add_argument("--jacobian", default="toeplitz", choices=["toeplitz", "columns"])
    a = ap.parse_args()
    import os
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:       # under torch.distributed.run: the Jacobian assembly is shared out over the ranks (one GPU each)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
    if a.permanent is not None:
        out = solve_permanent(a.n_a, a.n_e, a.T, a.permanent, verbose=a.verbose)[0]
    else:
        out, x = solve(a.n_a, a.n_e, a.T, a.shock, verbose=a.verbose, cold=a.cold, inner=a.inner, jacobian=a.jacobian)
    out["n_gpus"] = world
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()

