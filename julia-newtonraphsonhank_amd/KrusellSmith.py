"""Krusell-Smith (1998) model plugin — the native counterpart of the reference's KrusellSmith.jl.

In the reference `ValueFunction(value_next, xVals, model)` is Julia code executed for every
period of BackwardIteration (KrusellSmith.jl:43-83). Here the YAML name "ValueFunction" resolves
to a *kernel family* of libhank_hip (HANK_VF_KRUSELL_SMITH): the EGM step lives in
csrc/hank_kernels.h (egm_X / egm_Y and their tangent forms) and is never evaluated on the host
along the transition path.

The plugin object also declares which entries of xVals the household reads (r, w —
KrusellSmith.jl:53-54) and carries `host_steady_state_step`, a Float64 numpy EGM step used ONLY by
the host steady-state solver (SteadyState.jl stays on the host per the north star). The transition
path (BackwardIteration / ForwardIteration / JVP) never calls it.
"""
from __future__ import annotations

import numpy as np

from .ModelParser import register_function

HANK_VF_KRUSELL_SMITH = 0


def exogenousZ(T: int, ρ: float = 0.9, σ: float = 0.1) -> np.ndarray:
    """AR(1) path for aggregate productivity starting at Z0 = 1 (KrusellSmith.jl:14-20); random,
    unseeded, like the reference (np.random.seed to reproduce)."""
    Z = np.ones(T)
    for t in range(1, T):
        Z[t] = ρ * Z[t - 1] + σ * np.sqrt(1 - ρ ** 2) * np.random.randn()
    return Z


class _KSValueFunction:
    name = "ValueFunction"
    value_fn_id = HANK_VF_KRUSELL_SMITH
    household_inputs = ("r", "w")   # rows of xVals the household block reads
    outputs = ("KD", "C")           # one policy per heterogeneous variable a model may list: the reference's KD, and consumption (the
                                    # c_grid of KrusellSmith.jl:79 as a second policy: hank_get_het_outputs; not returned by the reference's plugin)
    endogenous_dim, exogenous_dim = "wealth", "productivity"

    def derived_policy(self, key: str, policy, xVals: dict, model):
        """a heterogeneous variable other than the policy variable from the savings policy (n_a, n_e), Float64 or `Dual`."""
        if key != "C":
            raise KeyError(key)
        grid = model.heterogeneity["wealth"].grid
        z = model.heterogeneity["productivity"].grid
        return (1.0 + xVals["r"]) * grid[:, None] + xVals["w"] * z[None, :] - policy

    def host_steady_state_step(self, value_next: np.ndarray, xVals: dict, model) -> dict:
        """one Float64 EGM step for the host steady-state VFI (same algebra as KrusellSmith.jl:59-80)."""
        grid = model.heterogeneity["wealth"].grid
        z = model.heterogeneity["productivity"].grid
        Π = model.heterogeneity["productivity"].transition
        β, γ, bc = model.params.β, model.params.γ, model.params.borrow_cons
        r, w = xVals["r"], xVals["w"]
        cmat = (β * (value_next @ Π.T)) ** (-1.0 / γ)
        s = (cmat - w * z[None, :] + grid[:, None]) / (1.0 + r)
        if not np.all(np.diff(s, axis=0) > 0):
            raise ValueError("knot-vectors must be unique and sorted in increasing order")
        g = np.empty_like(s)
        for e in range(z.size):
            g[:, e] = np.interp(grid, s[:, e], grid)   # flat outside [s_1, s_n]
        g = np.maximum(g, bc)
        c = (1.0 + r) * grid[:, None] + w * z[None, :] - g
        return {"Value": (1.0 + r) * c ** (-γ), "KD": g, "C": c}

    def __call__(self, *a, **k):
        raise RuntimeError("ValueFunction is a native kernel family (libhank_hip); on the transition "
                           "path it is invoked by BackwardIteration on the GPU, not called from Python.")


ValueFunction = _KSValueFunction()
register_function("ValueFunction", ValueFunction)
register_function("exogenousZ", exogenousZ)
