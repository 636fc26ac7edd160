"""Sequence-space Jacobian at the steady state — the J̅ that y_Iteration uses as GMRES operator.

The reference assembles it with Boehl's block-Toeplitz recursion from n_endog ForwardDiff JVPs
through BackwardIteration plus n_endog Zygote pullbacks through ForwardIteration
(SteadyStateJacobian.jl:41-410) and validates columns against full-pipeline JVPs
(test_SteadyState.jl:194-231, abs tol 1e-5). With a batched native JVP the same matrix is obtained
directly as n = n_endog·(T-1) unit-tangent JVPs of the full pipeline at the steady-state path,
carried as tangent batches on the GPU — no reverse mode.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .GeneralStructures import SequenceModel, vars_of_type
from .NewtonRaphson import LinearizedFunction
from ._threads import host_algebra


@host_algebra
def getSteadyStateJacobian(ss, model: SequenceModel, chunk: int = 512, drop_tol: float = 0.0, device_batch: int = 256, group=None):
    """n x n sparse Jacobian of F at the constant steady-state path (SteadyStateJacobian.jl:41-65).
    `chunk` unit tangents are pushed per call; only those that move the household inputs (r, w) reach the GPU,
    in device batches padded to `device_batch` so that one tangent workspace serves every call."""
    cs = model.compspec
    if len(model.equations) != cs.n_endog:
        raise AssertionError(f"System is not square: {len(model.equations)} equations but {cs.n_endog} endogenous "
                             "variables. Newton-Raphson requires n_eq == n_endog.")
    P = cs.T - 1
    n = cs.n_endog * P
    endog_keys = vars_of_type(model, "endogenous")
    exog_keys = vars_of_type(model, "exogenous")
    x_ss = np.tile(np.array([ss.vars[k] for k in endog_keys]), P)
    exog_ss = {k: np.full(P, float(ss.vars[k])) for k in exog_keys}
    lin = LinearizedFunction(x_ss, exog_ss, model, ss, ss)
    # unit tangents in chunks; under torch.distributed with W > 1 ranks the chunks of a pass are shared out over the
    # ranks (one GPU each) and all-gathered (parallel.assemble_columns) — every rank ends with the whole matrix
    from .parallel import assemble_columns
    J = assemble_columns(lambda E: lin.jvp(E, pad_to=device_batch), n, chunk, group)
    if drop_tol > 0:
        J[np.abs(J) < drop_tol] = 0.0
    return sp.csc_matrix(J)
