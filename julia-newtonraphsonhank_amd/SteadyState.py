"""Steady-state solver — host-side Python mirror of SteadyState.jl (north star: stays on the host).

It manufactures the boundary inputs of the GPU hot path: `ss.value` (terminal marginal value),
`ss.D` (initial distribution) and `ss.vars`. Algorithm as in the reference: Newton on the free
endogenous prices (find_ss, SteadyState.jl:184-233) around an inner VFI on the value function
(get_xVals, :111-154) and the stationary distribution by sparse linear solve
(invariant_dist, ForwardIteration.jl:436-442). The price Jacobian is taken by forward differences
instead of ForwardDiff.jacobian — only the fixed point matters.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import scipy.sparse as sp

from .Aggregation import Residuals
from .ForwardIteration import make_endogenous_transition
from .GeneralStructures import SequenceModel, invariant_dist, var_names, vars_of_type
from ._threads import host_algebra


@dataclass
class SteadyState:
    """SteadyState.jl:21-27."""
    vars: Dict[str, float]
    policies: Dict[str, np.ndarray]
    Λ: sp.csc_matrix
    D: np.ndarray
    value: np.ndarray


class SSAssembler:
    """variable roles + padded-matrix assembly for the steady-state Newton solve (SteadyState.jl:55-93)."""

    def __init__(self, model: SequenceModel, ss_spec, vfi_tol: float | None = None, vfi: str = "auto"):
        """`vfi`: where the inner value-function fixed point runs — "device" (hank_vfi: the EGM step kernels of the hot
        path iterated on the GPU), "host" (numpy), or "auto" = the device when an MI355X is present. Everything else of
        the steady state (price Newton, lottery matrix, invariant_dist) is host work either way (north star)."""
        self.model, self.ss_spec = model, ss_spec
        if vfi not in ("auto", "device", "host"):
            raise ValueError(f"vfi must be 'auto', 'device' or 'host' (got {vfi!r})")
        from .hip import device_available
        self.vfi_on_device = vfi == "device" or (vfi == "auto" and device_available())
        if vfi == "auto" and self.vfi_on_device:
            # a model the device block cannot take (more than one exogenous dimension, a grid it refuses) keeps the host
            # loop, which handles it through the kron of the transitions. Errors of the device sweeps themselves
            # (KnotsNotSorted / DomainError at a price iterate) surface from F(p) as they do on the host.
            from .BackwardIteration import household_block
            from .hip import HANK_ERR_BAD_ARG, HANK_ERR_NO_DEVICE, HankHIPError
            try:
                household_block(model)
            except ValueError:
                self.vfi_on_device = False
            except HankHIPError as e:
                # only "this model / this machine cannot": a device fault (out of memory, a refused launch, a failed sweep)
                # must not turn into a silent hundredfold slowdown on the host loop
                if e.code not in (HANK_ERR_BAD_ARG, HANK_ERR_NO_DEVICE):
                    raise
                self.vfi_on_device = False
        self.vfi_steps = 0
        self.all_keys = var_names(model)
        self.free_keys = tuple(k for k in vars_of_type(model, "endogenous") if k not in ss_spec.fixed)
        self.n_free = len(self.free_keys)
        endog = [d for d in model.heterogeneity.values() if d.dim_type == "endogenous"]
        exog = [d for d in model.heterogeneity.values() if d.dim_type == "exogenous"]
        if len(endog) != 1:
            raise ValueError("SSAssembler: exactly one endogenous heterogeneity dimension is currently supported")
        self.endog_dim = endog[0]
        self.n_exog = int(np.prod([d.n for d in exog])) if exog else 1
        Λ_exog = sp.identity(self.endog_dim.n, format="csc")
        for d in exog:
            Λ_exog = sp.kron(sp.csc_matrix(d.transition.T), Λ_exog, format="csc")
        self.Λ_exog = Λ_exog
        # the reference stops the VFI at compspec.ε; a tighter default makes ss.value a clean fixed point
        self.vfi_tol = vfi_tol if vfi_tol is not None else min(model.compspec.ε, 1e-11)
        self._value_warm = None

    def get_xVals(self, p_vec) -> Tuple[np.ndarray, np.ndarray, dict]:
        """full length-n_v aggregate vector for price iterate p_vec + converged marginal value
        (SteadyState.jl:111-154). VFI starts from ones (:132) or from the last converged value."""
        model = self.model
        xv = {k: 0.0 for k in self.all_keys}
        for i, k in enumerate(self.free_keys):
            xv[k] = float(p_vec[i])
        for k, v in self.ss_spec.fixed.items():
            xv[k] = float(v)
        vf = model.value_fn
        value = self._value_warm if self._value_warm is not None else np.ones((self.endog_dim.n, self.n_exog))
        if self.vfi_on_device:
            from .BackwardIteration import household_block
            from .hip import DomainError, KnotsNotSortedError
            hb = household_block(model)
            try:
                v, pol, steps, _ = hb.vfi(value, [xv[k] for k in vf.household_inputs], self.vfi_tol, 10_000)
            except (KnotsNotSortedError, DomainError) as e:      # what the host step raises / turns non-finite: find_ss halves the step
                raise ValueError(str(e)) from e
            self.vfi_steps += steps
            res = {"Value": v, self.endog_dim.policy_var: pol}
            for k in vars_of_type(model, "heterogeneous"):
                if k not in res:
                    res[k] = vf.derived_policy(k, pol, xv, model)
        else:
            res = vf.host_steady_state_step(value, xv, model)
            for _ in range(10_000):
                value_new = res["Value"]
                tol = np.max(np.abs(value_new - value))
                value = value_new
                self.vfi_steps += 1
                if tol < self.vfi_tol:
                    break
                res = vf.host_steady_state_step(value, xv, model)
        if self.vfi_on_device and self.endog_dim.n * self.n_exog > 4000:
            # chains this large take the power method on the host too (invariant_dist): the same iteration, run with the
            # forward step kernel of the hot path, warm-started from the last iterate
            D, _ = hb.stationary_dist(res[self.endog_dim.policy_var], getattr(self, "_D_warm", None))
        else:
            Λ_endog = make_endogenous_transition(res[self.endog_dim.policy_var], self.endog_dim, self.n_exog)
            D = invariant_dist((self.Λ_exog @ Λ_endog).T, D0=getattr(self, "_D_warm", None))
        self._D_warm = D
        for k in vars_of_type(model, "heterogeneous"):
            xv[k] = float(res[k].reshape(-1, order="F") @ D)
        return np.array([xv[k] for k in self.all_keys]), res["Value"], res

    def __call__(self, p_vec) -> np.ndarray:
        cs = self.model.compspec
        T_pad = 1 + cs.max_lag + cs.max_lead
        xVals, value, _ = self.get_xVals(p_vec)
        self._last_value = value
        return np.tile(xVals[:, None], (1, T_pad))


@host_algebra
def find_ss(model: SequenceModel, ss_spec, label: str, verbose: bool = False, vfi_tol=None, vfi: str = "auto") -> SteadyState:
    """Newton–Raphson on the free endogenous variables with step halving (SteadyState.jl:184-233)."""
    from .NewtonRaphson import warm_linear_solver
    from .GeneralStructures import vars_of_type
    warm_linear_solver(len(vars_of_type(model, "endogenous")) * (model.compspec.T - 1))      # (library start-up behind this solve)
    asm = SSAssembler(model, ss_spec, vfi_tol, vfi)

    def F(p):
        return Residuals(asm(p), model)

    def safe_eval(q):
        try:
            z = F(q)
            return z if np.all(np.isfinite(z)) else np.full(len(z), np.inf)
        except (ValueError, FloatingPointError):
            return np.full(asm.n_free, np.inf)

    p = np.array([ss_spec.guesses.get(k, 1.0) for k in asm.free_keys], dtype=np.float64)
    ε = model.compspec.ε
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        z = F(p)
        it, max_iter = 0, 100
        while np.linalg.norm(z) > ε and it < max_iter:
            if verbose:
                print(f"  [{label}] Iteration {it}: residual norm = {np.linalg.norm(z)}")
            asm._value_warm = getattr(asm, "_last_value", None)   # warm-start the inner VFI
            J = np.empty((len(z), asm.n_free))
            for j in range(asm.n_free):
                h = 1e-6 * max(1.0, abs(p[j]))
                q = p.copy()
                q[j] += h
                J[:, j] = (F(q) - z) / h
            step = np.linalg.solve(J, z)
            η, z_norm = 1.0, np.linalg.norm(z)
            p_new = p - η * step
            z_new = safe_eval(p_new)
            while (not np.isfinite(np.linalg.norm(z_new))) or np.linalg.norm(z_new) > z_norm:
                η /= 2
                if not η > 1e-8:
                    break
                p_new = p - η * step
                z_new = safe_eval(p_new)
            p, z = p_new, z_new
            it += 1
    if it == max_iter:
        import warnings
        warnings.warn(f"find_ss [{label}]: did not converge in {max_iter} iterations (residual norm: {np.linalg.norm(z)})")
    asm._value_warm = getattr(asm, "_last_value", None)
    xVals, ss_value, _ = asm.get_xVals(p)
    vars_ = {k: float(v) for k, v in zip(asm.all_keys, xVals)}
    # one more value-function call at the converged value for clean policies (:222-225)
    if asm.vfi_on_device:
        from .BackwardIteration import household_block
        _, pol = household_block(model).backward_step(ss_value, [vars_[k] for k in model.value_fn.household_inputs])
        res = {asm.endog_dim.policy_var: pol}
        for k in vars_of_type(model, "heterogeneous"):
            if k not in res:
                res[k] = model.value_fn.derived_policy(k, pol, vars_, model)
    else:
        res = model.value_fn.host_steady_state_step(ss_value, vars_, model)
    het_keys = vars_of_type(model, "heterogeneous")
    policies = {k: res[k] for k in het_keys}
    Λ_endog = make_endogenous_transition(policies[asm.endog_dim.policy_var], asm.endog_dim, asm.n_exog)
    Λss = (asm.Λ_exog @ Λ_endog).tocsc()
    if asm.vfi_on_device and asm.endog_dim.n * asm.n_exog > 4000:
        from .BackwardIteration import household_block
        D, _ = household_block(model).stationary_dist(policies[asm.endog_dim.policy_var], getattr(asm, "_D_warm", None))
    else:
        D = invariant_dist(Λss.T, D0=getattr(asm, "_D_warm", None))
    return SteadyState(vars_, policies, Λss, D, ss_value)


def get_SteadyStates(model: SequenceModel, verbose: bool = False, vfi_tol=None, vfi: str = "auto"):
    """both steady states (SteadyState.jl:245-259); one solve when the specs are the same object."""
    ss_initial = find_ss(model, model.ss_initial, "initial", verbose, vfi_tol, vfi)
    if model.ss_initial is model.ss_ending:
        return ss_initial, ss_initial
    return ss_initial, find_ss(model, model.ss_ending, "ending", verbose, vfi_tol, vfi)
