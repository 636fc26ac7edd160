"""BackwardIteration — same call surface as the reference's BackwardIteration.jl:46-116, body on the GPU.

    BackwardIteration(xVec_endog, exog_paths, model, ss_end) -> {het_var: [mat_1, …, mat_{T-1}]}

The T-1 sequential EGM steps (and, when `xVec_endog` is a `Dual`, their N partials) run as
hand-written HIP kernels behind the C ABI (hank_primal / hank_jvp); this function only picks the
household inputs out of x (get_xvals_at_t, BackwardIteration.jl:70-82 — for KS: r_t, w_t), hands
them over, and wraps the device-resident policy sequence.
"""
from __future__ import annotations

import numpy as np

from .dual import Dual
from .GeneralStructures import SequenceModel, var_names, vars_of_type
from .hip import HouseholdBlock


def household_block(model: SequenceModel) -> HouseholdBlock:
    """the model's device context (created on first use; one per model, bound to the current GPU)."""
    if model._hip_block is None:
        endog = [d for d in model.heterogeneity.values() if d.dim_type == "endogenous"]
        exog = [d for d in model.heterogeneity.values() if d.dim_type == "exogenous"]
        if len(endog) != 1:
            raise ValueError(f"exactly one endogenous dimension is currently supported (got {len(endog)})")
        if len(exog) != 1:
            raise ValueError(f"the native household block supports one exogenous dimension (got {len(exog)})")
        p = model.params
        model._hip_block = HouseholdBlock(endog[0].grid, exog[0].grid, exog[0].transition, p.β, p.γ,
                                          p.borrow_cons, model.compspec.T, model.value_fn.value_fn_id)
    return model._hip_block


def household_inputs(xVec_endog, exog_paths, model: SequenceModel):
    """rows of xVals the value function reads, per period: (n_hh, P) values and (n_hh, P, N) partials."""
    cs = model.compspec
    P = cs.T - 1
    endog_keys = vars_of_type(model, "endogenous")
    is_dual = isinstance(xVec_endog, Dual)
    xv = (xVec_endog.v if is_dual else np.asarray(xVec_endog, dtype=np.float64)).reshape(cs.n_endog, P, order="F")
    xp = xVec_endog.p.reshape(cs.n_endog, P, -1, order="F") if is_dual else None
    names = model.value_fn.household_inputs
    xhh = np.empty((len(names), P))
    dxhh = np.zeros((len(names), P, xp.shape[2])) if is_dual else None
    for k, name in enumerate(names):
        if name in endog_keys:
            j = endog_keys.index(name)
            xhh[k] = xv[j]
            if is_dual:
                dxhh[k] = xp[j]
        elif name in exog_paths:
            xhh[k] = np.asarray(exog_paths[name], dtype=np.float64)   # exogenous: zero partials
        else:
            raise KeyError(f"value function input '{name}' is neither endogenous nor exogenous in this model")
    return xhh, dxhh


def policy_variable(model: SequenceModel) -> str:
    """the heterogeneous variable that moves the distribution: the endogenous dimension's `policy_var`
    (ForwardIteration.jl:297-300)."""
    return next(d.policy_var for d in model.heterogeneity.values() if d.dim_type == "endogenous")


class PolicySequences(dict):
    """BackwardIteration's return value: {het_var: list of P (n_a x n_e) matrices}; the matrices are
    fetched from HBM lazily. Carries the tag ForwardIteration uses to stay on the fused path.

    The reference's 4-argument call (NewtonRaphson.jl:78) does not know `ss_initial`, and the fused device
    sweep needs D_0: such a call is DEFERRED — nothing runs until ForwardIteration supplies `ss_initial`
    (one sweep in all, NewtonRaphson.jl:78-79) or until a policy matrix is actually read (then the sweep runs
    with a placeholder D_0; policies do not depend on it)."""

    def __init__(self, hb, het_keys, is_dual, N, pending, model=None):
        super().__init__()
        self._hb, self._het_keys, self._is_dual, self._N = hb, het_keys, is_dual, N
        self._model = model                # (a second heterogeneous variable is derived from the policy variable: _fetch)
        self._pending = pending            # {"xhh", "dxhh", "value"} until the device sweep has run
        self._D_used, self._generation = None, None
        self._fetched = False
        for k in het_keys:
            dict.__setitem__(self, k, None)

    def _run(self, D0):
        """the ONE fused device sweep of this BackwardIteration (+ ForwardIteration) pair."""
        hb, pend = self._hb, self._pending
        hb.set_boundary(pend["value"], D0)
        if pend["dxhh"] is not None:      # a Dual pass carries value and partials together, like the reference's JVP
            agg, dagg = hb.primal_jvp(pend["xhh"], pend["dxhh"])
        else:
            agg, dagg = hb.primal(pend["xhh"]), None
        hb._generation = getattr(hb, "_generation", 0) + 1
        hb._last = {"agg": agg, "dagg": dagg, "D0": D0, "xhh": pend["xhh"], "dxhh": pend["dxhh"], "value": pend["value"]}
        if len(self._het_keys) > 1:        # every heterogeneous variable's aggregate, reduced by the same sweeps
            hb._last["het"] = hb.het_outputs(len(self._model.value_fn.outputs), pend["dxhh"])
        self._D_used, self._generation, self._pending = D0, hb._generation, None

    def _fetch(self):
        if self._fetched:
            return
        hb = self._hb
        if self._pending is not None:
            self._run(np.full(hb.G, 1.0 / hb.G))
        if getattr(hb, "_generation", None) != self._generation:
            raise RuntimeError("policy sequences were overwritten by a later BackwardIteration on this model")
        pol = hb.policy_seq()                      # (n_a, n_e, P)
        if self._is_dual:
            dpol = hb.dpolicy_seq(self._N)         # (n_a, n_e, P, N)
            seq = [Dual(pol[:, :, t], dpol[:, :, t, :]) for t in range(hb.P)]
        else:
            seq = [pol[:, :, t] for t in range(hb.P)]
        pol_key = self._het_keys[0] if self._model is None else policy_variable(self._model)
        for k in self._het_keys:
            if k == pol_key:
                dict.__setitem__(self, k, seq)
                continue
            # another heterogeneous variable: the plugin derives it from the policy variable and the period's inputs (the
            # reference's value_fn returns it next to the policy, BackwardIteration.jl:108-111)
            vf, last = self._model.value_fn, hb._last
            other = []
            for t in range(hb.P):
                xv = {name: (Dual(last["xhh"][j, t], last["dxhh"][j, t, :]) if self._is_dual else last["xhh"][j, t])
                      for j, name in enumerate(vf.household_inputs)}
                other.append(vf.derived_policy(k, seq[t], xv, self._model))
            dict.__setitem__(self, k, other)
        self._fetched = True

    def __getitem__(self, k):
        self._fetch()
        return dict.__getitem__(self, k)

    def values(self):
        self._fetch()
        return dict.values(self)

    def items(self):
        self._fetch()
        return dict.items(self)


def BackwardIteration(xVec_endog, exog_paths, model: SequenceModel, ss_end, ss_initial=None):
    """Backward iteration over the T-1 transition periods (BackwardIteration.jl:46-116).

    With the reference's four arguments the device sweep is deferred to ForwardIteration (see
    PolicySequences); `ss_initial` (optional, not in the reference signature) runs it here instead."""
    cs = model.compspec
    P = cs.T - 1
    n = (xVec_endog.v if isinstance(xVec_endog, Dual) else np.asarray(xVec_endog)).size
    if n != cs.n_endog * P:
        raise ValueError(f"xVec_endog has length {n}, expected n_endog*(T-1) = {cs.n_endog * P}")
    het_keys = vars_of_type(model, "heterogeneous")
    for k in het_keys:
        if k not in model.value_fn.outputs:
            raise KeyError(f"BackwardIteration: value_fn return is missing key :{k} (got keys: {model.value_fn.outputs})")
    hb = household_block(model)
    xhh, dxhh = household_inputs(xVec_endog, exog_paths, model)
    if model.value_fn.household_inputs[0] == "r" and not np.all(1.0 + xhh[0] > 0.0):   # raised here, not at the deferred sweep
        bad = int(np.flatnonzero(~(1.0 + xhh[0] > 0.0))[0]) + 1
        from .hip import DomainError, HANK_ERR_DOMAIN
        raise DomainError(HANK_ERR_DOMAIN, f"1 + r must be positive (period {bad})")
    seqs = PolicySequences(hb, het_keys, dxhh is not None, 0 if dxhh is None else dxhh.shape[2],
                           {"xhh": xhh, "dxhh": dxhh, "value": np.array(ss_end.value, dtype=np.float64, copy=True)}, model)
    if ss_initial is not None:
        seqs._run(np.asarray(ss_initial.D, dtype=np.float64))
    return seqs
