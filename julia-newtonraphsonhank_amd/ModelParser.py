"""YAML parsing, equation compilation and model construction — Python mirror of ModelParser.jl.

KrusellSmith.yaml is accepted unchanged. Equation strings are Julia syntax
("Y = Z * KS(-1)^α"); they are translated to a Python AST with the same rules as
`transform_expr` (ModelParser.jl:54-119): variables -> row slices, VAR(-k)/VAR(+k) -> shift_lag /
shift_lead, parameters -> params.<name>, arithmetic broadcast element-wise over time, other calls
(log, exp, sqrt) kept. The compiled closure works on float matrices and on `Dual` matrices.
Host-only (north_star: ModelParser.jl stays on the host).
"""
from __future__ import annotations

import ast
from pathlib import Path
from types import SimpleNamespace
from typing import Dict, Sequence, Tuple

import numpy as np
import yaml

from . import dual as _dual
from .dual import Dual
from . import GeneralStructures as gs
from .GeneralStructures import (ComputationalSpec, HeterogeneityDimension, SequenceModel,
                                SteadyStateSpec, Variable, shift_lag, shift_lead)

# functions a YAML file may reference by name (the reference resolves them in Main, :404-413).
_FUNCTION_REGISTRY: Dict[str, object] = {
    "double_exponential": gs.double_exponential,
    "rouwenhorst_discretization": gs.rouwenhorst_discretization,
}


def register_function(name: str, fn) -> None:
    """make `fn` resolvable from YAML (`function:`, `seq_function:`, `grid_function:`)."""
    _FUNCTION_REGISTRY[name] = fn


def _lookup_fn(name: str):
    if name not in _FUNCTION_REGISTRY:
        raise NameError(f"Function '{name}' not found in scope. Check that it is defined in the "
                        "function_file specified in your YAML.")
    return _FUNCTION_REGISTRY[name]


def _julia_to_python(expr: str) -> str:
    return expr.replace("^", "**")


def _lag_of(node: ast.Call, var_set) -> int | None:
    """VAR(-k) / VAR(+k) / VAR(0) pattern (ModelParser.jl:72-88): returns the signed shift."""
    if isinstance(node.func, ast.Name) and node.func.id in var_set and len(node.args) == 1 and not node.keywords:
        a = node.args[0]
        if isinstance(a, ast.UnaryOp) and isinstance(a.operand, ast.Constant) and isinstance(a.operand.value, int):
            if isinstance(a.op, ast.USub):
                return -a.operand.value
            if isinstance(a.op, ast.UAdd):
                return a.operand.value
        if isinstance(a, ast.Constant) and isinstance(a.value, int):
            return a.value
    return None


def detect_max_lag_lead(equations: Sequence[str], var_syms) -> Tuple[int, int]:
    """(max_lag, max_lead) over all equations (ModelParser.jl:137-172)."""
    var_set = set(var_syms)
    max_lag = max_lead = 0
    for eq in equations:
        parts = eq.split("=", 1)
        if len(parts) != 2:
            continue
        for part in parts:
            for node in ast.walk(ast.parse(_julia_to_python(part.strip()), mode="eval")):
                if isinstance(node, ast.Call):
                    k = _lag_of(node, var_set)
                    if k is not None:
                        if k < 0:
                            max_lag = max(max_lag, -k)
                        elif k > 0:
                            max_lead = max(max_lead, k)
    return max_lag, max_lead


class _Transform(ast.NodeTransformer):
    """transform_expr (ModelParser.jl:54-119) on the Python AST."""

    def __init__(self, var_indices: Dict[str, int], param_names):
        self.var_indices, self.param_names = var_indices, set(param_names)

    def visit_Call(self, node: ast.Call):
        k = _lag_of(node, self.var_indices)
        if k is not None:
            row = ast.Subscript(value=ast.Name("xMat", ast.Load()), slice=ast.Constant(self.var_indices[node.func.id]), ctx=ast.Load())
            if k == 0:
                return row
            fn = "shift_lag" if k < 0 else "shift_lead"
            return ast.Call(func=ast.Name(fn, ast.Load()), args=[row, ast.Constant(abs(k))], keywords=[])
        node.args = [self.visit(a) for a in node.args]
        return node

    def visit_Name(self, node: ast.Name):
        if node.id in self.var_indices:
            return ast.Subscript(value=ast.Name("xMat", ast.Load()), slice=ast.Constant(self.var_indices[node.id]), ctx=ast.Load())
        if node.id in self.param_names:
            return ast.Attribute(value=ast.Name("params", ast.Load()), attr=node.id, ctx=ast.Load())
        return node


def transform_expr(expr: str, var_indices: Dict[str, int], param_names) -> ast.Expression:
    tree = ast.parse(_julia_to_python(expr.strip()), mode="eval")
    tree = _Transform(var_indices, param_names).visit(tree)
    return ast.fix_missing_locations(tree)


def compile_residuals(equations: Sequence[str], var_syms: Tuple[str, ...], param_names):
    """equation strings -> closure (xMat, params) -> residual vector (ModelParser.jl:217-259).

    xMat is the padded n_v x T_pad matrix; each `LHS = RHS` gives LHS - RHS over all padded
    columns, sliced to the valid middle range, stacked equation-major within a period
    (all equations at t=1, then t=2, ...)."""
    var_indices = {s: i for i, s in enumerate(var_syms)}
    max_lag, max_lead = detect_max_lag_lead(equations, var_syms)
    codes = []
    for eq in equations:
        parts = eq.split("=", 1)
        if len(parts) != 2 or "=" in parts[1]:
            raise ValueError(f"Equation must contain exactly one '=': {eq}")
        lhs = transform_expr(parts[0], var_indices, param_names)
        rhs = transform_expr(parts[1], var_indices, param_names)
        codes.append((compile(lhs, f"<lhs of {eq!r}>", "eval"), compile(rhs, f"<rhs of {eq!r}>", "eval")))
    env = {"shift_lag": shift_lag, "shift_lead": shift_lead, "log": _dual.log, "exp": _dual.exp,
           "sqrt": _dual.sqrt, "pi": np.pi, "__builtins__": {}}

    def residuals_fn(xMat, params):
        T_pad = xMat.shape[1]
        lo, hi = max_lag, T_pad - max_lead
        scope = dict(env, xMat=xMat, params=params)
        rows = []
        for lhs, rhs in codes:
            r = eval(lhs, scope) - eval(rhs, scope)  # noqa: S307 - our own compiled AST
            if not isinstance(r, Dual):
                r = np.broadcast_to(np.asarray(r, dtype=np.float64), (T_pad,))
            rows.append(r[lo:hi])
        if any(isinstance(r, Dual) for r in rows):
            N = next(r.N for r in rows if isinstance(r, Dual))
            rows = [r if isinstance(r, Dual) else Dual.constant(r, N) for r in rows]
            v = np.stack([r.v for r in rows], axis=0)          # n_eq x P
            p = np.stack([r.p for r in rows], axis=0)          # n_eq x P x N
            return Dual(v.reshape(-1, order="F"), p.reshape(-1, p.shape[-1], order="F"))
        return np.stack(rows, axis=0).reshape(-1, order="F")

    residuals_fn.max_lag, residuals_fn.max_lead = max_lag, max_lead
    return residuals_fn


def _parse_number(v):
    return int(v) if isinstance(v, bool) is False and isinstance(v, int) else (float(v) if isinstance(v, float) else v)


def _parse_ss_spec(spec) -> SteadyStateSpec:
    fixed = {str(k): float(v) for k, v in (spec.get("fixed") or {}).items()}
    guesses = {str(k): float(v) for k, v in (spec.get("guesses") or {}).items()}
    return SteadyStateSpec(fixed, guesses)


def _build_dimension_from_yaml(d) -> HeterogeneityDimension:
    """grid-function contract of ModelParser.jl:452-511 with the same validation messages."""
    dim_type, name, fn_name = str(d["type"]), d["name"], d["grid_function"]
    params = d["params"]
    n = int(params["n"])
    policy_var = d.get("policy_var")
    result = _lookup_fn(fn_name)(**{str(k): v for k, v in params.items()})
    if dim_type == "endogenous":
        if not (isinstance(result, np.ndarray) and result.ndim == 1):
            raise TypeError(f"Grid function '{fn_name}' for endogenous dimension '{name}' must return a Vector, got {type(result)}.")
        if len(result) != n:
            raise ValueError(f"Grid function '{fn_name}' for endogenous dimension '{name}': expected {n} grid points (params.n = {n}), got {len(result)}.")
        return HeterogeneityDimension("endogenous", n, np.asarray(result, dtype=np.float64), None, policy_var)
    if dim_type == "exogenous":
        if not (isinstance(result, tuple) and len(result) == 2):
            raise TypeError(f"Grid function '{fn_name}' for exogenous dimension '{name}' must return a 2-tuple (grid, transition_matrix), got {type(result)}.")
        grid, Π = result
        grid, Π = np.asarray(grid, dtype=np.float64), np.asarray(Π, dtype=np.float64)
        if grid.ndim != 1 or len(grid) != n:
            raise ValueError(f"Grid from '{fn_name}' for exogenous dimension '{name}': expected {n} points (params.n = {n}), got {grid.shape}.")
        if Π.shape != (n, n):
            raise ValueError(f"Transition matrix from '{fn_name}' for exogenous dimension '{name}': expected {n}×{n}, got {Π.shape}.")
        return HeterogeneityDimension("exogenous", n, grid, Π, None)
    raise ValueError(f"Unknown dimension type '{dim_type}' for dimension '{name}'. Expected 'endogenous' or 'exogenous'.")


def build_model_from_yaml(file_path: str, overrides: dict | None = None) -> SequenceModel:
    """YAML -> SequenceModel (ModelParser.jl:296-379).

    `function_file` names the reference's Julia plugin file; here the plugin of the same stem
    (e.g. KrusellSmith.jl -> the KrusellSmith module of this package) registers the native kernel
    family behind the YAML's `function: "ValueFunction"`.
    `overrides` (not in the reference) lets benchmarks resize a spec without editing the YAML:
    {"T": 300, "dimensions": {"wealth": {"n": 2000}, "productivity": {"n": 11}},
     "steady_states": {"ending": {"fixed": {"Z": 1.03}, "guesses": {...}}}}  (adds / replaces steady-state blocks)."""
    with open(file_path, "r", encoding="utf-8") as fh:
        y = yaml.safe_load(fh)
    overrides = overrides or {}
    # 0. the model plugin
    stem = Path(y["file"]["function_file"]).stem
    import importlib
    try:
        importlib.import_module(f"{__package__}.{stem}")
    except ModuleNotFoundError as e:
        raise NameError(f"no native plugin for function_file '{y['file']['function_file']}'") from e
    # 1. parameters
    mp = y["parameters"]["model"]
    pnames = tuple(str(p["name"]) for p in mp)
    params = SimpleNamespace(**{str(p["name"]): _parse_number(p["value"]) for p in mp})
    cs = {str(p["name"]): p["value"] for p in (y.get("parameters", {}).get("computational") or [])}
    T = int(overrides.get("T", cs.get("T", 150)))
    ε = float(overrides.get("ε", cs.get("ε", 1e-6)))
    dx = float(cs.get("dx", 1e-8))
    # 2. heterogeneity dimensions
    heterogeneity = {}
    for d in y["dimensions"]:
        d = dict(d)
        d["params"] = dict(d["params"], **overrides.get("dimensions", {}).get(d["name"], {}))
        heterogeneity[str(d["name"])] = _build_dimension_from_yaml(d)
    # 3. variables: endogenous -> heterogeneous -> exogenous (ModelParser.jl:357-359)
    vs = y["variables"]
    endog = [Variable(str(v["name"]), "endogenous", v.get("description", "")) for v in vs.get("endogenous") or []]
    het_raw = vs.get("heterogeneous") or []
    het_fn = [v for v in het_raw if "function" in v]
    if len(het_fn) != 1:
        raise ValueError("The 'heterogeneous' variables section must contain exactly one 'function' entry "
                         f"(got {len(het_fn)}). This function maps ∂V/∂a' → (Value=∂V/∂a, <het vars>...).")
    value_fn = _lookup_fn(het_fn[0]["function"])
    het = [Variable(str(v["name"]), "heterogeneous", v.get("description", "")) for v in het_raw if "name" in v]
    exog = [Variable(str(v["name"]), "exogenous", v.get("description", ""),
                     _lookup_fn(v["seq_function"]) if "seq_function" in v else None)
            for v in vs.get("exogenous") or []]
    variables = {v.name: v for v in [*endog, *het, *exog]}
    all_names = tuple(variables.keys())
    # 4. equations
    equations = tuple(str(e) for e in y["equations"])
    param_names = set(pnames) | {"T", "ε", "dx", "n_v"}
    max_lag, max_lead = detect_max_lag_lead(equations, all_names)
    residuals_fn = compile_residuals(equations, all_names, param_names)
    compspec = ComputationalSpec(T, ε, dx, len(variables), len(endog), max_lag, max_lead)
    # 5. steady states
    ss = dict(y["steady_states"], **(overrides.get("steady_states") or {}))
    ss_initial = _parse_ss_spec(ss["initial"])
    ss_ending = _parse_ss_spec(ss["ending"]) if "ending" in ss else ss_initial
    return SequenceModel(variables, equations, compspec, params, residuals_fn, ss_initial, ss_ending,
                         heterogeneity, value_fn)
