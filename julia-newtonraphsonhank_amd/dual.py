"""Host-side dual numbers for the aggregate (n_v x T_pad) algebra.

The reference runs its whole pipeline on ForwardDiff.Dual{Tag,Float64,N}
(ForwardDiff.jl/src/dual.jl:14-21). In this build the household block carries its partials on the
GPU (hank_jvp); only the tiny aggregate layer — assemble_full_xMat + the YAML-compiled residual
equations (6 x 300 numbers) — needs dual arithmetic on the host. `Dual` is a numpy-backed
array-of-duals with exactly the rules the compiled equations can reach:
+, -, *, /, ^ (dual.jl:495-581), unary minus, and the DiffRules unary functions log/exp/sqrt.
Comparisons act on values only (dual.jl:395-404).
"""
from __future__ import annotations

import numpy as np


class Dual:
    """value array `v` of shape S and partials `p` of shape S + (N,)."""

    __array_priority__ = 1000  # numpy defers binary ops to us

    def __init__(self, v, p):
        self.v = np.asarray(v, dtype=np.float64)
        self.p = np.asarray(p, dtype=np.float64)
        if self.p.shape[:-1] != self.v.shape:
            raise ValueError(f"partials shape {self.p.shape} does not extend value shape {self.v.shape}")

    # -- construction -----------------------------------------------------------------------
    @property
    def N(self) -> int:
        return self.p.shape[-1]

    @property
    def shape(self):
        return self.v.shape

    @staticmethod
    def constant(v, N: int) -> "Dual":
        v = np.asarray(v, dtype=np.float64)
        return Dual(v, np.zeros(v.shape + (N,)))

    @staticmethod
    def seed(primal, tangent) -> "Dual":
        """primal + t*tangent with t = Dual(0,1) per direction (GeneralStructures.jl:546-547)."""
        primal = np.asarray(primal, dtype=np.float64)
        tangent = np.asarray(tangent, dtype=np.float64)
        if tangent.shape == primal.shape:
            tangent = tangent[..., None]
        return Dual(primal, tangent)

    def _lift(self, other) -> "Dual | None":
        if isinstance(other, Dual):
            return other
        return None

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        return Dual(self.v[idx], self.p[idx + (slice(None),)])

    def __len__(self):
        return len(self.v)

    def copy(self):
        return Dual(self.v.copy(), self.p.copy())

    # -- arithmetic (ForwardDiff rules) -------------------------------------------------------
    def __add__(self, o):
        d = self._lift(o)
        if d is None:
            return Dual(self.v + o, np.broadcast_to(self.p, np.broadcast(self.v, o).shape + (self.N,)).copy())
        return Dual(self.v + d.v, self.p + d.p)

    __radd__ = __add__

    def __sub__(self, o):
        d = self._lift(o)
        if d is None:
            return Dual(self.v - o, np.broadcast_to(self.p, np.broadcast(self.v, o).shape + (self.N,)).copy())
        return Dual(self.v - d.v, self.p - d.p)

    def __rsub__(self, o):
        return Dual(o - self.v, np.broadcast_to(-self.p, np.broadcast(self.v, o).shape + (self.N,)).copy())

    def __neg__(self):
        return Dual(-self.v, -self.p)

    def __pos__(self):
        return self

    def __mul__(self, o):
        d = self._lift(o)
        if d is None:
            o = np.asarray(o, dtype=np.float64)
            return Dual(self.v * o, self.p * o[..., None])
        # value vx*vy ; partials vy*px + vx*py (partials.jl:117-119, :219-221)
        return Dual(self.v * d.v, d.v[..., None] * self.p + self.v[..., None] * d.p)

    __rmul__ = __mul__

    def __truediv__(self, o):
        d = self._lift(o)
        if d is None:
            o = np.asarray(o, dtype=np.float64)
            return Dual(self.v / o, self.p / o[..., None])
        # _div_partials (partials.jl:84-86)
        return Dual(self.v / d.v, (1.0 / d.v)[..., None] * self.p + (-(self.v / (d.v * d.v)))[..., None] * d.p)

    def __rtruediv__(self, o):
        o = np.asarray(o, dtype=np.float64)
        divv = o / self.v
        return Dual(divv, (-(divv / self.v))[..., None] * self.p)

    def __pow__(self, y):
        if isinstance(y, Dual):
            # Dual^Dual (dual.jl:547-561)
            expv = self.v ** y.v
            powval = y.v * self.v ** (y.v - 1.0)
            with np.errstate(divide="ignore", invalid="ignore"):
                logval = np.where((self.v == 0) & (y.v > 0), 0.0, expv * np.log(self.v))
            logval = np.where(np.all(y.p == 0, axis=-1), 1.0, logval)
            return Dual(expv, powval[..., None] * self.p + logval[..., None] * y.p)
        y = float(y)
        if np.any((self.v < 0) & (y != np.floor(y))):
            raise ValueError("DomainError: negative base under a non-integer power")
        expv = self.v ** y
        if y == 0.0:
            return Dual(expv, np.zeros_like(self.p))
        # Dual^Real: partials * y * v^(y-1), zero where the partials are all zero (dual.jl:563-572)
        newp = (self.p * y) * (self.v ** (y - 1.0))[..., None]
        newp = np.where(np.all(self.p == 0, axis=-1, keepdims=True), 0.0, newp)
        return Dual(expv, newp)

    def __rpow__(self, x):
        # Real^Dual (dual.jl:574-579)
        x = np.asarray(x, dtype=np.float64)
        expv = x ** self.v
        with np.errstate(divide="ignore", invalid="ignore"):
            deriv = np.where((x == 0) & (self.v > 0), 0.0, expv * np.log(x))
        return Dual(expv, deriv[..., None] * self.p)

    # comparisons on values only
    def __lt__(self, o):
        return self.v < (o.v if isinstance(o, Dual) else o)

    def __gt__(self, o):
        return self.v > (o.v if isinstance(o, Dual) else o)

    def __repr__(self):
        return f"Dual(v={self.v!r}, N={self.N})"


def value(x):
    return x.v if isinstance(x, Dual) else np.asarray(x, dtype=np.float64)


def partials(x, N: int | None = None):
    if isinstance(x, Dual):
        return x.p
    x = np.asarray(x, dtype=np.float64)
    return np.zeros(x.shape + (N or 0,))


def _unary(fun, dfun):
    def f(x):
        if isinstance(x, Dual):
            return Dual(fun(x.v), dfun(x.v)[..., None] * x.p)
        return fun(np.asarray(x, dtype=np.float64))
    return f


# DiffRules unary rules reachable from equation strings (ModelParser.jl:108-110 leaves other calls as is)
log = _unary(np.log, lambda v: 1.0 / v)
exp = _unary(np.exp, np.exp)
sqrt = _unary(np.sqrt, lambda v: 0.5 / np.sqrt(v))
