"""ctypes harness of the CPU oracle (oracle/hank_oracle.c) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package never does (tests/test_layout.py enforces it).

Duals are numpy arrays with a trailing axis of length 1+N: [..., 0] = value, [..., 1:] = partials
(the AoS layout of ForwardDiff's Dual{T,Float64,N}).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB = _DIR / "libhank_oracle.so"
_REF = _DIR / "_ref" / "libref_cppdual.so"
SUPPORTED_N = (1, 2, 3, 4, 8, 16, 32)

ORC_OK, ORC_ERR_KNOTS, ORC_ERR_DOMAIN = 0, 3, 4


class orc_model(C.Structure):
    _fields_ = [("n_a", C.c_int32), ("n_e", C.c_int32), ("a", C.POINTER(C.c_double)),
                ("z", C.POINTER(C.c_double)), ("Pi", C.POINTER(C.c_double)),
                ("beta", C.c_double), ("gamma", C.c_double), ("borrow_cons", C.c_double)]


def build(force: bool = False) -> None:
    """compile the oracle (and oracle/_ref when /root/reference is present) with make."""
    if force or not _LIB.exists() or _LIB.stat().st_mtime < (_DIR / "hank_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(_DIR), "libhank_oracle.so"], check=True, capture_output=True)
    if Path("/root/reference/ForwardDiff.jl/benchmarks/cpp").is_dir() and (force or not _REF.exists()):
        subprocess.run(["make", "-C", str(_DIR), "ref"], check=True, capture_output=True)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB.exists():
            build()
        _lib = C.CDLL(str(_LIB))
    return _lib


def ref_lib() -> C.CDLL | None:
    """the reference's own C++ dual classes (oracle/_ref), or None when it was never built."""
    return C.CDLL(str(_REF)) if _REF.exists() else None


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fn(name: str, N: int):
    if N not in SUPPORTED_N:
        raise ValueError(f"oracle compiled for N in {SUPPORTED_N}, got {N}")
    f = getattr(lib(), f"{name}_n{N}")
    return f


def pad_N(N: int) -> int:
    """smallest compiled partial count >= N (extra partials are carried as zeros)."""
    for n in SUPPORTED_N:
        if n >= N:
            return n
    raise ValueError(f"N={N} exceeds the oracle's largest compiled chunk {SUPPORTED_N[-1]}")


class Oracle:
    """CPU restatement of the household block + KS residuals for one model."""

    def __init__(self, a_grid, z_grid, Pi, beta, gamma, borrow_cons):
        self.a = np.ascontiguousarray(a_grid, dtype=np.float64)
        self.z = np.ascontiguousarray(z_grid, dtype=np.float64)
        self.Pi = np.asfortranarray(np.asarray(Pi, dtype=np.float64))
        self.n_a, self.n_e = self.a.size, self.z.size
        self.G = self.n_a * self.n_e
        self.m = orc_model(self.n_a, self.n_e, _dp(self.a), _dp(self.z), _dp(self.Pi),
                           float(beta), float(gamma), float(borrow_cons))

    # duals: (n_a, n_e, 1+N) logically; memory = [e][a][1+N] (column-major matrix of AoS duals)
    def _mat_to_mem(self, M: np.ndarray, N: int) -> np.ndarray:
        M = np.asarray(M, dtype=np.float64)
        if M.ndim == 2:
            M = np.concatenate([M[..., None], np.zeros(M.shape + (N,))], axis=-1)
        assert M.shape == (self.n_a, self.n_e, 1 + N), M.shape
        return np.ascontiguousarray(M.transpose(1, 0, 2))

    def _mem_to_mat(self, mem: np.ndarray, N: int) -> np.ndarray:
        return mem.reshape(self.n_e, self.n_a, 1 + N).transpose(1, 0, 2).copy()

    @staticmethod
    def _scalar(x, N: int) -> np.ndarray:
        x = np.atleast_1d(np.asarray(x, dtype=np.float64))
        out = np.zeros(1 + N)
        out[: x.size] = x
        return out

    def value_function(self, value_next, r, w, N: int, tr=None):
        """ValueFunction (KrusellSmith.jl:43-83). Returns (status, Value, KD) as (n_a,n_e,1+N).
        `tr` (a dual, like r and w): the lump-sum transfer of the one-asset HANK family (not in the reference)."""
        vin = self._mat_to_mem(value_next, N)
        V = np.empty_like(vin)
        KD = np.empty_like(vin)
        if tr is None:
            f = _fn("orc_value_function", N)
            f.restype = C.c_int
            st = f(C.byref(self.m), _dp(vin), _dp(self._scalar(r, N)), _dp(self._scalar(w, N)), _dp(V), _dp(KD))
        else:
            f = _fn("orc_value_function_tr", N)
            f.restype = C.c_int
            st = f(C.byref(self.m), _dp(vin), _dp(self._scalar(r, N)), _dp(self._scalar(w, N)), _dp(self._scalar(tr, N)), _dp(V), _dp(KD))
        return st, self._mem_to_mat(V, N), self._mem_to_mat(KD, N)

    def backward_iteration(self, xr, xw, ss_end_value, N: int, xt=None):
        """BackwardIteration (BackwardIteration.jl:46-116). xr/xw: (P,1+N). -> (status, (P,n_a,n_e,1+N)).
        `xt` (P,1+N): the transfer path of the one-asset HANK family."""
        xr = np.ascontiguousarray(xr, dtype=np.float64)
        xw = np.ascontiguousarray(xw, dtype=np.float64)
        P = xr.shape[0]
        assert xr.shape == (P, 1 + N) and xw.shape == (P, 1 + N)
        vT = np.ascontiguousarray(np.asarray(ss_end_value, dtype=np.float64).T)  # [e][a]
        pol = np.empty((P, self.n_e, self.n_a, 1 + N))
        if xt is None:
            f = _fn("orc_backward_iteration", N)
            f.restype = C.c_int
            st = f(C.byref(self.m), P, _dp(xr), _dp(xw), _dp(vT), _dp(pol))
        else:
            xt = np.ascontiguousarray(xt, dtype=np.float64)
            assert xt.shape == (P, 1 + N)
            f = _fn("orc_backward_iteration_tr", N)
            f.restype = C.c_int
            st = f(C.byref(self.m), P, _dp(xr), _dp(xw), _dp(xt), _dp(vT), _dp(pol))
        return st, pol.transpose(0, 2, 1, 3).copy()

    def transition_step(self, policy, D_prev, N: int):
        """transition_step (ForwardIteration.jl:95-99). policy/D_prev: (n_a,n_e[,1+N])."""
        p = self._mat_to_mem(policy, N)
        d = self._mat_to_mem(np.asarray(D_prev).reshape((self.n_a, self.n_e) + np.asarray(D_prev).shape[2:], order="F")
                             if np.asarray(D_prev).ndim == 1 else D_prev, N)
        out = np.empty_like(p)
        _fn("orc_transition_step", N)(C.byref(self.m), _dp(p), _dp(d), _dp(out))
        return self._mem_to_mat(out, N)

    def forward_iteration(self, policy_seq, ss_init_D, N: int, return_D: bool = False):
        """ForwardIteration (ForwardIteration.jl:253-311). policy_seq: (P,n_a,n_e,1+N) -> agg (P,1+N)."""
        ps = np.ascontiguousarray(np.asarray(policy_seq, dtype=np.float64).transpose(0, 2, 1, 3))
        P = ps.shape[0]
        D0 = np.ascontiguousarray(np.asarray(ss_init_D, dtype=np.float64).reshape((self.n_a, self.n_e), order="F").T)
        agg = np.empty((P, 1 + N))
        Dseq = np.empty((P, self.n_e, self.n_a, 1 + N)) if return_D else None
        _fn("orc_forward_iteration", N)(C.byref(self.m), P, _dp(ps), _dp(D0), _dp(agg),
                                        _dp(Dseq) if return_D else None)
        if return_D:
            return agg, Dseq.transpose(0, 2, 1, 3).copy()
        return agg

    def household_block(self, xr, xw, ss_end_value, ss_init_D, N: int, xt=None):
        """ForwardIteration(BackwardIteration(...)) -> (status, agg (P,1+N), policy_seq)."""
        st, pol = self.backward_iteration(xr, xw, ss_end_value, N, xt)
        agg = self.forward_iteration(pol, ss_init_D, N)
        return st, agg, pol

    def household_block_het(self, xr, xw, ss_end_value, ss_init_D, N: int, xt=None):
        """the household block of a value function with TWO heterogeneous variables, savings and consumption (the c_grid of
        KrusellSmith.jl:79 returned as a policy): BackwardIteration keeps one policy sequence per variable
        (BackwardIteration.jl:99-112), ForwardIteration aggregates each with the same D_t (ForwardIteration.jl:303-307).
        -> (status, agg (2, P, 1+N), policy_seq, cons_seq)."""
        st, pol = self.backward_iteration(xr, xw, ss_end_value, N, xt)
        P = pol.shape[0]
        ps = np.ascontiguousarray(pol.transpose(0, 2, 1, 3))
        cons = np.empty_like(ps)
        xr = np.ascontiguousarray(xr, dtype=np.float64); xw = np.ascontiguousarray(xw, dtype=np.float64)
        xt_ = None if xt is None else np.ascontiguousarray(xt, dtype=np.float64)
        _fn("orc_consumption_policy", N)(C.byref(self.m), P, _dp(xr), _dp(xw), None if xt_ is None else _dp(xt_), _dp(ps), _dp(cons))
        seqs = np.ascontiguousarray(np.stack([ps, cons]))
        D0 = np.ascontiguousarray(np.asarray(ss_init_D, dtype=np.float64).reshape((self.n_a, self.n_e), order="F").T)
        agg = np.empty((2, P, 1 + N))
        _fn("orc_forward_iteration_het", N)(C.byref(self.m), P, 2, _dp(seqs), _dp(D0), _dp(agg))
        return st, agg, pol, cons.transpose(0, 2, 1, 3).copy()

    def ks_full_function(self, x, Z, alpha, delta, KS_ss_start, ss_end_value, ss_init_D, N: int):
        """fullFunction of y_Iteration for KrusellSmith.yaml (NewtonRaphson.jl:77-83).
        x: (4, P, 1+N) duals of (Y, KS, r, w). Returns (status, F (4,P,1+N), agg (P,1+N))."""
        x = np.asarray(x, dtype=np.float64)
        P = x.shape[1]
        assert x.shape == (4, P, 1 + N)
        xm = np.ascontiguousarray(x.transpose(1, 0, 2))  # [t][k][1+N] == column-major (4,P) of duals
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        vT = np.ascontiguousarray(np.asarray(ss_end_value, dtype=np.float64).T)
        D0 = np.ascontiguousarray(np.asarray(ss_init_D, dtype=np.float64).reshape((self.n_a, self.n_e), order="F").T)
        out = np.empty((P, 4, 1 + N))
        agg = np.empty((P, 1 + N))
        f = _fn("orc_ks_full_function", N)
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                      C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                      C.POINTER(C.c_double)]
        st = f(C.cast(C.byref(self.m), C.c_void_p), P, float(alpha), float(delta), _dp(xm), _dp(Z),
               float(KS_ss_start), _dp(vT), _dp(D0), _dp(out), _dp(agg))
        return st, out.transpose(1, 0, 2).copy(), agg

    def ks_jvp(self, x, y, Z, alpha, delta, KS_ss_start, ss_end_value, ss_init_D):
        """JVP(fullFunction, x, y) for a batch of tangents y (4,P,N): seeds x + t*y exactly like
        GeneralStructures.jl:546-547 (one partial per direction). Returns (F (4P,), J·y (4P,N))."""
        y = np.asarray(y, dtype=np.float64)
        if y.ndim == 2:
            y = y[:, :, None]
        Nreq = y.shape[2]
        N = pad_N(Nreq)
        P = y.shape[1]
        xd = np.zeros((4, P, 1 + N))
        xd[..., 0] = np.asarray(x, dtype=np.float64).reshape(4, P, order="F") if np.asarray(x).ndim == 1 else x
        xd[..., 1:1 + Nreq] = y
        st, F, _ = self.ks_full_function(xd, Z, alpha, delta, KS_ss_start, ss_end_value, ss_init_D, N)
        if st != ORC_OK:
            raise RuntimeError(f"oracle status {st}")
        Fv = F[..., 0].reshape(-1, order="F")
        J = F[..., 1:1 + Nreq].reshape(4 * P, Nreq, order="F")
        return Fv, J


def dual_binop(op: int, x, y, N: int) -> np.ndarray:
    out = np.empty(1 + N)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    _fn("orc_dual_binop", N)(int(op), _dp(x), _dp(y), _dp(out))
    return out
