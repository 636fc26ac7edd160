"""ctypes binding of the C ABI in include/hank_hip.h (libhank_hip.so, hand-written HIP for gfx950).

There is no CPU fallback on this path: if the shared library is missing or no MI355X is usable the
constructors raise — the product never silently computes the household block anywhere else.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

import os

# HANK_HIP_LIB: a dev knob (instrumented builds of the same library, e.g. `make stamp`)
_LIB_PATH = Path(os.environ.get("HANK_HIP_LIB") or Path(__file__).resolve().parent / "libhank_hip.so")
_lib = None

HANK_OK = 0
HANK_ERR_NO_DEVICE, HANK_ERR_BAD_ARG, HANK_ERR_KNOTS, HANK_ERR_DOMAIN = 1, 2, 3, 4
HANK_ERR_NOT_READY, HANK_ERR_NONMONOTONE, HANK_ERR_NOMEM, HANK_ERR_SWEEP, HANK_ERR_LAUNCH = 5, 6, 7, 8, 9
HANK_VF_KRUSELL_SMITH = 0

# the symbols include/hank_hip.h declares (tests check that every one is exported)
ABI_SYMBOLS = (
    "hank_create", "hank_create_on", "hank_gather_columns", "hank_destroy", "hank_last_error", "hank_n_hh", "hank_set_stream", "hank_sync",
    "hank_set_boundary", "hank_primal", "hank_jvp", "hank_primal_dev", "hank_jvp_dev", "hank_check",
    "hank_primal_jvp", "hank_primal_jvp_dev",
    "hank_get_policy_seq", "hank_get_dpolicy_seq", "hank_get_dist_seq", "hank_get_het_outputs", "hank_get_het_outputs_dev",
    "hank_get_grid_aggregates", "hank_get_grid_aggregates_dev", "hank_backward_step",
    "hank_backward_step_dual", "hank_forward_step", "hank_forward_step_dual", "hank_last_timings", "hank_stats", "hank_info", "hank_vfi", "hank_stationary_dist", "hank_fake_news", "hank_device_available",
)


class HankHIPError(RuntimeError):
    """Base class: a non-zero status from libhank_hip (message = hank_last_error)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"[hank_hip status {code}] {msg}")
        self.code = code


class KnotsNotSortedError(HankHIPError):
    """Interpolations.jl's 'knot-vectors must be unique and sorted' (KrusellSmith.jl:69-72)."""


class DomainError(HankHIPError):
    """Julia's DomainError: negative base under a non-integer power (KrusellSmith.jl:59, :80)."""


class NoDeviceError(HankHIPError):
    """No usable gfx950 device / HIP runtime failure."""


_ERR_CLASSES = {HANK_ERR_KNOTS: KnotsNotSortedError, HANK_ERR_DOMAIN: DomainError,
                HANK_ERR_NO_DEVICE: NoDeviceError}


class hank_model(C.Structure):
    _fields_ = [("n_a", C.c_int32), ("n_e", C.c_int32), ("T", C.c_int32), ("value_fn_id", C.c_int32),
                ("a_grid", C.POINTER(C.c_double)), ("z_grid", C.POINTER(C.c_double)),
                ("Pi", C.POINTER(C.c_double)),
                ("beta", C.c_double), ("gamma", C.c_double), ("borrow_cons", C.c_double)]


def library_path() -> Path:
    return _LIB_PATH


def load_library() -> C.CDLL:
    """dlopen libhank_hip.so (built by __graft_entry__.build() / csrc/Makefile). Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise RuntimeError(
            f"{_LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(str(_LIB_PATH))
    dp, vp, i32 = C.POINTER(C.c_double), C.c_void_p, C.c_int32
    lib.hank_create.argtypes = [C.POINTER(hank_model), C.POINTER(vp)]
    lib.hank_create_on.argtypes = [C.POINTER(hank_model), i32, C.POINTER(vp)]
    lib.hank_destroy.argtypes = [vp]
    lib.hank_gather_columns.argtypes = [C.POINTER(vp), i32, C.POINTER(vp), C.POINTER(i32), vp]
    lib.hank_last_error.argtypes = [vp]
    lib.hank_last_error.restype = C.c_char_p
    lib.hank_n_hh.argtypes = [vp]
    lib.hank_set_stream.argtypes = [vp, vp]
    lib.hank_sync.argtypes = [vp]
    lib.hank_set_boundary.argtypes = [vp, dp, dp]
    lib.hank_primal.argtypes = [vp, dp, dp]
    lib.hank_jvp.argtypes = [vp, dp, i32, dp]
    lib.hank_primal_dev.argtypes = [vp, vp, vp]
    lib.hank_jvp_dev.argtypes = [vp, vp, i32, vp]
    lib.hank_check.argtypes = [vp]
    lib.hank_primal_jvp.argtypes = [vp, dp, dp, i32, dp, dp]
    lib.hank_primal_jvp_dev.argtypes = [vp, vp, vp, i32, vp, vp]
    lib.hank_get_policy_seq.argtypes = [vp, dp]
    lib.hank_get_dpolicy_seq.argtypes = [vp, i32, dp]
    lib.hank_get_dist_seq.argtypes = [vp, dp]
    lib.hank_get_het_outputs.argtypes = [vp, i32, dp, i32, dp, dp]
    lib.hank_get_het_outputs_dev.argtypes = [vp, i32, vp, i32, vp, vp]
    lib.hank_get_grid_aggregates.argtypes = [vp, dp, i32, dp]
    lib.hank_get_grid_aggregates_dev.argtypes = [vp, vp, i32, vp]
    lib.hank_backward_step.argtypes = [vp, dp, dp, dp, dp]
    lib.hank_backward_step_dual.argtypes = [vp, dp, dp, dp, dp, i32, dp, dp, dp, dp]
    lib.hank_forward_step.argtypes = [vp, dp, dp, dp, dp]
    lib.hank_forward_step_dual.argtypes = [vp, dp, dp, dp, dp, i32, dp, dp, dp, dp]
    lib.hank_last_timings.argtypes = [vp, dp, C.POINTER(i32)]
    lib.hank_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.hank_info.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.hank_vfi.argtypes = [vp, dp, C.c_double, i32, dp, dp, C.POINTER(i32), dp]
    lib.hank_stationary_dist.argtypes = [vp, dp, dp, C.c_double, i32, i32, C.POINTER(i32)]
    lib.hank_fake_news.argtypes = [vp, dp, dp]
    lib.hank_device_available.argtypes = []
    for name in ABI_SYMBOLS:
        if name != "hank_last_error":
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def device_available() -> bool:
    """True when libhank_hip.so is built and the current HIP device is an MI355X (gfx950)."""
    try:
        return bool(load_library().hank_device_available())
    except (RuntimeError, OSError):
        return False


def gather_columns(blocks, d_block_ptrs, N_k, d_out_ptr: int):
    """hank_gather_columns: the (P, N_k) column blocks the contexts `blocks` hold in their own GPUs' memory (device pointers
    `d_block_ptrs`, as written by jvp_dev / primal_jvp_dev) assembled as one (P, sum N_k) matrix at `d_out_ptr` on blocks[0]'s device
    — over xGMI where the devices differ. Asynchronous: blocks[0].sync() before reading."""
    lib = load_library()
    n = len(blocks)
    ctxs = (C.c_void_p * n)(*[b._ctx for b in blocks])
    ptrs = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in d_block_ptrs])
    nk = (C.c_int32 * n)(*[int(v) for v in N_k])
    rc = lib.hank_gather_columns(ctxs, n, ptrs, nk, C.c_void_p(int(d_out_ptr)))
    if rc != HANK_OK:
        raise _ERR_CLASSES.get(rc, HankHIPError)(rc, lib.hank_last_error(blocks[0]._ctx).decode())


def _f(x, shape=None) -> np.ndarray:
    """contiguous column-major float64 copy (Julia's memory layout), optionally shape-checked."""
    arr = np.asfortranarray(np.asarray(x, dtype=np.float64))
    if shape is not None and tuple(arr.shape) != tuple(shape):
        raise ValueError(f"expected array of shape {tuple(shape)}, got {tuple(arr.shape)}")
    return arr


def _p(arr: np.ndarray):
    return arr.ctypes.data_as(C.POINTER(C.c_double))


class HouseholdBlock:
    """One hank_ctx: the device-resident household block of a SequenceModel.

    Mirrors what `BackwardIteration` / `ForwardIteration` read from `model` (grids, Π, β, γ,
    borrow_cons, T; GeneralStructures.jl:216-226) and from `ss_end` / `ss_initial`.
    All matrices are (n_a, n_e); numpy arrays of that shape are accepted in any memory order.
    """

    def __init__(self, a_grid, z_grid, Pi, beta, gamma, borrow_cons, T, value_fn_id=HANK_VF_KRUSELL_SMITH, device=None):
        """device: HIP device ordinal of the context (hank_create_on); None = the calling thread's current device."""
        self._lib = load_library()
        self.device = device
        self._ctx = C.c_void_p()
        self.a_grid = np.ascontiguousarray(a_grid, dtype=np.float64)
        self.z_grid = np.ascontiguousarray(z_grid, dtype=np.float64)
        self.n_a, self.n_e = self.a_grid.size, self.z_grid.size
        self.Pi = _f(Pi, (self.n_e, self.n_e))
        self.T, self.P, self.G = int(T), int(T) - 1, self.n_a * self.n_e
        self._params = (float(beta), float(gamma), float(borrow_cons), value_fn_id)
        self._boundary = None
        m = hank_model(self.n_a, self.n_e, self.T, value_fn_id, _p(self.a_grid), _p(self.z_grid),
                       _p(self.Pi), float(beta), float(gamma), float(borrow_cons))
        if device is None:
            rc = self._lib.hank_create(C.byref(m), C.byref(self._ctx))
        else:
            rc = self._lib.hank_create_on(C.byref(m), int(device), C.byref(self._ctx))
        if rc != HANK_OK:
            msg = self._lib.hank_last_error(self._ctx).decode() if self._ctx else "hank_create failed"
            if not msg and rc == HANK_ERR_NO_DEVICE:
                msg = "no HIP device available"
            if self._ctx:
                self._lib.hank_destroy(self._ctx)
                self._ctx = C.c_void_p()
            raise _ERR_CLASSES.get(rc, HankHIPError)(rc, msg or "no HIP device available (gfx950 required)")
        self.n_hh = self._lib.hank_n_hh(self._ctx)
        # sweeps issued through this context, by entry point (tests assert "ONE sweep per fullFunction")
        self.calls = {"primal": 0, "jvp": 0, "primal_jvp": 0}

    # -- plumbing ---------------------------------------------------------------------------
    def _chk(self, rc: int):
        if rc != HANK_OK:
            raise _ERR_CLASSES.get(rc, HankHIPError)(rc, self._lib.hank_last_error(self._ctx).decode())

    def clone(self, device=None) -> "HouseholdBlock":
        """A second, independent context of the same model and boundary (its own device memory, graphs and
        stream): independent tangent batches — Jacobian column chunks — can be in flight on both. `device`: put the
        clone on another GPU of the node (default: this context's)."""
        beta, gamma, bc, vf = self._params
        other = HouseholdBlock(self.a_grid, self.z_grid, self.Pi, beta, gamma, bc, self.T, vf, device=self.device if device is None else device)
        if self._boundary is not None:
            other.set_boundary(*self._boundary)
        return other

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.hank_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream_handle: int | None):
        self._chk(self._lib.hank_set_stream(self._ctx, C.c_void_p(hip_stream_handle or 0)))

    def sync(self):
        self._chk(self._lib.hank_sync(self._ctx))

    def check(self):
        self._chk(self._lib.hank_check(self._ctx))

    # -- boundary + fused sweeps ------------------------------------------------------------
    def set_boundary(self, ss_end_value, ss_init_D):
        v = _f(ss_end_value, (self.n_a, self.n_e))
        d = _f(np.asarray(ss_init_D, dtype=np.float64).reshape((self.n_a, self.n_e), order="F"))
        self._chk(self._lib.hank_set_boundary(self._ctx, _p(v), _p(d)))
        self._boundary = (v, d)

    def primal(self, xhh) -> np.ndarray:
        x = _f(xhh, (self.n_hh, self.P))
        agg = np.empty(self.P)
        self.calls["primal"] += 1
        self._chk(self._lib.hank_primal(self._ctx, _p(x), _p(agg)))
        return agg

    def jvp(self, dxhh) -> np.ndarray:
        dx = np.asarray(dxhh, dtype=np.float64)
        if dx.ndim == 2:
            dx = dx[:, :, None]
        N = dx.shape[2]
        dx = _f(dx, (self.n_hh, self.P, N))
        out = np.empty((self.P, N), order="F")
        self.calls["jvp"] += 1
        self._chk(self._lib.hank_jvp(self._ctx, _p(dx), N, _p(out)))
        return out

    def primal_jvp(self, xhh, dxhh):
        """value and N partials in one dual-sweep pass (what JVP(fullFunction, x, y) does in the reference)."""
        x = _f(xhh, (self.n_hh, self.P))
        dx = np.asarray(dxhh, dtype=np.float64)
        if dx.ndim == 2:
            dx = dx[:, :, None]
        N = dx.shape[2]
        dx = _f(dx, (self.n_hh, self.P, N))
        agg = np.empty(self.P)
        dagg = np.empty((self.P, N), order="F")
        self.calls["primal_jvp"] += 1
        self._chk(self._lib.hank_primal_jvp(self._ctx, _p(x), _p(dx), N, _p(agg), _p(dagg)))
        return agg, dagg

    def primal_jvp_dev(self, d_xhh_ptr: int, d_dxhh_ptr: int, N: int, d_agg_ptr: int = 0, d_dagg_ptr: int = 0):
        self._chk(self._lib.hank_primal_jvp_dev(self._ctx, C.c_void_p(d_xhh_ptr), C.c_void_p(d_dxhh_ptr), int(N),
                                                C.c_void_p(d_agg_ptr), C.c_void_p(d_dagg_ptr)))

    def primal_dev(self, d_xhh_ptr: int, d_agg_ptr: int = 0):
        self._chk(self._lib.hank_primal_dev(self._ctx, C.c_void_p(d_xhh_ptr), C.c_void_p(d_agg_ptr)))

    def jvp_dev(self, d_dxhh_ptr: int, N: int, d_dagg_ptr: int = 0):
        self._chk(self._lib.hank_jvp_dev(self._ctx, C.c_void_p(d_dxhh_ptr), int(N), C.c_void_p(d_dagg_ptr)))

    def last_timings(self):
        ms = (C.c_double * 6)()
        ln = (C.c_int32 * 6)()
        self._chk(self._lib.hank_last_timings(self._ctx, ms, ln))
        names = ("primal_backward", "primal_forward", "tangent_backward", "tangent_forward", "dual_backward", "dual_forward")
        return {k: {"ms": ms[i], "launches": ln[i]} for i, k in enumerate(names)}

    def stats(self):
        """counters of the context (include/hank_hip.h: hank_stats)."""
        out = (C.c_int64 * 8)()
        self._chk(self._lib.hank_stats(self._ctx, out))
        names = ("sweep_launches", "tangent_workspaces_allocated", "graphs_captured", "schedule", "fallbacks", "vfi_iterations",
                 "primal_memo_hits", "primal_sweeps")
        return {k: int(out[i]) for i, k in enumerate(names)}

    def info(self):
        """which kernel family served the last tangent sweep and how the context chooses (include/hank_hip.h: hank_info)."""
        out = (C.c_int64 * 8)()
        self._chk(self._lib.hank_info(self._ctx, out))
        names = ("last_tangent_family", "wide_mode", "wide_min", "wide_supported", "xjvp_max", "record_diet", "record_bytes", "reserved")
        d = {k: int(out[i]) for i, k in enumerate(names)}
        d["last_tangent_family_name"] = ("launch-per-period", "xcd-persistent", "on-chip-wide")[d["last_tangent_family"]]
        return d

    def policy_seq(self) -> np.ndarray:
        """(n_a, n_e, P): policy matrix of every period (BackwardIteration's return value)."""
        out = np.empty((self.n_a, self.n_e, self.P), order="F")
        self._chk(self._lib.hank_get_policy_seq(self._ctx, _p(out)))
        return out

    def dist_seq(self) -> np.ndarray:
        out = np.empty((self.n_a, self.n_e, self.P), order="F")
        self._chk(self._lib.hank_get_dist_seq(self._ctx, _p(out)))
        return out

    def dpolicy_seq(self, N: int) -> np.ndarray:
        """(n_a, n_e, P, N): partials of the policy sequence of the last jvp."""
        out = np.empty((self.n_a, self.n_e, self.P, N), order="F")
        self._chk(self._lib.hank_get_dpolicy_seq(self._ctx, int(N), _p(out)))
        return out

    def het_outputs(self, n_het: int = 2, dxhh=None):
        """every heterogeneous variable's aggregate of the last sweeps (hank_get_het_outputs; ForwardIteration.jl:303-307):
        output 0 the policy variable (KD / A), output 1 consumption. -> agg (P, n_het), and dagg (P, n_het, N) when `dxhh`
        (n_hh, P, N) — the input of the last tangent sweep — is given (else None)."""
        agg = np.empty((self.P, n_het), order="F")
        if dxhh is None:
            self._chk(self._lib.hank_get_het_outputs(self._ctx, int(n_het), None, 0, _p(agg), None))
            return agg, None
        dxhh = np.asfortranarray(dxhh, dtype=np.float64)
        if dxhh.ndim != 3 or dxhh.shape[:2] != (self.n_hh, self.P):
            raise ValueError(f"dxhh must be ({self.n_hh}, {self.P}, N), got {dxhh.shape}")
        N = dxhh.shape[2]
        dagg = np.empty((self.P, n_het, N), order="F")
        self._chk(self._lib.hank_get_het_outputs(self._ctx, int(n_het), _p(dxhh), N, _p(agg), _p(dagg)))
        return agg, dagg

    def het_outputs_dev(self, n_het: int, d_dxhh: int, N: int, d_agg: int, d_dagg: int):
        """device-pointer form (asynchronous on the context's stream)."""
        self._chk(self._lib.hank_get_het_outputs_dev(self._ctx, int(n_het), d_dxhh or None, int(N), d_agg or None, d_dagg or None))

    def grid_aggregates(self, N: int = 0):
        """the grid-weighted aggregate sum_pt a(pt) D_t(pt) of the last primal sweep, and (N > 0) its partials (P, N) of the
        last tangent sweep (hank_get_grid_aggregates)."""
        agg2 = np.empty(self.P)
        dagg2 = np.empty((self.P, N), order="F") if N > 0 else None
        self._chk(self._lib.hank_get_grid_aggregates(self._ctx, _p(agg2), int(N), _p(dagg2) if N > 0 else None))
        return agg2, dagg2

    def fake_news(self):
        """the household block's Jacobian at the steady state from its Toeplitz structure (hank_fake_news):
        -> F (P, P, n_hh), Dv (P, n_hh); `SteadyStateJacobian.household_jacobian` turns them into d agg_t / d xhh_{k,s}."""
        F = np.empty((self.P, self.P, self.n_hh), order="F")
        Dv = np.empty((self.P, self.n_hh), order="F")
        self._chk(self._lib.hank_fake_news(self._ctx, _p(F), _p(Dv)))
        return F, Dv

    def vfi(self, value0, xhh_t, tol: float, max_iter: int = 10_000):
        """device-resident inner fixed point of the steady state (hank_vfi; SteadyState.jl:132-141):
        -> (value, policy, steps, last sup-norm)."""
        v = _f(np.array(value0, dtype=np.float64, copy=True), (self.n_a, self.n_e))
        x = _f(xhh_t, (self.n_hh,))
        pol = np.empty((self.n_a, self.n_e), order="F")
        it, nrm = C.c_int32(), C.c_double()
        self._chk(self._lib.hank_vfi(self._ctx, _p(x), float(tol), int(max_iter), _p(v), _p(pol), C.byref(it), C.byref(nrm)))
        return v, pol, it.value, nrm.value

    def stationary_dist(self, policy, D0=None, tol: float = 1e-15, max_iter: int = 500_000, check_every: int = 25):
        """stationary distribution of the steady state by the power method on the device (hank_stationary_dist):
        -> (D as a length-G vector summing to one, steps)."""
        p = _f(policy, (self.n_a, self.n_e))
        d = np.full(self.G, 1.0 / self.G) if D0 is None else np.asarray(D0, dtype=np.float64).reshape(-1, order="F") / np.sum(D0)
        d = _f(d.reshape((self.n_a, self.n_e), order="F"))
        it = C.c_int32()
        self._chk(self._lib.hank_stationary_dist(self._ctx, _p(p), _p(d), float(tol), int(max_iter), int(check_every), C.byref(it)))
        out = d.reshape(-1, order="F")
        return out / out.sum(), it.value

    # -- granular steps ---------------------------------------------------------------------
    def backward_step(self, value_next, xhh_t):
        v = _f(value_next, (self.n_a, self.n_e))
        x = _f(xhh_t, (self.n_hh,))
        vo = np.empty((self.n_a, self.n_e), order="F")
        po = np.empty((self.n_a, self.n_e), order="F")
        self._chk(self._lib.hank_backward_step(self._ctx, _p(v), _p(x), _p(vo), _p(po)))
        return vo, po

    def backward_step_dual(self, value_next, dvalue_next, xhh_t, dxhh_t):
        dv = np.asarray(dvalue_next, dtype=np.float64)
        N = dv.shape[-1]
        v = _f(value_next, (self.n_a, self.n_e))
        dv = _f(dv, (self.n_a, self.n_e, N))
        x = _f(xhh_t, (self.n_hh,))
        dx = _f(dxhh_t, (self.n_hh, N))
        vo = np.empty((self.n_a, self.n_e), order="F")
        po = np.empty((self.n_a, self.n_e), order="F")
        dvo = np.empty((self.n_a, self.n_e, N), order="F")
        dpo = np.empty((self.n_a, self.n_e, N), order="F")
        self._chk(self._lib.hank_backward_step_dual(self._ctx, _p(v), _p(dv), _p(x), _p(dx), N,
                                                    _p(vo), _p(dvo), _p(po), _p(dpo)))
        return vo, dvo, po, dpo

    def forward_step(self, policy, D_prev):
        p = _f(policy, (self.n_a, self.n_e))
        d = _f(np.asarray(D_prev, dtype=np.float64).reshape((self.n_a, self.n_e), order="F"))
        do = np.empty((self.n_a, self.n_e), order="F")
        agg = C.c_double()
        self._chk(self._lib.hank_forward_step(self._ctx, _p(p), _p(d), _p(do), C.byref(agg)))
        return do, agg.value

    def forward_step_dual(self, policy, dpolicy, D_prev, dD_prev):
        dp_ = np.asarray(dpolicy, dtype=np.float64)
        N = dp_.shape[-1]
        p = _f(policy, (self.n_a, self.n_e))
        dp_ = _f(dp_, (self.n_a, self.n_e, N))
        d = _f(np.asarray(D_prev, dtype=np.float64).reshape((self.n_a, self.n_e), order="F"))
        dd = _f(np.asarray(dD_prev, dtype=np.float64).reshape((self.n_a, self.n_e, N), order="F"))
        do = np.empty((self.n_a, self.n_e), order="F")
        ddo = np.empty((self.n_a, self.n_e, N), order="F")
        agg = C.c_double()
        dagg = np.empty(N)
        self._chk(self._lib.hank_forward_step_dual(self._ctx, _p(p), _p(dp_), _p(d), _p(dd), N,
                                                   _p(do), _p(ddo), C.byref(agg), _p(dagg)))
        return do, ddo, agg.value, dagg
