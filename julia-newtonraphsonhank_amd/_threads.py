"""Host-side algebra between device calls: keep the BLAS / OpenMP pools small.

The drivers alternate short device calls with small host algebra (a 22 000-element dot product, an LU solve of a
1 196² matrix, the residual layer). On a many-core host the default pool (one thread per core: 64 on the MI355X box)
costs far more than it computes: woken after a few milliseconds of idling, a dot product of 22 000 numbers takes 60 ms
instead of 6 µs, and the spinning workers compete with the runtime's launch threads — the cold steady state at 2000×11
took 3.2 s with the default pool and 1.35 s with 8 threads, the Newton solve 0.38 s and 0.25 s (scripts/dev_ss_profile.py).
`HANK_HOST_THREADS` overrides the cap (default 8); without threadpoolctl installed nothing is changed."""
from __future__ import annotations

import contextlib
import functools
import os


_CONTROLLER = None


def host_threads():
    """context manager CAPPING the BLAS / OpenMP thread pools at HANK_HOST_THREADS for the enclosed host algebra: a pool that
    is already smaller (OMP_NUM_THREADS=1 under torchrun, say) is left as it is — `threadpool_limits(limits=n)` alone would
    raise it to n."""
    global _CONTROLLER
    try:
        from threadpoolctl import ThreadpoolController
    except ImportError:                      # optional dependency: run with the pools as they are
        return contextlib.nullcontext()
    try:
        n = int(os.environ.get("HANK_HOST_THREADS", "8"))
    except ValueError:
        n = 8
    if n <= 0:
        return contextlib.nullcontext()
    # the scan is cheap next to the algebra it guards and a BLAS / OpenMP runtime loaded later (a late torch import, say) must
    # be capped too: re-scan when the set of loaded libraries has changed
    fresh = ThreadpoolController()
    if _CONTROLLER is None or len(fresh.lib_controllers) != len(_CONTROLLER.lib_controllers):
        _CONTROLLER = fresh
    over = [lib for lib in _CONTROLLER.lib_controllers if (lib.num_threads or 0) > n]
    if not over:
        return contextlib.nullcontext()
    limits = {lib.prefix: n for lib in over if lib.prefix}
    return _CONTROLLER.limit(limits=limits) if limits else contextlib.nullcontext()


def host_algebra(fn):
    """decorator: run a host driver under `host_threads()`."""
    @functools.wraps(fn)
    def wrapped(*a, **k):
        with host_threads():
            return fn(*a, **k)
    return wrapped
