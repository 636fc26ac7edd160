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


def host_threads():
    """context manager capping the BLAS / OpenMP thread pools for the enclosed host algebra."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:                      # optional dependency: run with the pools as they are
        return contextlib.nullcontext()
    try:
        n = int(os.environ.get("HANK_HOST_THREADS", "8"))
    except ValueError:
        n = 8
    if n <= 0:
        return contextlib.nullcontext()
    return threadpool_limits(limits=n)


def host_algebra(fn):
    """decorator: run a host driver under `host_threads()`."""
    @functools.wraps(fn)
    def wrapped(*a, **k):
        with host_threads():
            return fn(*a, **k)
    return wrapped
