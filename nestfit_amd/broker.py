"""Callback-coalescing broker: many sampler threads, one GPU batch.

MultiNest asks for one likelihood per ``LogLike`` call (reference:
nestfit/core/cmultinest.pxd:27-28, trampoline ``mn_loglikelihood``
nestfit/core/core.pyx:622-624).  ``LikelihoodBroker.loglikelihood`` is that call made
blocking and thread-safe: concurrent callers are gathered by the engine
(``nfa_broker_*`` in include/nestfit_amd.h) into one launch.  ctypes releases the GIL for
the duration of the call, so plain Python threads are enough to fill batches.
"""
import ctypes as C

import numpy as np

from . import _ffi
from .core import _as_inplace_vector


class BrokerClient(C.Structure):
    """``context`` argument for ``nfa_broker_callback`` (MultiNest LogLike signature)."""
    _fields_ = [('broker', C.c_void_p), ('pix', C.c_int32)]


class LikelihoodBroker:
    """
    Parameters
    ----------
    runner : AmmoniaRunner | DiazenyliumRunner | GaussianRunner | CubeRunner
        Must not be used directly while the broker serves it.
    max_batch : int
        Launch as soon as this many calls are queued.
    max_wait_us : int
        Longest time the first caller of a batch waits for company.
    n_clients : int
        Number of sampler threads that call concurrently (0 = unknown); when known, a batch
        launches as soon as every client has a call queued instead of waiting out the timer.
    """

    def __init__(self, runner, max_batch=4096, max_wait_us=200, n_clients=0):
        lib = _ffi.engine()
        self.runner = runner
        self.ndim = int(runner.ndim)
        h = C.c_void_p()
        _ffi.check(lib.nfa_broker_create(C.byref(h), runner._run.handle, int(max_batch),
                                         int(max_wait_us), int(n_clients)))
        self.handle = h

    def loglikelihood(self, utheta, pix=-1):
        """Blocking lnL of one unit-cube point (overwritten with the physical parameters,
        core.pyx:558-561); callable from any thread."""
        utheta = _as_inplace_vector(utheta)
        if utheta.shape[0] != self.ndim:
            raise ValueError(f'Invalid shape for ncomp={self.runner.ncomp}: {utheta.shape[0]}')
        lnl = C.c_double()
        _ffi.check(_ffi.load().nfa_broker_loglike(self.handle, int(pix), _ffi.dptr(utheta),
                                                  C.byref(lnl)))
        return lnl.value

    def set_clients(self, n_clients):
        _ffi.check(_ffi.load().nfa_broker_set_clients(self.handle, int(n_clients)))

    def client(self, pix=-1):
        """(callback, context) pair for a sampler that takes a C ``LogLike`` pointer."""
        ctx = BrokerClient(self.handle.value, int(pix))
        return _ffi.load().nfa_broker_callback, ctx

    def stats(self):
        out = (C.c_int64 * 3)()
        _ffi.check(_ffi.load().nfa_broker_stats(self.handle, out))
        n_batches, n_evals, largest = (int(v) for v in out)
        return dict(n_batches=n_batches, n_evals=n_evals, largest_batch=largest,
                    mean_batch=(n_evals / n_batches if n_batches else 0.0))

    def close(self):
        if getattr(self, 'handle', None) is not None:
            _ffi.check(_ffi.load().nfa_broker_destroy(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
