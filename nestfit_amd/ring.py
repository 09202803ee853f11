"""Cross-process transport for the callback broker: sampler processes without a GPU
context, one serving process with the runner.

The reference fits a cube with one process per stripe (nestfit/main.py:516-523), each driving its own
MultiNest instance, which asks for one likelihood per ``LogLike`` call (nestfit/core/cmultinest.pxd:27-28,
trampoline ``mn_loglikelihood`` nestfit/core/core.pyx:622-624).  ``RingClient.loglikelihood`` is that call
for a process that never touches the GPU: the point goes into the process's slot of a POSIX
shared-memory ring (``nfa_ring_*`` in include/nestfit_amd.h), ``RingServer.serve`` -- in the one process
that owns the runner -- gathers the posted points of all processes into one launch and writes theta and
lnL back.  A client loads only ``libnestfit_amd_ring.so`` (host code, no HIP).
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from .core import _as_inplace_vector

_RING_SIGNATURES = ('nfa_ring_create', 'nfa_ring_create_multi', 'nfa_ring_attach', 'nfa_ring_close', 'nfa_ring_stop',
                    'nfa_ring_ndim', 'nfa_ring_slot', 'nfa_ring_max_points', 'nfa_ring_loglike', 'nfa_ring_loglike_many',
                    'nfa_ring_callback', 'nfa_ring_poll', 'nfa_ring_complete', 'nfa_ring_stats')
_ring_lib = None


def ring_library():
    """libnestfit_amd_ring.so (the transport alone) with its signatures bound."""
    global _ring_lib
    if _ring_lib is None:
        from .build import OUT_RING
        if not OUT_RING.exists():
            raise _ffi.EngineError(f'{OUT_RING} is missing: run `python -m nestfit_amd.build`')
        lib = C.CDLL(str(OUT_RING))
        for name in _RING_SIGNATURES:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = _ffi.SIGNATURES[name]
        lib.nfa_ring_last_error.restype = C.c_char_p
        _ring_lib = lib
    return _ring_lib


def _check(lib, rc, what):
    if rc != 0:
        msg = lib.nfa_ring_last_error().decode() if hasattr(lib, 'nfa_ring_last_error') else ''
        raise _ffi.EngineError(f'{what}: error {rc} {msg}'.strip())


class RingClientContext(C.Structure):
    """``context`` argument for ``nfa_ring_callback`` (MultiNest LogLike signature)."""
    _fields_ = [('ring', C.c_void_p), ('pix', C.c_int32)]


class RingClient:
    """A sampler process's end of the ring `name`: holds one slot until `close`.

    `wait_ms`: how long to wait for the serving process to create the ring."""

    def __init__(self, name, wait_ms=10000):
        self._lib = ring_library()
        h = C.c_void_p()
        _check(self._lib, self._lib.nfa_ring_attach(C.byref(h), os.fsencode(name), int(wait_ms)), f'attaching to ring {name}')
        self.handle = h
        self.ndim = self._lib.nfa_ring_ndim(h)
        self.slot = self._lib.nfa_ring_slot(h)
        self.max_points = self._lib.nfa_ring_max_points(h)

    def loglikelihood(self, utheta, pix=-1):
        """Blocking lnL of one unit-cube point (overwritten with the physical parameters, core.pyx:558-561)."""
        utheta = _as_inplace_vector(utheta)
        if utheta.shape[0] != self.ndim:
            raise ValueError(f'Invalid shape for ndim={self.ndim}: {utheta.shape[0]}')
        lnl = C.c_double()
        _check(self._lib, self._lib.nfa_ring_loglike(self.handle, int(pix), _ffi.dptr(utheta), C.byref(lnl)), 'ring loglike')
        return lnl.value

    def loglikelihood_many(self, U, pix=-1):
        """Blocking lnL of k <= `max_points` unit-cube points posted together (rows of U, overwritten with the physical
        parameters): for samplers whose next proposals do not depend on each other's likelihoods."""
        if not (isinstance(U, np.ndarray) and U.dtype == np.float64 and U.flags.c_contiguous and U.ndim == 2):
            raise TypeError('U must be a C-contiguous float64 array of shape (k, ndim): it is overwritten in place')
        if U.shape[1] != self.ndim:
            raise ValueError(f'Invalid shape for ndim={self.ndim}: {U.shape[1]}')
        lnl = np.empty(U.shape[0])
        _check(self._lib, self._lib.nfa_ring_loglike_many(self.handle, int(pix), _ffi.dptr(U), _ffi.dptr(lnl), int(U.shape[0])),
               'ring loglike_many')
        return lnl

    def callback(self, pix=-1):
        """(function pointer, context) for a C sampler: MultiNest's `LogLike` and its `context`."""
        self._ctx = RingClientContext(self.handle, int(pix))
        return self._lib.nfa_ring_callback, C.byref(self._ctx)

    def close(self):
        if self.handle:
            self._lib.nfa_ring_close(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class RingServer:
    """The serving end: creates the ring `name` with `n_slots` slots (one per sampler process).

    With a `runner` (AmmoniaRunner, CubeRunner, ...; it must not be used by anyone else meanwhile) `serve`
    runs the engine's native loop; `poll` / `complete` let a server put any evaluator between them."""

    def __init__(self, name, n_slots, runner=None, ndim=None, max_points=1):
        self.runner = runner
        self.ndim = int(runner.ndim if runner is not None else ndim)
        self._lib = _ffi.load() if runner is not None else ring_library()
        h = C.c_void_p()
        _check(self._lib, self._lib.nfa_ring_create_multi(C.byref(h), os.fsencode(name), int(n_slots), self.ndim, int(max_points)),
               f'creating ring {name}')
        self.handle, self.name, self.n_slots, self.max_points = h, name, int(n_slots), int(max_points)
        cap = self._cap = 1024
        self._slots = np.zeros(cap, dtype=np.int32)
        self._pix = np.zeros(cap, dtype=np.int32)
        self._U = np.zeros((cap, self.ndim))

    def serve_device(self, lifetime_ms=20, idle_ms=1000, runner=None):
        """Serve from a resident kernel (`nfa_ring_serve_device`): its workgroups poll the slots themselves and answer
        a posted point without a launch or a host thread in the round trip.  One point per slot; returns when the ring
        was stopped or nothing was served for `idle_ms`.  Run it in a thread like `serve`."""
        runner = runner if runner is not None else self.runner
        if runner is None:
            raise ValueError('serve_device needs a runner')
        _ffi.check(self._lib.nfa_ring_serve_device(self.handle, runner._run.handle, int(lifetime_ms), int(idle_ms)))

    def serve(self, max_wait_us=50, max_batches=0, idle_ms=1000, runner=None):
        """Serve until `stop`, `max_batches` (> 0) batches, or `idle_ms` without a request.  ctypes releases the
        GIL for the call: run it in a thread to keep the interpreter free.  Several threads may serve one ring,
        each with a `runner` of its own (same spectra and priors): the loops claim the posted points between
        them and their launches overlap on the GPU -- more sampler processes than one launch per round feeds."""
        runner = runner if runner is not None else self.runner
        if runner is None:
            raise ValueError('serve needs a runner; use poll / complete with your own evaluator')
        _ffi.check(self._lib.nfa_ring_serve(self.handle, runner._run.handle, int(max_wait_us), int(max_batches),
                                            int(idle_ms)))

    def serve_in_threads(self, runners, max_wait_us=50, idle_ms=1000):
        """One serving thread per runner; returns the started threads (join them after `stop`)."""
        import threading
        threads = [threading.Thread(target=self.serve, kwargs=dict(max_wait_us=max_wait_us, idle_ms=idle_ms, runner=r))
                   for r in runners]
        for t in threads:
            t.start()
        return threads

    def poll(self, max_batch=128, max_wait_us=50, idle_ms=1000):
        """(slots, pix, U, stopped) of the points gathered (the points of one request are neighbouring rows with the same
        slot); U is a view valid until the next poll."""
        n, stopped = C.c_int(), C.c_int()
        ip = C.POINTER(C.c_int32)
        _check(self._lib, self._lib.nfa_ring_poll(self.handle, min(int(max_batch), self._cap), int(max_wait_us), int(idle_ms),
                                                  self._slots.ctypes.data_as(ip), self._pix.ctypes.data_as(ip),
                                                  _ffi.dptr(self._U), C.byref(n), C.byref(stopped)), 'ring poll')
        return self._slots[:n.value], self._pix[:n.value], self._U[:n.value], bool(stopped.value)

    def complete(self, slots, U, lnL, rc=0):
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        U = np.ascontiguousarray(U, dtype=np.float64)
        lnL = np.ascontiguousarray(lnL, dtype=np.float64)
        _check(self._lib, self._lib.nfa_ring_complete(self.handle, int(slots.size), slots.ctypes.data_as(C.POINTER(C.c_int32)),
                                                      _ffi.dptr(U), _ffi.dptr(lnL), int(rc)), 'ring complete')

    def stop(self):
        self._lib.nfa_ring_stop(self.handle)

    @property
    def stats(self):
        out = (C.c_int64 * 4)()
        self._lib.nfa_ring_stats(self.handle, out)
        return {'batches': out[0], 'evals': out[1], 'largest_batch': out[2], 'clients': out[3]}

    def close(self):
        if self.handle:
            self._lib.nfa_ring_close(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.stop()
        self.close()
