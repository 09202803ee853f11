"""PCIe-inclusive rate of the host-pointer entry point (GPU box): U and lnL travel over
PCIe on every call, nfa_runner_loglike_batch synchronises before returning."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import nestfit_amd as na
from nestfit_amd.synth import TRUTH_2COMP, freq_axis
n = 1024
rng = np.random.default_rng(5)
axes = [freq_axis(t, n) for t in (1, 2)]
spec = []
for t, x in zip((1, 2), axes):
    s = na.AmmoniaSpectrum(x, np.zeros(n), 0.2, t)
    na.amm_predict(s, TRUTH_2COMP)
    spec.append([x, s.get_spec() + rng.normal(0, 0.2, n), 0.2, t])
run = na.AmmoniaRunner.from_data(spec, na.get_irdc_priors(), ncomp=2)
from nestfit_amd import _ffi
for point in (1, 0):
    _ffi.set_option('point', point)
    u0 = rng.uniform(size=12)
    for _ in range(20):
        run.loglikelihood(u0.copy())
    bufs = [u0.copy() for _ in range(2000)]
    t0 = time.perf_counter()
    for b in bufs:
        run.loglikelihood(b)
    dt = (time.perf_counter() - t0) / len(bufs)
    print(f'single point, {"point kernel" if point else "batch kernels as a graph"}: {dt*1e6:.1f} us per call', flush=True)
_ffi.set_option('point', 1)
# the same without Python in the loop: MultiNest's LogLike (nfa_loglike_callback) called from native code
import ctypes as C
tl = _ffi.test_engine()
sec, last = np.zeros(1), np.zeros(1)
for point in (1, 0):
    _ffi.set_option('point', point)
    for n_calls in (50, 5000):
        _ffi.test_check(tl.nfa_test_callback_latency(_ffi.loglike_callback_address(), run._run.handle, 12, _ffi.dptr(u0),
                                                     n_calls, _ffi.dptr(last), _ffi.dptr(sec)))
    print(f'native callback, {"point kernel" if point else "batch kernels as a graph"}: {sec[0] / 5000 * 1e6:.1f} us per call '
          f'(lnL {last[0]:.6f})', flush=True)
_ffi.set_option('point', 1)
for B in (1, 16, 64, 128, 129, 400, 4096, 65536):
    U = rng.uniform(size=(B, 12))
    run.loglikelihood_batch(U.copy())
    reps = 200 if B <= 4096 else 20
    bufs = [U.copy() for _ in range(reps)]
    t0 = time.perf_counter()
    for b in bufs:
        run.loglikelihood_batch(b)
    dt = (time.perf_counter() - t0) / reps
    print(f'host API B={B}: {dt*1e6:.1f} us per call, {B/dt/1e6:.3f} M evals/s (PCIe + sync inclusive)')
# the same calls with buffers the device addresses itself (nestfit_amd.pinned_empty): no copies in or out
import nestfit_amd as _na
for B in (400, 4096, 16384, 65536):
    U = rng.uniform(size=(B, 12))
    pu, pl = _na.pinned_empty((B, 12)), _na.pinned_empty(B)
    reps = 200 if B <= 4096 else 20
    pu[...] = U
    run.loglikelihood_batch(pu, out=pl)
    t_fill = time.perf_counter()
    for _ in range(reps):
        pu[...] = U
    t_fill = (time.perf_counter() - t_fill) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        pu[...] = U                                  # (the caller's own write of the unit cube: timed and subtracted)
        run.loglikelihood_batch(pu, out=pl)
    dt = (time.perf_counter() - t0) / reps - t_fill
    print(f'host API, pinned buffers, B={B}: {dt*1e6:.1f} us per call, {B/dt/1e6:.3f} M evals/s (sync inclusive)')

