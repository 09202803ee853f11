#!/usr/bin/env python3
"""Device-resident throughput of the sibling models at the shape of config C2:
B = 4096 unit-cube rows per step, 1024 channels per spectrum, 2 components.
Same timing rules as bench.py (inputs in HBM, stream lanes, sync on both sides)."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import nestfit_amd as na                                   # noqa: E402
from nestfit_amd import _ffi, gaussian                     # noqa: E402
from scipy import stats                                    # noqa: E402

CKMS = 299792.458


def priors(ranges, size=500):
    x = np.linspace(0, 1, size)
    return na.PriorTransformer([
        na.Prior(na.Distribution(lo + x * (hi - lo), stats.uniform(lo, hi - lo).pdf(lo + x * (hi - lo))), k)
        for k, (lo, hi) in enumerate(ranges)])


def timed(runner, ndim, B=4096, steps=200, warmup=20):
    lib = _ffi.engine()
    rh = runner._run.handle
    n_total = steps + warmup
    U = np.random.default_rng(3).uniform(size=(B, ndim))
    d_U, d_L = C.c_void_p(), C.c_void_p()
    _ffi.check(lib.nfa_malloc(C.byref(d_U), n_total * B * ndim * 8))
    _ffi.check(lib.nfa_malloc(C.byref(d_L), n_total * B * 8))
    for k in range(n_total):
        _ffi.check(lib.nfa_memcpy_h2d(C.c_void_p(d_U.value + k * B * ndim * 8), U.ctypes.data_as(C.c_void_p),
                                      B * ndim * 8))

    def step(k):
        _ffi.check(lib.nfa_runner_loglike_batch_dev(rh, None, C.c_void_p(d_U.value + k * B * ndim * 8),
                                                    C.c_void_p(d_L.value + k * B * 8), B))
    for k in range(warmup):
        step(k)
    _ffi.check(lib.nfa_runner_synchronize(rh)); _ffi.check(lib.nfa_device_synchronize())
    t0 = time.perf_counter()
    for k in range(warmup, n_total):
        step(k)
    _ffi.check(lib.nfa_runner_synchronize(rh)); _ffi.check(lib.nfa_device_synchronize())
    dt = time.perf_counter() - t0
    _ffi.check(lib.nfa_free(d_U)); _ffi.check(lib.nfa_free(d_L))
    return B * steps / dt, dt / steps * 1e6


def main():
    rng = np.random.default_rng(1)
    n = 1024
    for mode in ('fast', 'table'):
        na.set_exp_mode(mode)
        # N2H+ 1-0 + 2-1
        args = []
        for t, nu0 in ((1, 93173.7637e6), (2, 186344.8420e6)):
            x = nu0 * (1.0 - np.linspace(20, -20, n) / CKMS)
            args.append([x, rng.normal(0, 0.2, n), 0.2, t])
        ut = priors([(-6, 6), (2.8, 20), (-1.5, 1.0), (0.1, 1.5)])
        r = na.DiazenyliumRunner.from_data(args, ut, ncomp=2)
        v, us = timed(r, 8)
        print(f'N2H+ (1-0)+(2-1) 2x1024 ch 2 comp, {mode}: {v/1e6:.1f} M evals/s, {us:.1f} us/step')
        # Gaussian
        nu0 = 110.201354e9
        x = nu0 * (1.0 - np.linspace(30, -30, n) / CKMS)
        utg = priors([(-20, 20), (0.2, 3.0), (0.0, 5.0)])
        g = gaussian.GaussianRunner.from_data([x, rng.normal(0, 0.3, n), 0.3, nu0], utg, ncomp=2)
        v, us = timed(g, 6)
        print(f'Gaussian 1x1024 ch 2 comp, {mode}: {v/1e6:.1f} M evals/s, {us:.1f} us/step')
    na.set_exp_mode('fast')


if __name__ == '__main__':
    main()
