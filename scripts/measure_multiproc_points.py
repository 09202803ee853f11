#!/usr/bin/env python3
"""The reference's way of using more than one core: one process per longitude stripe, each with its own
serial sampler calling LogLike point by point (nestfit/main.py:516-523).  Here: N processes on ONE GPU,
each with its own runner (its own HIP context and hardware queue), each timing serial calls of
nfa_loglike_callback from native code.  Point kernels of different processes run side by side (one
workgroup each), so the aggregate rate says what sharing a GPU between stripe processes gives without any
broker between them."""
import multiprocessing as mp
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def worker(rank, n_calls, start, out):
    import nestfit_amd as na
    from nestfit_amd import _ffi
    from nestfit_amd.synth import TRUTH_2COMP, freq_axis
    rng = np.random.default_rng(rank)
    args = []
    for t in (1, 2):
        x = freq_axis(t, 1024)
        s = na.AmmoniaSpectrum(x, np.zeros(1024), 0.2, t)
        na.amm_predict(s, TRUTH_2COMP)
        args.append([x, s.get_spec() + rng.normal(0, 0.2, 1024), 0.2, t])
    run = na.AmmoniaRunner.from_data(args, na.get_irdc_priors(size=500, vsys=0.0), ncomp=2)
    u = rng.uniform(size=run.ndim)
    tl = _ffi.test_engine()
    sec, last = np.zeros(1), np.zeros(1)
    cb = _ffi.loglike_callback_address()
    _ffi.test_check(tl.nfa_test_callback_latency(cb, run._run.handle, run.ndim, _ffi.dptr(u), 200, _ffi.dptr(last), _ffi.dptr(sec)))
    start.wait()
    t0 = time.perf_counter()
    _ffi.test_check(tl.nfa_test_callback_latency(cb, run._run.handle, run.ndim, _ffi.dptr(u), n_calls, _ffi.dptr(last), _ffi.dptr(sec)))
    out.put((rank, t0, time.perf_counter(), float(sec[0]), float(last[0])))


def main():
    ctx = mp.get_context('spawn')
    n_calls = 20000
    for n_proc in (1, 2, 4, 6):                     # at most 6 processes may use the card on this pool
        start = ctx.Barrier(n_proc)
        out = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(k, n_calls, start, out)) for k in range(n_proc)]
        for p in procs:
            p.start()
        res = [out.get(timeout=600) for _ in procs]
        for p in procs:
            p.join()
        t_first = min(r[1] for r in res)
        t_last = max(r[2] for r in res)
        per_call = np.mean([r[3] for r in res]) / n_calls
        print(f'{n_proc} processes: {n_proc * n_calls / (t_last - t_first) / 1e3:7.1f} k evals/s in all, '
              f'{per_call * 1e6:.1f} us per call in each', flush=True)


if __name__ == '__main__':
    main()
