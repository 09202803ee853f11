#!/usr/bin/env python3
"""Throughput of the callback-coalescing broker against the number of sampler threads
(config C2 pixel, one LogLike-shaped blocking call per evaluation; Python threads, the GIL is
released inside the call)."""
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import nestfit_amd as na                                   # noqa: E402
from nestfit_amd.broker import LikelihoodBroker            # noqa: E402
from nestfit_amd.synth import TRUTH_2COMP, freq_axis       # noqa: E402


def main():
    rng = np.random.default_rng(0)
    args = []
    for t in (1, 2):
        x = freq_axis(t, 1024)
        s = na.AmmoniaSpectrum(x, np.zeros(1024), 0.2, t)
        na.amm_predict(s, TRUTH_2COMP)
        args.append([x, s.get_spec() + rng.normal(0, 0.2, 1024), 0.2, t])
    ut = na.get_irdc_priors(size=500, vsys=0.0)
    run = na.AmmoniaRunner.from_data(args, ut, ncomp=2)
    u = rng.uniform(size=run.ndim)
    t0 = time.perf_counter()
    for _ in range(300):
        run.loglikelihood(u.copy())
    direct = 300 / (time.perf_counter() - t0)
    print(f'direct one-point calls (no broker): {direct/1e3:.1f} k evals/s')
    for n_threads in (1, 8, 32, 128, 512):
        n_calls = max(20, 20000 // n_threads)
        broker = LikelihoodBroker(run, max_batch=4096, max_wait_us=500, n_clients=n_threads)
        U = rng.uniform(size=(n_threads, n_calls, run.ndim))

        def sampler(k):
            for j in range(n_calls):
                broker.loglikelihood(U[k, j])
        th = [threading.Thread(target=sampler, args=(k,)) for k in range(n_threads)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        st = broker.stats()
        print(f'{n_threads:4d} threads: {n_threads*n_calls/dt/1e3:8.1f} k evals/s, mean batch {st["mean_batch"]:.1f}, '
              f'largest {st["largest_batch"]}')
        broker.close()
    # native threads (no GIL): what a compiled sampler would see
    import ctypes as C
    from nestfit_amd import _ffi
    for n_threads in (1, 8, 32, 128, 512, 2048):
        n_calls = max(20, 100000 // n_threads)
        broker = LikelihoodBroker(run, max_batch=4096, max_wait_us=500, n_clients=n_threads)
        U = rng.uniform(size=(n_threads, n_calls, run.ndim))
        lnL = np.empty((n_threads, n_calls))
        sec = C.c_double()
        _ffi.test_check(_ffi.test_engine().nfa_test_broker_storm(broker.handle, _ffi.broker_loglike_address(), n_threads, n_calls, None, _ffi.dptr(U),
                                                     _ffi.dptr(lnL), C.byref(sec)))
        st = broker.stats()
        print(f'native {n_threads:4d} threads: {n_threads*n_calls/sec.value/1e3:8.1f} k evals/s, '
              f'mean batch {st["mean_batch"]:.1f}, largest {st["largest_batch"]}')
        broker.close()


if __name__ == '__main__':
    main()
