#!/usr/bin/env python3
"""BASELINE config 5 with the built-in sampler: nested sampling (400 live points) of every pixel of
a 32x32 synthetic NH3 (1,1)+(2,2) cube, all pixels in lock-step on one GPU; a few pixels are
repeated with the CPU oracle as likelihood (same sampler, same seed) for the evidence comparison
and the CPU rate.  usage: measure_sampler.py [side=32] [ncomp=1] [nlive=400] [n_cpu_pix=4] [host|gpu] [time_limit|0] [batch_target]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import nestfit_amd as na                                   # noqa: E402
from nestfit_amd import sampler                            # noqa: E402
from nestfit_amd.cube import CubeRunner                    # noqa: E402
from nestfit_amd.synth import freq_axis                    # noqa: E402


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    ncomp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    nlive = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    n_cpu = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    n_pix, n, noise = side * side, 512, 0.1
    rng = np.random.default_rng(0)
    axes = [freq_axis(1, n), freq_axis(2, n)]
    ut = na.get_irdc_priors(size=500, vsys=0.0)
    # truth per pixel: smooth velocity gradient, amplitude falling off from the centre
    lon, lat = np.indices((side, side))
    r = np.hypot(lon - side / 2, lat - side / 2) / (side / 2)
    truths = np.zeros((n_pix, 6 * ncomp))
    for c in range(ncomp):
        truths[:, c] = (-1.0 + 2.0 * lon.ravel() / side) + 1.5 * c
        truths[:, ncomp + c] = 12.0 + 3 * c
        truths[:, 2 * ncomp + c] = 5.0 + c
        truths[:, 3 * ncomp + c] = 14.6 - 0.6 * r.ravel()
        truths[:, 4 * ncomp + c] = 0.4
        truths[:, 5 * ncomp + c] = 0.0
    dummy = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=ncomp)
    model, _ = dummy_predict(dummy, truths)
    data = model + rng.normal(0, noise, model.shape)
    cube = CubeRunner(axes, (1, 2), data, np.full((n_pix, 2), noise), ut, ncomp=ncomp)
    t0 = time.perf_counter()
    last = [t0]

    def progress(n_active, it):
        if time.perf_counter() - last[0] > 30:
            last[0] = time.perf_counter()
            print(f'  ... {n_active} pixels active, iteration {it}', flush=True)
    host = len(sys.argv) > 5 and sys.argv[5] == 'host'
    if host:
        res = sampler.fit_pixels(cube, np.arange(n_pix), nlive=nlive, tol=0.5, efr=0.3, seed=1, device=False,
                                 progress=progress)
    else:
        res = sampler.fit_pixels(cube, np.arange(n_pix), nlive=nlive, tol=0.5, efr=0.3, seed=1,
                                 progress=lambda n_active, it: print(f'  ... {n_active} pixels active, '
                                                                     f'{time.perf_counter() - t0:.0f} s', flush=True),
                                 time_limit=float(sys.argv[6]) if len(sys.argv) > 6 and float(sys.argv[6]) > 0 else None,
                                 **({'batch_target': int(sys.argv[7])} if len(sys.argv) > 7 else {}))
    dt = time.perf_counter() - t0
    evals = sum(r.n_evals for r in res)
    iters = np.array([r.n_iter for r in res])
    dlnz = np.array([r.lnZ for r in res]) - cube.null_lnZ
    print(f'rounds {res[0].rounds}, sampler state ' + ('on the host (numpy twin)' if host else 'on the device'))
    print(f'GPU: {n_pix} pixels, ncomp={ncomp}, nlive={nlive}: {dt:.1f} s wall, {evals/1e6:.1f} M likelihood '
          f'evaluations ({evals/dt/1e6:.2f} M evals/s end to end), {n_pix/dt:.1f} pixels/s, '
          f'iterations per pixel {iters.min()}..{iters.max()}, detections (dlnZ > 11): {(dlnz > 11).sum()}')
    if n_cpu < 1:
        return
    # CPU oracle on a few pixels, same sampler and seed
    from oracle import nfo
    ps = nfo.PriorSet(ut.lower())
    pick = np.linspace(0, n_pix - 1, n_cpu).astype(int)
    runners = []
    for p in pick:
        specs = [nfo.AmmoniaSpectrum(axes[k], data[p, k * n:(k + 1) * n], noise, t, native=True)
                 for k, t in enumerate((1, 2))]
        runners.append(nfo.AmmoniaRunner(specs, ps, ncomp=ncomp, native=True))

    def cpu_loglike(pix, U):
        out = np.empty(U.shape[0])
        for q in np.unique(pix):
            m = pix == q
            sub = U[m]
            out[m] = runners[q].loglikelihood_batch(sub)
            U[m] = sub
        return out
    t0 = time.perf_counter()
    ref = sampler.run_nested(cpu_loglike, cube.ndim, n_cpu, nlive=nlive, tol=0.5, efr=0.3, seed=2)
    dtc = time.perf_counter() - t0
    gpu_same = sampler.fit_pixels(cube, pick, nlive=nlive, tol=0.5, efr=0.3, seed=2)
    ev_c = sum(r.n_evals for r in ref)
    print(f'CPU oracle (1 core, reference flags), {n_cpu} pixels: {dtc:.1f} s, {n_cpu/dtc:.3f} pixels/s, '
          f'{ev_c/dtc/1e3:.1f} k evals/s')
    for p, a, b in zip(pick, gpu_same, ref):
        print(f'  pixel {p}: lnZ gpu {a.lnZ:.3f} cpu {b.lnZ:.3f} +- {b.lnZ_err:.3f}; '
              f'v mean gpu {a.param_constr[0, 0]:.4f} cpu {b.param_constr[0, 0]:.4f} truth {truths[p, 0]:.4f}')


def dummy_predict(dummy, truths):
    """Noise-free model spectra of all pixels through the engine's batched predict."""
    import ctypes as C
    from nestfit_amd import _ffi
    B = truths.shape[0]
    spec = np.empty((B, dummy._ss.chan_tot))
    lnl = np.empty(B)
    th = np.ascontiguousarray(truths)
    _ffi.check(_ffi.load().nfa_runner_predict_batch(dummy._run.handle, None, _ffi.dptr(th), B, _ffi.dptr(spec),
                                                    _ffi.dptr(lnl)))
    return spec, lnl


if __name__ == '__main__':
    main()
