#!/usr/bin/env python3
"""Evidence of config 5's 1024 two-component pixels with 64 and with 128 walkers per pixel (engine option
sampler_walkers) against a run with 64 walkers and 140 steps per walk: mean difference and its error."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import nestfit_amd as na                                   # noqa: E402
from nestfit_amd import _ffi, sampler                      # noqa: E402
from nestfit_amd.cube import CubeRunner                    # noqa: E402
from nestfit_amd.synth import freq_axis                    # noqa: E402

side, n, noise, nlive, ncomp = 32, 512, 0.1, 400, 2
n_pix = side * side
rng = np.random.default_rng(0)
rng.normal(0, noise, (n_pix, 2 * n))                       # (the bench draws the one-component cube's noise first)
axes = [freq_axis(1, n), freq_axis(2, n)]
ut = na.get_irdc_priors(size=500, vsys=0.0)
lon, lat = np.indices((side, side))
r = np.hypot(lon - side / 2, lat - side / 2) / (side / 2)
truths = np.zeros((n_pix, 6 * ncomp))
for c in range(ncomp):
    truths[:, c] = (-1.0 + 2.0 * lon.ravel() / side) + 1.5 * c
    truths[:, ncomp + c], truths[:, 2 * ncomp + c] = 12.0 + 3 * c, 5.0 + c
    truths[:, 3 * ncomp + c], truths[:, 4 * ncomp + c] = 14.6 - 0.6 * r.ravel(), 0.4
probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=ncomp)
model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
cube = CubeRunner(axes, (1, 2), model + rng.normal(0, noise, model.shape), np.full((n_pix, 2), noise), ut, ncomp=ncomp)


def run(walkers, seed, **kw):
    _ffi.set_option('sampler_walkers', walkers)
    t0 = time.perf_counter()
    res = sampler.fit_pixels(cube, np.arange(n_pix), nlive=nlive, tol=0.5, efr=0.3, seed=seed, **kw)
    return time.perf_counter() - t0, np.array([x.lnZ for x in res]), np.array([x.n_evals for x in res]), np.mean([x.lnZ_err for x in res])


refs = []
for seed in (11, 12, 13, 14):
    dt, lnz, ev, err = run(64, seed, n_steps=140)
    refs.append(lnz)
    print(f'reference (64 walkers, 140 steps, seed {seed}): {dt:.2f} s, {ev.mean() / 1e3:.0f} k evaluations per pixel, per-pixel lnZ error {err:.3f}', flush=True)
ref = np.mean(refs, axis=0)
steps = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [100]
for walkers in ((64, 128) if len(sys.argv) <= 1 else (128,)):
    for n_steps in steps:
        ds = []
        for seed in (1, 2, 3):
            dt, lnz, ev, _ = run(walkers, seed, n_steps=n_steps)
            d = lnz - ref
            ds.append(d.mean())
            print(f'{walkers} walkers, {n_steps} steps, seed {seed}: {dt:.2f} s, {ev.mean() / 1e3:.0f} k evaluations per pixel; lnZ - reference: '
                  f'{d.mean():+.4f} +- {d.std() / np.sqrt(n_pix):.4f}', flush=True)
        print(f'   mean over the seeds: {np.mean(ds):+.4f} +- {0.0099 / np.sqrt(len(ds)):.4f}', flush=True)
