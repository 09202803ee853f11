#!/usr/bin/env python3
"""Evidence of config 5's 1024 one-component pixels with one bounding ellipsoid per pixel and with up to four
(engine option sampler_ellipsoids), against a cautious one-ellipsoid run (efr 0.05): mean difference and its error."""
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

side, n, noise, nlive = 32, 512, 0.1, 400
n_pix = side * side
rng = np.random.default_rng(0)
axes = [freq_axis(1, n), freq_axis(2, n)]
ut = na.get_irdc_priors(size=500, vsys=0.0)
lon, lat = np.indices((side, side))
r = np.hypot(lon - side / 2, lat - side / 2) / (side / 2)
truths = np.zeros((n_pix, 6))
truths[:, 0] = (-1.0 + 2.0 * lon.ravel() / side); truths[:, 1] = 12.0; truths[:, 2] = 5.0
truths[:, 3] = 14.6 - 0.6 * r.ravel(); truths[:, 4] = 0.4
probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=1)
model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
cube = CubeRunner(axes, (1, 2), model + rng.normal(0, noise, model.shape), np.full((n_pix, 2), noise), ut, ncomp=1)


def run(ell, seed, **kw):
    _ffi.set_option('sampler_ellipsoids', ell)
    t0 = time.perf_counter()
    res = sampler.fit_pixels(cube, np.arange(n_pix), nlive=nlive, tol=0.5, seed=seed, method='reject', **kw)
    return time.perf_counter() - t0, np.array([x.lnZ for x in res]), np.array([x.n_evals for x in res]), np.mean([x.lnZ_err for x in res])


dt0, ref, ev0, err = run(1, 11, efr=0.05)
print(f'cautious reference (one ellipsoid, efr 0.05, seed 11): {dt0:.2f} s, {ev0.mean() / 1e3:.0f} k evaluations per pixel, per-pixel lnZ error {err:.3f}')
faint = r.ravel() >= 0.7
for ell, name in ((1, 'one ellipsoid'), (0, 'up to four ellipsoids')):
    for seed in (1, 2):
        dt, lnz, ev, _ = run(ell, seed, efr=0.3)
        d = lnz - ref
        print(f'{name}, efr 0.3, seed {seed}: {dt:.2f} s, {ev.mean() / 1e3:.0f} k evaluations per pixel; lnZ - reference: '
              f'{d.mean():+.4f} +- {d.std() / np.sqrt(n_pix):.4f} (faint half: {d[faint].mean():+.4f} +- {d[faint].std() / np.sqrt(faint.sum()):.4f})')
