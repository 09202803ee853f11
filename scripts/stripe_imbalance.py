#!/usr/bin/env python3
"""How uneven are static longitude stripes (i_lon % N, nestfit/main.py:565-571) when the work per pixel is a
nested-sampling run?  One GPU fits the BASELINE config-5 cube (32x32, 400 live points; brightness falling
off with radius, a velocity gradient along longitude) with one and with two components; the likelihood
evaluations each pixel needed are summed per stripe for N = 2, 4, 8: the slowest stripe over the mean is
what an N-GPU fit of this cube would wait for (SURVEY 8e asks for dynamic queues above 10 %)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import nestfit_amd as na                                   # noqa: E402
from nestfit_amd import sampler                            # noqa: E402
from nestfit_amd.cube import CubeRunner, get_multiproc_indices   # noqa: E402
from nestfit_amd.synth import freq_axis                    # noqa: E402

side, n, noise, nlive = 32, 512, 0.1, 400
n_pix = side * side
rng = np.random.default_rng(0)
axes = [freq_axis(1, n), freq_axis(2, n)]
ut = na.get_irdc_priors(size=500, vsys=0.0)
lon, lat = np.indices((side, side))
r = np.hypot(lon - side / 2, lat - side / 2) / (side / 2)
for ncomp in (1, 2):
    truths = np.zeros((n_pix, 6 * ncomp))
    for c in range(ncomp):
        truths[:, c] = (-1.0 + 2.0 * lon.ravel() / side) + 1.5 * c
        truths[:, ncomp + c], truths[:, 2 * ncomp + c] = 12.0 + 3 * c, 5.0 + c
        truths[:, 3 * ncomp + c], truths[:, 4 * ncomp + c] = 14.6 - 0.6 * r.ravel(), 0.4
    probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=ncomp)
    model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
    cube = CubeRunner(axes, (1, 2), model + rng.normal(0, noise, model.shape), np.full((n_pix, 2), noise), ut, ncomp=ncomp)
    res = sampler.fit_pixels(cube, np.arange(n_pix), nlive=nlive, tol=0.5, efr=0.3, seed=1)
    evals = np.array([x.n_evals for x in res], dtype=np.float64).reshape(side, side)      # [i_lon, i_lat]
    print(f'{ncomp} component(s): evaluations per pixel min {evals.min():.0f} / mean {evals.mean():.0f} / max {evals.max():.0f}')
    for world in (2, 4, 8):
        per = np.array([evals[ix[0], ix[1]].sum() for ix in get_multiproc_indices((side, side), world)])
        print(f'   N = {world}: stripe work max / mean = {per.max() / per.mean():.3f}, min / mean = {per.min() / per.mean():.3f}')
