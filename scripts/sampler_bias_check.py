#!/usr/bin/env python3
"""Does the ellipsoid rule bias the evidence?  The same 32x32 synthetic cube fitted with the default
target efficiency (efr = 0.3: bounding ellipsoid, volume >= X / 0.3) and with a very cautious one
(efr = 0.02: volume >= 50 X); the mean lnZ difference over 1024 pixels has a standard error of
~0.007, far below the per-pixel error of ~0.16."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import nestfit_amd as na                                   # noqa: E402
from nestfit_amd import sampler                            # noqa: E402
from nestfit_amd.cube import CubeRunner                    # noqa: E402
from nestfit_amd.synth import freq_axis                    # noqa: E402
from measure_sampler import dummy_predict                  # noqa: E402

side, n, noise = 32, 512, 0.1
n_pix = side * side
rng = np.random.default_rng(0)
axes = [freq_axis(1, n), freq_axis(2, n)]
ut = na.get_irdc_priors(size=500, vsys=0.0)
lon, lat = np.indices((side, side))
r = np.hypot(lon - side / 2, lat - side / 2) / (side / 2)
truths = np.zeros((n_pix, 6))
truths[:, 0] = -1.0 + 2.0 * lon.ravel() / side
truths[:, 1], truths[:, 2], truths[:, 3], truths[:, 4] = 12.0, 5.0, 14.6 - 0.6 * r.ravel(), 0.4
dummy = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=1)
model, _ = dummy_predict(dummy, truths)
cube = CubeRunner(axes, (1, 2), model + rng.normal(0, noise, model.shape), np.full((n_pix, 2), noise), ut, ncomp=1)
out = {}
for efr, enl in ((0.3, 1.0), (0.3, 1.25), (0.3, 2.0), (0.02, 1.25)):
    t0 = time.perf_counter()
    res = sampler.fit_pixels(cube, np.arange(n_pix), nlive=400, tol=0.5, efr=efr, seed=11, enlarge=enl)
    out[(efr, enl)] = np.array([x.lnZ for x in res])
    err = np.mean([x.lnZ_err for x in res])
    print(f'efr {efr} enlarge {enl}: {time.perf_counter()-t0:.1f} s, {sum(x.n_evals for x in res)/1e6:.0f} M evals, mean lnZ_err {err:.3f}')
for key in ((0.3, 1.0), (0.3, 1.25), (0.3, 2.0)):
    d = out[key] - out[(0.02, 1.25)]
    print(f'mean lnZ{key} - lnZ(efr 0.02) = {d.mean():+.4f} +- {d.std(ddof=1)/np.sqrt(n_pix):.4f}  (scatter {d.std(ddof=1):.3f})')
