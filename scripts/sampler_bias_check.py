#!/usr/bin/env python3
"""Do the sampler's shortcuts bias the evidence?  The same 32x32 synthetic cube fitted with the
default settings (automatic switch to constrained walks, ellipsoid volume >= max(1.5 V_bound, X / 0.3)),
with rejection only, with walks only, and with a very cautious rejection-only reference
(efr = 0.02: volume >= 50 X); the mean lnZ difference over 1024 pixels has a standard error of
~0.007, far below the per-pixel error of ~0.18."""
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

ncomp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
side = int(sys.argv[3]) if len(sys.argv) > 3 else 32
n, noise = 512, 0.1
n_pix = side * side
rng = np.random.default_rng(0)
axes = [freq_axis(1, n), freq_axis(2, n)]
ut = na.get_irdc_priors(size=500, vsys=0.0)
lon, lat = np.indices((side, side))
r = np.hypot(lon - side / 2, lat - side / 2) / (side / 2)
truths = np.zeros((n_pix, 6 * ncomp))
for c in range(ncomp):
    truths[:, c] = (-1.0 + 2.0 * lon.ravel() / side) + 1.5 * c
    truths[:, ncomp + c], truths[:, 2 * ncomp + c] = 12.0 + 3 * c, 5.0 + c
    truths[:, 3 * ncomp + c], truths[:, 4 * ncomp + c] = 14.6 - 0.6 * r.ravel(), 0.4
dummy = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=ncomp)
model, _ = dummy_predict(dummy, truths)
cube = CubeRunner(axes, (1, 2), model + rng.normal(0, noise, model.shape), np.full((n_pix, 2), noise), ut, ncomp=ncomp)
out = {}
runs = [('auto', 0.3, 1.5, 25), ('auto', 0.3, 1.5, 50), ('walk', 0.3, 1.5, 25), ('walk', 0.3, 1.5, 50), ('walk', 0.3, 1.5, 100),
        ('reject', 0.3, 1.5, 0), ('reject', 0.02, 1.5, 0)]
if len(sys.argv) > 1 and sys.argv[1] == 'short':
    nd = 5 * ncomp                                   # sampled dimensions with get_irdc_priors
    runs = [('auto', 0.3, 1.5, 10 * nd), ('walk', 0.3, 1.5, 10 * nd), ('walk', 0.3, 1.5, 4 * nd), ('reject', 0.3, 1.5, 0)]
if len(sys.argv) > 4:                                # e.g. "sampler_bias_check.py steps 2 24 100,120,150": walk lengths against rejection only
    lengths = [int(v) for v in sys.argv[4].split(',')]
    runs = [('walk', 0.3, 1.5, k) for k in lengths] + [('auto', 0.3, 1.5, k) for k in lengths] + [('reject', 0.3, 1.5, 0)]
for method, efr, enl, steps in runs:
    t0 = time.perf_counter()
    res = sampler.fit_pixels(cube, np.arange(n_pix), nlive=400, tol=0.5, efr=efr, seed=11, enlarge=enl, method=method,
                             n_steps=max(steps, 1))
    out[(method, efr, enl, steps)] = np.array([x.lnZ for x in res])
    err = np.mean([x.lnZ_err for x in res])
    print(f'{method} efr {efr} enlarge {enl} n_steps {steps}: {time.perf_counter()-t0:.1f} s, {sum(x.n_evals for x in res)/1e6:.0f} M evals, '
          f'mean lnZ_err {err:.3f}', flush=True)
ref = out[runs[-1]]
for key in runs[:-1]:
    d = out[key] - ref
    print(f'mean lnZ{key} - lnZ{runs[-1]} = {d.mean():+.4f} +- {d.std(ddof=1)/np.sqrt(n_pix):.4f}  (scatter {d.std(ddof=1):.3f})')
