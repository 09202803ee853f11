"""GPU: the two-component run of rounds 2-4's config-5 cube (synth.c5r4_cube) under the three named settings of the sampler."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import nestfit_amd as na
from nestfit_amd import sampler
from nestfit_amd.cube import CubeRunner
from nestfit_amd.synth import c5r4_cube
na.set_exp_mode('fast')
axes, data, noise, ut = c5r4_cube(2)
cube = CubeRunner(axes, (1, 2), data, np.full((1024, 2), noise), ut, ncomp=2)
sampler.fit_pixels(cube, np.arange(64), nlive=400, tol=0.5, efr=0.3, seed=1, maxiter=200)
for prec in ('speed', 'default', 'evidence'):
    t0 = time.perf_counter()
    res = sampler.fit_pixels(cube, np.arange(1024), nlive=400, tol=0.5, efr=0.3, seed=1, precision=prec)
    dt = time.perf_counter() - t0
    print(f'precision {prec:9s}: {dt:6.2f} s, {np.mean([r.n_evals for r in res]) / 1e3:7.1f} k evaluations per pixel, mean lnZ {np.mean([r.lnZ for r in res]):.4f}', flush=True)
