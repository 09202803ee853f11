"""GPU: reference evidences of the two pixels of scripts/proto_intersection.py (two components; ntot 14.4 "bright" and 14.0
"faint", the same noise realisation) from the shipped device sampler with long walks: 128 independent runs per pixel.
    python scripts/sampler_reference_lnz.py [n_steps] [copies] [exp mode]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import nestfit_amd as na
from nestfit_amd import sampler
from nestfit_amd.cube import CubeRunner
from nestfit_amd.synth import freq_axis

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 140
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 128
na.set_exp_mode(sys.argv[3] if len(sys.argv) > 3 else 'table')
n, noise, ncomp = 512, 0.1, 2
axes = [freq_axis(1, n), freq_axis(2, n)]
ut = na.get_irdc_priors(size=500, vsys=0.0)
rng = np.random.default_rng(0)
eps = np.concatenate([rng.normal(0, noise, n), rng.normal(0, noise, n)])
truths = np.array([[-0.5, 1.0, 12.0, 15.0, 5.0, 6.0, nt, nt + 0.2, 0.4, 0.4, 0.0, 0.0] for nt in (14.4, 14.0)])
probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=ncomp)
model, _ = probe.predict_batch(np.zeros(2, dtype=np.int32), truths)
data = np.repeat(model + eps, copies, axis=0)
cube = CubeRunner(axes, (1, 2), data, np.full((2 * copies, 2), noise), ut, ncomp=ncomp)
t0 = time.perf_counter()
res = sampler.fit_pixels(cube, np.arange(2 * copies), nlive=400, tol=0.5, efr=0.3, seed=7, method='walk', n_steps=n_steps)
dt = time.perf_counter() - t0
for g, name in enumerate(('ntot 14.4', 'ntot 14.0')):
    r = res[g * copies:(g + 1) * copies]
    lz = np.array([x.lnZ for x in r]); ev = np.array([x.n_evals for x in r]); it = np.array([x.n_iter for x in r])
    print(f'{name}: walks of {n_steps} steps, {copies} runs: lnZ {lz.mean():.3f} +- {lz.std(ddof=1) / np.sqrt(copies):.3f} (scatter {lz.std(ddof=1):.3f}), '
          f'iterations {it.mean():.0f}, evals/iteration {ev.sum() / it.sum():.1f}', flush=True)
print(f'{dt:.1f} s')
