"""GPU: the reference evidences of tests/test_sampler_bias.py -> tests/golden/sampler_bias_reference.json.

The first 256 pixels (16 x 16... the first eight longitudes) of the two-component cube of nestfit_amd.synth.c5r4_cube,
400 live points, tol 0.5, efr 0.3: rejection sampling from the sheared ellipsoid ALONE -- no boxes, no pair ellipses, no
walks -- with a safety factor of 8 on its volume, where nothing that holds prior mass can be cut off (factors
of 8 and 16 agree; 5-8 M evaluations per pixel), `seeds` independent runs.  Stored: per pixel the mean lnZ and its
standard error, the seeds, the evaluations per pixel.
    python scripts/sampler_bias_reference.py [seeds] [exp mode]"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import nestfit_amd as na
from nestfit_amd import sampler
from nestfit_amd.cube import CubeRunner
from nestfit_amd.synth import c5r4_cube

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else 'fast'
na.set_exp_mode(mode)
N_SUB = 256
axes, data, noise, ut = c5r4_cube(2)
cube = CubeRunner(axes, (1, 2), np.ascontiguousarray(data[:N_SUB]), np.full((N_SUB, 2), noise), ut, ncomp=2)
lnz, evals = [], []
for k in range(n_seeds):
    t0 = time.perf_counter()
    res = sampler.fit_pixels(cube, np.arange(N_SUB), nlive=400, tol=0.5, efr=0.3, seed=101 + k, method='reject', shear=8.0, frames=-1, pairs=0)
    lnz.append([r.lnZ for r in res]); evals.append(np.mean([r.n_evals for r in res]))
    print(f'seed {101 + k}: mean lnZ {np.mean(lnz[-1]):.4f}, {evals[-1] / 1e6:.2f} M evaluations per pixel, {time.perf_counter() - t0:.1f} s', flush=True)
lnz = np.array(lnz)
out = {'what': 'bound-free rejection (sheared ellipsoid x 8, no boxes, no pair ellipses, no walks), nlive 400, tol 0.5, efr 0.3',
       'cube': 'nestfit_amd.synth.c5r4_cube(2), pixels 0..255', 'exp_mode': mode, 'seeds': [101 + k for k in range(n_seeds)],
       'evals_per_pixel': float(np.mean(evals)), 'mean_lnZ': float(lnz.mean()), 'mean_lnZ_se': float(lnz.mean(axis=1).std(ddof=1) / np.sqrt(n_seeds)),
       'lnZ': lnz.mean(axis=0).tolist(), 'lnZ_se': (lnz.std(axis=0, ddof=1) / np.sqrt(n_seeds)).tolist()}
dst = ROOT / 'gpurun_out' / 'r05' / 'sampler_bias_reference.json'
dst.parent.mkdir(parents=True, exist_ok=True)
dst.write_text(json.dumps(out))
print(f'cube mean lnZ {out["mean_lnZ"]:.4f} +- {out["mean_lnZ_se"]:.4f} ({n_seeds} seeds) -> {dst}')
# the three named settings against it, four seeds each
for prec in ('speed', 'default', 'evidence'):
    got, ev, t0 = [], [], time.perf_counter()
    for k in range(4):
        res = sampler.fit_pixels(cube, np.arange(N_SUB), nlive=400, tol=0.5, efr=0.3, seed=11 + k, precision=prec)
        got.append([r.lnZ for r in res]); ev.append(np.mean([r.n_evals for r in res]))
    got = np.array(got)
    d = got.mean(axis=0) - lnz.mean(axis=0)
    print(f'{prec:9s}: bias {d.mean():+.4f} +- {d.std(ddof=1) / np.sqrt(d.size):.4f}, {np.mean(ev) / 1e3:.0f} k evaluations per pixel, {(time.perf_counter() - t0) / 4:.2f} s per run', flush=True)
