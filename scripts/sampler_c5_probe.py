"""GPU: the two-component run of BASELINE config 5 (bench.py --workload C5) under sampler settings given on the
command line: python scripts/sampler_c5_probe.py <side> key=value ... (frames, margin, method, n_steps, batch_target)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import nestfit_amd as na
from nestfit_amd import _ffi, sampler
from nestfit_amd.cube import CubeRunner
from nestfit_amd.synth import freq_axis

side = int(sys.argv[1])
kw = {}
for a in sys.argv[2:]:
    k, v = a.split('=')
    kw[k] = v if k == 'method' else float(v) if k in ('margin', 'pairs', 'shear') else int(v)
n, noise, nlive, ncomp = 512, 0.1, 400, 2
n_pix = side * side
rng = np.random.default_rng(0)
axes = [freq_axis(1, n), freq_axis(2, n)]
ut = na.get_irdc_priors(size=500, vsys=0.0)
lon, lat = np.indices((side, side))
r = np.hypot(lon - side / 2, lat - side / 2) / (side / 2)
truths = np.zeros((n_pix, 12))
for c in range(ncomp):
    truths[:, c] = (-1.0 + 2.0 * lon.ravel() / side) + 1.5 * c
    truths[:, ncomp + c], truths[:, 2 * ncomp + c] = 12.0 + 3 * c, 5.0 + c
    truths[:, 3 * ncomp + c], truths[:, 4 * ncomp + c] = 14.6 - 0.6 * r.ravel(), 0.4
probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=ncomp)
model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
cube = CubeRunner(axes, (1, 2), model + rng.normal(0, noise, model.shape), np.full((n_pix, 2), noise), ut, ncomp=ncomp)
for key in ('walk_factor', 'refit_every', 'parts', 'kmax', 'ktarget', 'ratio_max'):
    if key in kw:
        _ffi.set_option('sampler_' + key, kw.pop(key))
_ffi.check(_ffi.load().nfa_device_synchronize())
t0 = time.perf_counter()
res = sampler.fit_pixels(cube, np.arange(n_pix), nlive=nlive, tol=0.5, efr=0.3, **{"seed": 1, **kw})
dt = time.perf_counter() - t0
ev = np.array([x.n_evals for x in res]); it = np.array([x.n_iter for x in res]); lz = np.array([x.lnZ for x in res])
print(f'{side}x{side} {kw}: {dt:.2f} s, evals/pixel {ev.mean():.0f} (min {ev.min()} max {ev.max()}), iterations {it.mean():.0f}, '
      f'evals/iteration {ev.sum() / it.sum():.1f}, rounds {res[0].rounds}, mean lnZ {lz.mean():.3f}')
print('   timings', {k: round(v, 3) for k, v in getattr(res[0], 'timings', {}).items()})
import os; os.makedirs('gpurun_out/r04', exist_ok=True)
np.save(f'gpurun_out/r04/c5probe_lnZ_{side}_' + '_'.join(f'{k}{v}' for k, v in kw.items()) + '.npy', lz)
