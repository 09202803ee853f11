import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import nestfit_amd as na
from nestfit_amd import sampler
from nestfit_amd.cube import CubeRunner
from nestfit_amd.synth import freq_axis
side, n, noise, nlive = 32, 512, 0.1, 400
n_pix = side * side
rng = np.random.default_rng(0)
axes = [freq_axis(1, n), freq_axis(2, n)]
ut = na.get_irdc_priors(size=500, vsys=0.0)
lon, lat = np.indices((side, side))
r = np.hypot(lon - side / 2, lat - side / 2) / (side / 2)
ncomp = 1
truths = np.zeros((n_pix, 6))
truths[:, 0] = (-1.0 + 2.0 * lon.ravel() / side); truths[:, 1] = 12.0; truths[:, 2] = 5.0
truths[:, 3] = 14.6 - 0.6 * r.ravel(); truths[:, 4] = 0.4
probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=1)
model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
cube = CubeRunner(axes, (1, 2), model + rng.normal(0, noise, model.shape), np.full((n_pix, 2), noise), ut, ncomp=1)
def run(pix, **kw):
    t0 = time.perf_counter()
    res = sampler.fit_pixels(cube, np.asarray(pix), nlive=nlive, tol=0.5, efr=0.3, seed=1, **kw)
    return time.perf_counter() - t0, np.array([x.n_evals for x in res]), np.array([x.n_iter for x in res]), res[0].rounds
sel = np.arange(0, n_pix, 16)          # 64 pixels spread over the cube
dt, ev, it, rd = run(np.arange(n_pix))
print(f'all 1024 pixels, batch_target 262144: {dt:.2f} s, evals/pixel {ev.mean()/1e3:.0f} k (the 64 probe pixels: {ev[sel].mean()/1e3:.0f} k), iterations {it.mean()/1e3:.1f} k, rounds {rd}')
print('   evals/iteration by radius quartile:', [round(float((ev/it)[(r.ravel() >= a) & (r.ravel() < b)].mean()), 1) for a, b in ((0, .35), (.35, .7), (.7, 1.0), (1.0, 2.0))])
for bt in (64 * 256, 64 * 64, 64 * 1024):
    for method in ('auto', 'reject'):
        dt, ev, it, rd = run(sel, batch_target=bt, method=method)
        print(f'64 probe pixels alone, {bt // 64} candidates per pixel and round, method {method}: {dt:.2f} s, evals/pixel {ev.mean()/1e3:.0f} k, iterations {it.mean()/1e3:.1f} k, rounds {rd}')
dt, ev, it, rd = run(np.arange(n_pix), method='reject')
print(f'all 1024 pixels, rejection only: {dt:.2f} s, evals/pixel {ev.mean()/1e3:.0f} k, rounds {rd}')
