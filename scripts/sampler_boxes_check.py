"""GPU: the box vetoes of the device sampler against the numpy twin (same seed: same run), two components.
    python scripts/sampler_boxes_check.py [n_pix] [nlive] [maxiter]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import nestfit_amd as na
from nestfit_amd import sampler
from nestfit_amd.cube import CubeRunner
from nestfit_amd.synth import freq_axis

n_pix = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nlive = int(sys.argv[2]) if len(sys.argv) > 2 else 400
maxiter = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
n, noise = 256, 0.1
rng = np.random.default_rng(3)
axes = [freq_axis(1, n), freq_axis(2, n)]
ut = na.get_irdc_priors(size=500, vsys=0.0)
truths = np.tile(np.array([-0.5, 1.0, 12.0, 15.0, 5.0, 6.0, 14.4, 14.6, 0.4, 0.4, 0.0, 0.0]), (n_pix, 1))
truths[:, 6] += rng.uniform(-0.3, 0.3, n_pix)
na.set_exp_mode('table')
probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=2)
model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
cube = CubeRunner(axes, (1, 2), model + rng.normal(0, noise, model.shape), np.full((n_pix, 2), noise), ut, ncomp=2)
kw = dict(nlive=nlive, tol=0.5, efr=0.3, seed=7, maxiter=maxiter, batch_target=4096)
for label, extra in (('boxes (default)', {}), ('no boxes', dict(frames=-1))):
    t0 = time.time(); dev = sampler.fit_pixels(cube, np.arange(n_pix), device=True, **kw, **extra); t1 = time.time()
    twin = sampler.fit_pixels(cube, np.arange(n_pix), device=False, **kw, **extra); t2 = time.time()
    print(label, 'device %.1f s, twin %.1f s' % (t1 - t0, t2 - t1))
    for p in range(n_pix):
        d, t = dev[p], twin[p]
        print(f'  pixel {p}: device iters {d.n_iter} evals {d.n_evals} lnZ {d.lnZ:.4f} rounds {d.rounds} | twin iters {t.n_iter} evals {t.n_evals} lnZ {t.lnZ:.4f} rounds {t.rounds}'
              f' | {"SAME" if (d.n_iter, d.n_evals) == (t.n_iter, t.n_evals) else "DIFFERENT"}')
