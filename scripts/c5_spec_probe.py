"""GPU: the two-component stage of tests/test_configs_at_size.py::test_config5_as_specified (32 x 32 pixels of config 3's
generator, 1024 channels, table mode) with progress lines: python scripts/c5_spec_probe.py [key=value ...]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import nestfit_amd as na
from nestfit_amd import sampler
from nestfit_amd.cube import CubeRunner
import test_configs_at_size as T

kw = {}
for a in sys.argv[1:]:
    k, v = a.split('=')
    kw[k] = v if k == 'method' else float(v) if k in ('margin', 'shear') else int(v)
ncomp = kw.pop('ncomp', 2)
mode = 'table'
side, n, noise = 32, 1024, 0.2
stack, truths, model, data, axes, ut = T._c5_stack(na, side, n, noise)
na.set_exp_mode(mode)
cube = CubeRunner(axes, (1, 2), data, np.full((side * side, 2), noise), ut, ncomp=ncomp)
t0 = time.perf_counter()
last = [0.0]
def progress(n_active, rounds):
    t = time.perf_counter() - t0
    if t - last[0] > 5:
        last[0] = t
        print(f'  {t:6.1f} s: {n_active} pixels active, round {rounds}', flush=True)
res = sampler.fit_pixels(cube, np.arange(side * side), nlive=400, tol=0.5, efr=0.3, seed=5, progress=progress, time_limit=150, **kw)
dt = time.perf_counter() - t0
ev = np.array([x.n_evals for x in res]); it = np.array([x.n_iter for x in res])
print(f'{kw}: {dt:.1f} s, evals/pixel {ev.mean():.0f} (max {ev.max()}), iterations {it.mean():.0f} (max {it.max()}), rounds {res[0].rounds}')
worst = np.argsort(-ev)[:5]
print('   slowest pixels', worst, ev[worst], it[worst], truths[worst][:, [6, 7]])
