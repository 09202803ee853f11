"""GPU: one pixel of config 5 as specified through the numpy twin (GPU likelihood), with the round's state printed."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import nestfit_amd as na
from nestfit_amd import sampler
from nestfit_amd.cube import CubeRunner
import test_configs_at_size as T
pix = int(sys.argv[1])
kw = {}
for a in sys.argv[2:]:
    k, v = a.split('=')
    kw[k] = v if k == 'method' else float(v) if k in ('margin', 'shear') else int(v)
side, n, noise = 32, 1024, 0.2
stack, truths, model, data, axes, ut = T._c5_stack(na, side, n, noise)
na.set_exp_mode('table')
cube = CubeRunner(axes, (1, 2), data, np.full((side * side, 2), noise), ut, ncomp=2)
t0 = time.perf_counter()
class P:
    def __call__(self, n_active, it):
        pass
    def detail(self, d):
        if d['rnd'] % 200 == 0 or (d['n_iter'][0] > 21000 and d['rnd'] % 20 == 0):
            L = d['Llive'][0]
            print(f"round {d['rnd']:6d} iter {d['n_iter'][0]:6d} evals {d['n_evals'][0]:8d} walk {int(d['walk'][0])} cube {int(d['use_cube'][0])} lnvol {d['lnvol'][0]:8.2f} Kr {d['Kr']} "
                  f"rj scan/acc/raw/val {[int(x[0]) for x in d['rj']]} Lmin {L.min():.6f} Lmax-Lmin {L.max() - L.min():.3e} spread u {d['Ulive'][0].std(axis=0).min():.2e} [{time.perf_counter() - t0:.0f} s]", flush=True)
res = sampler.fit_pixels(cube, np.array([pix]), nlive=400, tol=0.5, efr=0.3, seed=5, device=False, progress=P(), batch_target=4096, **kw)
r = res[0]
print(f'pixel {pix}: lnZ {r.lnZ:.3f} iterations {r.n_iter} evals {r.n_evals} rounds {r.rounds}')
