"""GPU: a few pixels of config 5 as specified on the device and through the twin, counters compared every 8 rounds."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import nestfit_amd as na
from nestfit_amd import sampler
from nestfit_amd.cube import CubeRunner
import test_configs_at_size as T
pixels = np.array([int(x) for x in sys.argv[1].split(',')])
side, n, noise = 32, 1024, 0.2
stack, truths, model, data, axes, ut = T._c5_stack(na, side, n, noise)
na.set_exp_mode('table')
cube = CubeRunner(axes, (1, 2), data, np.full((side * side, 2), noise), ut, ncomp=2)
kw = dict(nlive=400, tol=0.5, efr=0.3, seed=5, batch_target=int(sys.argv[2]) if len(sys.argv) > 2 else 8192)
dev = {}
class PD:
    def __init__(self): self.t0 = time.perf_counter(); self.last = None; self.same = 0
    def __call__(self, n_active, it): pass
    def counts(self, ni, ne, rounds):
        dev[rounds] = (ni.copy(), ne.copy())
        if self.last is not None and np.array_equal(self.last, ni): self.same += 1
        else: self.same = 0
        self.last = ni.copy()
        if rounds % 800 == 0 or self.same in (50, 51): print(f'device round {rounds}: iters {ni} evals {ne} [{time.perf_counter() - self.t0:.0f} s]' + (' STUCK' if self.same >= 50 else ''), flush=True)
        return 1
res = sampler.fit_pixels(cube, pixels, device=True, progress=PD(), time_limit=float(sys.argv[3]) if len(sys.argv) > 3 else 60, **kw)
print('device:', [(r.n_iter, r.n_evals) for r in res], 'rounds', res[0].rounds, flush=True)
class PT:
    def __init__(self): self.bad = False; self.badr = 10**9; self.t0 = time.perf_counter()
    def __call__(self, n_active, it): pass
    def detail(self, d):
        r = d['rnd']
        if r in dev and not self.bad:
            ni, ne = dev[r]
            if not (np.array_equal(ni, d['n_iter']) and np.array_equal(ne, d['n_evals'])):
                self.bad = True
                print(f'FIRST DIFFERENCE at round {r}: device iters {ni} evals {ne}; twin iters {d["n_iter"]} evals {d["n_evals"]} walk {d["walk"].astype(int)} cube {d["use_cube"].astype(int)} '
                      f'rj {[x.tolist() for x in d["rj"]]} Kr {d["Kr"]}', flush=True)
        if r % 800 == 0: print(f'twin round {r}: iters {d["n_iter"]} walk {d["walk"].astype(int)} Kr {d["Kr"]} [{time.perf_counter() - self.t0:.0f} s]', flush=True)
        if self.bad and r % 8 == 0 and r in dev and r < self.badr + 80:
            print(f'   round {r}: device iters {dev[r][0]} twin {d["n_iter"]} walk {d["walk"].astype(int)} rj {[x.tolist() for x in d["rj"]]}', flush=True)
        if self.bad and self.badr == 10**9: self.badr = r
        if self.bad and r > self.badr + 100: raise SystemExit(0)
twin = sampler.fit_pixels(cube, pixels, device=False, progress=PT(), **kw)
print('twin:', [(r.n_iter, r.n_evals) for r in twin])
