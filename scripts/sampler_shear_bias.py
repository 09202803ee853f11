"""GPU: evidence bias and cost of the sheared bound on the two pixels of scripts/proto_intersection.py (two components;
ntot 14.4 "bright", 14.0 "faint"; the same noise realisation): `copies` independent runs per pixel and setting, against
walks of 140 steps.
    python scripts/sampler_shear_bias.py <copies> <exp mode> shear:frames[:margin[:method]] ..."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import nestfit_amd as na
from nestfit_amd import sampler
from nestfit_amd.cube import CubeRunner
from nestfit_amd.synth import freq_axis

copies = int(sys.argv[1])
na.set_exp_mode(sys.argv[2])
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


def run(label, **kw):
    t0 = time.perf_counter()
    res = sampler.fit_pixels(cube, np.arange(2 * copies), nlive=400, tol=0.5, efr=0.3, seed=7, **kw)
    dt = time.perf_counter() - t0
    out = []
    for g in range(2):
        r = res[g * copies:(g + 1) * copies]
        lz = np.array([x.lnZ for x in r]); ev = np.array([x.n_evals for x in r]); it = np.array([x.n_iter for x in r])
        out.append((lz.mean(), lz.std(ddof=1) / np.sqrt(copies), ev.sum() / it.sum(), ev.mean()))
    return label, dt, out


ref = run('walks of 140 steps', method='walk', n_steps=140)
rows = [ref, run('shipped default (shear 2.5, 32 box frames, pairs 1.75)')]
for tok in sys.argv[3:]:
    f = tok.split(':')
    kw = dict(shear=float(f[0]), frames=int(f[1]))
    if len(f) > 2 and f[2]:
        kw['margin'] = float(f[2])
    if len(f) > 3:
        kw['method'] = f[3]
    rows.append(run(f'shear {f[0]} frames {f[1]}' + (f' margin {f[2]}' if len(f) > 2 and f[2] else '') + (f' {f[3]}' if len(f) > 3 else ''), **kw))
for label, dt, out in rows:
    print(f'{label:48s} {dt:6.1f} s   ' + '   '.join(
        f'{name}: lnZ {m:.3f} ({m - ref[2][g][0]:+.3f} +- {np.hypot(e, ref[2][g][1]):.3f}) evals/iter {epi:5.1f} evals {evm / 1e3:5.0f} k'
        for g, (name, (m, e, epi, evm)) in enumerate(zip(('bright', 'faint'), out))), flush=True)
