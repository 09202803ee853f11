"""GPU: evidence bias and cost of settings of the bound's free rejections against the committed reference
(tests/golden/sampler_bias_reference.json, or gpurun_out/r05/...): python scripts/sampler_bias_scan.py margin:pairs:shear[:method] ..."""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import nestfit_amd as na
from nestfit_amd import sampler
from nestfit_amd.cube import CubeRunner
from nestfit_amd.synth import c5r4_cube
ref_file = next(p for p in (ROOT / 'tests' / 'golden' / 'sampler_bias_reference.json', ROOT / 'gpurun_out' / 'r05' / 'sampler_bias_reference.json') if p.exists())
ref = json.loads(ref_file.read_text())
na.set_exp_mode(ref['exp_mode'])
N_SUB = 256
axes, data, noise, ut = c5r4_cube(2)
cube = CubeRunner(axes, (1, 2), np.ascontiguousarray(data[:N_SUB]), np.full((N_SUB, 2), noise), ut, ncomp=2)
want = np.array(ref['lnZ'])
for tok in sys.argv[1:]:
    f = tok.split(':')
    kw = {}
    if f[0] not in ('', '-'): kw['margin'] = float(f[0])
    if f[1] not in ('', '-'): kw['pairs'] = float(f[1])
    if len(f) > 2 and f[2] not in ('', '-'): kw['shear'] = float(f[2])
    if len(f) > 3 and f[3]: kw['method'] = f[3]
    if len(f) > 4 and f[4]: kw['frames'] = int(f[4])
    got, ev, t0 = [], [], time.perf_counter()
    for k in range(4):
        res = sampler.fit_pixels(cube, np.arange(N_SUB), nlive=400, tol=0.5, efr=0.3, seed=11 + k, **kw)
        got.append([r.lnZ for r in res]); ev.append(np.mean([r.n_evals for r in res]))
    d = np.array(got).mean(axis=0) - want
    print(f'{tok:28s}: bias {d.mean():+.4f} +- {d.std(ddof=1) / np.sqrt(d.size):.4f}, {np.mean(ev) / 1e3:.0f} k evaluations per pixel, {(time.perf_counter() - t0) / 4:.2f} s per run', flush=True)
