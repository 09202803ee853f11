"""The evidence bias of the built-in sampler's named settings (nestfit_amd.sampler.PRECISION), pinned.

A bound that is fitted to the live points cuts prior mass the live points do not show, and the cut shows in lnZ: the
sheared ellipsoid's safety factor and the margins of its free rejections (boxes, pair ellipses) trade evaluations for
it.  Reference: 256 pixels of the two-component test cube (nestfit_amd.synth.c5r4_cube), rejection sampling from the
sheared ellipsoid alone with a factor of 8 on its volume -- no boxes, no pair ellipses, no walks, 8.4 M evaluations per
pixel --, eight seeds; per-pixel means committed as tests/golden/sampler_bias_reference.json by
scripts/sampler_bias_reference.py (cube mean -531.111 +- 0.008).  A pixel's own lnZ_err is 0.25; the reference's
model selection works with a threshold of 11 (nestfit/main.py:464-469)."""
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).parent / 'golden' / 'sampler_bias_reference.json'


@pytest.fixture(scope='module')
def bias_cube(engine):
    from nestfit_amd.cube import CubeRunner
    from nestfit_amd.synth import c5r4_cube
    ref = json.loads(GOLDEN.read_text())
    engine.set_exp_mode(ref['exp_mode'])
    axes, data, noise, ut = c5r4_cube(2)
    n_sub = len(ref['lnZ'])
    cube = CubeRunner(axes, (1, 2), np.ascontiguousarray(data[:n_sub]), np.full((n_sub, 2), noise), ut, ncomp=2)
    return cube, np.array(ref['lnZ']), float(ref['mean_lnZ_se'])


def _bias(cube, want, seeds, **kw):
    from nestfit_amd import sampler
    got, evals = [], []
    for seed in seeds:
        res = sampler.fit_pixels(cube, np.arange(want.size), nlive=400, tol=0.5, efr=0.3, seed=seed, **kw)
        got.append([r.lnZ for r in res])
        evals.append(np.mean([r.n_evals for r in res]))
    d = np.array(got).mean(axis=0) - want
    return float(d.mean()), float(d.std(ddof=1) / np.sqrt(d.size)), float(np.mean(evals))


def test_evidence_bias_of_the_named_settings(engine, bias_cube):
    """default: mean lnZ within +0.04 of bound-free rejection (measured +0.031); 'evidence': within +0.03 (+0.018);
    'speed' (round 4's default) is the +0.075 it was measured at -- each to two standard errors of the comparison, which
    are printed (the comparison is paired per pixel, so what is left is the runs' own scatter: 0.25 / sqrt(256 seeds))."""
    cube, want, ref_se = bias_cube
    try:
        rows = {}
        for name, seeds in (('default', range(11, 17)), ('evidence', range(11, 15)), ('speed', range(11, 14))):
            b, se, ev = _bias(cube, want, seeds, precision=name)
            se = float(np.hypot(se, ref_se))
            rows[name] = (b, se, ev)
            print(f'precision {name:9s}: lnZ bias {b:+.4f} +- {se:.4f}, {ev / 1e3:.0f} k evaluations per pixel ({len(seeds)} seeds x {want.size} pixels)')
        b, se, ev = rows['default']
        assert -0.03 <= b <= 0.04 + 2 * se
        b, se, _ = rows['evidence']
        assert -0.03 <= b <= 0.03 + 2 * se
        b, se, ev_speed = rows['speed']
        assert 0.03 <= b <= 0.075 + 3 * se                       # (what the speed costs; a doubling would show)
        assert ev_speed < rows['default'][2] < rows['evidence'][2]
    finally:
        engine.set_exp_mode('fast')


def test_precision_is_recorded_in_the_store(engine, tmp_path):
    """A store says which setting its evidences were sampled with (attribute sampler_precision of the table file)."""
    from nestfit_amd.fitter import CubeFitter
    from nestfit_amd.store import HdfStore
    from nestfit_amd.synth import c5_stack
    stack, *_, ut = c5_stack(4, 128, 0.2)
    for prec in (None, 'evidence'):
        kw = {'nlive': 60, 'tol': 1.0, 'efr': 0.3, 'seed': 3, 'maxiter': 300}
        if prec:
            kw['precision'] = prec
        fitter = CubeFitter(stack, ut, engine.AmmoniaRunner, lnZ_thresh=11, ncomp_max=1, nlive_snr_fact=0, mn_kwargs=kw)
        name = str(tmp_path / f'store_{prec}')
        fitter.fit_cube(name, nproc=1)
        with HdfStore(name) as store:
            assert store.hdf.attrs['sampler_precision'] == (prec or 'default')
