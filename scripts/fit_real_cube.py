#!/usr/bin/env python3
"""End to end on real data: the reference's EVLA NH3 (1,1)+(2,2) cutout cubes (20 x 20 pixels x 379
channels, tests/golden/ = nestfit/test/data/) read by the FITS reader, every pixel fitted with up
to two components by the cube driver on the device sampler, results written in the reference's
store layout.  usage: fit_real_cube.py [store_name=/tmp/nestfit_amd_cutout] [ncomp_max=2] [nlive_quantum=1] [groups]
(`groups`: one lock-step run per distinct number of live points instead of one run for the whole stripe)"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import nestfit_amd as na                                   # noqa: E402
from nestfit_amd.cubeio import CubeStack, DataCube, SimpleCube   # noqa: E402
from nestfit_amd.fitter import CubeFitter                  # noqa: E402
from nestfit_amd.store import HdfStore                     # noqa: E402


def main():
    store_name = sys.argv[1] if len(sys.argv) > 1 else '/tmp/nestfit_amd_cutout'
    ncomp_max = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    quantum = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # 1: every pixel exactly the reference's number of live points
    rms = 0.35                                             # nestfit/test/__init__.py:12
    t0 = time.perf_counter()
    stack = CubeStack([
        DataCube(SimpleCube.read(ROOT / 'tests' / 'golden' / f'ammonia_{t}{t}_cutout.fits')[:-1], rms, trans_id=t)
        for t in (1, 2)])
    t_read = time.perf_counter() - t0
    ut = na.get_irdc_priors(size=500, vsys=63.7)           # G23.481: v_lsr ~ 63.7 km/s
    fitter = CubeFitter(stack, ut, na.AmmoniaRunner, lnZ_thresh=11, ncomp_max=ncomp_max,
                        mn_kwargs={'nlive': 100, 'tol': 1.0, 'efr': 0.3, 'seed': 1}, nlive_snr_fact=5,
                        nlive_quantum=quantum)
    fitter.one_group = 'groups' not in sys.argv[4:]
    t0 = time.perf_counter()
    fitter.fit_cube(store_name, nproc=1)
    t_fit = time.perf_counter() - t0
    with HdfStore(store_name) as store:
        groups = list(store.iter_pix_groups())
        nbest = np.array([g.attrs['nbest'] for g in groups])
        gain = np.array([g['1'].attrs['global_lnZ'] - g['1'].attrs['null_lnZ'] for g in groups])
        v1 = np.array([g['1']['map_params'][0] for g in groups if g.attrs['nbest'] >= 1])
        n_runs = sum(len([k for k in g.keys()]) for g in groups)
        evals = sum(int(g[k].attrs['n_samples']) for g in groups for k in g.keys())
    print(f'nlive_quantum {quantum}, {"one lock-step run" if fitter.one_group else "a run per live-point count"}: read 2 cubes in {t_read:.2f} s; fitted {len(groups)} pixels ({n_runs} runs) in {t_fit:.1f} s '
          f'= {len(groups)/t_fit:.1f} pixels/s; posterior samples stored {evals}')
    print(f'nbest histogram: {np.bincount(nbest, minlength=ncomp_max + 1).tolist()}; '
          f'median lnZ gain of N=1 over the null model {np.median(gain):.1f}; '
          f'MAP velocity of component 1: median {np.median(v1):.2f} km/s, 16-84 % {np.percentile(v1, 16):.2f}..{np.percentile(v1, 84):.2f}')


if __name__ == '__main__':
    main()
