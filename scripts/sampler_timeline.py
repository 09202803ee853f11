#!/usr/bin/env python3
"""Active pixels against wall time of config 5's two nested-sampling runs (the bench's cube): where the tail of slow
pixels begins and how long it lasts.  usage: sampler_timeline.py [ncomp=2]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import nestfit_amd as na                                   # noqa: E402
from nestfit_amd import sampler                            # noqa: E402
from nestfit_amd.cube import CubeRunner                    # noqa: E402
from nestfit_amd.synth import freq_axis                    # noqa: E402


def main():
    ncomp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    side, n, noise, nlive = 32, 512, 0.1, 400
    n_pix = side * side
    rng = np.random.default_rng(0)
    axes = [freq_axis(1, n), freq_axis(2, n)]
    ut = na.get_irdc_priors(size=500, vsys=0.0)
    lon, lat = np.indices((side, side))
    r = np.hypot(lon - side / 2, lat - side / 2) / (side / 2)
    for _ in range(ncomp - 1):                             # the bench draws the one-component cube's noise first
        rng.normal(0, noise, (n_pix, 2 * n))
    truths = np.zeros((n_pix, 6 * ncomp))
    for c in range(ncomp):
        truths[:, c] = (-1.0 + 2.0 * lon.ravel() / side) + 1.5 * c
        truths[:, ncomp + c], truths[:, 2 * ncomp + c] = 12.0 + 3 * c, 5.0 + c
        truths[:, 3 * ncomp + c], truths[:, 4 * ncomp + c] = 14.6 - 0.6 * r.ravel(), 0.4
    probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=ncomp)
    model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
    cube = CubeRunner(axes, (1, 2), model + rng.normal(0, noise, model.shape), np.full((n_pix, 2), noise), ut, ncomp=ncomp)
    t0 = time.perf_counter()
    log = []

    def progress(n_active, _):
        log.append((time.perf_counter() - t0, n_active))

    res = sampler.fit_pixels(cube, np.arange(n_pix), nlive=nlive, tol=0.5, efr=0.3, seed=1, progress=progress)
    dt = time.perf_counter() - t0
    print(f'{ncomp} component(s): {dt:.2f} s, {np.mean([x.n_evals for x in res]) / 1e3:.0f} k evaluations and '
          f'{np.mean([x.n_iter for x in res]) / 1e3:.1f} k iterations per pixel, {res[0].rounds} rounds')
    for t, a in log[:: max(1, len(log) // 24)]:
        print(f'   {t:6.2f} s  {a:5d} active')
    for thr in (512, 256, 64, 16):
        t_thr = next((t for t, a in log if a <= thr), dt)
        print(f'   <= {thr:4d} pixels active from {t_thr:.2f} s on: {dt - t_thr:.2f} s of the run')


if __name__ == '__main__':
    main()
