"""BASELINE configs 3 and 5 at their full size on one GPU, against the CPU oracle.

C3: the 128 x 128-pixel cube of 2-component NH3 (1,1)+(2,2) spectra, 2 x 1024 channels (268 MB in
HBM, SURVEY.md 8d generator): one (pixel, unit-cube row) item per pixel through
`CubeRunner.loglikelihood_batch`; 256 pixels against the oracle, all 16384 through size-independent
properties (permutation equivariance, batch-split invariance, null evidence against numpy).
C5: nested sampling of a 32 x 32 cube with 400 live points per pixel on the device sampler in the
bit-faithful table mode: every pixel is a detection (main.py:464-469 threshold), and the same 1024
lock-step runs replayed by the numpy twin -- whose likelihood for 8 of the pixels is the CPU oracle --
make the same decisions (iteration and evaluation counts equal, lnZ to 1e-10).
"""
import numpy as np
import pytest

from nestfit_amd.synth import freq_axis, param_sampler_draw

pytestmark = pytest.mark.gpu
LNL_RTOL = {'table': 1e-9, 'fast': 1e-6}


@pytest.fixture(scope='module')
def c3_cube(engine, nfo):
    """The config-3 cube: per-pixel truths from the ParamSampler ranges (default_rng(11)), engine model
    spectra (fast mode) + normal noise of 0.2 K; resident on the GPU as ONE spectra set."""
    from nestfit_amd.cube import CubeRunner
    side, n, noise = 128, 1024, 0.2
    n_pix = side * side
    rng = np.random.default_rng(11)
    truths = np.array([param_sampler_draw(rng) for _ in range(n_pix)])
    axes = [freq_axis(t, n) for t in (1, 2)]
    ut = engine.get_irdc_priors(size=500, vsys=0.0)
    engine.set_exp_mode('fast')
    probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=2)
    data = np.empty((n_pix, 2 * n))
    for a in range(0, n_pix, 4096):
        data[a:a + 4096], _ = probe.predict_batch(np.zeros(min(4096, n_pix - a), dtype=np.int32), truths[a:a + 4096])
    data += rng.normal(0, noise, data.shape)
    cube = CubeRunner(axes, (1, 2), data, np.full((n_pix, 2), noise), ut, ncomp=2)
    assert data.nbytes == 268435456 and cube.n_pix == 16384 and cube.n_chan_tot == 2048
    return cube, data, axes, ut, noise


@pytest.mark.parametrize('mode', ['table', 'fast'])
def test_config3_cube_at_size(engine, nfo, c3_cube, mode):
    cube, data, axes, ut, noise = c3_cube
    engine.set_exp_mode(mode)
    try:
        n_pix, n = cube.n_pix, 1024
        pix = np.arange(n_pix, dtype=np.int32)
        U = np.random.default_rng(7).uniform(size=(n_pix, 12))
        U1 = U.copy()
        l1 = cube.loglikelihood_batch(pix, U1)                       # one item per pixel
        assert np.isfinite(l1).all()
        # 256 pixels against the oracle (its own prior transform, model spectra and chi^2)
        ps = nfo.PriorSet(ut.lower())
        for p in np.random.default_rng(3).choice(n_pix, 256, replace=False):
            run = nfo.AmmoniaRunner([nfo.AmmoniaSpectrum(axes[k], data[p, k * n:(k + 1) * n], noise, t)
                                     for k, t in enumerate((1, 2))], ps, ncomp=2)
            u = U[p].copy()
            assert l1[p] == pytest.approx(run.loglikelihood(u), rel=LNL_RTOL[mode])
            np.testing.assert_allclose(U1[p], u, rtol=1e-11, atol=1e-13)
        # every pixel: items are independent of their company and of their place in the batch
        perm = np.random.default_rng(8).permutation(n_pix)
        U2 = U[perm].copy()
        assert np.array_equal(cube.loglikelihood_batch(pix[perm], U2), l1[perm])
        assert np.array_equal(U2, U1[perm])
        U3 = U.copy()
        parts = [cube.loglikelihood_batch(pix[a:b], U3[a:b]) for a, b in ((0, 5000), (5000, 5001), (5001, n_pix))]
        assert np.array_equal(np.concatenate(parts), l1)
        # null evidence of every pixel (core.pyx:517-520) against numpy
        np.testing.assert_allclose(cube.null_lnZ, -(data ** 2).sum(axis=1) / (2 * noise ** 2), rtol=1e-12)
        # a pixel's result does not depend on which other pixels the cube holds: the stripe a rank of an
        # 8-GPU run would own (i_lon % 8 == 3) as its own spectra set
        from nestfit_amd.cube import CubeRunner, shard_pixels
        lon, lat = shard_pixels((128, 128), 3, 8)
        mine = lon * 128 + lat
        stripe = CubeRunner(axes, (1, 2), data[mine], np.full((mine.size, 2), noise), ut, ncomp=2)
        Us = U[mine].copy()
        assert np.array_equal(stripe.loglikelihood_batch(np.arange(mine.size, dtype=np.int32), Us), l1[mine])
    finally:
        engine.set_exp_mode('fast')


def _c5_cube(engine, side, n, noise):
    """The cube bench.py --workload C5 times: one-component truths with a velocity gradient and a
    radial column-density fall-off, every pixel well above the noise."""
    from nestfit_amd.cube import CubeRunner
    n_pix = side * side
    axes = [freq_axis(1, n), freq_axis(2, n)]
    ut = engine.get_irdc_priors(size=500, vsys=0.0)
    lon, lat = np.indices((side, side))
    r = np.hypot(lon - side / 2, lat - side / 2) / (side / 2)
    truths = np.zeros((n_pix, 6))
    truths[:, 0] = -1.0 + 2.0 * lon.ravel() / side
    truths[:, 1], truths[:, 2] = 12.0, 5.0
    truths[:, 3], truths[:, 4] = 14.6 - 0.6 * r.ravel(), 0.4
    probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=1)
    model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
    data = model + np.random.default_rng(0).normal(0, noise, model.shape)
    return CubeRunner(axes, (1, 2), data, np.full((n_pix, 2), noise), ut, ncomp=1), data, axes, ut, truths


def test_config5_sampler_at_size(engine, nfo):
    from nestfit_amd import sampler
    side, n, noise, nlive = 32, 512, 0.1, 400
    engine.set_exp_mode('fast')
    cube, data, axes, ut, truths = _c5_cube(engine, side, n, noise)
    n_pix = side * side
    pix = np.arange(n_pix)
    mask = cube.utrans.free_mask(1)
    try:
        engine.set_exp_mode('table')
        # ---- the full run: 1024 pixels x 400 live points to convergence on the device
        res = sampler.fit_pixels(cube, pix, nlive=nlive, tol=0.5, efr=0.3, seed=1)
        gain = np.array([r.lnZ for r in res]) - cube.null_lnZ
        assert (gain > 11).all(), f'{(gain <= 11).sum()} pixels below the evidence threshold'      # main.py:464-469
        assert all(np.isfinite(r.lnZ) and r.lnZ_err > 0 and r.n_iter > nlive for r in res)
        v = np.array([r.param_constr[0][0] for r in res])
        s = np.array([r.param_constr[1][0] for r in res])
        assert (np.abs(v - truths[:, 0]) < 5 * s + 0.02).mean() > 0.99                             # velocities recovered
        # ---- the same 1024 lock-step runs for their first 1500 iterations, device sampler against the
        # numpy twin; the twin's likelihood for 8 pixels is the CPU oracle, for the others the engine
        check = np.random.default_rng(5).choice(n_pix, 8, replace=False)
        ps = nfo.PriorSet(ut.lower())
        oracle = {int(p): nfo.AmmoniaRunner([nfo.AmmoniaSpectrum(axes[k], data[p, k * n:(k + 1) * n], noise, t, native=False)
                                             for k, t in enumerate((1, 2))], ps, ncomp=1) for p in check}

        def loglike(px, U):
            out = cube.loglikelihood_batch(px.astype(np.int32), U)        # transforms U in place
            return out

        def loglike_hybrid(px, U):
            U0 = U.copy()
            out = loglike(px, U)
            for p, run in oracle.items():
                m = px == p
                if m.any():
                    sub = U0[m]
                    out[m] = run.loglikelihood_batch(sub)
                    U[m] = sub
            return out

        kw = dict(nlive=nlive, tol=0.5, efr=0.3, seed=1, maxiter=1500, free_mask=mask)
        dev = sampler.fit_pixels(cube, pix, **kw)
        twin = sampler.run_nested(loglike_hybrid, cube.ndim, n_pix, **kw)
        for p in check:
            g, r = dev[p], twin[p]
            assert g.n_iter == r.n_iter == 1500 and g.n_evals == r.n_evals, (p, g.n_evals, r.n_evals)
            assert g.lnZ == pytest.approx(r.lnZ, rel=1e-10)
            np.testing.assert_allclose(g.posterior, r.posterior, rtol=1e-8, atol=1e-12)
        # and the other 1016 pixels (engine likelihood on both sides): identical counts everywhere
        assert all(a.n_evals == b.n_evals and a.n_iter == b.n_iter for a, b in zip(dev, twin))
    finally:
        engine.set_exp_mode('fast')
