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


def _c5_stack(engine, side, n, noise):
    """BASELINE config 5 as SURVEY.md 8d defines it (the generator bench.py --workload C5 uses: nestfit_amd/synth.py)."""
    from nestfit_amd.synth import c5_stack
    return c5_stack(side, n, noise)


@pytest.fixture(scope='module')
def c5(engine):
    return _c5_stack(engine, 32, 1024, 0.2)


def _c5_component_snr(truths, axes, ut, n, noise):
    """Peak brightness of every true component alone against the noise, per pixel: how many components are there to find."""
    from nestfit_amd.cube import CubeRunner
    n_pix = truths.shape[0]
    single = np.zeros((n_pix, 2))
    probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=1)
    for c in range(2):
        one, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), np.ascontiguousarray(truths[:, c::2]))
        single[:, c] = one.max(axis=1) / noise
    return single


def test_config5_as_specified(engine, c5, tmp_path):
    """SURVEY.md 8d C5 through the cube driver, at size: 32 x 32 pixels of the C3 generator (1024 channels, two-component
    truths), 400 live points, tol 0.5, efr 0.3, fixed seed, ncomp_max = 2 with the lnZ_thresh = 11 loop of
    main.py:452-469, in the table mode (the reference's arithmetic).  The stored evidences say how many components
    every pixel needs.  (The device sampler against the oracle-fed twin: the next test, on a quarter of the cube.)"""
    from nestfit_amd.fitter import CubeFitter
    from nestfit_amd.store import HdfStore
    side, n, noise, nlive = 32, 1024, 0.2, 400
    n_pix = side * side
    try:
        stack, truths, model, data, axes, ut = c5
        engine.set_exp_mode('table')
        fitter = CubeFitter(stack, ut, engine.AmmoniaRunner, lnZ_thresh=11, ncomp_max=2, nlive_snr_fact=0,
                            mn_kwargs={'nlive': nlive, 'tol': 0.5, 'efr': 0.3, 'seed': 5})
        fitter.fit_cube(str(tmp_path / 'c5'), nproc=1)
        with HdfStore(str(tmp_path / 'c5')) as store:
            groups = {(g.attrs['i_lon'], g.attrs['i_lat']): g for g in store.iter_pix_groups()}
            assert len(groups) == n_pix and store.hdf.attrs['n_max_components'] == 2 and store.hdf.attrs['lnZ_threshold'] == 11
            assert 'nlive_quantum' not in store.hdf.attrs                    # exactly the reference's live points
            nbest = np.array([groups[(p // side, p % side)].attrs['nbest'] for p in range(n_pix)])
            gain1 = np.array([groups[(p // side, p % side)]['1'].attrs['global_lnZ'] - groups[(p // side, p % side)]['1'].attrs['null_lnZ']
                              for p in range(n_pix)])
            for p in range(n_pix):
                g = groups[(p // side, p % side)]
                assert g['1'].attrs['n_live'] == nlive and g['1'].attrs['n_chan_tot'] == 2 * n
                # the loop of main.py:452-469: N + 1 components are tried exactly when N gained lnZ_thresh
                assert ('2' in g) == (gain1[p] >= 11)
                if '2' in g:
                    gain2 = g['2'].attrs['global_lnZ'] - g['1'].attrs['global_lnZ']
                    assert (nbest[p] == 2) == (gain2 >= 11)
                    assert g['2']['posteriors'].shape[1] == 14 and g['2'].attrs['n_params'] == 12
                else:
                    assert nbest[p] == 0
        single = _c5_component_snr(truths, axes, ut, n, noise)
        # The generator's ranges (ntot 13..16, tex 2.8..12 K, separations down to 0.16 km/s) leave about half of the
        # pixels with a second component that is faint or blended: nbest = 2 is asserted where two components are
        # there to be found -- both clear of the noise and further apart than their blended width -- and nbest >= 1
        # wherever one is; the table printed says what the thresholds do.
        fwhm_blend = 2.355 * np.sqrt(truths[:, 8] * truths[:, 9])
        dv = np.abs(truths[:, 1] - truths[:, 0])
        print(f'C5 as specified: nbest = 0 / 1 / 2 on {(nbest == 0).sum()} / {(nbest == 1).sum()} / {(nbest == 2).sum()} of {n_pix} pixels')
        for snr_min in (3, 5, 8):
            for sep in (0.5, 1.0, 1.2):
                sel = (single.min(axis=1) >= snr_min) & (dv >= sep * fwhm_blend)
                print(f'   both components >= {snr_min} sigma, separation >= {sep} blended FWHM: {sel.sum():4d} pixels, '
                      f'nbest = 2 on {(nbest[sel] == 2).mean() if sel.any() else float("nan"):.3f}')
        want2 = (single.min(axis=1) >= 8) & (dv >= 1.0 * fwhm_blend)
        assert want2.sum() >= 40 and (nbest[want2] == 2).mean() >= 0.9
        assert (nbest[single.max(axis=1) >= 8] >= 1).mean() >= 0.99
    finally:
        engine.set_exp_mode('fast')


@pytest.mark.parametrize('ncomp', [1, 2])
def test_config5_device_sampler_against_the_oracle_fed_twin(engine, nfo, c5, ncomp):
    """The nested-sampling runs of config 5's first 256 pixels (a quarter of the cube: the numpy twin is the slow side),
    one and two components, table mode: the device sampler against the numpy twin whose likelihood for 8 two-component
    pixels is the CPU oracle, over the first 1000 iterations (the oracle needs ~40 us per evaluation) -- the same
    decisions (iteration and evaluation counts equal for EVERY pixel), lnZ to 1e-10, posteriors to 1e-8."""
    from nestfit_amd import sampler
    from nestfit_amd.cube import CubeRunner
    n, noise, nlive, n_sub = 1024, 0.2, 400, 256
    try:
        stack, truths, model, data, axes, ut = c5
        engine.set_exp_mode('table')
        single = _c5_component_snr(truths[:n_sub], axes, ut, n, noise)
        fwhm_blend = 2.355 * np.sqrt(truths[:n_sub, 8] * truths[:n_sub, 9])
        want2 = (single.min(axis=1) >= 8) & (np.abs(truths[:n_sub, 1] - truths[:n_sub, 0]) >= fwhm_blend)
        check = np.random.default_rng(5).choice(np.flatnonzero(want2), 8, replace=False)
        ps = nfo.PriorSet(ut.lower())
        pix = np.arange(n_sub)
        sub = np.ascontiguousarray(data[:n_sub])
        cube = CubeRunner(axes, (1, 2), sub, np.full((n_sub, 2), noise), ut, ncomp=ncomp)
        oracle = {int(p): nfo.AmmoniaRunner([nfo.AmmoniaSpectrum(axes[k], sub[p, k * n:(k + 1) * n], noise, t, native=False)
                                             for k, t in enumerate((1, 2))], ps, ncomp=ncomp) for p in check}

        def loglike_hybrid(px, U):
            U0 = U.copy()
            out = cube.loglikelihood_batch(px.astype(np.int32), U)          # transforms U in place
            for p, run in oracle.items():
                m = px == p
                if m.any():
                    rows = U0[m]
                    out[m] = run.loglikelihood_batch(rows)
                    U[m] = rows
            return out

        kw = dict(nlive=nlive, tol=0.5, efr=0.3, seed=5, maxiter=1000, free_mask=cube.utrans.free_mask(ncomp))
        dev = sampler.fit_pixels(cube, pix, **kw)
        twin = sampler.run_nested(loglike_hybrid, cube.ndim, n_sub, **kw)
        for p in check:
            g, r = dev[p], twin[p]
            assert g.n_iter == r.n_iter == 1000 and g.n_evals == r.n_evals, (ncomp, p, g.n_evals, r.n_evals)
            assert g.lnZ == pytest.approx(r.lnZ, rel=1e-10)
            np.testing.assert_allclose(g.posterior, r.posterior, rtol=1e-8, atol=1e-12)
        assert all(a.n_evals == b.n_evals and a.n_iter == b.n_iter for a, b in zip(dev, twin))
    finally:
        engine.set_exp_mode('fast')
