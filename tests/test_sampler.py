"""Built-in batched nested sampler.

CPU part: analytic evidences, reproducibility, the reference-shaped Dumper / run_multinest front
end.  GPU part: the same sampler, same seed, fed by the HIP engine and by the CPU oracle."""
import numpy as np
import pytest

from nestfit_amd import sampler
from nestfit_amd.synth import freq_axis


def _gauss_problem(centres, sigma):
    centres = np.atleast_2d(centres)

    def loglike(pix, U):
        d2 = ((U[:, None, :] - centres[None]) ** 2).sum(axis=2)
        return np.logaddexp.reduce(-0.5 * d2 / sigma ** 2, axis=1)
    return loglike


def test_gaussian_evidence_and_posterior():
    ndim, sigma, P = 4, 0.05, 8
    res = sampler.run_nested(_gauss_problem(np.full(ndim, 0.5), sigma), ndim, P, nlive=200, tol=0.1,
                             efr=0.5, seed=12)
    truth = ndim * np.log(sigma * np.sqrt(2 * np.pi))
    lnZ = np.array([r.lnZ for r in res])
    err = np.array([r.lnZ_err for r in res])
    assert np.all(np.abs(lnZ - truth) < 4 * err + 0.05), (lnZ, truth, err)
    assert abs(lnZ.mean() - truth) < 4 * err.mean() / np.sqrt(P) + 0.03
    # the sample scatter of lnZ over the runs is what lnZ_err claims
    assert 0.4 < lnZ.std(ddof=1) / err.mean() < 2.0
    for r in res:
        assert r.posterior.shape == (r.n_samples, ndim + 2)
        assert r.posterior[:, -1].sum() == pytest.approx(1.0, abs=1e-12)
        assert r.n_samples == r.n_iter + r.n_live
        mean, sig, best, mapp = r.param_constr
        np.testing.assert_allclose(mean, 0.5, atol=0.01)
        np.testing.assert_allclose(sig, sigma, rtol=0.2)
        np.testing.assert_allclose(best, 0.5, atol=0.03)
        assert r.max_loglike == pytest.approx(-0.5 * r.posterior[:, -2].min())
        assert r.information == pytest.approx(-truth - ndim / 2, abs=0.5)     # H of a Gaussian in a box


def test_two_modes_in_one_ellipsoid():
    ndim, sigma = 3, 0.04
    centres = np.array([[0.3, 0.3, 0.3], [0.7, 0.7, 0.7]])
    res = sampler.run_nested(_gauss_problem(centres, sigma), ndim, 4, nlive=300, tol=0.1, efr=0.3, seed=5)
    truth = np.log(2) + ndim * np.log(sigma * np.sqrt(2 * np.pi))
    for r in res:
        assert abs(r.lnZ - truth) < 4 * r.lnZ_err + 0.05
        w, x = r.posterior[:, -1], r.posterior[:, 0]
        assert w[x < 0.5].sum() == pytest.approx(0.5, abs=0.15)               # both modes populated


def test_walk_mode_evidence_and_switch():
    """Constrained random walks instead of rejection sampling: same evidence within the error, about
    n_steps evaluations per iteration; 'auto' switches when rejection gets inefficient."""
    ndim, sigma, P = 6, 0.04, 6
    f = _gauss_problem(np.full(ndim, 0.5), sigma)
    truth = ndim * np.log(sigma * np.sqrt(2 * np.pi))
    res = sampler.run_nested(f, ndim, P, nlive=150, tol=0.1, efr=0.5, seed=21, method='walk', n_steps=20,
                             batch_target=512)
    lnZ = np.array([r.lnZ for r in res])
    err = np.mean([r.lnZ_err for r in res])
    assert abs(lnZ.mean() - truth) < 4 * err / np.sqrt(P) + 0.05, (lnZ - truth, err)
    assert np.all(np.abs(lnZ - truth) < 4.5 * err)
    per_iter = np.mean([r.n_evals / r.n_iter for r in res])
    assert 15 < per_iter < 30
    np.testing.assert_allclose(res[0].param_constr[1], sigma, rtol=0.25)
    # two narrow modes in opposite corners: ONE ellipsoid around both accepts almost nothing and 'auto' turns to walks
    # (walk_factor = 2: the switch at 1 in 2 n_steps, the default from seven sampled dimensions on; with five the default
    # waits for 1 in 64 n_steps -- on the GPU a rejection round is one large batch, a walk cycle n_steps small ones) ...
    D, sig = 5, 0.01
    f2 = _gauss_problem(np.array([[0.25] * D, [0.75] * D]), sig)
    truth2 = np.log(2) + D * np.log(sig * np.sqrt(2 * np.pi))
    kw = dict(nlive=100, tol=0.5, seed=4, batch_target=512)
    auto = sampler.run_nested(f2, D, 1, method='auto', n_steps=20, walk_factor=2, ellipsoids=1, **kw)[0]
    rej = sampler.run_nested(f2, D, 1, method='reject', ellipsoids=1, **kw)[0]
    assert auto.n_evals < 0.8 * rej.n_evals
    # ... and the default bound of a five-dimensional fit, up to four ellipsoids, puts one around each mode (mmodal)
    multi = sampler.run_nested(f2, D, 1, method='auto', **kw)[0]
    assert multi.n_evals < 0.4 * rej.n_evals and multi.n_evals < 0.7 * auto.n_evals
    for r in (auto, rej, multi):
        assert abs(r.lnZ - truth2) < 4 * r.lnZ_err
        w, x = r.posterior[:, -1], r.posterior[:, 0]
        assert 0.02 < w[x < 0.5].sum() < 0.98                                # both modes kept (100 live points: noisy)


def test_dummy_dimensions_are_integrated_out():
    """free_mask: slots the likelihood ignores are not sampled; the evidence is the one of the
    reduced problem, the full-length rows carry u = 0.5 there."""
    sigma = 0.05
    def like5(pix, U):                                   # depends on slots 0, 2, 3 only
        assert (U[:, 1] == 0.5).all() and (U[:, 4] == 0.5).all()
        return -0.5 * ((U[:, [0, 2, 3]] - 0.5) ** 2).sum(axis=1) / sigma ** 2
    res = sampler.run_nested(like5, 5, 6, nlive=150, tol=0.1, efr=0.5, seed=9, free_mask=[1, 0, 1, 1, 0])
    truth = 3 * np.log(sigma * np.sqrt(2 * np.pi))
    lnZ = np.array([r.lnZ for r in res])
    err = np.mean([r.lnZ_err for r in res])
    assert abs(lnZ.mean() - truth) < 4 * err / np.sqrt(6) + 0.03
    assert res[0].posterior.shape[1] == 7 and (res[0].posterior[:, 1] == 0.5).all()
    full = sampler.run_nested(lambda pix, U: -0.5 * ((U[:, [0, 2, 3]] - 0.5) ** 2).sum(axis=1) / sigma ** 2,
                              5, 6, nlive=150, tol=0.1, efr=0.5, seed=9)
    assert np.mean([r.n_evals for r in res]) < 1.1 * np.mean([r.n_evals for r in full])     # no dearer (three dimensions or five: both cheap), same answer
    assert abs(np.mean([r.lnZ for r in full]) - truth) < 4 * err / np.sqrt(6) + 0.03


def test_free_mask_of_the_reference_prior_sets():
    import nestfit_amd as na
    assert na.get_irdc_priors().free_mask(2).tolist() == [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0]      # orth constant
    assert na.get_synth_priors().free_mask(1).tolist() == [1, 1, 0, 1, 1, 0]                        # tex duplicated


def test_seed_reproducibility_and_independent_pixels():
    f = _gauss_problem(np.full(3, 0.4), 0.1)
    a = sampler.run_nested(f, 3, 3, nlive=60, seed=7)
    b = sampler.run_nested(f, 3, 3, nlive=60, seed=7)
    c = sampler.run_nested(f, 3, 3, nlive=60, seed=8)
    for x, y in zip(a, b):
        assert x.lnZ == y.lnZ and np.array_equal(x.posterior, y.posterior)
    assert a[0].lnZ != c[0].lnZ
    assert len({r.lnZ for r in a}) == 3


def test_pixels_with_live_points_of_their_own():
    """One lock-step run, a number of live points per pixel (the cube driver's nlive + int(5 SNR), main.py:445-447): every
    pixel's run is the run it would have had alone with that number -- same random stream, same decisions."""
    f = _gauss_problem(np.full(3, 0.4), 0.1)
    nl = np.array([60, 75, 90, 75])
    mixed = sampler.run_nested(f, 3, 4, nlive=nl, seed=7)
    for p, r in enumerate(mixed):
        assert r.n_live == nl[p] and r.n_samples == r.n_iter + nl[p]
        assert r.posterior[:, -1].sum() == pytest.approx(1.0, abs=1e-12)
    # pixels 1 and 3 (both 75) against a run in which everybody has 75: pixel p's stream is keyed by p alone
    same = sampler.run_nested(f, 3, 4, nlive=75, seed=7)
    for p in (1, 3):
        assert mixed[p].lnZ == same[p].lnZ and np.array_equal(mixed[p].posterior, same[p].posterior)
    assert mixed[0].lnZ != same[0].lnZ
    truth = 3 * np.log(0.1 * np.sqrt(2 * np.pi))
    assert all(abs(r.lnZ - truth) < 4 * r.lnZ_err + 0.1 for r in mixed)


def test_maxiter_and_logzero():
    f = _gauss_problem(np.full(2, 0.5), 0.1)
    r = sampler.run_nested(f, 2, 2, nlive=50, maxiter=30, seed=1)
    assert all(x.n_iter == 30 and x.n_samples == 80 for x in r)
    r0 = sampler.run_nested(f, 2, 1, nlive=50, maxiter=0, seed=1)[0]
    assert r0.n_iter == 0 and r0.n_samples == 50

    def nan_loglike(pix, U):
        out = _gauss_problem(np.full(2, 0.5), 0.1)(pix, U)
        out[U[:, 0] < 0.2] = np.nan                                           # treated as logZero
        return out
    r = sampler.run_nested(nan_loglike, 2, 1, nlive=80, seed=3)[0]
    assert np.isfinite(r.lnZ) and r.posterior[r.posterior[:, -1] > 1e-6, 0].min() >= 0.2


class _FakeRunner:
    ndim = n_params = 3
    ncomp = 1
    null_lnZ = -50.0
    n_chan_tot = 100
    run_lnZ = float('nan')

    def loglikelihood_batch(self, U):
        return _gauss_problem(np.full(3, 0.5), 0.1)(None, U)


def test_run_multinest_front_end_writes_the_reference_layout():
    """Attributes and datasets of mn_dump (core.pyx:627-687, docs/store_spec.rst)."""
    group = sampler.MemoryGroup()
    dumper = sampler.Dumper(group)
    runner = _FakeRunner()
    res = sampler.run_multinest(runner, dumper, nlive=80, tol=0.5, efr=0.5, seed=4)
    assert runner.run_lnZ == res.lnZ
    for key in ('ncomp', 'null_lnZ', 'n_chan_tot', 'n_samples', 'n_live', 'n_params', 'global_lnZ',
                'global_lnZ_err', 'max_loglike', 'marg_cols', 'marg_quantiles', 'BIC', 'AIC', 'AICc',
                'null_BIC', 'null_AIC', 'null_AICc'):
        assert key in group.attrs, key
    k, n = 3.0, 100.0
    assert group.attrs['BIC'] == pytest.approx(np.log(n) * k - 2 * res.max_loglike)
    assert group.attrs['null_AICc'] == pytest.approx(2 * k + 100.0 + (2 * k * k + 2 * k) / (n - k - 1))
    assert group['posteriors'].dtype == np.float32
    assert group['posteriors'].shape == (res.n_samples, 5)
    assert group['marginals'].shape == (15, 3)
    assert len(group.attrs['marg_cols']) == 15
    np.testing.assert_allclose(group['marginals'][4], 0.5, atol=0.2)          # p50 row, unweighted like the reference
    assert group['bestfit_params'].shape == (3,) and group['map_params'].shape == (3,)
    dumper.append_attributes(extra=1)
    dumper.append_datasets(extra=np.arange(3))
    dumper.flush()
    assert group.attrs['extra'] == 1 and group['extra'].tolist() == [0, 1, 2]
    quiet = sampler.Dumper(sampler.MemoryGroup(), no_dump=True)
    sampler.run_multinest(runner, quiet, nlive=60, seed=4)
    assert not quiet.group.attrs and not quiet.group.datasets
    with pytest.raises(AssertionError):
        sampler.run_multinest(runner, dumper, nlive=0)
    with pytest.raises(AssertionError):
        sampler.run_multinest(runner, dumper, efr=1.5)
    with pytest.raises(ValueError, match='clustering parameters'):
        sampler.run_multinest(runner, dumper, nClsPar=4)


# ---------------------------------------------------------------------------- GPU
def _cube(engine, nfo, n_pix, seed, ntot=None):
    """Small synthetic cube: NH3 (1,1)+(2,2), 1 component, 256 channels, varying amplitude."""
    from nestfit_amd.cube import CubeRunner
    rng = np.random.default_rng(seed)
    n, noise = 256, 0.15
    axes = [freq_axis(1, n), freq_axis(2, n)]
    truths = []
    data = np.empty((n_pix, 2 * n))
    cpu_runners = []
    ut = engine.get_irdc_priors(size=300, vsys=0.0)
    ps = nfo.PriorSet(ut.lower())
    for p in range(n_pix):
        th = np.array([rng.uniform(-1, 1), rng.uniform(10, 18), rng.uniform(4, 8), rng.uniform(14.2, 14.8),
                       rng.uniform(0.3, 0.8), 0.0])
        if ntot is not None:
            th[3] = ntot
        truths.append(th)
        specs = []
        for k, t in enumerate((1, 2)):
            s = nfo.AmmoniaSpectrum(axes[k], np.zeros(n), noise, t)
            nfo.amm_predict(s, th)
            d = s.get_spec() + rng.normal(0, noise, n)
            data[p, k * n:(k + 1) * n] = d
            specs.append(nfo.AmmoniaSpectrum(axes[k], d, noise, t))
        cpu_runners.append(nfo.AmmoniaRunner(specs, ps, ncomp=1))
    cube = CubeRunner(axes, (1, 2), data, np.full((n_pix, 2), noise), ut, ncomp=1)
    return cube, cpu_runners, np.array(truths), data, axes


@pytest.mark.gpu
def test_sampler_on_gpu_matches_the_same_sampler_on_the_oracle(engine, nfo):
    """Same seed, same proposals: in the bit-faithful table mode the likelihoods agree to ~1e-13, so
    every accept/reject decision and hence the whole run is the same; in fast mode (1e-7) the
    evidences still agree far inside their error."""
    n_pix = 3
    cube, cpu_runners, truths, _, _ = _cube(engine, nfo, n_pix, seed=21)

    def cpu_loglike(pix, U):
        out = np.empty(U.shape[0])
        for p in np.unique(pix):
            m = pix == p
            sub = U[m]
            out[m] = cpu_runners[p].loglikelihood_batch(sub)
            U[m] = sub
        return out

    # the constant-prior slot (orth) is not sampled: 5 of the 6 dimensions
    mask = cube.utrans.free_mask(1)
    assert mask.tolist() == [1, 1, 1, 1, 1, 0]
    kw = dict(nlive=60, tol=0.5, efr=0.3, seed=33, free_mask=mask)
    ref = sampler.run_nested(cpu_loglike, cube.ndim, n_pix, **kw)
    assert all((r.posterior[:, 5] == 0).all() for r in ref)       # theta of the constant slot
    try:
        engine.set_exp_mode('table')
        # host twin (numpy rounds, GPU likelihood) and device-resident sampler, same seed
        for device in (False, True):
            got = sampler.fit_pixels(cube, np.arange(n_pix), device=device, **kw)
            for g, r in zip(got, ref):
                assert g.n_iter == r.n_iter and g.n_evals == r.n_evals, (device, g.n_iter, r.n_iter)
                assert 0 <= g.rounds - r.rounds < 32           # the device looks up every 32 rounds (check_every)
                assert g.lnZ == pytest.approx(r.lnZ, rel=1e-10)
                np.testing.assert_allclose(g.posterior, r.posterior, rtol=1e-8, atol=1e-12)
        # constrained random walks from the first round on: twin and device take the same steps
        kww = dict(kw, method='walk', n_steps=7, maxiter=400)
        ref_w = sampler.run_nested(cpu_loglike, cube.ndim, n_pix, **kww)
        got_w = sampler.fit_pixels(cube, np.arange(n_pix), **kww)
        for g, r in zip(got_w, ref_w):
            assert g.n_iter == r.n_iter and g.n_evals == r.n_evals, (g.n_iter, r.n_iter, g.n_evals, r.n_evals)
            assert g.lnZ == pytest.approx(r.lnZ, rel=1e-10)
            np.testing.assert_allclose(g.posterior, r.posterior, rtol=1e-8, atol=1e-12)
        # 400 live points: 128 walkers per pixel, two per lane of the update wave (ns_walkers_for)
        kw2 = dict(kw, nlive=400, method='walk', n_steps=5, maxiter=500)
        ref_2 = sampler.run_nested(cpu_loglike, cube.ndim, n_pix, **kw2)
        got_2 = sampler.fit_pixels(cube, np.arange(n_pix), **kw2)
        for g, r in zip(got_2, ref_2):
            assert g.n_iter == r.n_iter == 500 and g.n_evals == r.n_evals, (g.n_iter, r.n_iter, g.n_evals, r.n_evals)
            assert g.lnZ == pytest.approx(r.lnZ, rel=1e-10)
        # every pixel its own number of live points, one lock-step group (nfa_sampler_set_pixel_nlive): device = twin,
        # rejection rounds and walks alike
        for extra in (dict(), dict(method='walk', n_steps=7, maxiter=500)):
            kwn = dict(kw, nlive=np.array([60, 71, 83]), **extra)
            ref_n = sampler.run_nested(cpu_loglike, cube.ndim, n_pix, **kwn)
            got_n = sampler.fit_pixels(cube, np.arange(n_pix), **kwn)
            for g, r, n in zip(got_n, ref_n, (60, 71, 83)):
                assert g.n_live == r.n_live == n
                assert g.n_iter == r.n_iter and g.n_evals == r.n_evals, (n, g.n_iter, r.n_iter, g.n_evals, r.n_evals)
                assert g.lnZ == pytest.approx(r.lnZ, rel=1e-10)
                np.testing.assert_allclose(g.posterior, r.posterior, rtol=1e-8, atol=1e-12)
        engine.set_exp_mode('fast')
        fast = sampler.fit_pixels(cube, np.arange(n_pix), **kw)
        for g, r in zip(fast, ref):
            assert abs(g.lnZ - r.lnZ) < 0.5 * r.lnZ_err + 1e-3
    finally:
        engine.set_exp_mode('fast')
    null = cube.null_lnZ
    for p, r in enumerate(ref):
        assert r.lnZ - null[p] > 11                                   # main.py:464-469 threshold: a detection
        mean, sig = r.param_constr[0], r.param_constr[1]
        assert abs(mean[0] - truths[p, 0]) < 5 * sig[0] + 0.02        # velocity recovered


@pytest.mark.gpu
def test_run_multinest_front_end_on_the_engine(engine, nfo):
    cube, cpu_runners, truths, data, axes = _cube(engine, nfo, 1, seed=5)
    n = 256
    spec_data = [[axes[k], data[0, k * n:(k + 1) * n], 0.15, t] for k, t in enumerate((1, 2))]
    ut = engine.get_irdc_priors(size=300, vsys=0.0)
    runner = engine.AmmoniaRunner.from_data(spec_data, ut, ncomp=1)
    group = sampler.MemoryGroup()
    res = sampler.run_multinest(runner, sampler.Dumper(group), nlive=60, seed=2)
    assert group.attrs['ncomp'] == 1 and group.attrs['n_params'] == 6
    assert group.attrs['global_lnZ'] == res.lnZ == runner.run_lnZ
    assert res.lnZ - runner.null_lnZ > 11
    assert group['posteriors'].shape[1] == 8


@pytest.mark.gpu
def test_device_sampler_limits_and_errors(engine, nfo):
    from nestfit_amd import _ffi
    cube, cpu_runners, truths, data, axes = _cube(engine, nfo, 4, seed=8)
    pix = np.arange(4)
    capped = sampler.run_nested_device(cube, pix, nlive=40, maxiter=25, seed=3)
    assert all(r.n_iter == 25 and r.n_samples == 65 for r in capped)
    none = sampler.run_nested_device(cube, pix, nlive=40, maxiter=0, seed=3)
    assert all(r.n_iter == 0 and r.n_samples == 40 for r in none)
    full = sampler.run_nested_device(cube, pix, nlive=40, cap_iter=30, seed=3)     # dead-point buffer full
    assert all(r.n_iter == 30 for r in full)
    twin = sampler.fit_pixels(cube, pix, nlive=40, cap_iter=30, seed=3, device=False)
    for a, b in zip(full, twin):
        assert a.n_evals == b.n_evals and a.lnZ == pytest.approx(b.lnZ, rel=1e-10)
    # a subset and a permutation of pixels: every pixel's stream depends on its slot, not on company
    sub = sampler.run_nested_device(cube, np.array([2, 0]), nlive=40, maxiter=25, seed=3)
    assert sub[0].lnZ != capped[2].lnZ                       # slot 0 stream on pixel 2: a different run
    with pytest.raises(engine.EngineError, match='nlive'):
        sampler.run_nested_device(cube, pix, nlive=cube.ndim + 2 + 9000, seed=1)
    with pytest.raises(engine.EngineError, match='pixel index'):
        sampler.run_nested_device(cube, np.array([7]), nlive=40, seed=1)


@pytest.mark.gpu
def test_mmodal_on_the_device(engine, nfo):
    """One bounding ellipsoid (mmodal = False) against up to four (the default with five sampled dimensions) on a faint
    pixel, whose posterior is a curved ridge: the same evidence within the errors for fewer evaluations;
    device = twin for both."""
    cube, cpu_runners, truths, _, _ = _cube(engine, nfo, 2, seed=5, ntot=14.0)
    mask = cube.utrans.free_mask(1)
    kw = dict(nlive=200, tol=0.5, efr=0.3, seed=3, free_mask=mask, method='reject')
    try:
        engine.set_exp_mode('table')

        def cpu_loglike(pix, U):
            out = np.empty(U.shape[0])
            for p in np.unique(pix):
                m = pix == p
                sub = U[m]
                out[m] = cpu_runners[p].loglikelihood_batch(sub)
                U[m] = sub
            return out

        res = {}
        for ell in (1, 4):
            dev = sampler.fit_pixels(cube, np.arange(2), ellipsoids=ell, **kw)
            twin = sampler.run_nested(cpu_loglike, cube.ndim, 2, ellipsoids=ell, **kw)
            for g, r in zip(dev, twin):
                assert g.n_iter == r.n_iter and g.n_evals == r.n_evals, (ell, g.n_iter, r.n_iter, g.n_evals, r.n_evals)
                assert g.lnZ == pytest.approx(r.lnZ, rel=1e-10)
            res[ell] = dev
        for a, b in zip(res[1], res[4]):
            assert abs(a.lnZ - b.lnZ) < 4 * np.hypot(a.lnZ_err, b.lnZ_err)
        # (200 live points leave room for few cuts -- a cluster below 28 points is not cut: the 400 of config 5 gain
        # a factor 2.2 on its faint pixels, profiles/r03/sweep_ellipsoids.txt)
        assert sum(b.n_evals for b in res[4]) < 0.95 * sum(a.n_evals for a in res[1]), [(a.n_evals, b.n_evals) for a, b in zip(res[1], res[4])]
    finally:
        engine.set_exp_mode('fast')


@pytest.mark.gpu
def test_device_refit_paths_large_nlive(engine, nfo):
    """The ellipsoid refit stages the live points in LDS when they fit (nlive * ndim * 8 <= 96 KB) and
    falls back to wave reductions over global memory otherwise: both against the numpy twin."""
    cube, cpu_runners, truths, _, _ = _cube(engine, nfo, 1, seed=3)
    mask = cube.utrans.free_mask(1)

    def cpu_loglike(pix, U):
        return cpu_runners[0].loglikelihood_batch(U)
    try:
        engine.set_exp_mode('table')
        for nlive in (1500, 2600):                    # 5 sampled dimensions: 60 KB staged / 104 KB not
            kw = dict(nlive=nlive, seed=12, maxiter=nlive // 2, free_mask=mask, batch_target=4096)
            ref = sampler.run_nested(cpu_loglike, cube.ndim, 1, **kw)[0]
            got = sampler.fit_pixels(cube, np.zeros(1, dtype=np.int32), **kw)[0]
            assert got.n_iter == ref.n_iter == nlive // 2 and got.n_evals == ref.n_evals
            assert got.lnZ == pytest.approx(ref.lnZ, rel=1e-10)
            np.testing.assert_allclose(got.posterior, ref.posterior, rtol=1e-8, atol=1e-12)
    finally:
        engine.set_exp_mode('fast')


@pytest.mark.gpu
def test_sampler_on_sibling_models(engine, nfo):
    """The device sampler is model-agnostic: N2H+ (4 parameters) and Gaussian (3) cubes."""
    from scipy import stats
    from nestfit_amd.cube import CubeRunner
    CKMS = 299792.458

    def priors(ranges, size=200):
        x = np.linspace(0, 1, size)
        return engine.PriorTransformer([
            engine.Prior(engine.Distribution(lo + x * (hi - lo), stats.uniform(lo, hi - lo).pdf(lo + x * (hi - lo))), k)
            for k, (lo, hi) in enumerate(ranges)])
    rng = np.random.default_rng(4)
    n, n_pix, noise = 256, 3, 0.1
    # N2H+ 1-0
    nu0 = 93173.7637e6
    x = nu0 * (1.0 - np.linspace(15, -15, n) / CKMS)
    truth = np.array([0.5, 7.0, 0.3, 0.4])
    sc = nfo.DiazenyliumSpectrum(x, np.zeros(n), noise, 1)
    nfo.nnhp_predict(sc, truth)
    data = sc.get_spec()[None, :] + rng.normal(0, noise, (n_pix, n))
    ut = priors([(-4, 4), (2.8, 20), (-1.5, 1.0), (0.1, 1.5)])
    cube = CubeRunner([x], [1], data, np.full((n_pix, 1), noise), ut, ncomp=1, model=1)
    res = sampler.fit_pixels(cube, np.arange(n_pix), nlive=80, seed=6)
    for p, r in enumerate(res):
        assert r.lnZ - cube.null_lnZ[p] > 11 and r.posterior.shape[1] == 6
        mean, sig = r.param_constr[0], r.param_constr[1]
        assert abs(mean[0] - truth[0]) < 5 * sig[0] + 0.02 and abs(mean[3] - truth[3]) < 5 * sig[3] + 0.02
    # Gaussian line
    nu0 = 110.201354e9
    x = nu0 * (1.0 - np.linspace(20, -20, n) / CKMS)
    sg = nfo.Spectrum(x, np.zeros(n), noise, rest_freq=nu0)
    nfo.gauss_predict(sg, np.array([-2.0, 1.2, 1.5]))
    data = sg.get_spec()[None, :] + rng.normal(0, noise, (n_pix, n))
    utg = priors([(-15, 15), (0.2, 3.0), (0.0, 5.0)])
    cube = CubeRunner([x], [1], data, np.full((n_pix, 1), noise), utg, ncomp=1, model=2, rest_freqs=[nu0])
    res = sampler.fit_pixels(cube, np.arange(n_pix), nlive=80, seed=7)
    for p, r in enumerate(res):
        assert r.lnZ - cube.null_lnZ[p] > 11
        np.testing.assert_allclose(r.param_constr[0], [-2.0, 1.2, 1.5], atol=0.15)


def test_marginals_are_numpy_quantiles_bit_for_bit():
    """Dumper.calc_marginals sorts once and interpolates like numpy's default method: the values np.quantile gives
    (what the reference stores, core.pyx:596-598), to the last bit -- ties, tiny and odd sample counts, NaN columns."""
    from nestfit_amd import sampler
    d = sampler.Dumper(sampler.MemoryGroup())
    rng = np.random.default_rng(17)
    for n in (1, 2, 3, 16, 401, 2999, 3000):
        for n_par in (3, 6, 12):
            post = rng.normal(size=(n, n_par + 2))
            if n > 3:
                post[rng.integers(0, n, 7)] = post[0]
            want = np.quantile(post[:, :n_par], d.quantiles, axis=0)
            got = d.calc_marginals(post)
            assert got.shape == (15, n_par) and np.array_equal(got, want), (n, n_par)
            got32 = d.calc_marginals(post.astype(np.float32))
            assert np.array_equal(got32, np.quantile(post.astype(np.float32)[:, :n_par].astype(np.float64), d.quantiles, axis=0))
    post = rng.normal(size=(50, 8))
    post[3, 2] = np.nan
    assert np.array_equal(d.calc_marginals(post), np.quantile(post[:, :6], d.quantiles, axis=0), equal_nan=True)


def test_box_vetoes_of_a_one_ellipsoid_bound():
    """frames=K: proposals outside the bounding boxes of the live points (unit-cube axes, the ellipsoid's frame, K
    fixed rotations of it) are dropped before they are evaluated.  The frames are orthogonal; on a region no ellipsoid
    bounds well -- a flat likelihood inside a small cube, eight dimensions -- the evidence stays inside its error and a
    third of the evaluations go; on a Gaussian, which the ellipsoid bounds as well as anything, nothing changes."""
    Q = sampler._frames(8, 5)
    for k in range(5):
        np.testing.assert_allclose(Q[k].T @ Q[k], np.eye(8), atol=1e-12)
    assert not np.allclose(Q[0], Q[1])
    D, half = 8, 0.11

    def cube_like(pix, U):      # a gentle slope inside |u - 0.5| < half in every coordinate (no plateau), a steep wall outside
        d = np.maximum(np.abs(U - 0.5).max(axis=1) - half, 0.0)
        return -0.5 * (d / 0.004) ** 2 - 0.5 * (((U - 0.5) / 0.2) ** 2).sum(axis=1)
    kw = dict(nlive=200, tol=0.1, efr=0.3, seed=11, method='reject', batch_target=1024)
    plain = sampler.run_nested(cube_like, D, 2, **kw)
    boxed = sampler.run_nested(cube_like, D, 2, frames=16, **kw)
    # the wall is steep, the slope gentle: Z is close to the integral of the Gaussian slope over the cube
    from math import erf, log, pi, sqrt
    truth = D * log(0.2 * sqrt(2 * pi) * erf(half / (0.2 * sqrt(2))))
    for r in plain + boxed:
        assert abs(r.lnZ - truth) < 4 * r.lnZ_err + 0.25, (r.lnZ, truth, r.lnZ_err)       # (the soft wall adds ~ D * 0.03)
    assert sum(r.n_evals for r in boxed) < 0.67 * sum(r.n_evals for r in plain)
    f = _gauss_problem(np.full(D, 0.5), 0.05)
    # (every pixel the round's share of proposals, so that the two runs draw from the same places of the random stream)
    g0 = sampler.run_nested(f, D, 1, k_target=0, **kw)[0]
    g1 = sampler.run_nested(f, D, 1, frames=16, k_target=0, **kw)[0]
    assert abs(g1.n_iter - g0.n_iter) < 0.01 * g0.n_iter and abs(g1.n_evals - g0.n_evals) < 0.05 * g0.n_evals and abs(g1.lnZ - g0.lnZ) < 0.1


def test_shear_in_front_of_the_ellipsoid():
    """shear=e: the bound is an ellipsoid around the live points AFTER a volume-preserving polynomial shear.  The map and
    its inverse are each other's; one Cholesky factorisation of the monomials' Gram matrix gives every coordinate's
    least-squares coefficients; a bent ridge comes out straight; and on a ten-dimensional likelihood whose ridge is a
    parabola the sampler needs a fraction of the evaluations for the same evidence."""
    rng = np.random.default_rng(5)
    comp = np.arange(10) % 2
    mono, start = sampler._shear_monomials(comp)
    assert mono.shape == (36, 2) and list(start) == [1, 3, 5, 8, 11, 15, 19, 24, 29, 35]
    assert all(tuple(mono[start[j]]) == (j, -1) for j in range(10))
    U = rng.uniform(0.3, 0.7, size=(400, 10))
    U[:, 6] = 0.5 + 8.0 * (U[:, 4] - 0.5) ** 2 + 0.004 * rng.normal(size=400)          # coordinate 6 bends with coordinate 4 (same component)
    mu, sg, beta = sampler._fit_shear(U, mono, start)
    W = sampler._shear_fwd(U, mu, sg, beta, mono, start)
    np.testing.assert_allclose(sampler._shear_inv(W, mu, sg, beta, mono, start), U, atol=1e-13)
    Z = (U - mu) / sg
    for j in (3, 6, 9):                                                                 # = ridge regression, coordinate by coordinate
        F = sampler._shear_phi(Z, mono, start[j])
        ref = np.linalg.solve(F.T @ F + sampler._NS_SHEAR_RIDGE * 400 * np.eye(start[j]), F.T @ Z[:, j])
        np.testing.assert_allclose(beta[j, :start[j]], ref, atol=1e-9)
    assert W[:, 6].std() < 0.05 and abs(W[:, 4].std() - 1.0) < 0.05                     # the bend is gone, the rest untouched
    # a unit Jacobian: the sheared image of a box has the box's volume (Monte Carlo over the unit cube)
    X = rng.uniform(size=(200000, 10))
    Wx = sampler._shear_fwd(X, mu, sg, beta, mono, start)
    lo, hi = np.quantile(W, 0.1, axis=0), np.quantile(W, 0.9, axis=0)
    inside = np.all((Wx[:, [4, 6]] >= lo[[4, 6]]) & (Wx[:, [4, 6]] <= hi[[4, 6]]), axis=1).mean()
    assert inside == pytest.approx(np.prod((hi - lo)[[4, 6]] * sg[[4, 6]]), rel=0.05)

    def ridge(pix, T):            # 12 slots, slots 10 and 11 dummies; slot 6 follows a parabola in slot 4, slot 7 in slot 5
        U = T[:, :10] - 0.5
        d = U.copy()
        d[:, 6] -= 3.0 * U[:, 4] ** 2 - 0.05
        d[:, 7] += 3.0 * U[:, 5] ** 2 - 0.05
        s = np.full(10, 0.08)
        s[[6, 7]] = 0.01
        return -0.5 * ((d / s) ** 2).sum(axis=1)
    fm = np.array([1] * 10 + [0, 0])
    kw = dict(nlive=200, tol=0.5, efr=0.3, seed=3, method='reject', batch_target=512, free_mask=fm)
    plain = sampler.run_nested(ridge, 12, 1, shear=0, **kw)
    bent = sampler.run_nested(ridge, 12, 1, shear=4.0, frames=-1, **kw)
    both = sampler.run_nested(ridge, 12, 1, **kw)                  # the default of this shape: shear 2.5, 32 box frames, pair ellipses
    truth = float(np.sum(np.log(np.sqrt(2 * np.pi) * np.array([0.08] * 8 + [0.01] * 2))))     # (the Gaussians fit into the cube)
    for r in plain + bent + both:
        assert abs(r.lnZ - truth) < 4 * r.lnZ_err + 0.3, (r.lnZ, truth, r.lnZ_err)
    assert sum(r.n_evals for r in bent) < 0.5 * sum(r.n_evals for r in plain)
    assert sum(r.n_evals for r in both) < sum(r.n_evals for r in bent)
    # the pair ellipses: every live point inside every one of them, a point half as far again outside some
    Wp = rng.normal(size=(300, 10)) @ rng.normal(size=(10, 10))
    pt = sampler._fit_pairs(Wp, 1.75)
    assert pt.shape == (45, 5) and sampler._pair_veto(Wp, pt).all() and not sampler._pair_veto(1.6 * (Wp - Wp.mean(axis=0)) + Wp.mean(axis=0), pt).all()
    fewer = sampler.run_nested(ridge, 12, 1, pairs=1.75, **kw)
    none = sampler.run_nested(ridge, 12, 1, pairs=0, **kw)
    assert fewer[0].n_evals < none[0].n_evals and abs(fewer[0].lnZ - truth) < 4 * fewer[0].lnZ_err + 0.3
    # shapes the device has no shear for run without it: the same result as shear=None
    f5 = _gauss_problem(np.full(5, 0.5), 0.05)
    a5, b5 = sampler.run_nested(f5, 5, 1, nlive=100, seed=2, ellipsoids=1)[0], sampler.run_nested(f5, 5, 1, nlive=100, seed=2, ellipsoids=1, shear=4.0)[0]
    assert (a5.n_iter, a5.n_evals, a5.lnZ) == (b5.n_iter, b5.n_evals, b5.lnZ)


@pytest.mark.gpu
def test_shear_on_the_device_follows_the_twin(engine, nfo):
    """Two velocity components, shear on (with and without boxes): device and twin take the same decisions."""
    from nestfit_amd.cube import CubeRunner
    n_pix, n, noise = 3, 128, 0.1
    rng = np.random.default_rng(3)
    axes = [freq_axis(1, n), freq_axis(2, n)]
    ut = engine.get_irdc_priors(size=500, vsys=0.0)
    truths = np.tile(np.array([-0.5, 1.0, 12.0, 15.0, 5.0, 6.0, 14.4, 14.6, 0.4, 0.4, 0.0, 0.0]), (n_pix, 1))
    truths[:, 6] += np.array([0.0, -0.4, 0.2])
    try:
        engine.set_exp_mode('table')
        probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=2)
        model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
        cube = CubeRunner(axes, (1, 2), model + rng.normal(0, noise, model.shape), np.full((n_pix, 2), noise), ut, ncomp=2)
        for extra in (dict(shear=4.0, method='reject', maxiter=2400), dict(shear=4.0, frames=32, maxiter=1800), dict(shear=2.5, frames=8, n_steps=20)):
            kw = dict(nlive=150, tol=0.5, efr=0.3, seed=7, batch_target=2048, **{'maxiter': 900, **extra})
            dev = sampler.fit_pixels(cube, np.arange(n_pix), device=True, **kw)
            twin = sampler.fit_pixels(cube, np.arange(n_pix), device=False, **kw)
            for d, t in zip(dev, twin):
                assert (d.n_iter, d.n_evals) == (t.n_iter, t.n_evals), (extra, d.n_iter, t.n_iter, d.n_evals, t.n_evals)
                assert d.lnZ == pytest.approx(t.lnZ, rel=1e-10)
        kw = dict(nlive=150, tol=0.5, efr=0.3, seed=7, maxiter=2400, batch_target=2048, method='reject')
        with_shear = sampler.fit_pixels(cube, np.arange(n_pix), shear=4.0, frames=-1, **kw)
        without = sampler.fit_pixels(cube, np.arange(n_pix), shear=0, **kw)
        assert sum(r.n_evals for r in with_shear) < 0.8 * sum(r.n_evals for r in without)
    finally:
        engine.set_exp_mode('fast')


@pytest.mark.gpu
def test_live_points_that_do_not_fit_in_lds(engine, nfo):
    """1300 live points in ten dimensions are 104 KB: the refit workgroup reads them from global memory (block-wide sums
    instead of the staged passes), shear and boxes stay off -- device and twin alike."""
    from nestfit_amd.cube import CubeRunner
    n_pix, n, noise = 2, 128, 0.1
    rng = np.random.default_rng(6)
    axes = [freq_axis(1, n), freq_axis(2, n)]
    ut = engine.get_irdc_priors(size=500, vsys=0.0)
    truths = np.tile(np.array([-0.5, 1.0, 12.0, 15.0, 5.0, 6.0, 14.4, 14.6, 0.4, 0.4, 0.0, 0.0]), (n_pix, 1))
    try:
        engine.set_exp_mode('table')
        probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=2)
        model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
        cube = CubeRunner(axes, (1, 2), model + rng.normal(0, noise, model.shape), np.full((n_pix, 2), noise), ut, ncomp=2)
        kw = dict(nlive=1300, tol=0.5, efr=0.3, seed=3, maxiter=14000, batch_target=4096, method='reject')      # (well past the rounds in which the unit cube is the bound)
        dev = sampler.fit_pixels(cube, np.arange(n_pix), device=True, **kw)
        twin = sampler.fit_pixels(cube, np.arange(n_pix), device=False, **kw)
        for d, t in zip(dev, twin):
            assert (d.n_iter, d.n_evals) == (t.n_iter, t.n_evals), (d.n_iter, t.n_iter, d.n_evals, t.n_evals)
            assert d.lnZ == pytest.approx(t.lnZ, rel=1e-10)
    finally:
        engine.set_exp_mode('fast')


@pytest.mark.gpu
def test_shear_with_three_components_follows_the_twin(engine, nfo):
    """Fifteen sampled dimensions (three velocity components): the defaults -- shear, boxes, a pixel's own share of
    proposals -- on the device and in the twin, decision for decision."""
    from nestfit_amd.cube import CubeRunner
    n_pix, n, noise = 2, 128, 0.1
    rng = np.random.default_rng(4)
    axes = [freq_axis(1, n), freq_axis(2, n)]
    ut = engine.get_irdc_priors(size=500, vsys=0.0)
    assert ut.free_mask(3).tolist() == [1] * 15 + [0] * 3
    truths = np.tile(np.array([-1.5, 0.0, 1.6, 12.0, 14.0, 15.0, 5.0, 5.5, 6.0, 14.4, 14.5, 14.6, 0.35, 0.4, 0.35, 0.0, 0.0, 0.0]), (n_pix, 1))
    truths[1, 9:12] -= 0.3
    try:
        engine.set_exp_mode('table')
        probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=3)
        model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
        cube = CubeRunner(axes, (1, 2), model + rng.normal(0, noise, model.shape), np.full((n_pix, 2), noise), ut, ncomp=3)
        for extra in (dict(method='reject', maxiter=2000),):        # (the twin's fifteen-dimensional rounds are slow in Python)
            kw = dict(nlive=200, tol=0.5, efr=0.3, seed=11, batch_target=2048, **extra)
            dev = sampler.fit_pixels(cube, np.arange(n_pix), device=True, **kw)
            twin = sampler.fit_pixels(cube, np.arange(n_pix), device=False, **kw)
            for d, t in zip(dev, twin):
                assert (d.n_iter, d.n_evals) == (t.n_iter, t.n_evals), (extra, d.n_iter, t.n_iter, d.n_evals, t.n_evals)
                assert d.lnZ == pytest.approx(t.lnZ, rel=1e-10)
        kw = dict(nlive=200, tol=0.5, efr=0.3, seed=11, batch_target=2048, method='reject', maxiter=2600)
        plain = sampler.fit_pixels(cube, np.arange(n_pix), shear=0, frames=-1, **kw)
        assert sum(r.n_evals for r in dev) > 0 and sum(r.n_evals for r in sampler.fit_pixels(cube, np.arange(n_pix), **kw)) < 0.8 * sum(r.n_evals for r in plain)
    finally:
        engine.set_exp_mode('fast')


@pytest.mark.gpu
def test_box_vetoes_on_the_device_follow_the_twin(engine, nfo):
    """Two velocity components (ten sampled dimensions), boxes on: the device sampler and the numpy twin fed by the
    GPU's table-mode likelihood take the same decisions -- iteration and evaluation counts equal, lnZ to 1e-10 -- in
    rejection rounds with vetoes, through the switch to walks and back."""
    from nestfit_amd.cube import CubeRunner
    n_pix, n, noise = 3, 128, 0.1
    rng = np.random.default_rng(3)
    axes = [freq_axis(1, n), freq_axis(2, n)]
    ut = engine.get_irdc_priors(size=500, vsys=0.0)
    truths = np.tile(np.array([-0.5, 1.0, 12.0, 15.0, 5.0, 6.0, 14.4, 14.6, 0.4, 0.4, 0.0, 0.0]), (n_pix, 1))
    truths[:, 6] += np.array([0.0, -0.4, 0.2])
    try:
        engine.set_exp_mode('table')
        probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=2)
        model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
        cube = CubeRunner(axes, (1, 2), model + rng.normal(0, noise, model.shape), np.full((n_pix, 2), noise), ut, ncomp=2)
        for extra in (dict(frames=8), dict(frames=32, margin=1.5, n_steps=20), dict(frames=0, method='reject')):
            kw = dict(nlive=150, tol=0.5, efr=0.3, seed=7, maxiter=900, batch_target=2048, shear=0, **extra)
            dev = sampler.fit_pixels(cube, np.arange(n_pix), device=True, **kw)
            twin = sampler.fit_pixels(cube, np.arange(n_pix), device=False, **kw)
            for d, t in zip(dev, twin):
                assert (d.n_iter, d.n_evals) == (t.n_iter, t.n_evals), (extra, d.n_iter, t.n_iter, d.n_evals, t.n_evals)
                assert d.lnZ == pytest.approx(t.lnZ, rel=1e-10)
        # and the vetoes do veto: fewer evaluations than the same run without them
        kw = dict(nlive=150, tol=0.5, efr=0.3, seed=7, maxiter=900, batch_target=2048, method='reject', shear=0)
        with_boxes = sampler.fit_pixels(cube, np.arange(n_pix), frames=32, **kw)
        without = sampler.fit_pixels(cube, np.arange(n_pix), **kw)
        assert sum(r.n_evals for r in with_boxes) < 0.8 * sum(r.n_evals for r in without)
    finally:
        engine.set_exp_mode('fast')
