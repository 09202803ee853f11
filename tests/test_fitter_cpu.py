"""The cube driver (nestfit_amd/fitter.py, reference nestfit/main.py:380-526) without a GPU: its
`fit_backend` hook is given the numpy twin of the sampler fed by the CPU oracle, so the component
-count rule, the store layout and the N > 1 path (one process per stripe, gloo, world size 2) run
here."""
import socket

import numpy as np
import pytest

from nestfit_amd import sampler
from nestfit_amd.cubeio import CubeStack, DataCube, SimpleCube
from nestfit_amd.store import HdfStore
from nestfit_amd.synth import freq_axis

N_CHAN, NOISE = 96, 0.1


def _oracle_backend(fitter, lon, lat, ncomp, nlive, kw):
    from oracle import nfo
    ps = nfo.PriorSet(fitter.utrans.lower())
    runners = []
    for i, j in zip(lon, lat):
        spec_data, has_nans = fitter.stack.get_spec_data(i, j)
        assert not has_nans
        runners.append(nfo.AmmoniaRunner([nfo.AmmoniaSpectrum(*sd) for sd in spec_data], ps, ncomp=ncomp))

    def loglike(pix, U):
        out = np.empty(U.shape[0])
        for q in np.unique(pix):
            m = pix == q
            sub = U[m]
            out[m] = runners[q].loglikelihood_batch(sub)
            U[m] = sub
        return out
    res = sampler.run_nested(loglike, 6 * ncomp, len(runners), nlive=nlive, batch_target=64, **kw)
    return res, np.array([r.null_lnZ for r in runners]), sum(r.n_chan_tot for r in runners[:1])


def _stack(n_lon=4, n_lat=1, seed=0):
    """Synthetic (1,1)+(2,2) cubes: a line in the pixels with even i_lon, noise only elsewhere, one
    NaN pixel."""
    from oracle import nfo
    rng = np.random.default_rng(seed)
    cubes = []
    for t in (1, 2):
        x = freq_axis(t, N_CHAN, 12.0)
        data = rng.normal(0, NOISE, (N_CHAN, n_lat, n_lon))
        s = nfo.AmmoniaSpectrum(x, np.zeros(N_CHAN), NOISE, t)
        nfo.amm_predict(s, np.array([0.3, 14.0, 6.0, 14.7, 0.5, 0.0]))
        for i in range(0, n_lon, 2):
            data[:, :, i] += s.get_spec()[:, None]
        if t == 1:
            data[5, 0, n_lon - 1] = np.nan
        hdr = {'SIMPLE': True, 'BITPIX': -64, 'NAXIS': 3, 'NAXIS1': n_lon, 'NAXIS2': n_lat, 'NAXIS3': N_CHAN,
               'BUNIT': 'K', 'CTYPE1': 'RA---SIN', 'CTYPE2': 'DEC--SIN', 'CTYPE3': 'FREQ', 'CUNIT3': 'Hz',
               'CRVAL3': float(x[0]), 'CDELT3': float(x[1] - x[0]), 'CRPIX3': 1.0, 'RESTFRQ': float(x.mean())}
        cubes.append(DataCube(SimpleCube(hdr, data), NOISE, trans_id=t))
    return CubeStack(cubes)


def _fitter(stack):
    import nestfit_amd as na
    from nestfit_amd.fitter import CubeFitter
    ut = na.get_irdc_priors(size=200, vsys=0.0)
    return CubeFitter(stack, ut, na.AmmoniaRunner, lnZ_thresh=11, ncomp_max=2,
                      mn_kwargs={'nlive': 24, 'tol': 1.0, 'seed': 3, 'maxiter': 250}, nlive_snr_fact=0,
                      nlive_quantum=1, fit_backend=_oracle_backend)


def _check_store(path, n_lon):
    with HdfStore(path) as store:
        groups = {(g.attrs['i_lon'], g.attrs['i_lat']): g for g in store.iter_pix_groups()}
        assert sorted(groups) == [(i, 0) for i in range(n_lon - 1)]          # the NaN pixel is skipped
        for (i, _), g in groups.items():
            one = g['1']
            gain = one.attrs['global_lnZ'] - one.attrs['null_lnZ']
            if i % 2 == 0:                                                   # a line: detected, N = 2 tried
                assert g.attrs['nbest'] >= 1 and gain >= 11 and '2' in g
                assert g['2'].attrs['n_params'] == 12 and g['2']['posteriors'].shape[1] == 14
            else:                                                            # noise: rejected at N = 1
                assert g.attrs['nbest'] == 0 and gain < 11 and '2' not in g
            assert one.attrs['n_chan_tot'] == 2 * N_CHAN and one['marginals'].shape == (15, 6)
        assert store.hdf.attrs['n_max_components'] == 2 and store.hdf.attrs['model_name'] == 'ammonia'
        assert store.hdf.attrs['naxis1'] == n_lon


def test_fit_cube_single_process(tmp_path, capsys):
    stack = _stack()
    _fitter(stack).fit_cube(str(tmp_path / 'run'), nproc=1)
    assert 'SKIP: has NaN values' in capsys.readouterr().out
    _check_store(str(tmp_path / 'run'), 4)
    with pytest.raises(ValueError, match='must be greater than or equal to the number of processes'):
        _fitter(stack).fit_cube(str(tmp_path / 'run2'), nproc=5)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, store_name):
    import torch.distributed as dist
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    _fitter(_stack()).fit_cube(store_name, nproc=world, rank=rank)        # stripe i_lon % world == rank
    dist.barrier()
    if rank == 0:
        with HdfStore(store_name) as store:
            store.link_files()
    dist.barrier()
    dist.destroy_process_group()


def test_fit_cube_one_process_per_stripe_gloo(tmp_path):
    import torch.multiprocessing as mp
    store_name = str(tmp_path / 'run')
    mp.spawn(_worker, args=(2, _free_port(), store_name), nprocs=2, join=True)
    _check_store(store_name, 4)
    from nestfit_amd.store import FILE_SUFFIXES, store_format
    sfx = FILE_SUFFIXES[store_format()]
    assert sorted(p.name for p in (tmp_path / 'run.store').iterdir()) == [f'chunk0{sfx}', f'chunk1{sfx}', f'table{sfx}']
    # same seed, same pixels, same slots within their stripe -> the two-process run equals the
    # one-process run stripe by stripe
    one = str(tmp_path / 'one')
    _fitter(_stack()).fit_cube(one, nproc=2)
    with HdfStore(store_name) as a, HdfStore(one) as b:
        for ga in a.iter_pix_groups():
            gb = b.hdf[ga.name]
            assert ga['1'].attrs['global_lnZ'] == gb['1'].attrs['global_lnZ']
            np.testing.assert_array_equal(ga['1']['posteriors'], gb['1']['posteriors'])


def test_masked_beam_pixel_gets_an_nbest_zero_group(tmp_path, capsys):
    """A pixel whose noise is infinite (masked primary beam, NoiseMap.from_pbimg) has NaN-free data: the
    reference does not skip it (main.py:437-441 looks for NaNs only) and its flat likelihood ends in
    nbest = 0.  The driver writes that group without sampling; truncated runs are flagged in the store."""
    from nestfit_amd.cubeio import NoiseMap
    stack = _stack(n_lon=3)
    pb = np.ones((1, 3))                         # FITS order (lat, lon)
    pb[0, 1] = np.nan                            # the beam response is masked at i_lon = 1
    nmap = NoiseMap.from_pbimg(NOISE, pb)
    assert np.isinf(nmap.get_noise(1, 0)) and nmap.get_noise(0, 0) == NOISE
    for dc in stack.cubes:
        dc.noise_map = nmap
    glon, glat = stack.good_pixels()
    blon, blat = stack.masked_beam_pixels()
    assert glon.tolist() == [0] and blon.tolist() == [1] and blat.tolist() == [0]     # i_lon = 2 holds the NaN
    _fitter(stack).fit_cube(str(tmp_path / 'run'), nproc=1)
    out = capsys.readouterr().out
    assert '(1, 0) infinite noise: nbest = 0 without sampling' in out and '(2, 0) SKIP: has NaN values' in out
    with HdfStore(str(tmp_path / 'run')) as store:
        groups = {(g.attrs['i_lon'], g.attrs['i_lat']): g for g in store.iter_pix_groups()}
        assert sorted(groups) == [(0, 0), (1, 0)]
        assert groups[(1, 0)].attrs['nbest'] == 0 and list(groups[(1, 0)]) == []
        one = groups[(0, 0)]['1']
        assert one.attrs['truncated'] in (True, False)
        assert one.attrs['n_samples'] == one['posteriors'].shape[0]


def test_truncated_flag_and_default_cap():
    """A run stopped by its iteration cap before the evidence tolerance is met says so; the host twin and
    the device sampler share the default cap."""
    def loglike(pix, U):
        return -0.5 * np.sum(((U - 0.5) / 0.01) ** 2, axis=1)
    short = sampler.run_nested(loglike, 2, 1, nlive=50, seed=1, maxiter=30)[0]
    full = sampler.run_nested(loglike, 2, 1, nlive=50, seed=1)[0]
    assert short.truncated and short.n_iter == 30
    assert not full.truncated and full.n_iter < sampler.default_cap_iter(50) == 3000
    names, q = sampler.marginal_quantile_table()
    assert names[:3] == ['min', 'p01', 'p10'] and names[-2:] == ['3s_lo', '3s_hi'] and len(names) == q.size == 15
    assert q[9] == 1.58655254e-1 and q[10] == 0.84134475 and q[13] == 1.34989803e-3 and q[14] == 0.99865010
    bic, aic, aicc = sampler.information_criteria(2048, 12, -1000.0)
    assert bic == pytest.approx(np.log(2048) * 12 + 2000) and aic == 2024.0 and aicc == pytest.approx(2024 + 312 / 2035)


def test_lock_step_runs_are_cut_by_the_spread_of_live_points():
    """One very bright pixel (nlive + 5 SNR) must not size the lock-step run of a whole stripe: pixels are cut into runs
    whose largest count is at most twice the smallest; the reference test cube's spread (100 .. 180) stays one run."""
    from nestfit_amd.fitter import _runs_within_a_factor
    nl = np.array([100, 180, 120, 400, 90, 1000, 95, 181])
    runs = _runs_within_a_factor(nl, 2.0)
    assert [r.tolist() for r in runs] == [[0, 1, 2, 4, 6], [7], [3], [5]]
    assert sorted(np.concatenate(runs).tolist()) == list(range(8))
    for r in runs:
        assert nl[r].max() <= 2 * nl[r].min()
    assert len(_runs_within_a_factor(np.arange(100, 181), 2.0)) == 1


@pytest.mark.parametrize('workers', [1, 3])
def test_every_pixel_of_a_run_gets_its_own_live_points(tmp_path, workers):
    """Pixels whose counts differ by more than a factor of two are fitted in several lock-step runs, side by side on
    worker threads on the device path: whichever branch submits them, a run's pixels must arrive with THEIR counts (the
    parallel branch once handed every run the whole stripe's array, so that all runs but the first took the first
    pixels' counts).  The device path is stood in for by a stub that records what it is given."""
    import nestfit_amd.fitter as fmod
    stack = _stack(n_lon=7)
    fit = _fitter(stack)
    fit.nlive_snr_fact = 5
    fit.fit_backend = None                                   # the device path's branch (threads), with the stub below
    fit.group_workers = workers
    want = {0: 24, 1: 500, 2: 30, 3: 40, 4: 1200, 5: 26}     # three runs: {24, 26, 30, 40}, {500}, {1200}
    fit._nlive = lambda lon, lat: np.array([want[int(i)] for i in lon], dtype=np.int64)
    seen = {}

    def stub(lon, lat, ncomp, nlive, kw):
        assert np.shape(nlive) == (lon.size,)
        for i, n in zip(lon.tolist(), np.asarray(nlive).tolist()):
            seen.setdefault(i, set()).add(n)
        return _oracle_backend(fit, lon, lat, ncomp, np.minimum(nlive, 24), kw)
    fit._fit_on_device = stub
    lon, lat = np.arange(7), np.zeros(7, dtype=int)
    fit.fit((lon, lat), tmp_path / 'chunk0.npz')
    assert seen == {i: {n} for i, n in want.items()}
    assert len(fmod._runs_within_a_factor(np.array(list(want.values())), 2.0)) == 3


def test_posterior_spread_survives_a_large_offset():
    """The posterior standard deviation of a parameter whose mean is 1e7 times its spread (raw second moments cancel
    there: round 4's form gave 0 or noise): moments are taken about the sample of the largest weight."""
    rng = np.random.default_rng(1)
    n = 4000
    t = 1.0e7 + rng.normal(0, 1.0, n)
    w = rng.uniform(0.5, 1.5, n)
    w /= w.sum()
    post = np.column_stack([t, 3.0 + 0.1 * rng.normal(size=n), np.zeros(n), w])
    r = sampler.NestedResult(post, 0.0, 0.1, 0.0, 400, n, n, 1.0)
    want = np.sqrt(w @ (post[:, :2] - w @ post[:, :2]) ** 2)
    np.testing.assert_allclose(r.param_constr[1], want, rtol=1e-6)
    np.testing.assert_allclose(r.param_constr[0], w @ post[:, :2], rtol=1e-14)
    # ... and the device's statistics (mean, second moment about the MAP row, MAP row) give the same through from_stats
    imap = int(np.argmax(w))
    c = post[imap, :2]
    stats = np.concatenate([[0.0, 0.0, 1.0, 0.0, 0.0, w.sum()], w @ post[:, :2], w @ (post[:, :2] - c) ** 2, post[0, :2], c])
    r2 = sampler.NestedResult.from_stats(post, stats, 400, n, n)
    np.testing.assert_allclose(r2.param_constr[1], want, rtol=1e-6)
