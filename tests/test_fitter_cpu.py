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
    assert sorted(p.name for p in (tmp_path / 'run.store').iterdir()) == ['chunk0.npz', 'chunk1.npz', 'table.npz']
    # same seed, same pixels, same slots within their stripe -> the two-process run equals the
    # one-process run stripe by stripe
    one = str(tmp_path / 'one')
    _fitter(_stack()).fit_cube(one, nproc=2)
    with HdfStore(store_name) as a, HdfStore(one) as b:
        for ga in a.iter_pix_groups():
            gb = b.hdf[ga.name]
            assert ga['1'].attrs['global_lnZ'] == gb['1'].attrs['global_lnZ']
            np.testing.assert_array_equal(ga['1']['posteriors'], gb['1']['posteriors'])
