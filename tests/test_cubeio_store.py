"""Cube ingestion, result store and cube driver: the reference's own tests
for DataCube / CubeStack (nestfit/test/test_main.py:12-71) on its own FITS cubes
(tests/golden/ammonia_*_cutout.fits = nestfit/test/data/), the store layout of
docs/store_spec.rst, and an end-to-end fit of real pixels on the GPU."""
import numpy as np
import pytest

from nestfit_amd import cubeio
from nestfit_amd.cubeio import CubeStack, DataCube, NoiseMap, NoiseMapUniform, SimpleCube
from nestfit_amd.store import Group, HdfStore, StoreFile

from conftest import ROOT

DATA_PATH = ROOT / 'tests' / 'golden'
NH3_RMS_K = 0.35                                  # nestfit/test/__init__.py:12
CKMS = 299792.458


def get_ammonia_cube(trans_id=1):
    """nestfit/test/__init__.py:15-27"""
    assert trans_id in (1, 2)
    transition = f'{trans_id}' * 2
    cube = SimpleCube.read(DATA_PATH / f'ammonia_{transition}_cutout.fits')
    cube = cube[:-1]  # last channel contains NaNs
    return cube


@pytest.fixture
def nmap():
    return NoiseMapUniform(NH3_RMS_K)


@pytest.fixture
def dcube(nmap):
    return DataCube(get_ammonia_cube(trans_id=1), nmap, trans_id=1)


@pytest.fixture
def stack():
    return CubeStack([
        DataCube(get_ammonia_cube(trans_id=1), NH3_RMS_K, trans_id=1),
        DataCube(get_ammonia_cube(trans_id=2), NH3_RMS_K, trans_id=2),
    ])


# ---- the reference's tests, verbatim in substance -------------------------------------------
def test_noise_map_uniform(nmap):
    rms = nmap.rms
    assert nmap.get_noise(1, 1) == rms
    assert nmap.shape is None


class TestDataCube:
    def test_read(self):
        cube = get_ammonia_cube(trans_id=1)
        assert DataCube(cube, NH3_RMS_K, trans_id=1)

    def test_properties(self, dcube):
        assert dcube.trans_id == 1
        assert dcube.dv
        assert dcube.shape == (20, 20, 379)
        assert dcube.spatial_shape == (20, 20)
        assert dcube.nchan == 379
        assert dcube.full_header
        assert dcube.simple_header
        xarr, arr, noise, trans_id, has_nans = dcube.get_spec_data(1, 1)
        assert not has_nans
        assert xarr[1] > xarr[0]  # ascending
        assert not np.any(np.isnan(arr))
        assert not np.isnan(noise)


class TestCubeStack:
    def test_properties(self, stack):
        assert stack.full_header
        assert stack.simple_header
        assert stack.shape == (20, 20, 379)
        assert stack.spatial_shape == (20, 20)

    def test_get_arrays(self, stack):
        all_spec_data, any_nans = stack.get_spec_data(1, 1)
        assert not any_nans
        assert all_spec_data

    def test_get_max_snr(self, stack):
        assert stack.get_max_snr(1, 1) > 0


# ---- what the reader itself must get right -----------------------------------------------------
def test_fits_header_and_axes(dcube):
    hdr, data = cubeio.read_fits(DATA_PATH / 'ammonia_11_cutout.fits')
    assert (hdr['BITPIX'], hdr['NAXIS'], hdr['NAXIS1'], hdr['NAXIS2'], hdr['NAXIS3']) == (-64, 3, 20, 20, 380)
    assert hdr['BUNIT'] == 'K' and hdr['CTYPE3'] == 'VRAD' and hdr['CUNIT3'] == 'm s-1'
    assert hdr['RESTFRQ'] == 23694495500.0 and hdr['CRVAL3'] == 33780.00000004
    assert hdr['TELESCOP'] == 'EVLA' and hdr['SIMPLE'] is True
    assert data.shape == (380, 20, 20) and not np.isnan(data[:-1]).any()
    # radio convention: nu = nu0 (1 - v/c); the reader flips the descending frequencies
    v = (hdr['CRVAL3'] + (np.arange(379) + 1 - hdr['CRPIX3']) * hdr['CDELT3']) * 1e-3
    nu = hdr['RESTFRQ'] * (1 - v / CKMS)
    np.testing.assert_allclose(dcube.xarr, nu[::-1], rtol=1e-15)
    np.testing.assert_allclose(dcube.varr, v[::-1], rtol=0, atol=1e-9)
    assert dcube.varr[0] > dcube.varr[1]
    assert dcube.dv == pytest.approx(0.1581330992788, rel=1e-12)
    # (s, b, l) -> (l, b, s), spectral axis reversed
    assert np.array_equal(dcube.data[3, 7, :], data[:-1, 7, 3][::-1])
    sh = dcube.simple_header
    assert sh['NAXIS'] == 2 and sh['WCSAXES'] == 2 and sh['CTYPE1'] == 'RA---SIN' and 'CRVAL3' not in sh


def test_fits_reader_other_encodings(tmp_path):
    """BITPIX 16 with BSCALE/BZERO, a FREQ axis in GHz, a degenerate Stokes axis, quoted strings."""
    def card(k, v):
        if isinstance(v, str):
            v = "'" + v.replace("'", "''").ljust(8) + "'"
            return f'{k:<8}= {v:<20}'.ljust(80)
        if isinstance(v, bool):
            v = 'T' if v else 'F'
        return f'{k:<8}= {str(v):>20}'.ljust(80)
    raw = np.arange(2 * 3 * 4, dtype='>i2').reshape(1, 2, 3, 4)
    cards = [card('SIMPLE', True), card('BITPIX', 16), card('NAXIS', 4), card('NAXIS1', 4), card('NAXIS2', 3),
             card('NAXIS3', 2), card('NAXIS4', 1), card('BSCALE', 0.5), card('BZERO', 10.0), card('BUNIT', 'K'),
             card('CTYPE3', 'FREQ'), card('CUNIT3', 'GHz'), card('CRVAL3', 23.7), card('CDELT3', -0.001),
             card('CRPIX3', 1.0), card('RESTFRQ', 23.6944955e9), card('OBSERVER', "O'Neil"),
             'COMMENT something'.ljust(80), 'END'.ljust(80)]
    blob = ''.join(cards).encode('ascii')
    blob += b' ' * (-len(blob) % 2880)
    body = raw.tobytes()
    body += b'\0' * (-len(body) % 2880)
    path = tmp_path / 'small.fits'
    path.write_bytes(blob + body)
    hdr, data = cubeio.read_fits(path)
    assert hdr['OBSERVER'] == "O'Neil" and 'COMMENT' not in hdr
    np.testing.assert_array_equal(data, raw.astype(float) * 0.5 + 10.0)
    cube = SimpleCube(hdr, data)
    assert cube.shape == (2, 3, 4)
    dc = DataCube(cube, 0.1, trans_id=1)
    np.testing.assert_allclose(dc.xarr, [23.699e9, 23.7e9])           # flipped to ascending
    assert dc.data.shape == (4, 3, 2) and dc.data[1, 2, 0] == data[0, 1, 2, 1]
    with pytest.raises(ValueError, match='only K'):
        DataCube(SimpleCube(dict(hdr, BUNIT='mJy/pixel'), data), 0.1)
    with pytest.raises(ValueError, match='BMAJ'):
        DataCube(SimpleCube(dict(hdr, BUNIT='Jy/beam'), data), 0.1)
    # Jy/beam -> K: 1.222e6 S / (nu_GHz^2 theta_maj theta_min [arcsec^2]) per channel
    hj = dict(hdr, BUNIT='Jy/beam', BMAJ=3.95 / 3600, BMIN=2.87 / 3600)
    dj = DataCube(SimpleCube(hj, data), 0.1)
    k = 1.222e6 / (np.array([23.699, 23.7]) ** 2 * 3.95 * 2.87)
    np.testing.assert_allclose(dj.data / dc.data, np.broadcast_to(k, dc.data.shape), rtol=5e-4)


def test_noise_map_from_pbimg_and_nan_pixels():
    cube = get_ammonia_cube(1)
    pb = np.ones((1, 1, 20, 20))
    pb[0, 0, 4, 2] = np.nan                       # (lat 4, lon 2) masked in the primary beam
    nm = NoiseMap.from_pbimg(NH3_RMS_K, pb)
    assert nm.shape == (20, 20) and nm.get_noise(2, 4) == np.inf and nm.get_noise(4, 2) == NH3_RMS_K
    dc = DataCube(cube, nm, trans_id=1)
    dc.data[5, 6, 10] = np.nan
    st = CubeStack([dc])
    assert st.get_spec_data(5, 6)[1]
    lon, lat = st.good_pixels()
    assert lon.size == 398 and (5, 6) not in set(zip(lon, lat)) and (2, 4) not in set(zip(lon, lat))
    with pytest.raises(ValueError, match='Cannot parse shape'):
        NoiseMap.from_pbimg(1.0, np.ones(5))


# ---- store -------------------------------------------------------------------------------------
_HAVE_HDF5 = __import__('nestfit_amd.hdf5', fromlist=['available']).available()
_SUFFIXES = ['.npz', pytest.param('.hdf', marks=pytest.mark.skipif(not _HAVE_HDF5, reason='no libhdf5'))]
_FORMATS = ['npz', pytest.param('hdf5', marks=pytest.mark.skipif(not _HAVE_HDF5, reason='no libhdf5'))]


@pytest.mark.parametrize('suffix', _SUFFIXES)
def test_group_tree_and_file_round_trip(tmp_path, suffix):
    f = StoreFile(tmp_path / f'chunk0{suffix}')
    g = f.require_group('/pix/3/4')
    sub = g.create_group('1')
    sub.attrs['global_lnZ'] = -12.5
    sub.attrs['marg_cols'] = ['min', 'max']
    sub.attrs['marg_quantiles'] = np.array([0.0, 1.0])
    sub.create_dataset('posteriors', data=np.arange(12, dtype='float32').reshape(3, 4))
    g.attrs['nbest'] = 1
    with pytest.raises(ValueError, match='already exists'):
        g.create_group('1')
    assert '1' in g and '/pix/3/4/1/posteriors' in f and list(f['/pix']) == ['3']
    f.close()
    with pytest.raises(ValueError):
        f.flush()
    r = StoreFile(tmp_path / f'chunk0{suffix}', 'r')
    sub = r['/pix/3/4/1']
    assert sub.attrs['global_lnZ'] == -12.5 and sub.attrs['marg_cols'] == ['min', 'max']
    np.testing.assert_array_equal(sub.attrs['marg_quantiles'], [0.0, 1.0])
    assert sub['posteriors'].dtype == np.float32 and sub['posteriors'].shape == (3, 4)
    assert r['/pix/3/4'].attrs['nbest'] == 1
    with pytest.raises(FileNotFoundError):
        StoreFile(tmp_path / f'nope{suffix}', 'r')


@pytest.mark.parametrize('fmt', _FORMATS)
def test_hdfstore_layout(tmp_path, stack, fmt, monkeypatch):
    import nestfit_amd as na
    monkeypatch.setenv('NFA_STORE_FORMAT', fmt)
    sfx = {'npz': '.npz', 'hdf5': '.hdf'}[fmt]

    class _Fitter:
        lnZ_thresh, ncomp_max, mn_kwargs = 11, 2, {'nlive': 100}
    with HdfStore(str(tmp_path / 'run'), nchunks=2) as store:
        assert store.store_dir.name == 'run.store' and store.nchunks == 2 and store.model is None
        assert [p.name for p in store.chunk_paths] == [f'chunk0{sfx}', f'chunk1{sfx}'] and store.file_format == fmt
        store.insert_header(stack)
        store.insert_fitter_pars(_Fitter())
        store.insert_model_metadata(na.AmmoniaRunner)
        for k, path in enumerate(store.chunk_paths):
            with StoreFile(path) as ch:
                ch.require_group(f'/pix/{k}/5').create_group('1').attrs['global_lnZ'] = float(k)
        store.link_files()
        assert sorted(g.name for g in store.iter_pix_groups()) == ['/pix/0/5', '/pix/1/5']
        assert store.find_first_valid_group().attrs['global_lnZ'] == 0.0
        store.create_dataset('nbest', np.zeros((20, 20)), group=store.dpath)
        with pytest.warns(RuntimeWarning, match='Deleting dataset'):
            store.create_dataset('nbest', np.ones((20, 20)), group=store.dpath)
        assert store.hdf['/products/nbest'].sum() == 400
    monkeypatch.setenv('NFA_STORE_FORMAT', 'npz')               # an existing store keeps its own format
    again = HdfStore(str(tmp_path / 'run'))
    assert again.file_format == fmt and (tmp_path / 'run.store' / f'table{sfx}').exists()
    assert again.nchunks == 2 and again.model is na.MODELS['ammonia']
    assert again.hdf.attrs['par_names'] == ['voff', 'trot', 'tex', 'ntot', 'sigm', 'orth']
    assert again.hdf.attrs['lnZ_threshold'] == 11 and again.hdf.attrs['naxis1'] == 20
    assert again.read_header(full=False)['CTYPE1'] == 'RA---SIN'
    assert again.read_header()['TELESCOP'] == 'EVLA'
    assert len(list(again.iter_pix_groups())) == 2            # links rebuilt from the chunk files
    again.reset_pix_links()
    assert '/pix' not in again.hdf
    again.close()
    again.close()                                             # prints, does not raise (main.py:283-288)


# ---- GPU: real pixels end to end ------------------------------------------------------------------
@pytest.mark.gpu
def test_real_cube_likelihood_and_fit(engine, nfo, stack, tmp_path):
    from nestfit_amd.fitter import CubeFitter
    ut = engine.get_irdc_priors(size=300, vsys=63.7)
    runner, lon, lat = stack.to_device(ut, ncomp=1)
    assert runner.n_pix == 400 and lon.size == 400
    # likelihood of real pixels against the oracle
    rng = np.random.default_rng(2)
    pix = rng.integers(0, 400, 64).astype(np.int32)
    U = rng.uniform(size=(64, 6))
    Ug = U.copy()
    lg = runner.loglikelihood_batch(pix, Ug)
    ps = nfo.PriorSet(ut.lower())
    for k in range(0, 64, 8):
        spec_data, _ = stack.get_spec_data(lon[pix[k]], lat[pix[k]])
        rc = nfo.AmmoniaRunner([nfo.AmmoniaSpectrum(*sd) for sd in spec_data], ps, ncomp=1)
        u = U[k].copy()
        assert rc.loglikelihood(u) == pytest.approx(lg[k], rel=1e-6)
        np.testing.assert_allclose(Ug[k], u, rtol=1e-9, atol=1e-10)
        assert rc.null_lnZ == pytest.approx(runner.null_lnZ[pix[k]], rel=1e-12)
    # the cube driver on a 3 x 2 corner of the map, two stripes
    sub = CubeStack([_crop(dc, 3, 2) for dc in stack])
    fitter = CubeFitter(sub, ut, engine.AmmoniaRunner, lnZ_thresh=11, ncomp_max=2,
                        mn_kwargs={'nlive': 40, 'tol': 1.0, 'seed': 5, 'maxiter': 600}, nlive_snr_fact=5)
    fitter.fit_cube(str(tmp_path / 'cutout'), nproc=2)
    with HdfStore(str(tmp_path / 'cutout')) as store:
        groups = list(store.iter_pix_groups())
        assert len(groups) == 6 and store.hdf.attrs['model_name'] == 'ammonia'
        assert store.hdf.attrs['multinest_kwargs'].startswith('{')
        for g in groups:
            assert set(g.attrs) == {'i_lon', 'i_lat', 'nbest'} and 0 <= g.attrs['nbest'] <= 2
            one = g['1']
            assert one.attrs['ncomp'] == 1 and one.attrs['n_params'] == 6 and one.attrs['n_chan_tot'] == 758
            assert one['posteriors'].shape == (one.attrs['n_samples'], 8)
            assert one.attrs['n_live'] == 40 + int(5 * sub.get_max_snr(g.attrs['i_lon'], g.attrs['i_lat']))   # main.py:445-447
    # a stripe is one lock-step run whose pixels have live points of their own (fitter.one_group); the scheme before
    # it -- a run per distinct number of live points, side by side on threads -- gives the same store to the last bit
    # whether the runs go side by side or one after the other
    assert fitter.one_group and fitter.group_workers > 1
    fitter.one_group = False
    fitter.fit_cube(str(tmp_path / 'cutout_groups'), nproc=2)
    with HdfStore(str(tmp_path / 'cutout_groups')) as store:
        first = {g.name: (g.attrs['nbest'], g['1'].attrs['global_lnZ'], g['1']['posteriors'].copy()) for g in store.iter_pix_groups()}
    fitter.group_workers = 1
    fitter.fit_cube(str(tmp_path / 'cutout_serial'), nproc=2)
    with HdfStore(str(tmp_path / 'cutout_serial')) as store:
        for g in store.iter_pix_groups():
            nbest, lnz, post = first[g.name]
            assert g.attrs['nbest'] == nbest and g['1'].attrs['global_lnZ'] == lnz and np.array_equal(g['1']['posteriors'], post)
            assert one['marginals'].shape == (15, 6) and one['bestfit_params'].shape == (6,)
            assert np.isfinite(one.attrs['global_lnZ'])
            if g.attrs['nbest'] >= 1:
                assert one.attrs['global_lnZ'] - one.attrs['null_lnZ'] >= 11 and '2' in g
            else:
                assert '2' not in g


def _crop(dc, n_lon, n_lat):
    new = object.__new__(DataCube)
    new.__dict__.update(dc.__dict__)
    new.data = dc.data[:n_lon, :n_lat, :].copy()
    new.shape = new.data.shape
    new.spatial_shape = (n_lon, n_lat)
    return new
