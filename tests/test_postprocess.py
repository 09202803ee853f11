"""Post-processing of a fitted store (nestfit_amd/postprocess.py; reference nestfit/main.py:664-1276): the
aggregation and convolution steps on a store fitted without a GPU (sampler twin on the oracle), every
product in the shape and order of docs/store_spec.rst:97-122, values checked against the per-pixel
groups they come from; the two steps on the hot path against per-pixel oracle predictions (CPU backend
here, the GPU batch in test_postprocess_on_device)."""
import numpy as np
import pytest

from nestfit_amd import postprocess as pp
from nestfit_amd.store import HdfStore

from test_fitter_cpu import N_CHAN, _fitter, _stack


def _oracle_predictor(stack):
    from oracle import nfo

    def predict(lon, lat, theta, want_spectra):
        spec = np.empty((theta.shape[0], sum(dc.nchan for dc in stack.cubes)))
        peak = np.empty((theta.shape[0], stack.n_cubes))
        tot = np.empty_like(peak)
        for k, th in enumerate(theta):
            off = 0
            for t, dc in enumerate(stack.cubes):
                s = nfo.AmmoniaSpectrum(dc.xarr, np.zeros(dc.nchan), 1.0, dc.trans_id)
                nfo.amm_predict(s, th)
                spec[k, off:off + dc.nchan] = s.get_spec()
                peak[k, t], tot[k, t] = s.max_spec, s.sum_spec
                off += dc.nchan
        return (spec, None, None) if want_spectra else (None, peak, tot)
    return predict


@pytest.fixture(scope='module')
def fitted(tmp_path_factory):
    stack = _stack(n_lon=4, n_lat=2, seed=4)
    path = tmp_path_factory.mktemp('post') / 'run'
    _fitter(stack).fit_cube(str(path), nproc=2)
    return stack, str(path)


def test_smooth_map_ignores_nans_and_extends_the_border():
    img = np.arange(20.0).reshape(4, 5)
    k = pp.gaussian_kernel(0.7)
    assert k.shape == (7, 7) and k.sum() == pytest.approx(1.0)
    flat = pp.smooth_map(np.full((4, 5), 3.0), k)
    assert np.allclose(flat, 3.0)                                   # border repeated: a constant stays constant
    holed = img.copy(); holed[1, 2] = np.nan
    out = pp.smooth_map(holed, k)
    assert np.isfinite(out).all() and abs(out[1, 2] - img[1, 2]) < 0.5      # filled in from its neighbours
    one = pp.smooth_map(img, np.ones((1, 1)))
    assert np.array_equal(one, img)
    # zero fill, kernel not normalised: a plane of ones keeps the value sum(k) wherever any weight falls inside
    assert np.allclose(pp.smooth_map(np.ones((4, 5)), 2.0 * k, edge='constant', normalise=False), 2.0)


def test_take_by_components():
    data = np.arange(2 * 2 * 3, dtype=float).reshape(2, 2, 3)
    comps = np.array([[1, 2, 0], [-1, 2, 1]])
    out = pp.take_by_components(data, comps)
    assert out[0, 0] == data[0, 0, 0] and out[0, 1] == data[1, 0, 1] and out[0, 2] == data[0, 0, 2]
    assert np.isnan(out[1, 0]) and out[1, 1] == data[1, 1, 1]
    assert np.isnan(pp.take_by_components(data, comps, incl_zero=False)[0, 2])


def test_aggregation_and_convolution_products(fitted):
    stack, path = fitted
    with HdfStore(path) as store:
        pp.aggregate_run_attributes(store)
        pp.convolve_evidence(store, 0.6)
        pp.extended_masked_evidence(store, 0.6)
        pp.aggregate_run_products(store)
        pp.aggregate_run_pdfs(store)
        pp.convolve_post_pdfs(store, pp.gaussian_kernel(0.6))
        pp.quantize_conv_marginals(store)
        prod = store.hdf['/products']
        n_lon, n_lat, n_max, n_par = 4, 2, 2, 6
        nbest, evid = prod['nbest'][...], prod['evidence'][...]
        assert nbest.shape == (n_lat, n_lon) and evid.shape == (n_max + 1, n_lat, n_lon)
        groups = {(g.attrs['i_lon'], g.attrs['i_lat']): g for g in store.iter_pix_groups()}
        assert nbest[0, 3] == -1 and (3, 0) not in groups                       # the NaN pixel
        for (l, b), g in groups.items():
            assert nbest[b, l] == g.attrs['nbest']
            assert evid[0, b, l] == g['1'].attrs['null_lnZ'] and evid[1, b, l] == g['1'].attrs['global_lnZ']
            assert prod['BIC'][...][1, b, l] == g['1'].attrs['BIC'] and prod['AICc'][...][0, b, l] == g['1'].attrs['null_AICc']
            assert np.isnan(evid[2, b, l]) == ('2' not in g)
        conv_nbest = prod['conv_nbest'][...]
        assert conv_nbest.shape == nbest.shape and np.all(conv_nbest - nbest <= 1) and conv_nbest[0, 3] == -1
        assert prod['conv_evidence'][...].shape == evid.shape and prod['mext_evidence'][...].shape == nbest.shape
        # parameter cubes: the preferred run's vectors, parameter-major in the store, (m, p, b, l) here
        pmap, marg = prod['nbest_MAP'][...], prod['nbest_marginals'][...]
        assert pmap.shape == (n_max, n_par, n_lat, n_lon) and marg.shape == (n_max, n_par, 15, n_lat, n_lon)
        for (l, b), g in groups.items():
            n = conv_nbest[b, l]
            if n <= 0:
                assert np.isnan(pmap[:, :, b, l]).all()
                continue
            vec = np.asarray(g[f'{n}']['map_params'][...])
            for p in range(n_par):
                for m in range(n):
                    assert pmap[m, p, b, l] == vec[p * n + m]
                    assert np.array_equal(marg[m, p, :, b, l], np.asarray(g[f'{n}']['marginals'][...])[:, p * n + m])
            assert np.isnan(pmap[n:, :, b, l]).all()
        bins, pdfs = prod['pdf_bins'][...], prod['post_pdfs'][...]
        assert bins.shape == (n_par, pp.N_PDF_BINS - 1) and pdfs.shape == (n_max, n_max, n_par, pp.N_PDF_BINS - 1, n_lat, n_lon)
        assert pdfs.dtype == np.float32
        tot = np.nansum(pdfs, axis=3)
        filled = ~np.isnan(pdfs).all(axis=3)
        assert np.allclose(tot[filled], 1.0, atol=1e-5) and not filled[0, 1].any()      # run 1 has no component 2
        cpdf, cmarg = prod['conv_post_pdfs'][...], prod['conv_marginals'][...]
        assert cpdf.shape == pdfs.shape and cmarg.shape == (n_max, n_max, n_par, 15, n_lat, n_lon)
        assert np.array_equal(np.isnan(cpdf), np.isnan(pdfs))
        ok = filled[0, 0]
        med = cmarg[0, 0][:, 4][ok]                                    # medians of run 1, component 1
        lo, hi = cmarg[0, 0][:, 0][ok], cmarg[0, 0][:, 8][ok]
        assert np.all(lo <= med) and np.all(med <= hi)


def test_hot_path_steps_against_per_pixel_predictions(fitted):
    from oracle import nfo
    stack, path = fitted
    backend = _oracle_predictor(stack)
    with HdfStore(path) as store:
        if 'nbest_MAP' not in store.hdf['/products']:
            pp.aggregate_run_attributes(store); pp.convolve_evidence(store, None)
            pp.aggregate_run_products(store); pp.aggregate_run_pdfs(store)
        pp.deblend_hf_intensity(store, stack, predict_backend=backend)
        pp.generate_predicted_profiles(store, stack, predict_backend=backend)
        prod = store.hdf['/products']
        pmap = prod['nbest_MAP'][...]
        peak, integ, hfdb = prod['peak_intensity'][...], prod['integrated_intensity'][...], prod['hf_deblended'][...]
        assert peak.shape == (2, 2, 2, 4) and hfdb.shape == (2, 2, pp.N_PDF_BINS - 1, 2, 4) and hfdb.dtype == np.float32
        spec_cubes = [prod['model_spec'][f'trans{t}'][...] for t in (1, 2)]
        assert all(c.shape == (2, N_CHAN, 2, 4) and c.dtype == np.float32 for c in spec_cubes)
        seen = 0
        for m, b, l in np.ndindex(2, 2, 4):
            th = pmap[m, :, b, l]
            if np.isnan(th).any():
                assert np.isnan(peak[:, m, b, l]).all() and np.isnan(spec_cubes[0][m, :, b, l]).all()
                continue
            seen += 1
            for t, dc in enumerate(stack.cubes):
                s = nfo.AmmoniaSpectrum(dc.xarr, np.zeros(dc.nchan), 1.0, dc.trans_id)
                nfo.amm_predict(s, np.ascontiguousarray(th))
                assert peak[t, m, b, l] == s.max_spec and integ[t, m, b, l] == pytest.approx(s.sum_spec * dc.dv, rel=1e-14)
                assert np.array_equal(spec_cubes[t][m, :, b, l], s.get_spec().astype('float32'))
                # the deblended profile carries the integrated intensity
                vaxis = prod['pdf_bins'][...][0]
                if vaxis.min() < th[0] - 4 * th[4] and vaxis.max() > th[0] + 4 * th[4]:
                    assert hfdb[t, m, :, b, l].sum() == pytest.approx(integ[t, m, b, l], rel=2e-2)
        assert seen >= 2
        # FITS export of the deblended cubes: read back with the package's own reader
        from nestfit_amd.cubeio import read_fits
        paths = pp.create_fits_from_store(store, prefix=str(path) + '_out')
        assert len(paths) == 2
        hdr, data = read_fits(paths[1])
        assert data.shape == (pp.N_PDF_BINS - 1, 2, 4) and hdr['CTYPE3'] == 'VRAD' and hdr['BITPIX'] == -32
        vaxis = prod['pdf_bins'][...][0]
        assert hdr['CRVAL3'] == pytest.approx(vaxis[0]) and hdr['CDELT3'] == pytest.approx(vaxis[1] - vaxis[0])
        np.testing.assert_array_equal(data, np.nansum(hfdb[1], axis=0).astype(np.float32).astype(np.float64))


@pytest.mark.gpu
def test_postprocess_on_device(engine, nfo, fitted, tmp_path):
    """The device batch behind deblend_hf_intensity / generate_predicted_profiles gives what the CPU backend
    gives (table mode: peak and integrated intensity to 1e-11, spectra equal after rounding to float32)."""
    import shutil
    stack, path = fitted
    work = tmp_path / 'run.store'
    shutil.copytree(path + '.store', work)
    with HdfStore(str(tmp_path / 'run')) as store:
        pp.postprocess_run(store, stack, evid_kernel=0.6, post_kernel=pp.gaussian_kernel(0.6))
        dev = {k: np.array(store.hdf['/products'][k][...]) for k in ('peak_intensity', 'integrated_intensity', 'hf_deblended')}
        dev_spec = [np.array(store.hdf['/products']['model_spec'][f'trans{t}'][...]) for t in (1, 2)]
        backend = _oracle_predictor(stack)
        pp.deblend_hf_intensity(store, stack, predict_backend=backend)
        pp.generate_predicted_profiles(store, stack, predict_backend=backend)
        prod = store.hdf['/products']
        assert np.isfinite(dev['peak_intensity']).sum() >= 4
        for k, v in dev.items():
            np.testing.assert_allclose(v, prod[k][...], rtol=1e-6 if k == 'hf_deblended' else 1e-11, equal_nan=True)
        for t in range(2):
            np.testing.assert_allclose(dev_spec[t], prod['model_spec'][f'trans{t + 1}'][...], rtol=2e-7, atol=1e-30, equal_nan=True)


def test_smooth_map_against_a_direct_sum():
    """The two convolution flavours against their definition evaluated pixel by pixel: weights of the valid
    neighbours only, border repeated ('nearest') or empty ('constant'), scaled back by the kernel's sum where the
    kernel is not to be normalised."""
    rng = np.random.default_rng(9)
    img = rng.normal(size=(7, 9))
    img[rng.uniform(size=img.shape) < 0.2] = np.nan
    k = rng.uniform(0.1, 1.0, size=(3, 5))
    ky, kx = k.shape[0] // 2, k.shape[1] // 2

    def direct(edge, normalise):
        out = np.full(img.shape, np.nan)
        for b in range(img.shape[0]):
            for l in range(img.shape[1]):
                num = den = 0.0
                for dy in range(-ky, ky + 1):
                    for dx in range(-kx, kx + 1):
                        yy, xx = b - dy, l - dx                      # convolution: the kernel is flipped
                        if edge == 'nearest':
                            yy, xx = min(max(yy, 0), img.shape[0] - 1), min(max(xx, 0), img.shape[1] - 1)
                        elif not (0 <= yy < img.shape[0] and 0 <= xx < img.shape[1]):
                            continue
                        v = img[yy, xx]
                        if np.isfinite(v):
                            w = k[dy + ky, dx + kx]
                            num += w * v
                            den += w
                if den > 0:
                    out[b, l] = num / den * (1.0 if normalise else k.sum())
        return out
    for edge, normalise in (('nearest', True), ('constant', False), ('constant', True)):
        np.testing.assert_allclose(pp.smooth_map(img, k, edge=edge, normalise=normalise), direct(edge, normalise),
                                   rtol=1e-12, equal_nan=True)


def test_fits_writer_round_trip(tmp_path):
    from nestfit_amd.cubeio import read_fits, write_fits
    rng = np.random.default_rng(3)
    for dtype in (np.float32, np.float64):
        data = rng.normal(size=(5, 4, 3)).astype(dtype)
        hdr = {'BUNIT': 'K', 'CRVAL3': -12.5, 'CDELT3': 0.25, 'CRPIX3': 1, 'CTYPE3': 'VRAD', 'OBJECT': "it's a test",
               'FLAG': True, 'NAXIS': 99, 'SIMPLE': False}         # structural keywords come from the array, not from here
        path = tmp_path / f'cube_{np.dtype(dtype).name}.fits'
        write_fits(path, hdr, data)
        assert path.stat().st_size % 2880 == 0
        back_hdr, back = read_fits(path)
        assert np.array_equal(back, data.astype(np.float64))
        assert back_hdr['NAXIS'] == 3 and [back_hdr[f'NAXIS{k}'] for k in (1, 2, 3)] == [3, 4, 5]
        assert back_hdr['BITPIX'] == (-32 if dtype == np.float32 else -64) and back_hdr['SIMPLE'] is True
        assert back_hdr['CRVAL3'] == -12.5 and back_hdr['CDELT3'] == 0.25 and back_hdr['CTYPE3'] == 'VRAD'
        assert back_hdr['OBJECT'] == "it's a test" and back_hdr['FLAG'] is True
