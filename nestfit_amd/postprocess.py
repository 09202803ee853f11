"""Post-processing of a fitted store into dense map products (reference: nestfit/main.py:664-1276,
`postprocess_run` and the eight steps it chains; product names and axis orders from
docs/store_spec.rst:97-122).  Written from that specification; the per-pixel loops of the reference are
array operations here, and the two steps that touch the hot path -- `deblend_hf_intensity` and
`generate_predicted_profiles`, one model evaluation per (pixel, component) -- go through
`CubeRunner.predict_batch` / `peak_and_integrated` on the GPU in batches instead of one
`runner.predict` call per pixel (main.py:1106-1113, 1186-1191).

Axis codes (store_spec.rst:124-135): b latitude, l longitude, m component, p parameter, M quantile,
r run (1 .. n components), h PDF bin, t transition, S channel.  Arrays are built directly in the stored
order (..., b, l).
"""
import numpy as np

_MODEL_ID = {'ammonia': 0, 'diazenylium': 1, 'gaussian': 2}
N_PDF_BINS = 200                 # edges when `aggregate_run_pdfs` makes its own bins (main.py:905-917)
PDF_FLOOR = 1e-32                # zero-probability bins before the logarithm (main.py:980)
PREDICT_ROWS = 4096              # (pixel, component) rows per device batch


# ---------------------------------------------------------------------------------------------
#  small pieces
# ---------------------------------------------------------------------------------------------
def _nans(shape):
    return np.full(shape, np.nan)


def _map_shape(store):
    a = store.hdf.attrs
    return int(a['naxis1']), int(a['naxis2'])


def _runs(group):
    """(ncomp, run group) of a pixel group, ascending."""
    runs = [(int(group[name].attrs['ncomp']), group[name]) for name in group if str(name).isdigit()]
    return sorted(runs, key=lambda r: r[0])


def _product(store, name):
    return np.asarray(store.hdf[f'{store.dpath}/{name}'][...])


def gaussian_kernel(sigma):
    """Unit-sum Gaussian on an odd square grid of about 8 sigma across, sampled at the pixel centres (what
    the reference gets from `Gaussian2DKernel(sigma)`, main.py:742-743)."""
    half = int(np.ceil(8.0 * float(sigma))) // 2
    r = np.arange(-half, half + 1)
    g = np.exp(-0.5 * (r / float(sigma)) ** 2)
    k = np.outer(g, g)
    return k / k.sum()


def get_indep_info_kernel(sigma, nrad=1, sigma_taper=None):
    """Weights for combining a pixel with its neighbours by how much of their information is independent of
    it under a round Gaussian beam of standard deviation `sigma` pixels (main.py:613-661): one at the
    centre; elsewhere one minus the beam's response integrated over the neighbour's area (relative to the
    peak), divided by the pixels per beam (at least one); optionally tapered by a Gaussian of `sigma_taper`
    pixels.  Odd square of side 2 nrad + 1; not normalised."""
    from scipy.special import erf
    assert isinstance(nrad, int) and nrad >= 0
    if nrad == 0:
        return np.ones((1, 1))
    r = np.arange(-nrad, nrad + 1, dtype=np.float64)

    def strip(z):                                        # Gaussian mass between z - 1/2 and z + 1/2
        return 0.5 * (erf((z + 0.5) / (sigma * np.sqrt(2))) - erf((z - 0.5) / (sigma * np.sqrt(2))))
    beam_area = 2 * np.pi * sigma ** 2
    k = (1.0 - np.outer(strip(r), strip(r)) * beam_area) / max(1.0, beam_area)
    if sigma_taper is not None:
        k = k * np.exp(-0.5 * (r[:, None] ** 2 + r[None, :] ** 2) / sigma_taper ** 2)
    k[nrad, nrad] = 1.0
    return k


def _as_kernel(kernel):
    if kernel is None:
        return np.ones((1, 1))
    if isinstance(kernel, (int, float)):
        return gaussian_kernel(kernel)
    return np.asarray(getattr(kernel, 'array', kernel), dtype=np.float64)


def smooth_map(image, kernel, edge='nearest', normalise=True):
    """Kernel-weighted mean of a (b, l) map that ignores NaN pixels: sum(k d) / sum(k) over the valid
    neighbours (the reference's `convolve(..., boundary='extend')` with its default NaN interpolation,
    main.py:757, and `convolve_fft(..., normalize_kernel=False)` with zero fill, main.py:1007-1008: then the
    mean is scaled back by the kernel's sum).  `edge`: 'nearest' repeats the border, 'constant' leaves the
    outside empty (no weight)."""
    from scipy import ndimage
    k = _as_kernel(kernel)[::-1, ::-1]
    valid = np.isfinite(image)
    num = ndimage.correlate(np.where(valid, image, 0.0), k, mode=edge, cval=0.0)
    den = ndimage.correlate(valid.astype(np.float64), k, mode=edge, cval=0.0)
    with np.errstate(invalid='ignore', divide='ignore'):
        out = num / den
    out[den <= 0] = np.nan
    return out if normalise else out * k.sum()


def take_by_components(data, comps, axis=0, incl_zero=True):
    """data[..., b, l] picked along `axis` by the component-count map comps[b, l] (count n -> index n - 1);
    NaN where comps is -1 (no data) and, unless `incl_zero`, where it is 0 (main.py:529-562)."""
    comps = np.asarray(comps)
    index = np.clip(comps - 1, 0, None)
    index = index.reshape((1,) * (data.ndim - comps.ndim) + comps.shape)
    out = np.squeeze(np.take_along_axis(np.asarray(data, dtype=np.float64), index, axis=axis), axis=axis)
    out[..., comps < (0 if incl_zero else 1)] = np.nan
    return out


# ---------------------------------------------------------------------------------------------
#  aggregation of the per-pixel runs
# ---------------------------------------------------------------------------------------------
# product (m, b, l)  <-  attribute of run n (plane n), attribute of the one-component run that fills plane 0
ATTRIBUTE_MAPS = (
    ('evidence', 'global_lnZ', 'null_lnZ'),
    ('evidence_err', 'global_lnZ_err', None),
    ('BIC', 'BIC', 'null_BIC'),
    ('AIC', 'AIC', 'null_AIC'),
    ('AICc', 'AICc', 'null_AICc'),
)


def aggregate_run_attributes(store):
    """'nbest' (b, l) and the evidence / information-criterion maps (m, b, l), plane 0 = the null model
    (main.py:664-721)."""
    print(':: Aggregating store attributes')
    n_lon, n_lat = _map_shape(store)
    n_max = int(store.hdf.attrs['n_max_components'])
    maps = {name: _nans((n_max + 1, n_lat, n_lon)) for name, _, _ in ATTRIBUTE_MAPS}
    nbest = np.full((n_lat, n_lon), -1, dtype=np.int32)
    for group in store.iter_pix_groups():
        l, b = int(group.attrs['i_lon']), int(group.attrs['i_lat'])
        nbest[b, l] = group.attrs['nbest']
        for n, run in _runs(group):
            for name, attr, null_attr in ATTRIBUTE_MAPS:
                maps[name][n, b, l] = run.attrs[attr]
                if n == 1 and null_attr is not None:
                    maps[name][0, b, l] = run.attrs[null_attr]
    store.create_dataset('nbest', nbest, group=store.dpath)
    for name, _, _ in ATTRIBUTE_MAPS:
        store.create_dataset(name, maps[name], group=store.dpath)


def convolve_evidence(store, kernel):
    """Evidence maps smoothed over the sky and the component count chosen again from them: 'conv_evidence'
    (m, b, l), 'conv_nbest' (b, l) (main.py:724-774).  A count may grow by at most one over the local
    choice: no run exists beyond that."""
    print(':: Convolving evidence maps')
    thresh = float(store.hdf.attrs['lnZ_threshold'])
    evid, nbest = _product(store, 'evidence'), _product(store, 'nbest')
    conv = np.stack([smooth_map(plane, kernel, edge='nearest') for plane in evid])
    choice = np.zeros(nbest.shape, dtype=np.int32)
    for n in range(evid.shape[0] - 1):                   # every step has to pass: 0 -> 1 before 1 -> 2
        with np.errstate(invalid='ignore'):
            choice[(choice == n) & (conv[n + 1] - conv[n] > thresh)] += 1
    choice[nbest == -1] = -1
    choice = np.where(choice - nbest >= 2, nbest + 1, choice).astype(np.int32)
    store.create_dataset('conv_nbest', choice, group=store.dpath)
    store.create_dataset('conv_evidence', conv, group=store.dpath)


def extended_masked_evidence(store, kernel, conv=True, lnz_thresh=3):
    """'mext_evidence' (b, l): evidence of one component over none, smoothed a second time with the
    detections masked out, to bring out weak extended emission (main.py:777-816)."""
    print(':: Convolving masked evidence')
    evid = _product(store, 'evidence')
    ref = _product(store, 'conv_evidence' if conv else 'evidence')
    gain = ref[1] - ref[0]
    with np.errstate(invalid='ignore'):
        detected = gain > lnz_thresh
    smooth = [smooth_map(np.where(detected, np.nan, evid[n]), kernel, edge='nearest') for n in (0, 1)]
    out = smooth[1] - smooth[0]
    out[np.isnan(gain) | detected] = np.nan
    store.create_dataset('mext_evidence', out, group=store.dpath)


def aggregate_run_products(store):
    """Parameter cubes of the preferred run of every pixel: 'nbest_MAP' and 'nbest_bestfit' (m, p, b, l),
    'nbest_marginals' (m, p, M, b, l), 'marg_quantiles' (M) (main.py:819-882)."""
    print(':: Aggregating store products')
    n_lon, n_lat = _map_shape(store)
    n_max, n_par = int(store.hdf.attrs['n_max_components']), int(store.hdf.attrs['n_params'])
    choice = _product(store, 'conv_nbest')
    quantiles = np.asarray(store.find_first_valid_group().attrs['marg_quantiles'])
    cubes = {'nbest_MAP': ('map_params', _nans((n_max, n_par, n_lat, n_lon))),
             'nbest_bestfit': ('bestfit_params', _nans((n_max, n_par, n_lat, n_lon)))}
    margs = _nans((n_max, n_par, quantiles.size, n_lat, n_lon))
    for group in store.iter_pix_groups():
        l, b = int(group.attrs['i_lon']), int(group.attrs['i_lat'])
        n = int(choice[b, l])
        if n <= 0:
            continue
        run = group[f'{n}']
        for dset, cube in cubes.values():               # stored parameter-major: (p * m) -> (p, m)
            cube[:n, :, b, l] = np.asarray(run[dset][...]).reshape(n_par, n).T
        margs[:n, :, :, b, l] = np.asarray(run['marginals'][...]).reshape(quantiles.size, n_par, n).transpose(2, 1, 0)
    store.create_dataset('marg_quantiles', quantiles, group=store.dpath)
    for name, (_, cube) in cubes.items():
        store.create_dataset(name, cube, group=store.dpath)
    store.create_dataset('nbest_marginals', margs, group=store.dpath)


def aggregate_run_pdfs(store, par_bins=None):
    """Histograms of the posterior samples per parameter: 'pdf_bins' (p, h) = bin centres, 'post_pdfs'
    (r, m, p, h, b, l), each normalised to one (main.py:885-953).  `par_bins` (p, h + 1) = bin edges;
    by default N_PDF_BINS edges between the extremes of the marginals."""
    print(':: Aggregating store marginalized posterior PDFs')
    n_lon, n_lat = _map_shape(store)
    n_max, n_par = int(store.hdf.attrs['n_max_components']), int(store.hdf.attrs['n_params'])
    if par_bins is None:
        margs = _product(store, 'nbest_marginals')       # (m, p, M, b, l); quantile 0 = min, 8 = max
        lo = np.nanmin(margs[:, :, 0], axis=(0, 2, 3))
        hi = np.nanmax(margs[:, :, 8], axis=(0, 2, 3))
        par_bins = np.stack([np.linspace(a, b, N_PDF_BINS) for a, b in zip(lo, hi)])
    par_bins = np.asarray(par_bins, dtype=np.float64)
    n_bin = par_bins.shape[1] - 1
    pdfs = _nans((n_max, n_max, n_par, n_bin, n_lat, n_lon))
    for group in store.iter_pix_groups():
        l, b = int(group.attrs['i_lon']), int(group.attrs['i_lat'])
        for n, run in _runs(group):
            post = np.asarray(run['posteriors'][...], dtype=np.float64)
            for p in range(n_par):
                for m in range(n):                       # column of parameter p, component m
                    pdfs[n - 1, m, p, :, b, l] = np.histogram(post[:, p * n + m], bins=par_bins[p])[0]
    with np.errstate(invalid='ignore', divide='ignore'):
        pdfs /= np.nansum(pdfs, axis=3, keepdims=True)
    store.create_dataset('pdf_bins', 0.5 * (par_bins[:, :-1] + par_bins[:, 1:]), group=store.dpath)
    store.create_dataset('post_pdfs', pdfs.astype('float32'), group=store.dpath)


def convolve_post_pdfs(store, kernel, evid_weight=True):
    """'conv_post_pdfs' (r, m, p, h, b, l): the PDFs multiplied together over the sky with the kernel's
    weights (a kernel-weighted sum of their logarithms), optionally with every pixel's log PDF scaled by its
    evidence gain mapped to [0, 1] (main.py:956-1017)."""
    print(':: Convolving posterior PDFs')
    k = _as_kernel(kernel)
    pdfs = _product(store, 'post_pdfs').astype(np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        logp = np.log(np.where(pdfs == 0, PDF_FLOOR, pdfs))
    if evid_weight:
        evid = _product(store, 'evidence')
        gain = take_by_components(evid[1:], _product(store, 'conv_nbest')) - evid[0]
        gain = gain - np.nanmin(gain)
        gain = gain / np.nanmax(gain)
        logp = logp * gain                               # broadcasts over the trailing (b, l)
    out = np.zeros_like(logp)
    n_run, n_comp, n_par, n_bin = logp.shape[:4]
    for r in range(n_run):
        for m in range(r + 1):                           # run r + 1 has components 0 .. r
            for p in range(n_par):
                for h in range(n_bin):
                    out[r, m, p, h] = smooth_map(logp[r, m, p, h], k, edge='constant', normalise=False)
    with np.errstate(invalid='ignore', divide='ignore', over='ignore'):
        out = np.exp(out)
        out /= np.nansum(out, axis=3, keepdims=True)
    out[np.isnan(pdfs)] = np.nan
    store.create_dataset('conv_post_pdfs', out.astype('float32'), group=store.dpath)


def quantize_conv_marginals(store):
    """'conv_marginals' (r, m, p, M, b, l): the quantiles `marg_quantiles` of the convolved PDFs, by linear
    interpolation of their cumulative sums over the bin centres (main.py:1020-1061)."""
    print(':: Calculating convolved PDF quantiles')
    centres, quantiles = _product(store, 'pdf_bins'), _product(store, 'marg_quantiles')
    pdfs = np.moveaxis(_product(store, 'conv_post_pdfs').astype(np.float64), 3, -1)     # (r, m, p, b, l, h)
    with np.errstate(invalid='ignore', divide='ignore'):
        cdf = np.cumsum(pdfs, axis=-1) / np.sum(pdfs, axis=-1, keepdims=True)
    out = _nans(cdf.shape[:-1] + (quantiles.size,))
    for ix in np.ndindex(*cdf.shape[:-1]):
        out[ix] = np.interp(quantiles, cdf[ix], centres[ix[2]])
    store.create_dataset('conv_marginals', np.moveaxis(out, -1, 3).astype('float32'), group=store.dpath)


# ---------------------------------------------------------------------------------------------
#  the two steps on the hot path: one model evaluation per (pixel, component)
# ---------------------------------------------------------------------------------------------
def _device_predictor(store, stack):
    """predict(lon[B], lat[B], theta[B, p], want_spectra) -> (spectra[B, chan_tot] or None, peak[B, t],
    integrated[B, t]) on the GPU: every pixel's data stay where the fit left them conceptually -- a predict
    needs only the axes, so the spectra set is built over the requested pixels alone."""
    from .cube import CubeRunner
    model_id = _MODEL_ID[store.hdf.attrs['model_name']]
    xarrs = [dc.xarr for dc in stack.cubes]
    trans = [dc.trans_id for dc in stack.cubes]
    chan_tot = sum(len(x) for x in xarrs)
    # predictions do not read the data: one all-zero pixel serves every row
    extra = {}
    if model_id == 2:                                    # the Gaussian model has no transition table to take them from
        extra['rest_freqs'] = [float(dc.full_header.get('RESTFRQ', dc.full_header.get('RESTFREQ'))) for dc in stack.cubes]
    runner = CubeRunner(xarrs, trans, np.zeros((1, chan_tot)), np.ones((1, len(xarrs))), None, ncomp=1,
                        model=model_id, **extra)
    runner.set_exp_mode('table')                         # map products in the reference's own arithmetic: a one-off, not a rate

    scratch = {}

    def predict(lon, lat, theta, want_spectra):
        pix = np.zeros(theta.shape[0], dtype=np.int32)
        if want_spectra:
            # (one buffer the device addresses itself serves every batch: the callers use a batch's spectra before
            # they ask for the next)
            if scratch.get('rows', 0) < theta.shape[0]:
                from ._ffi import pinned_empty
                scratch['buf'], scratch['rows'] = pinned_empty((theta.shape[0], chan_tot)), theta.shape[0]
            spec, _ = runner.predict_batch(pix, theta, out=scratch['buf'][:theta.shape[0]])
            return spec, None, None
        peak, tot = runner.peak_and_integrated(pix, theta)
        return None, peak, tot
    return predict


def _map_rows(store):
    """Rows (l, b, m, theta[p]) of the MAP cube with all parameters finite."""
    pmap = _product(store, 'nbest_MAP')                  # (m, p, b, l)
    ok = np.all(np.isfinite(pmap), axis=1)               # (m, b, l)
    m, b, l = np.nonzero(ok)
    return pmap, l, b, m, np.ascontiguousarray(pmap[m, :, b, l])


def deblend_hf_intensity(store, stack, runner=None, predict_backend=None):
    """Peak and integrated intensity of every MAP component per transition, and line profiles with the
    hyperfine structure taken out (a Gaussian of the component's width carrying its integrated intensity,
    on the velocity grid of `pdf_bins`): 'peak_intensity', 'integrated_intensity' (t, m, b, l) in K and
    K km/s, 'hf_deblended' (t, m, S, b, l) (main.py:1064-1133).

    `runner` is accepted for the reference's signature (its one-component runner evaluated pixel by pixel);
    the model spectra come from one GPU batch per PREDICT_ROWS (pixel, component) rows.  `predict_backend`:
    callable(lon, lat, theta, want_spectra) standing in for the device (tests without a GPU)."""
    assert runner is None or getattr(runner, 'ncomp', 1) == 1
    print(':: Deblending HF structure in intensity map')
    predict = predict_backend or _device_predictor(store, stack)
    pmap, l, b, m, theta = _map_rows(store)
    n_spec = stack.n_cubes
    shape = (n_spec,) + pmap.shape[:1] + pmap.shape[2:]  # (t, m, b, l)
    peak, integ = _nans(shape), _nans(shape)
    for a in range(0, theta.shape[0], PREDICT_ROWS):
        s = slice(a, a + PREDICT_ROWS)
        _, pk, tot = predict(l[s], b[s], theta[s], False)
        peak[:, m[s], b[s], l[s]] = pk.T
        integ[:, m[s], b[s], l[s]] = tot.T
    integ *= np.array([dc.dv for dc in stack.cubes]).reshape(-1, 1, 1, 1)       # K -> K km/s
    vaxis = _product(store, 'pdf_bins')[0]
    dv_bin = abs(vaxis[1] - vaxis[0])
    vcen = pmap[:, store.model.IX_VCEN][None, :, None]   # (1, m, 1, b, l)
    sigm = pmap[:, store.model.IX_SIGM][None, :, None]
    with np.errstate(invalid='ignore', divide='ignore'):
        profile = (dv_bin / (sigm * np.sqrt(2 * np.pi))) * integ[:, :, None] \
                  * np.exp(-0.5 * ((vaxis.reshape(1, 1, -1, 1, 1) - vcen) / sigm) ** 2)
    store.create_dataset('peak_intensity', peak, group=store.dpath)
    store.create_dataset('integrated_intensity', integ, group=store.dpath)
    store.create_dataset('hf_deblended', profile.astype('float32'), group=store.dpath)


def generate_predicted_profiles(store, stack, runner=None, predict_backend=None):
    """Model spectra of every MAP component on the channels of each cube: 'model_spec/trans<ID>'
    (m, S, b, l), float32 (main.py:1136-1193).  The spectra-out mode of the likelihood kernel, the one shape
    of this path that is bound by memory traffic."""
    assert runner is None or getattr(runner, 'ncomp', 1) == 1
    print(':: Generating MAP model spectral profiles')
    predict = predict_backend or _device_predictor(store, stack)
    pmap, l, b, m, theta = _map_rows(store)
    n_max, n_lat, n_lon = pmap.shape[0], pmap.shape[2], pmap.shape[3]
    cubes = [np.full((n_max, dc.nchan, n_lat, n_lon), np.nan, dtype=np.float32) for dc in stack.cubes]
    edges = np.concatenate([[0], np.cumsum([dc.nchan for dc in stack.cubes])])
    for a in range(0, theta.shape[0], PREDICT_ROWS):
        s = slice(a, a + PREDICT_ROWS)
        spec, _, _ = predict(l[s], b[s], theta[s], True)
        for k, cube in enumerate(cubes):
            cube[m[s], :, b[s], l[s]] = spec[:, edges[k]:edges[k + 1]]
    for cube, dc in zip(cubes, stack.cubes):
        store.create_dataset(f'trans{dc.trans_id}', cube, group=f'{store.dpath}/model_spec')


def create_fits_from_store(store, prefix='source'):
    """The deblended cubes as FITS files, one per transition, the components added up, on the velocity grid of
    the PDF bins: `<prefix>_hf_deblended_trans<t>.fits` (main.py:1196-1237; peak / integrated intensity and the
    PDF products are a TODO there as well).  Returns the paths written."""
    from .cubeio import write_fits
    header = dict(store.read_header(full=True))
    vaxis = _product(store, 'pdf_bins')[store.model.IX_VCEN]
    cube = _product(store, 'hf_deblended')                 # (t, m, S, b, l)
    header.update(BUNIT='K', CRPIX3=1, CDELT3=float(vaxis[1] - vaxis[0]), CUNIT3='km/s', CTYPE3='VRAD',
                  CRVAL3=float(vaxis[0]), SPECSYS='LSRK')
    paths = []
    for t in range(cube.shape[0]):
        paths.append(f'{prefix}_hf_deblended_trans{t}.fits')
        write_fits(paths[-1], header, np.nansum(cube[t], axis=0).astype(np.float32))      # (S, b, l)
    return paths


def postprocess_run(store, stack, runner=None, par_bins=None, evid_kernel=None, post_kernel=None,
                    evid_weight=True, predict_backend=None):
    """All steps in the reference's order (main.py:1240-1276)."""
    aggregate_run_attributes(store)
    convolve_evidence(store, evid_kernel)
    aggregate_run_products(store)
    aggregate_run_pdfs(store, par_bins=par_bins)
    convolve_post_pdfs(store, post_kernel, evid_weight=evid_weight)
    quantize_conv_marginals(store)
    deblend_hf_intensity(store, stack, runner, predict_backend=predict_backend)
    generate_predicted_profiles(store, stack, runner, predict_backend=predict_backend)
