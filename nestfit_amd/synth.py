"""Synthetic inputs of SURVEY.md section 8(d): frequency axes and truth
parameters of the benchmark configurations.  Pure numpy, no model arithmetic
(synthetic *data* are made by the engine itself, see bench.py)."""
import numpy as np

CKMS = 299792.458
NU0 = {1: 23.6944955e9, 2: 23.722633335e9, 3: 23.8701296e9, 4: 24.1394169e9, 5: 24.53299e9,
       6: 25.05603e9, 7: 25.71518e9, 8: 26.51898e9, 9: 27.477943e9}

# get_test_spectra(kind=0) truth (reference: nestfit/synth_spectra.py:251-258)
TRUTH_2COMP = np.array([-1.0, 1.5, 10.0, 15.0, 4.0, 6.0, 14.5, 15.0, 0.3, 0.6, 0.0, 0.0])
TRUTH_1COMP = np.array([-1.0, 10.0, 4.0, 14.5, 0.3, 0.0])
# cold + hot 3-component model of config 4
TRUTH_3COMP = np.array([-2, .5, 3, 12, 25, 60, 5, 8, 20, 14.5, 14.8, 14.2, .3, .5, 1, .3, .3, .5])


def freq_axis(trans_id, n_chan, vhalf=30.0):
    """Ascending, uniform frequency axis nu0 (1 - v/c), v = linspace(+vh, -vh, N)."""
    v = np.linspace(vhalf, -vhalf, n_chan)
    return NU0[trans_id] * (1.0 - v / CKMS)


def param_sampler_draw(rng, vsep=(0.16, 3), trot=(3, 30), tex=(2.8, 12), ntot=(13, 16),
                       sigm=(0.15, 2), orth=(0, 0)):
    """Two-component truth drawn like ParamSampler.draw
    (reference: nestfit/synth_spectra.py:165-192)."""
    v = rng.uniform(*vsep)
    return np.concatenate([
        [0.0, v], rng.uniform(*trot, size=2), rng.uniform(*tex, size=2),
        rng.uniform(*ntot, size=2), rng.uniform(*sigm, size=2), rng.uniform(*orth, size=2)])


def c5_stack(side=32, n=1024, noise=0.2, seed=11, exp_mode='table'):
    """BASELINE config 5 as SURVEY.md 8d defines it: `side` x `side` pixels of config 3's generator (two-component
    truths from the ParamSampler ranges, default_rng(11), sigma = 0.2 K), NH3 (1,1)+(2,2) on `n` channels each, as
    the `CubeStack` of two `DataCube`s the cube driver takes (frequency axes from FITS-style headers).  The model
    spectra come from the engine (`exp_mode`; the mode in force before the call is put back).
    Returns (stack, truths, model, data, axes_hz, utrans)."""
    import nestfit_amd as na
    from .cube import CubeRunner
    from .cubeio import CubeStack, DataCube, SimpleCube
    n_pix = side * side
    rng = np.random.default_rng(seed)
    truths = np.array([param_sampler_draw(rng) for _ in range(n_pix)])
    ut = na.get_irdc_priors(size=500, vsys=0.0)
    headers, cubes_hz = [], []
    for t in (1, 2):
        f = freq_axis(t, n)
        hdr = {'BUNIT': 'K', 'CTYPE3': 'FREQ', 'CUNIT3': 'Hz', 'CRVAL3': float(f[0]), 'CDELT3': float((f[-1] - f[0]) / (n - 1)),
               'CRPIX3': 1.0, 'RESTFRQ': NU0[t], 'CTYPE1': 'RA---SIN', 'CTYPE2': 'DEC--SIN', 'CRVAL1': 270.0, 'CRVAL2': -20.0,
               'CDELT1': -1e-3, 'CDELT2': 1e-3, 'CRPIX1': 1.0, 'CRPIX2': 1.0, 'CUNIT1': 'deg', 'CUNIT2': 'deg',
               'NAXIS1': side, 'NAXIS2': side, 'NAXIS3': n, 'NAXIS': 3}
        headers.append(hdr)
        cubes_hz.append(SimpleCube(hdr, np.zeros((n, side, side))).spectral_axis_hz())
    before = na.get_exp_mode()
    na.set_exp_mode(exp_mode)
    try:
        probe = CubeRunner(cubes_hz, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=2)
        model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
    finally:
        na.set_exp_mode(before)
    data = model + rng.normal(0, noise, model.shape)
    dcubes = []
    for k, t in enumerate((1, 2)):
        # SimpleCube data are (chan, lat, lon); pixel p = i_lon * side + i_lat
        arr = data[:, k * n:(k + 1) * n].reshape(side, side, n).transpose(2, 1, 0)
        dcubes.append(DataCube(SimpleCube(headers[k], arr), noise, trans_id=t))
    return CubeStack(dcubes), truths, model, data, cubes_hz, ut


def c5r4_cube(ncomp, side=32, n=512, noise=0.1, seed=0):
    """The cube rounds 2-4 timed as "config 5" (bench.py --workload C5r4) and the sampler's evidence bias is measured
    on (tests/test_sampler_bias.py): side x side pixels, NH3 (1,1)+(2,2) on n channels, `ncomp` velocity components
    whose centre moves across the cube and whose column density falls from its middle, normal noise.
    Returns (axes, data[n_pix, 2 n], noise, utrans).  The noise of the two-component cube is the generator's SECOND
    draw (the first belongs to the one-component cube), as in the rounds that used it."""
    import nestfit_amd as na
    from .cube import CubeRunner
    n_pix = side * side
    rng = np.random.default_rng(seed)
    axes = [freq_axis(1, n), freq_axis(2, n)]
    ut = na.get_irdc_priors(size=500, vsys=0.0)
    lon, lat = np.indices((side, side))
    r = np.hypot(lon - side / 2, lat - side / 2) / (side / 2)
    for nc in range(1, ncomp):
        rng.normal(0, noise, (n_pix, 2 * n))                  # (the other cubes' noise draws)
    truths = np.zeros((n_pix, 6 * ncomp))
    for c in range(ncomp):
        truths[:, c] = (-1.0 + 2.0 * lon.ravel() / side) + 1.5 * c
        truths[:, ncomp + c], truths[:, 2 * ncomp + c] = 12.0 + 3 * c, 5.0 + c
        truths[:, 3 * ncomp + c], truths[:, 4 * ncomp + c] = 14.6 - 0.6 * r.ravel(), 0.4
    probe = CubeRunner(axes, (1, 2), np.zeros((1, 2 * n)), np.full((1, 2), noise), ut, ncomp=ncomp)
    model, _ = probe.predict_batch(np.zeros(n_pix, dtype=np.int32), truths)
    return axes, model + rng.normal(0, noise, model.shape), noise, ut
