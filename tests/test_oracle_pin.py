"""Pins the CPU oracle: (i) FastExp bit-for-bit against the reference's own
fastexp.c compiled as-is (oracle/_ref), (ii) every known answer the survey
captured from the compiled reference (tests/golden/survey_kat.json), (iii) the
reference's in-module tests (swift_convert, iemtex_interp, Distribution)."""
import ctypes as C

import numpy as np
import pytest

from nestfit_amd.synth import freq_axis


def _edge_floats():
    """Float inputs around every branch of FastExp (fastexp.c:259-273)."""
    f32 = np.float32
    xs = [0.0, -0.0, 1e-9, 1e-30, 1e-40, 0.01, 0.1, 1.0, 12.5, 31.9, 32.0, 40.0, 1e30,
          -1.0, -0.5, -20.0]
    for p in range(-8, 7):
        b = f32(2.0) ** f32(p)
        xs += [b, np.nextafter(b, f32(0)), np.nextafter(b, f32(100))]
    return np.array(xs, dtype=np.float32)


def test_fastexp_matches_reference_source_bit_for_bit(nfo):
    ref = nfo.ref_fastexp_lib()
    if ref is None:
        pytest.skip('oracle/_ref not built (reference tree absent)')
    rng = np.random.default_rng(1)
    # uniform in bit pattern over [2^-8, 64) plus edges: every (l, j0, j1, j2) family
    bits = rng.integers(np.float32(2.0**-8).view(np.uint32), np.float32(64.0).view(np.uint32),
                        size=400_000, dtype=np.uint32)
    x = np.concatenate([bits.view(np.float32), _edge_floats(),
                        rng.uniform(0, 13, 100_000).astype(np.float32)])
    mine = nfo.fastexp(x)
    theirs = np.array([ref.FastExp(C.c_float(float(v))) for v in x])
    assert np.array_equal(mine.view(np.uint64), theirs.view(np.uint64))


def test_fastexp_golden_set_g1(nfo):
    """4096 FastExp values recorded from the reference's own fastexp.c
    (tests/golden/make_g1_fastexp.py); needs no reference tree, so it also pins the oracle on the GPU box."""
    from pathlib import Path
    g = np.load(Path(__file__).parent / 'golden' / 'g1_fastexp.npz')
    x, y = g['x'], g['y']
    assert x.dtype == np.float32 and x.size == 4096 and (y == 0).sum() > 100 and (x < 2.0**-5).sum() > 10
    assert np.array_equal(nfo.fastexp(x).view(np.uint64), y.view(np.uint64))
    ref = nfo.ref_fastexp_lib()
    if ref is not None:                           # where the reference is present the fixture is re-derived
        again = np.array([ref.FastExp(C.c_float(float(v))) for v in x])
        assert np.array_equal(again.view(np.uint64), y.view(np.uint64))


def test_fastexp_index_fields_match_reference_layout(nfo):
    # x = (128+j0) 2^(l-12) + j1 2^(l-20) + j2 2^(l-28)  (fastexp.c:239-283)
    rng = np.random.default_rng(2)
    for _ in range(2000):
        l, j0, j1, j2 = rng.integers(0, 10), rng.integers(0, 128), rng.integers(0, 256), rng.integers(0, 256)
        x = np.float32((128 + j0) * 2.0**(l - 12) + j1 * 2.0**(l - 20) + j2 * 2.0**(l - 28))
        assert nfo.fastexp_indices(x) == (l, j0, j1, j2)
    assert nfo.fastexp_indices(np.float32(0.03125))[0] == 0
    assert nfo.fastexp_indices(np.nextafter(np.float32(0.03125), np.float32(0)))[0] == -1
    assert nfo.fastexp_indices(np.float32(32.0))[0] == 10


def test_fastexp_reference_flag_noise_is_tiny(nfo):
    ref = nfo.ref_fastexp_lib(fast_math=True)
    if ref is None:
        pytest.skip('oracle/_ref not built (reference tree absent)')
    x = np.random.default_rng(3).uniform(0, 33, 50_000).astype(np.float32)
    mine = nfo.fastexp(x)
    theirs = np.array([ref.FastExp(C.c_float(float(v))) for v in x])
    ok = theirs != 0
    assert np.array_equal(mine == 0, theirs == 0)
    assert np.max(np.abs(mine[ok] - theirs[ok]) / theirs[ok]) < 1e-15


def test_fastexp_survey_known_answers(nfo, kat):
    got = nfo.fast_expn(np.array(kat['fastexp']['x']))
    np.testing.assert_allclose(got, kat['fastexp']['out'], rtol=2e-16, atol=0)
    # FastExp(x) is exp(-(float)x) to 4.07e-8 (SURVEY hard part 1)
    x = np.random.default_rng(4).uniform(0, 31.9, 100_000)
    ref = np.exp(-x.astype(np.float32).astype(np.float64))
    assert np.max(np.abs(nfo.fast_expn(x) - ref) / ref) < 4.1e-8


def test_iemtex_known_answers_and_reference_test(nfo, kat):
    got = nfo.iemtex_interp(np.array(kat['iemtex_interp']['x']))
    np.testing.assert_allclose(got, kat['iemtex_interp']['out'], rtol=1e-15)
    # reference's own test (nestfit/models/hyperfine.pyx:147-152): max rel err < 1e-5
    lo, hi = nfo.lib().nfo_t0_xmin(), nfo.lib().nfo_t0_xmax()
    fine = np.linspace(lo, hi, 100_000)
    exact = 1 / (np.exp(fine) - 1)
    diffs = np.abs((nfo.iemtex_interp(fine) - exact) / exact)
    np.testing.assert_almost_equal(diffs.max(), 0, decimal=5)
    assert nfo.iemtex_index(lo) == -1 and nfo.iemtex_index(hi) == -1     # exact branch at the ends
    assert nfo.iemtex_index(np.nextafter(lo, 1)) == 0
    assert nfo.iemtex_index(np.nextafter(hi, 0)) in (998, 999)


def test_partition_and_swift_known_answers(nfo, kat):
    for c in kat['partition_func']:
        assert nfo.partition_func(c['para'], c['trot']) == pytest.approx(c['out'], rel=1e-15)
    for c in kat['partition_level']:
        assert nfo.partition_level(c['j'], c['trot']) == pytest.approx(c['out'], rel=1e-15, abs=0)
    # reference test_swift_convert (nestfit/models/ammonia.pyx:517-521)
    np.testing.assert_almost_equal(nfo.swift_convert(15), kat['swift_convert']['out'], decimal=8)


@pytest.mark.parametrize('idx', range(8))
def test_spectra_known_answers(nfo, kat, idx):
    c = kat['spectra'][idx]
    x = freq_axis(c['trans_id'], c['n_chan'], c['vhalf'])
    s = nfo.AmmoniaSpectrum(x, np.zeros(c['n_chan']), c['noise'], c['trans_id'])
    nfo.amm_predict(s, np.array(c['params'], dtype=float))
    p = s.get_spec()
    assert int((p != 0).sum()) == c['nnz']                 # window support: exact
    assert p[-1] == 0.0                                    # last channel never receives tau
    assert p.max() == pytest.approx(c['max'], rel=1e-12)
    if 'sum' in c:
        assert p.sum() == pytest.approx(c['sum'], rel=1e-12)
    assert s.loglikelihood == pytest.approx(c['lnL'], rel=1e-12)


def test_runner_irdc_known_answer(nfo, kat):
    import nestfit_amd as na
    c = kat['runner_irdc']
    ps = nfo.PriorSet(na.get_irdc_priors(size=500, vsys=0.0).lower())
    for n_chan, lnl_ref in c['lnL'].items():
        n = int(n_chan)
        spectra = [nfo.AmmoniaSpectrum(freq_axis(t, n), np.zeros(n), c['noise'], t) for t in (1, 2)]
        run = nfo.AmmoniaRunner(spectra, ps, ncomp=c['ncomp'])
        u = np.full(12, c['u'])
        lnl = run.loglikelihood(u)
        np.testing.assert_allclose(u, c['theta'], rtol=1e-13, atol=0)
        assert lnl == pytest.approx(lnl_ref, rel=1e-12)
        assert run.null_lnZ == 0.0


def test_distribution_reference_test():
    # reference test_distribution (nestfit/core/core.pyx:830-839)
    import nestfit_amd as na
    x = np.linspace(-4, 4, 201)
    d = na.Distribution(x, np.exp(-0.5 * x**2))
    eps = 1e-15
    assert abs(d.ppf[100]) < eps
    from oracle import nfo as o
    ps = o.PriorSet(na.PriorTransformer(np.array([na.Prior(d, 0)])).lower())
    assert abs(o.lib().nfo_dist_ppf_interp(ps._dists[0], 0.5)) < eps
    assert abs(o.lib().nfo_dist_cdf_interp(ps._dists[0], 0.5)) < eps


def test_edge_cases_of_the_model(nfo):
    n = 256
    x = freq_axis(1, n)
    s = nfo.AmmoniaSpectrum(x, np.zeros(n), 0.1, 1)
    # line fully off band: pred == 0 everywhere
    nfo.amm_predict(s, np.array([500.0, 10, 4, 14.5, 0.3, 0]))
    assert not s.get_spec().any()
    # sigma << channel: lower-edge FastExp underflow leaves the included channel at 0
    nfo.amm_predict(s, np.array([-1.0, 10, 4, 14.5, 0.005, 0]))
    lo, hi = s.hf_windows(-1.0, 0.005)
    assert ((hi - lo) <= 2).all()
    # ortho transition with orth = 0: tau_main = 0 -> log10 -> -inf -> pred == 0
    s3 = nfo.AmmoniaSpectrum(freq_axis(3, n), np.zeros(n), 0.1, 3)
    nfo.amm_predict(s3, np.array([-1.0, 10, 4, 14.5, 0.3, 0.0]))
    assert not s3.get_spec().any()
    # constructor asserts (core.pyx:502-504, ammonia.pyx:268)
    with pytest.raises(AssertionError):
        nfo.AmmoniaSpectrum(x[::-1].copy(), np.zeros(n), 0.1, 1)
    with pytest.raises(AssertionError):
        nfo.AmmoniaSpectrum(x, np.zeros(n), 0.0, 1)
    with pytest.raises(AssertionError):
        nfo.AmmoniaSpectrum(x, np.zeros(n), 0.1, 10)


def test_partition_function_reference_known_answers_old_constants(nfo):
    """The reference's own partition-function test (nestfit/models/ammonia.pyx:496-514): its three
    numbers come from pyspeckit with the Poynter & Kakar rotation constants and only hold when the
    module is compiled with __NEW_CONST = False.  The restatement built the same way
    (-DNFA_OLD_CONST) meets them to the reference's own precision (decimal=7); with the shipped
    constants it must NOT (that is why the reference skips its test)."""
    old = nfo.oldconst_lib()
    np.testing.assert_almost_equal(old.nfo_partition_level(1, 10.0), 0.29279893434489096, decimal=7)
    np.testing.assert_almost_equal(old.nfo_partition_level(2, 10.0), 0.007933862262432792, decimal=7)
    np.testing.assert_almost_equal(old.nfo_partition_func(1, 10.0), 0.30073281405688107, decimal=7)
    new = nfo.lib()
    assert abs(new.nfo_partition_level(1, 10.0) - 0.29279893434489096) > 1e-5
    assert abs(new.nfo_partition_func(1, 10.0) - 0.30073281405688107) > 1e-5


def test_reference_spectrum_fixture_reads_as_model_axis():
    """nestfit/test/data/test_spectrum_11.fits (kept under tests/golden/): a 380-channel VLA NH3 (1,1)
    profile on a descending FREQ axis.  `cubeio.read_spectrum` hands it over ascending, on the grid
    of the reference's 20 x 20 x 380 test cube (same CDELT / RESTFRQ as test/data/ammonia_11_cutout.fits)."""
    from pathlib import Path
    from nestfit_amd.cubeio import SimpleCube, read_spectrum
    gold = Path(__file__).parent / 'golden'
    x, d, hdr = read_spectrum(gold / 'test_spectrum_11.fits')
    assert x.shape == d.shape == (380,) and np.all(np.diff(x) > 0) and np.isfinite(d).all()
    assert float(hdr['RESTFRQ']) == 23.6944955e9
    assert np.diff(x) == pytest.approx(1.249826974487e4, rel=1e-9)
    assert x[-1] == pytest.approx(2.368986555182e10, rel=1e-12)          # CRVAL1 at CRPIX1 = 1, now the last channel
    cube = SimpleCube.read(gold / 'ammonia_11_cutout.fits')
    assert abs(np.diff(cube.spectral_axis_hz())).mean() == pytest.approx(1.249826974487e4, rel=2e-4)
    assert 0.02 < d.max() < 0.05 and abs(np.median(d)) < 0.005           # a line profile (Jy/beam) on a flat baseline
