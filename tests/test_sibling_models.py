"""Sibling models on the same kernels: N2H+ (nestfit/models/diazenylium.pyx)
and Gaussian (nestfit/models/gaussian.pyx).

CPU part: line data against the reference text, oracle sanity against closed forms.
GPU part: engine against the oracle through the C ABI, all three numerical modes.
There are no reference outputs for these two models among the survey's known answers, so the
oracle is pinned for them only through the shared pieces (FastExp, iemtex, windows) and the
literal line data: "parity unpinned beyond the restatement" (DESIGN.md section 2).
"""
import re
from pathlib import Path

import numpy as np
import pytest

CKMS = 299792.458
N2HP_NU = {1: 93173.7637e6, 2: 186344.8420e6, 3: 279511.8325e6}
REF = Path('/root/reference/nestfit/models/diazenylium.pyx')


def n2hp_axis(trans, n, vhalf=20.0):
    v = np.linspace(vhalf, -vhalf, n)
    return N2HP_NU[trans] * (1.0 - v / CKMS)


def _floats(txt):
    txt = re.sub(r'#.*', '', txt)
    return [float(v) for v in re.findall(r'[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?', txt)]


def _block(text, start, end=']'):
    a = text.index(start) + len(start)
    return text[a:text.index(end, a)]


# ---------------------------------------------------------------------------- CPU
@pytest.mark.skipif(not REF.exists(), reason='reference tree not present')
def test_n2hp_line_data_matches_reference_text(nfo):
    text = REF.read_text()
    lib = nfo.lib()
    nhf = [int(v) for v in _floats(_block(text, 'NHF = ['))]
    assert nhf == [lib.nfo_n2hp_nhf(t) for t in (1, 2, 3)] == [15, 40, 45]
    assert _floats(_block(text, 'NU = [')) == [lib.nfo_n2hp_nu(t) for t in (1, 2, 3)]
    for t in (1, 2, 3):
        v = _floats(_block(text, f'VOFF[{t-1}][:NHF[{t-1}]] = ['))
        w = _floats(_block(text, f'TAU_WTS[{t-1}][:NHF[{t-1}]] = ['))
        assert len(v) == len(w) == nhf[t - 1]
        assert v == [lib.nfo_n2hp_voff(t, i) for i in range(len(v))]
        assert w == [lib.nfo_n2hp_tau_wt(t, i) for i in range(len(w))]


def test_n2hp_weights_are_normalised(nfo):
    lib = nfo.lib()
    for t in (1, 2, 3):
        tot = sum(lib.nfo_n2hp_tau_wt(t, i) for i in range(lib.nfo_n2hp_nhf(t)))
        assert tot == pytest.approx(1.0, abs=2e-3)


def test_oracle_gaussian_against_closed_form(nfo):
    """peak * exp(-(nu - nu_cen)^2 / (2 width^2)) inside the exp(-12.5) window, zero outside
    (gaussian.pyx:29-50); FastExp is good to ~1e-7 relative."""
    nu0 = 110.2e9
    n = 777
    x = nu0 * (1.0 - np.linspace(25, -25, n) / CKMS)
    s = nfo.Spectrum(x, np.zeros(n), 0.1, rest_freq=nu0)
    th = np.array([3.0, -7.5, 0.9, 0.35, 2.0, -1.25])           # 2 comps: voff, sigm, peak
    nfo.gauss_predict(s, th)
    want = np.zeros(n)
    for c in range(2):
        width = th[2 + c] / CKMS * nu0
        cen = nu0 * (1 - th[c] / CKMS)
        arg = (x - cen) ** 2 * 0.5 / width ** 2
        cut = np.sqrt(12.5 / (0.5 / width ** 2))
        lo = int(np.floor((cen - x[0] - cut) / (x[1] - x[0])))
        hi = int(np.floor((cen - x[0] + cut) / (x[1] - x[0])))
        lo, hi = max(lo, 0), min(hi, n - 1)
        # FastExp takes its argument as a float (core/math.pxd:17)
        want[lo:hi] += th[4 + c] * np.exp(-arg[lo:hi].astype(np.float32).astype(np.float64))
    got = s.get_spec()
    assert np.array_equal(got == 0, want == 0)
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=0)
    assert s.loglikelihood == pytest.approx(-np.sum(want ** 2) / (2 * 0.1 ** 2), rel=1e-6)


def test_oracle_n2hp_thin_limit_and_saturation(nfo):
    """Optically thin: integrated Tb = (J(tex) - J(tbg)) * tau_main * sum(w) * sqrt(2 pi) sigma_nu / chan;
    very thick: line core saturates at J(tex) - J(tbg)."""
    n = 4096
    x = n2hp_axis(1, n, 25.0)
    s = nfo.DiazenyliumSpectrum(x, np.zeros(n), 0.1, trans_id=1)
    tex, sigm = 7.5, 0.4
    T0 = 6.62607015e-27 * N2HP_NU[1] / 1.380649e-16
    jdiff = T0 * (1 / np.expm1(T0 / tex) - 1 / np.expm1(T0 / 2.72548))
    nfo.nnhp_predict(s, np.array([0.0, tex, -4.0, sigm]))
    chan = x[1] - x[0]
    sig_nu = sigm / CKMS * N2HP_NU[1]
    wsum = sum(nfo.lib().nfo_n2hp_tau_wt(1, i) for i in range(15))
    want = jdiff * 1e-4 * wsum * np.sqrt(2 * np.pi) * sig_nu / chan
    assert s.sum_spec == pytest.approx(want, rel=2e-3)
    nfo.nnhp_predict(s, np.array([0.0, tex, 3.0, sigm]))
    assert s.max_spec == pytest.approx(jdiff, rel=1e-3)


def test_host_mirror_metadata():
    """Module-level aliases of the reference's model modules (diazenylium.pyx:234-264,
    gaussian.pyx:115-150, models/__init__.py:3-7)."""
    import nestfit_amd as na
    assert set(na.MODELS) == {'ammonia', 'diazenylium', 'gaussian'}
    d, g = na.MODELS['diazenylium'], na.MODELS['gaussian']
    assert (d.N, d.IX_VCEN, d.IX_SIGM, d.PAR_NAMES) == (4, 0, 3, ['voff', 'tex', 'ltau', 'sigm'])
    assert (g.N, g.IX_VCEN, g.IX_SIGM, g.PAR_NAMES) == (3, 0, 1, ['voff', 'sigm', 'peak'])
    assert d.get_par_names(2) == ['v1', 'v2', 'Tx1', 'Tx2', 'lt1', 'lt2', 's1', 's2']
    assert g.get_par_names() == ['v', 's', 'pk']
    assert d.ModelRunner is na.DiazenyliumRunner and g.ModelRunner is na.GaussianRunner
    assert d.model_predict is na.nnhp_predict and g.model_predict is na.gauss_predict


# ---------------------------------------------------------------------------- GPU
MODES = ['table', 'fast']
TB_RTOL = 1e-6
TB_ATOL_K = {'table': 0.0, 'fast': 4e-15}
TIGHT = {'table': 1e-11, 'fast': 5e-7}
LNL_RTOL = {'table': 1e-9, 'fast': 1e-6}


@pytest.fixture
def mode_guard(engine):
    yield
    engine.set_exp_mode('fast')


def _check_spec(pg, pc, mode, scale=None):
    """Zero pattern exact; values within 1e-6 of `scale` (default: the oracle value itself)."""
    assert np.array_equal(pg == 0, pc == 0)
    scale = np.abs(pc) if scale is None else scale
    nz = pc != 0
    worst = 0.0
    if nz.any():
        assert (np.abs(pg - pc) <= TB_RTOL * scale + TB_ATOL_K[mode]).all()
        big = scale > 1e-6
        if big.any():
            worst = float(np.max(np.abs(pg[big] - pc[big]) / scale[big]))
    return worst


@pytest.mark.gpu
def test_n2hp_window_indices_bit_exact(engine, nfo, mode_guard):
    from nestfit_amd import _ffi
    rng = np.random.default_rng(5)
    for trans in (1, 2, 3):
        for n in (300, 1024):
            x = n2hp_axis(trans, n)
            sg = engine.DiazenyliumSpectrum(x, np.zeros(n), 0.1, trans)
            sc = nfo.DiazenyliumSpectrum(x, np.zeros(n), 0.1, trans)
            run = sg._runner(1)
            nhf = nfo.lib().nfo_n2hp_nhf(trans)
            for _ in range(40):
                voff, sigm = rng.uniform(-25, 25), 10 ** rng.uniform(-2, 0.5)
                lo = np.zeros(64, dtype=np.int32)
                hi = np.zeros(64, dtype=np.int32)
                _ffi.test_check(_ffi.test_engine().nfa_test_windows(run.handle, 0, voff, sigm,
                                                        lo.ctypes.data_as(_ffi._ip),
                                                        hi.ctypes.data_as(_ffi._ip)))
                clo, chi = sc.hf_windows(voff, sigm)
                skipped = clo < 0
                assert np.array_equal(lo[:nhf][~skipped], clo[~skipped])
                assert np.array_equal(hi[:nhf][~skipped], chi[~skipped])
                assert (lo[:nhf][skipped] == hi[:nhf][skipped]).all()


@pytest.mark.gpu
@pytest.mark.parametrize('mode', MODES)
def test_nnhp_predict_grid(engine, nfo, mode, mode_guard):
    engine.set_exp_mode(mode)
    rng = np.random.default_rng(41)
    worst = 0.0
    for trans in (1, 2, 3):
        for n in (300, 1024, 2048):
            x = n2hp_axis(trans, n, 30.0 if n == 2048 else 20.0)
            data = rng.normal(0, 0.2, n)
            sg = engine.DiazenyliumSpectrum(x, data, 0.2, trans)
            sc = nfo.DiazenyliumSpectrum(x, data, 0.2, trans)
            assert sg.null_lnZ == pytest.approx(sc.null_lnZ, rel=1e-13)
            np.testing.assert_allclose(sg.tbg_arr, sc.tbg_arr, rtol=1e-14)
            for ncomp in (1, 2, 3):
                for _ in range(6):
                    th = np.concatenate([rng.uniform(-8, 8, ncomp), rng.uniform(2.8, 25, ncomp),
                                         rng.uniform(-2, 1.5, ncomp), 10 ** rng.uniform(-1.3, 0.3, ncomp)])
                    engine.nnhp_predict(sg, th)
                    nfo.nnhp_predict(sc, th)
                    worst = max(worst, _check_spec(sg.get_spec(), sc.get_spec(), mode))
                    assert sg.loglikelihood == pytest.approx(sc.loglikelihood, rel=LNL_RTOL[mode])
    print(f'n2hp {mode}: worst relative Tb error {worst:.2e}')
    assert worst < TIGHT[mode]


@pytest.mark.gpu
@pytest.mark.parametrize('mode', MODES)
def test_gauss_predict_grid(engine, nfo, mode, mode_guard):
    from nestfit_amd import gaussian
    engine.set_exp_mode(mode)
    rng = np.random.default_rng(43)
    worst = 0.0
    for nu0 in (23.6944955e9, 110.201354e9, 345.7959899e9):
        for n in (65, 300, 1024, 3000):
            x = nu0 * (1.0 - np.linspace(40, -40, n) / CKMS)
            data = rng.normal(0, 0.5, n)
            sg = gaussian.Spectrum(x, data, 0.5, rest_freq=nu0)
            sc = nfo.Spectrum(x, data, 0.5, rest_freq=nu0)
            assert sg.null_lnZ == pytest.approx(sc.null_lnZ, rel=1e-13)
            for ncomp in (1, 2, 4):
                for _ in range(6):
                    th = np.concatenate([rng.uniform(-45, 45, ncomp), 10 ** rng.uniform(-1.5, 1.0, ncomp),
                                         rng.uniform(-2, 8, ncomp)])
                    # components of opposite sign cancel: the forward-error scale of the sum is
                    # sum_c |peak_c e_c| (the oracle with |peak|), not the cancelled value
                    nfo.gauss_predict(sc, np.concatenate([th[:2 * ncomp], np.abs(th[2 * ncomp:])]))
                    scale = sc.get_spec()
                    engine.gauss_predict(sg, th)
                    nfo.gauss_predict(sc, th)
                    worst = max(worst, _check_spec(sg.get_spec(), sc.get_spec(), mode, scale))
                    assert sg.loglikelihood == pytest.approx(sc.loglikelihood, rel=LNL_RTOL[mode])
    print(f'gauss {mode}: worst relative error {worst:.2e}')
    assert worst < TIGHT[mode]


def _simple_priors(engine, ranges, size=200):
    from scipy import stats
    x = np.linspace(0, 1, size)
    return engine.PriorTransformer([
        engine.Prior(engine.Distribution(lo + x * (hi - lo), stats.uniform(lo, hi - lo).pdf(lo + x * (hi - lo))), k)
        for k, (lo, hi) in enumerate(ranges)])


@pytest.mark.gpu
@pytest.mark.parametrize('mode', MODES)
def test_sibling_runners_with_priors(engine, nfo, mode, mode_guard):
    """c_loglikelihood of DiazenyliumRunner / GaussianRunner (diazenylium.pyx:207-216,
    gaussian.pyx:96-100): unit cube in, theta (in place) and lnL out."""
    engine.set_exp_mode(mode)
    rng = np.random.default_rng(47)
    # N2H+: two transitions of one pixel, 2 components
    ut = _simple_priors(engine, [(-6, 6), (2.8, 20), (-1.5, 1.0), (0.1, 1.5)])
    ps = nfo.PriorSet(ut.lower())
    ncomp = 2
    args = []
    truth = np.array([-1.0, 2.0, 8.0, 5.0, 0.3, -0.2, 0.4, 0.7])
    for trans, n in ((1, 700), (2, 1024)):
        x = n2hp_axis(trans, n)
        sc = nfo.DiazenyliumSpectrum(x, np.zeros(n), 0.15, trans)
        nfo.nnhp_predict(sc, truth)
        args.append([x, sc.get_spec() + rng.normal(0, 0.15, n), 0.15, trans])
    rg = engine.DiazenyliumRunner.from_data(args, ut, ncomp=ncomp)
    rc = nfo.DiazenyliumRunner([nfo.DiazenyliumSpectrum(*a) for a in args], ps, ncomp=ncomp)
    assert (rg.ndim, rg.n_params, rg.n_spec, rg.n_chan_tot) == (8, 8, 2, 1724)
    assert rg.null_lnZ == pytest.approx(rc.null_lnZ, rel=1e-13)
    U = rng.uniform(size=(513, 8))
    Ug, Uc = U.copy(), U.copy()
    lg, lc = rg.loglikelihood_batch(Ug), rc.loglikelihood_batch(Uc)
    np.testing.assert_allclose(Ug, Uc, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(lg, lc, rtol=LNL_RTOL[mode])
    u1 = U[7].copy()
    assert rg.loglikelihood(u1) == pytest.approx(lc[7], rel=LNL_RTOL[mode])
    np.testing.assert_allclose(u1, Uc[7], rtol=1e-12, atol=1e-13)
    # one point and a handful go through the point kernel (N2H+ (1-0) has 15 lines; transitions with more than
    # 26 take the batch kernels): the bits of the 513-row batch either way
    assert rg.loglikelihood(U[7].copy()) == lg[7] and np.array_equal(u1, Ug[7])
    few = U[20:31].copy()
    assert np.array_equal(rg.loglikelihood_batch(few), lg[20:31]) and np.array_equal(few, Ug[20:31])
    with pytest.raises(ValueError, match='Invalid shape for ncomp=2'):
        rg.loglikelihood(np.zeros(6))
    with pytest.raises(ValueError, match='Invalid shape for ncomp=2'):
        rg.predict(np.zeros(12))
    rg.predict(truth)
    rc.predict(truth)
    for sg, sc in zip(rg.spectra, rc.spectra):
        _check_spec(sg.get_spec(), sc.get_spec(), mode)
    # Gaussian: one spectrum, 3 components
    from nestfit_amd import gaussian
    nu0 = 110.201354e9
    n = 1500
    x = nu0 * (1.0 - np.linspace(30, -30, n) / CKMS)
    utg = _simple_priors(engine, [(-20, 20), (0.2, 3.0), (0.0, 5.0)])
    psg = nfo.PriorSet(utg.lower())
    sc = nfo.Spectrum(x, np.zeros(n), 0.3, rest_freq=nu0)
    nfo.gauss_predict(sc, np.array([-5.0, 0.0, 6.0, 1.0, 0.5, 2.0, 3.0, 1.5, 0.8]))
    data = sc.get_spec() + rng.normal(0, 0.3, n)
    gg = gaussian.GaussianRunner.from_data([x, data, 0.3, nu0], utg, ncomp=3)
    gc = nfo.GaussianRunner(nfo.Spectrum(x, data, 0.3, rest_freq=nu0), psg, ncomp=3)
    assert (gg.ndim, gg.n_spec, gg.n_chan_tot) == (9, 1, n)
    U = rng.uniform(size=(300, 9))
    Ug, Uc = U.copy(), U.copy()
    lg, lc = gg.loglikelihood_batch(Ug), gc.loglikelihood_batch(Uc)
    np.testing.assert_allclose(Ug, Uc, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(lg, lc, rtol=LNL_RTOL[mode])
    ug = U[3].copy()
    assert gg.loglikelihood(ug) == lg[3] and np.array_equal(ug, Ug[3])           # point kernel, Gaussian model
    few = U[40:47].copy()
    assert np.array_equal(gg.loglikelihood_batch(few), lg[40:47]) and np.array_equal(few, Ug[40:47])


@pytest.mark.gpu
def test_sibling_model_constructor_errors(engine, nfo, mode_guard):
    from nestfit_amd import gaussian
    x = n2hp_axis(1, 64)
    with pytest.raises(AssertionError):
        engine.DiazenyliumSpectrum(x, np.zeros(64), 0.1, trans_id=4)         # diazenylium.pyx:128
    with pytest.raises(AssertionError):
        engine.DiazenyliumSpectrum(x, np.zeros(64), 0.0, trans_id=1)         # core.pyx:502
    with pytest.raises(AssertionError):
        gaussian.Spectrum(x[::-1].copy(), np.zeros(64), 0.1, rest_freq=N2HP_NU[1])   # core.pyx:504
    ut6 = engine.get_irdc_priors()
    s = engine.DiazenyliumSpectrum(x, np.zeros(64), 0.1, 1)
    with pytest.raises(engine.EngineError, match='prior program'):
        engine.DiazenyliumRunner([s], ut6, ncomp=1)
    # rest_freq = None -> 0 like the reference (core.pyx:510): every window is empty, pred = 0
    g = gaussian.Spectrum(x, np.ones(64), 0.1)
    engine.gauss_predict(g, np.array([0.0, 1.0, 3.0]))
    assert not g.get_spec().any()
