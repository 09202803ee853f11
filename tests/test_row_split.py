"""The likelihood kernel with 2 or 4 waves per (item, spectrum) unit against the one-wave form: same
model spectra and the same log-likelihoods bit for bit (chi^2 is the sum of four row blocks in a fixed
order whoever computes them), and against the oracle; the automatic choice for small launches."""
import numpy as np
import pytest

from nestfit_amd.synth import TRUTH_2COMP, TRUTH_3COMP, freq_axis

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mode', ['table', 'fast'])
def test_row_split_matches_one_wave_per_unit(engine, nfo, mode):
    from nestfit_amd import _ffi
    engine.set_exp_mode(mode)
    try:
        # (5000 channels = 79 rows: more than one group of 64 rows for the signal-free-row bookkeeping)
        for trans, n, ncomp, truth in (((1, 2), 1024, 2, TRUTH_2COMP), ((1, 2, 3), 200, 3, TRUTH_3COMP), ((9,), 70, 1, None),
                                       ((2, 1), 5000, 2, None)):
            rng = np.random.default_rng(5)
            axes = [freq_axis(t, n) for t in trans]
            spec_data = [[x, rng.normal(0, 0.2, n), 0.2, t] for x, t in zip(axes, trans)]
            ut = engine.get_irdc_priors(size=500, vsys=0.0)
            U = np.random.default_rng(7).uniform(size=(300, 6 * ncomp))
            out = {}
            for split in (1, 2, 4, 0):
                _ffi.set_option('lnl_split', split)
                run = engine.AmmoniaRunner.from_data(spec_data, ut, ncomp=ncomp)
                Us = U.copy()
                lnl = run.loglikelihood_batch(Us)
                spec, lp = run.predict_batch(Us[:32])
                one = run.loglikelihood(U[0].copy())
                out[split] = (lnl, Us, spec, lp, one)
            base = out[1]
            for split in (2, 4, 0):
                lnl, Us, spec, lp, one = out[split]
                assert np.array_equal(Us, base[1]) and np.array_equal(spec, base[2])
                assert np.array_equal(lnl, base[0]) and np.array_equal(lp, base[3])
                assert one == base[4] and one == lnl[0]
            cpu = nfo.AmmoniaRunner([nfo.AmmoniaSpectrum(*sd) for sd in spec_data], nfo.PriorSet(ut.lower()), ncomp=ncomp)
            Uc = U.copy()
            np.testing.assert_allclose(out[4][0], cpu.loglikelihood_batch(Uc), rtol=1e-9 if mode == 'table' else 1e-6)
    finally:
        _ffi.set_option('lnl_split', 0)
        engine.set_exp_mode('fast')


@pytest.mark.parametrize('trans,n,ncomp', [((1, 2), 1024, 2), ((1,), 512, 1), ((1, 2, 3), 512, 3)])
def test_unit_queue_matches_one_unit_per_wave(engine, nfo, trans, n, ncomp):
    """Table mode, launches of two and more units per wave slot of the device: resident workgroups whose waves draw
    the units from a queue (lnl_kernel_queue) against the one-wave-per-unit launch of the same rows -- the same bits,
    twice in a row (the launch leaves its counters zeroed for the next one), theta included; a sample against the
    oracle; spectra out through the same queue."""
    from nestfit_amd import _ffi
    engine.set_exp_mode('table')
    try:
        rng = np.random.default_rng(5)
        axes = [freq_axis(t, n) for t in trans]
        spec_data = [[x, rng.normal(0, 0.2, n), 0.2, t] for x, t in zip(axes, trans)]
        ut = engine.get_irdc_priors(size=500, vsys=0.0)
        # 2 * 512 workgroups * 16 waves = 16384 units per launch at least, and not a multiple of the queue's chunks of
        # 16 units (the host call sends 16384 rows and more as four launches)
        B = 16384 // len(trans) + 16 * 40 + 1 if len(trans) > 1 else 4 * (16384 + 16 * 10) + 4
        U = np.random.default_rng(11).uniform(size=(B, 6 * ncomp))
        out = {}
        for q in (0, 1, 1):
            _ffi.set_option('lnl_queue', q)
            run = engine.AmmoniaRunner.from_data(spec_data, ut, ncomp=ncomp)
            Us = U.copy()
            lnl = run.loglikelihood_batch(Us)
            lnl2 = run.loglikelihood_batch(U.copy())
            assert np.array_equal(lnl, lnl2)
            out.setdefault(q, []).append((lnl, Us))
        for lnl, Us in out[1]:
            assert np.array_equal(lnl, out[0][0][0]) and np.array_equal(Us, out[0][0][1])
        assert np.all(np.isfinite(out[1][0][0]))
        cpu = nfo.AmmoniaRunner([nfo.AmmoniaSpectrum(*sd) for sd in spec_data], nfo.PriorSet(ut.lower()), ncomp=ncomp)
        pick = np.random.default_rng(3).choice(B, 200, replace=False)
        Uc = U[pick].copy()
        np.testing.assert_allclose(out[1][0][0][pick], cpu.loglikelihood_batch(Uc), rtol=1e-9)
        if n <= 512:
            theta = out[0][0][1]
            spec = {}
            for q in (0, 1):
                _ffi.set_option('lnl_queue', q)
                run = engine.AmmoniaRunner.from_data(spec_data, ut, ncomp=ncomp)
                spec[q] = run.predict_batch(theta)
            assert np.array_equal(spec[0][0], spec[1][0]) and np.array_equal(spec[0][1], spec[1][1])
    finally:
        _ffi.set_option('lnl_queue', 1)
        engine.set_exp_mode('fast')
