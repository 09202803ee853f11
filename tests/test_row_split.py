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
