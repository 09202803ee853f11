"""Host-side mirror of the reference interface: names, shapes, error behaviour."""
import numpy as np
import pytest

import nestfit_amd as na
from nestfit_amd import ammonia, core


def test_module_metadata_matches_reference():
    # nestfit/models/ammonia.pyx:450-489
    assert ammonia.N == 6 and ammonia.IX_VCEN == 0 and ammonia.IX_SIGM == 4
    assert ammonia.NAME == 'ammonia'
    assert ammonia.model_predict is ammonia.amm_predict
    assert ammonia.ModelSpectrum is ammonia.AmmoniaSpectrum
    assert ammonia.ModelRunner is ammonia.AmmoniaRunner
    assert ammonia.PAR_NAMES == ['voff', 'trot', 'tex', 'ntot', 'sigm', 'orth']
    assert ammonia.get_par_names(2) == ['v1', 'v2', 'Tk1', 'Tk2', 'Tx1', 'Tx2', 'N1', 'N2',
                                        's1', 's2', 'o1', 'o2']
    assert ammonia.get_par_names() == ['v', 'Tk', 'Tx', 'N', 's', 'o']
    assert len(ammonia.TEX_LABELS) == len(ammonia.TEX_LABELS_WITH_UNITS) == 6


def test_prior_programs_lower_like_the_reference_sets():
    irdc = na.get_irdc_priors(size=500, vsys=1.0).lower()
    kinds = [p['kind'] for p in irdc['priors']]
    assert kinds == [core.KIND_RESOLVED_PLACEMENT, core.KIND_SIMPLE, core.KIND_SIMPLE,
                     core.KIND_SIMPLE, core.KIND_CONSTANT]
    assert irdc['n_param'] == 6 and len(irdc['dists']) == 5
    p0 = irdc['priors'][0]
    assert p0['p_ix'] == 0 and p0['p_ix2'] == 4 and p0['sub_kind'] == core.KIND_SIMPLE
    assert p0['sep_scale'] == pytest.approx(2.3548200450309493 * 1.2)
    assert irdc['dists'][p0['dist0']]['xmin'] == pytest.approx(-3.0)
    synth = na.get_synth_priors().lower()
    assert [p['kind'] for p in synth['priors']] == [core.KIND_RESOLVED_CENSEP, core.KIND_DUPLICATE,
                                                    core.KIND_SIMPLE, core.KIND_CONSTANT]
    assert synth['n_param'] == 6
    assert synth['priors'][1]['p_ix'] == 1 and synth['priors'][1]['p_ix2'] == 2


def test_distribution_tables():
    x = np.linspace(0, 2, 101)
    d = na.Distribution(x, np.ones_like(x))
    assert d.size == 101 and d.du == pytest.approx(0.01) and d.dx == pytest.approx(0.02)
    assert d.cdf[0] == 0 and d.cdf[-1] == 1
    np.testing.assert_allclose(d.ppf, x, atol=1e-9)
    with pytest.raises(AssertionError):
        na.Distribution(x[::-1], np.ones_like(x))


def test_transform_shape_error_matches_reference_message():
    # nestfit/core/core.pyx:478-483
    ut = na.get_irdc_priors(size=50)
    with pytest.raises(ValueError, match='Invalid shape for ncomp=2: 6'):
        ut.transform(np.full(6, 0.5), 2)
    with pytest.raises(ValueError):
        ut.transform(np.full(6, 0.5, dtype=np.float32), 1)       # double[::1] only


def test_constructor_asserts():
    with pytest.raises(AssertionError):
        na.Prior(None, -1)
    with pytest.raises(AssertionError):
        na.PriorTransformer(np.array([], dtype=object))
    x = np.linspace(23.69e9, 23.70e9, 64)
    with pytest.raises(AssertionError):
        na.AmmoniaSpectrum(x, np.zeros(64), -1.0, 1)             # noise > 0
    with pytest.raises(AssertionError):
        na.AmmoniaSpectrum(x[::-1].copy(), np.zeros(64), 0.1, 1)  # ascending axis
    with pytest.raises(AssertionError):
        na.AmmoniaSpectrum(x, np.zeros(64), 0.1, 0)              # trans_id 1..9
