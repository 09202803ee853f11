"""nestfit_amd -- MI355X-native NH3 hyperfine log-likelihood engine.

Drop-in for the hot path of autocorr/nestfit (``AmmoniaRunner.loglikelihood`` /
prior transform / ``amm_predict``); see DESIGN.md and INTEGRATION.md.
"""
from . import _ffi
from ._ffi import EngineError, device_count, get_exp_mode, pinned_empty, set_device, set_exp_mode
from .core import (ConstantPrior, CenSepPrior, Distribution, DuplicatePrior, OrderedPrior, Prior,
                   PriorTransformer, ResolvedCenSepPrior, ResolvedPlacementPrior, SpacedPrior)
from .ammonia import AmmoniaRunner, AmmoniaSpectrum, amm_predict
from .diazenylium import DiazenyliumRunner, DiazenyliumSpectrum, nnhp_predict
from .gaussian import GaussianRunner, gauss_predict
from . import ammonia, diazenylium, gaussian

# registry like nestfit/models/__init__.py:3-7
MODELS = {m.NAME: m for m in (ammonia, diazenylium, gaussian)}
from .prior_constructors import get_irdc_priors, get_synth_priors

__all__ = [
    'EngineError', 'device_count', 'set_device', 'set_exp_mode', 'get_exp_mode', 'pinned_empty',
    'Distribution', 'Prior', 'ConstantPrior', 'DuplicatePrior', 'OrderedPrior', 'SpacedPrior',
    'CenSepPrior', 'ResolvedCenSepPrior', 'ResolvedPlacementPrior', 'PriorTransformer',
    'AmmoniaSpectrum', 'AmmoniaRunner', 'amm_predict', 'get_irdc_priors', 'get_synth_priors',
    'DiazenyliumSpectrum', 'DiazenyliumRunner', 'nnhp_predict', 'GaussianRunner', 'gauss_predict',
    'MODELS',
]
