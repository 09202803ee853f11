"""Host-side mirror of ``nestfit.models.ammonia`` (reference:
nestfit/models/ammonia.pyx:244-489): same names, arguments and error
behaviour; all arithmetic runs in the HIP engine through the C ABI.
"""
import numpy as np

from ._model import (MODEL_AMMONIA, EngineRunner, EngineSpectrumMixin, _pix_ptr, _RunnerHandle,  # noqa: F401
                     _SpecSet, par_names)
from .core import HyperfineSpectrum

N_LEVELS = 9
N_PARAMS = 6


class AmmoniaSpectrum(EngineSpectrumMixin, HyperfineSpectrum):
    """NH3 (J,K) inversion spectrum (reference: ammonia.pyx:244-277).

    Parameters
    ----------
    xarr : array, Hz, ascending
    data : array, K
    noise : number, K
    trans_id : 1 -> (1,1) ... 9 -> (9,9)
    """
    MODEL = MODEL_AMMONIA

    def __init__(self, xarr, data, noise, trans_id=1):
        assert trans_id in range(1, N_LEVELS + 1)
        HyperfineSpectrum.__init__(self, xarr, data, noise, rest_freq=0.0, trans_id=trans_id)
        self._attach(trans_id)

    @property
    def tbg_arr(self):
        return self._ss.tbg()


def amm_predict(s, params, cold=False, lte=False):
    """Model spectrum of `s` for parameter-major `params` (reference:
    ammonia.pyx:326-366); result in ``s.get_spec()`` / ``s.loglikelihood``."""
    s._predict(params, N_PARAMS, cold, lte)


class AmmoniaRunner(EngineRunner):
    """Prior transform + model + log-likelihood for one pixel's spectra
    (reference: ammonia.pyx:369-447).

    Parameters
    ----------
    spectra : sequence of AmmoniaSpectrum
    utrans : PriorTransformer
    ncomp : int, number of velocity components
    cold : bool, Swift et al. (2005) Tkin -> Trot
    lte : bool, Tex = Trot
    """
    MODEL = MODEL_AMMONIA
    N_MODEL = N_PARAMS

    def __init__(self, spectra, utrans, ncomp=1, cold=False, lte=False):
        assert ncomp > 0
        self.spectra = list(spectra)
        self.cold = bool(cold)
        self.lte = bool(lte)
        self._setup(self.spectra, utrans, ncomp, self.cold, self.lte)

    @classmethod
    def from_data(cls, spec_data, utrans, **kwargs):
        spectra = np.array([AmmoniaSpectrum(*args) for args in spec_data])
        return cls(spectra, utrans, **kwargs)

    def get_spectra(self):
        return np.array(self.spectra)

    def predict(self, params):
        """Model spectra for physical `params` into every spectrum of the runner
        (reference: ammonia.pyx:437-447)."""
        params = self._check_params(params)
        for s in self.spectra:
            amm_predict(s, params, self.cold, self.lte)


# Aliases and metadata at module scope (reference: ammonia.pyx:450-489)
N = N_PARAMS
IX_VCEN = 0
IX_SIGM = 4
NAME = 'ammonia'
model_predict = amm_predict
ModelSpectrum = AmmoniaSpectrum
ModelRunner = AmmoniaRunner

PAR_NAMES = ['voff', 'trot', 'tex', 'ntot', 'sigm', 'orth']
PAR_NAMES_SHORT = ['v', 'Tk', 'Tx', 'N', 's', 'o']

TEX_LABELS = [
    r'$v_\mathrm{lsr}$',
    r'$T_\mathrm{rot}$',
    r'$T_\mathrm{ex}$',
    r'$\log(N_\mathrm{p})$',
    r'$\sigma_\mathrm{v}$',
    r'$f_\mathrm{o}$',
]

TEX_LABELS_WITH_UNITS = [
    r'$v_\mathrm{lsr} \ [\mathrm{km\, s^{-1}}]$',
    r'$T_\mathrm{rot} \ [\mathrm{K}]$',
    r'$T_\mathrm{ex} \ [\mathrm{K}]$',
    r'$\log(N) \ [\log(\mathrm{cm^{-2}})]$',
    r'$\sigma_\mathrm{v} \ [\mathrm{km\, s^{-1}}]$',
    r'$f_\mathrm{o}$',
]


def get_par_names(ncomp=None):
    return par_names(PAR_NAMES_SHORT, ncomp)
