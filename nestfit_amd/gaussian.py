"""Host-side mirror of ``nestfit.models.gaussian`` (reference:
nestfit/models/gaussian.pyx:17-150): sum of Gaussian components on one plain
``Spectrum`` with a rest frequency.  Parameters per component: voff, sigm, peak.
On the device it is the hyperfine kernel with one line of weight `peak` and no
radiative-transfer pass.
"""
import numpy as np

from . import core
from ._model import MODEL_GAUSSIAN, EngineRunner, EngineSpectrumMixin, par_names

N_PARAMS = 3


class Spectrum(EngineSpectrumMixin, core.Spectrum):
    """``nestfit.core.core.Spectrum`` (core.pyx:486-545) with its model values on the device."""
    MODEL = MODEL_GAUSSIAN

    def __init__(self, xarr, data, noise, rest_freq=None, trans_id=None):
        core.Spectrum.__init__(self, xarr, data, noise, rest_freq=rest_freq, trans_id=trans_id)
        self._attach(1, rest_freq=float(self.rest_freq))


def gauss_predict(s, params):
    """Model spectrum of `s` for parameter-major `params` (reference:
    gaussian.pyx:17-54); result in ``s.get_spec()`` / ``s.loglikelihood``."""
    s._predict(params, N_PARAMS)


class GaussianRunner(EngineRunner):
    """Prior transform + model + log-likelihood for one spectrum (reference:
    gaussian.pyx:57-112)."""
    MODEL = MODEL_GAUSSIAN
    N_MODEL = N_PARAMS

    def __init__(self, spectrum, utrans, ncomp=1):
        assert ncomp > 0
        self.spectrum = spectrum
        self._setup([spectrum], utrans, ncomp, rest_freqs=[float(spectrum.rest_freq)])

    @classmethod
    def from_data(cls, spec_data, utrans, **kwargs):
        return cls(Spectrum(*spec_data), utrans, **kwargs)

    def get_spectrum(self):
        return np.array(self.spectrum)

    def predict(self, params):
        params = self._check_params(params)
        gauss_predict(self.spectrum, params)


# Aliases and metadata at module scope (reference: gaussian.pyx:115-150)
N = N_PARAMS
IX_VCEN = 0
IX_SIGM = 1
NAME = 'gaussian'
model_predict = gauss_predict
ModelSpectrum = Spectrum
ModelRunner = GaussianRunner

PAR_NAMES = ['voff', 'sigm', 'peak']
PAR_NAMES_SHORT = ['v', 's', 'pk']

TEX_LABELS = [
    r'$v_\mathrm{lsr}$',
    r'$\sigma_\mathrm{v}$',
    r'$T_\mathrm{pk}$',
]

TEX_LABELS_WITH_UNITS = [
    r'$v_\mathrm{lsr} \ [\mathrm{km\, s^{-1}}]$',
    r'$\sigma_\mathrm{v} \ [\mathrm{km\, s^{-1}}]$',
    r'$T_\mathrm{pk} \ [\mathrm{K}]$',
]


def get_par_names(ncomp=None):
    return par_names(PAR_NAMES_SHORT, ncomp)
