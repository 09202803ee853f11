"""Host-side mirror of ``nestfit.models.diazenylium`` (reference:
nestfit/models/diazenylium.pyx:108-264): N2H+ J = 1-0, 2-1, 3-2 with 15 / 40 / 45
hyperfine lines on the same device kernels as ammonia (c_hf_predict,
nestfit/models/hyperfine.pyx:52-118).  Parameters per component: voff, tex, ltau, sigm.
"""
import numpy as np

from ._model import MODEL_DIAZENYLIUM, EngineRunner, EngineSpectrumMixin, par_names
from .core import HyperfineSpectrum

N_LEVELS = 3
N_PARAMS = 4


class DiazenyliumSpectrum(EngineSpectrumMixin, HyperfineSpectrum):
    """N2H+ spectrum (reference: diazenylium.pyx:108-136).

    Parameters
    ----------
    xarr : array, Hz, ascending
    data : array, K
    noise : number, K
    trans_id : 1 -> (1-0), 2 -> (2-1), 3 -> (3-2)
    """
    MODEL = MODEL_DIAZENYLIUM

    def __init__(self, xarr, data, noise, trans_id=1):
        assert trans_id in range(1, N_LEVELS + 1)
        # the reference passes rest_freq=self.trans.nu before trans is assigned: 0 (diazenylium.pyx:129-131)
        HyperfineSpectrum.__init__(self, xarr, data, noise, rest_freq=0.0, trans_id=trans_id)
        self._attach(trans_id)

    @property
    def tbg_arr(self):
        return self._ss.tbg()


def nnhp_predict(s, params):
    """Model spectrum of `s` for parameter-major `params` (reference:
    diazenylium.pyx:138-158); result in ``s.get_spec()`` / ``s.loglikelihood``."""
    s._predict(params, N_PARAMS)


class DiazenyliumRunner(EngineRunner):
    """Prior transform + model + log-likelihood (reference: diazenylium.pyx:161-231)."""
    MODEL = MODEL_DIAZENYLIUM
    N_MODEL = N_PARAMS

    def __init__(self, spectra, utrans, ncomp=1):
        assert ncomp > 0
        self.spectra = list(spectra)
        self._setup(self.spectra, utrans, ncomp)

    @classmethod
    def from_data(cls, spec_data, utrans, **kwargs):
        spectra = np.array([DiazenyliumSpectrum(*args) for args in spec_data])
        return cls(spectra, utrans, **kwargs)

    def get_spectra(self):
        return np.array(self.spectra)

    def predict(self, params):
        params = self._check_params(params)
        for s in self.spectra:
            nnhp_predict(s, params)


# Aliases and metadata at module scope (reference: diazenylium.pyx:234-264)
N = N_PARAMS
IX_VCEN = 0
IX_SIGM = 3
NAME = 'diazenylium'
model_predict = nnhp_predict
ModelSpectrum = DiazenyliumSpectrum
ModelRunner = DiazenyliumRunner

PAR_NAMES = ['voff', 'tex', 'ltau', 'sigm']
PAR_NAMES_SHORT = ['v', 'Tx', 'lt', 's']

TEX_LABELS = [
    r'$v_\mathrm{lsr}$',
    r'$T_\mathrm{ex}$',
    r'$\log(\tau_0)$',
    r'$\sigma_\mathrm{v}$',
]

TEX_LABELS_WITH_UNITS = [
    r'$v_\mathrm{lsr} \ [\mathrm{km\, s^{-1}}]$',
    r'$T_\mathrm{ex} \ [\mathrm{K}]$',
    r'$\log(\tau_0)$',
    r'$\sigma_\mathrm{v} \ [\mathrm{km\, s^{-1}}]$',
]


def get_par_names(ncomp=None):
    return par_names(PAR_NAMES_SHORT, ncomp)
