"""Host-side mirror of ``nestfit.models.ammonia`` (reference:
nestfit/models/ammonia.pyx:244-489): same names, arguments and error
behaviour; all arithmetic runs in the HIP engine through the C ABI.
"""
import ctypes as C

import numpy as np

from . import _ffi
from .core import HyperfineSpectrum, Runner, _as_inplace_matrix, _as_inplace_vector

N_LEVELS = 9
N_PARAMS = 6


class _SpecSet:
    """Owner of a device-resident set of spectra (one pixel or a cube)."""

    def __init__(self, xarrs, trans_ids, data, noise):
        """xarrs: list of 1-D axes; data [n_pix, sum(sizes)]; noise [n_pix, n_spec]."""
        lib = _ffi.engine()
        self.n_spec = len(xarrs)
        self.sizes = np.array([x.size for x in xarrs], dtype=np.int64)
        self.trans_ids = np.asarray(trans_ids, dtype=np.int32)
        self.xarrs = [np.ascontiguousarray(x, dtype=np.float64) for x in xarrs]
        data = np.ascontiguousarray(data, dtype=np.float64)
        noise = np.ascontiguousarray(noise, dtype=np.float64)
        self.n_pix = int(data.shape[0])
        self.chan_tot = int(self.sizes.sum())
        assert data.shape == (self.n_pix, self.chan_tot)
        assert noise.shape == (self.n_pix, self.n_spec)
        xp = (_ffi._dp * self.n_spec)(*[_ffi.dptr(x) for x in self.xarrs])
        h = C.c_void_p()
        _ffi.check(lib.nfa_specset_create(
            C.byref(h), self.n_spec, self.sizes.ctypes.data_as(_ffi._lp),
            self.trans_ids.ctypes.data_as(_ffi._ip), xp, self.n_pix, _ffi.dptr(data),
            _ffi.dptr(noise)))
        self.handle = h
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)

    def null_lnZ(self):
        out = np.empty((self.n_pix, self.n_spec))
        _ffi.check(_ffi.load().nfa_specset_null_lnz(self.handle, _ffi.dptr(out)))
        return out

    def tbg(self):
        out = np.empty(self.chan_tot)
        _ffi.check(_ffi.load().nfa_specset_tbg(self.handle, _ffi.dptr(out)))
        return out

    def __del__(self):
        if getattr(self, 'handle', None) is not None:
            try:
                _ffi.load().nfa_specset_destroy(self.handle)
            except Exception:
                pass
            self.handle = None


class _RunnerHandle:
    def __init__(self, specset, utrans, ncomp, cold, lte):
        lib = _ffi.engine()
        self.specset = specset
        self.utrans = utrans
        ph = utrans._device_handle() if utrans is not None else None
        h = C.c_void_p()
        _ffi.check(lib.nfa_runner_create(C.byref(h), specset.handle, ph, int(ncomp), int(bool(cold)),
                                         int(bool(lte))))
        self.handle = h

    def __del__(self):
        if getattr(self, 'handle', None) is not None:
            try:
                _ffi.load().nfa_runner_destroy(self.handle)
            except Exception:
                pass
            self.handle = None


def _pix_ptr(pix, B):
    if pix is None:
        return None, None
    pix = np.ascontiguousarray(pix, dtype=np.int32)
    assert pix.shape == (B,)
    return pix, pix.ctypes.data_as(_ffi._ip)


class AmmoniaSpectrum(HyperfineSpectrum):
    """NH3 (J,K) inversion spectrum (reference: ammonia.pyx:244-277).

    Parameters
    ----------
    xarr : array, Hz, ascending
    data : array, K
    noise : number, K
    trans_id : 1 -> (1,1) ... 9 -> (9,9)
    """

    def __init__(self, xarr, data, noise, trans_id=1):
        assert trans_id in range(1, N_LEVELS + 1)
        super().__init__(xarr, data, noise, rest_freq=0.0, trans_id=trans_id)
        self._ss = _SpecSet([self.xarr], [trans_id], self.data.reshape(1, -1),
                            np.array([[self.noise]]))
        self.null_lnZ = float(self._ss.null_lnZ()[0, 0])
        self._runners = {}

    @property
    def tbg_arr(self):
        return self._ss.tbg()

    def _runner(self, ncomp, cold, lte):
        key = (int(ncomp), bool(cold), bool(lte))
        if key not in self._runners:
            self._runners[key] = _RunnerHandle(self._ss, None, *key)
        return self._runners[key]

    # reference: core.pyx:532-545
    @property
    def sum_spec(self):
        return np.nansum(self._pred)

    @property
    def max_spec(self):
        return np.nanmax(self._pred)

    @property
    def loglikelihood(self):
        return self.null_lnZ if self._lnL is None else self._lnL

    def get_spec(self):
        return np.array(self._pred)


def amm_predict(s, params, cold=False, lte=False):
    """Model spectrum of `s` for parameter-major `params` (reference:
    ammonia.pyx:326-366); result in ``s.get_spec()`` / ``s.loglikelihood``."""
    params = np.ascontiguousarray(params, dtype=np.float64)
    if params.ndim != 1 or params.shape[0] == 0 or params.shape[0] % N_PARAMS != 0:
        raise ValueError(f'Invalid parameter vector length: {params.shape}')
    ncomp = params.shape[0] // N_PARAMS
    run = s._runner(ncomp, cold, lte)
    spec = np.empty((1, s.size))
    lnl = np.empty(1)
    _ffi.check(_ffi.load().nfa_runner_predict_batch(run.handle, None, _ffi.dptr(params), 1,
                                                    _ffi.dptr(spec), _ffi.dptr(lnl)))
    s._pred = spec[0]
    s._lnL = float(lnl[0])


class AmmoniaRunner(Runner):
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

    def __init__(self, spectra, utrans, ncomp=1, cold=False, lte=False):
        assert ncomp > 0
        spectra = list(spectra)
        self.n_model = N_PARAMS
        self.spectra = spectra
        self.utrans = utrans
        self.ncomp = int(ncomp)
        self.cold = bool(cold)
        self.lte = bool(lte)
        self.n_spec = len(spectra)
        self.n_params = self.n_model * self.ncomp
        self.ndim = self.n_params
        self.null_lnZ = 0.0
        self.n_chan_tot = 0
        for spec in spectra:
            self.null_lnZ += spec.null_lnZ
            self.n_chan_tot += spec.size
        self.run_lnZ = np.nan
        data = np.concatenate([s.data for s in spectra]).reshape(1, -1)
        noise = np.array([[s.noise for s in spectra]])
        self._ss = _SpecSet([s.xarr for s in spectra], [s.trans_id for s in spectra], data, noise)
        self._run = _RunnerHandle(self._ss, utrans, self.ncomp, self.cold, self.lte)

    @classmethod
    def from_data(cls, spec_data, utrans, **kwargs):
        spectra = np.array([AmmoniaSpectrum(*args) for args in spec_data])
        return cls(spectra, utrans, **kwargs)

    def loglikelihood(self, utheta):
        """lnL of one unit-cube point; `utheta` is overwritten with the physical
        parameters exactly like the reference (core.pyx:558-561)."""
        utheta = _as_inplace_vector(utheta)
        if utheta.shape[0] != self.ndim:
            raise ValueError(f'Invalid shape for ncomp={self.ncomp}: {utheta.shape[0]}')
        return float(self.loglikelihood_batch(utheta.reshape(1, -1))[0])

    def loglikelihood_batch(self, U):
        """lnL[B] for unit-cube rows U[B, ndim] (overwritten with parameters)."""
        U = _as_inplace_matrix(U)
        if U.shape[1] != self.ndim:
            raise ValueError(f'Invalid shape for ncomp={self.ncomp}: {U.shape[1]}')
        lnL = np.empty(U.shape[0])
        _ffi.check(_ffi.load().nfa_runner_loglike_batch(self._run.handle, None, _ffi.dptr(U),
                                                        _ffi.dptr(lnL), U.shape[0]))
        return lnL

    def get_spectra(self):
        return np.array(self.spectra)

    def predict(self, params):
        """Model spectra for physical `params` into every spectrum of the runner
        (reference: ammonia.pyx:437-447)."""
        params = np.ascontiguousarray(params, dtype=np.float64)
        if params.shape[0] != self.ndim:
            ncomp = self.ncomp
            shape = params.shape[0]
            raise ValueError(f'Invalid shape for ncomp={ncomp}: {shape}')
        for s in self.spectra:
            amm_predict(s, params, self.cold, self.lte)

    def predict_batch(self, theta, want_spectra=True):
        """spectra[B, n_chan_tot] and lnL[B] for parameter rows theta[B, ndim]."""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        if theta.ndim != 2 or theta.shape[1] != self.ndim:
            raise ValueError(f'Invalid shape for ncomp={self.ncomp}: {theta.shape}')
        B = theta.shape[0]
        spec = np.empty((B, self._ss.chan_tot)) if want_spectra else None
        lnl = np.empty(B)
        _ffi.check(_ffi.load().nfa_runner_predict_batch(
            self._run.handle, None, _ffi.dptr(theta), B,
            _ffi.dptr(spec) if want_spectra else None, _ffi.dptr(lnl)))
        return spec, lnl


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
    if ncomp is not None:
        return [f'{label}{n}' for label in PAR_NAMES_SHORT for n in range(1, ncomp + 1)]
    return PAR_NAMES_SHORT
