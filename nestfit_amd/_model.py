"""Engine-backed pieces shared by the model modules (ammonia, diazenylium, gaussian):
device-resident spectra, the runner handle and the reference's Spectrum / Runner
behaviour (nestfit/core/core.pyx:486-561) on top of the C ABI."""
import ctypes as C

import numpy as np

from . import _ffi
from .core import Runner, _as_inplace_matrix, _as_inplace_vector

MODEL_AMMONIA, MODEL_DIAZENYLIUM, MODEL_GAUSSIAN = 0, 1, 2


class _SpecSet:
    """Owner of a device-resident set of spectra (one pixel or a cube)."""

    def __init__(self, xarrs, trans_ids, data, noise, model=MODEL_AMMONIA, rest_freqs=None):
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
        self.model = int(model)
        self.rest_freqs = (None if rest_freqs is None
                           else np.ascontiguousarray(rest_freqs, dtype=np.float64))
        _ffi.check(lib.nfa_specset_create_model(
            C.byref(h), self.model, self.n_spec, self.sizes.ctypes.data_as(_ffi._lp),
            self.trans_ids.ctypes.data_as(_ffi._ip),
            None if self.rest_freqs is None else _ffi.dptr(self.rest_freqs), xp, self.n_pix,
            _ffi.dptr(data), _ffi.dptr(noise)))
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
    def __init__(self, specset, utrans, ncomp, cold=False, lte=False):
        lib = _ffi.engine()
        self.specset = specset
        self.utrans = utrans
        ph = utrans._device_handle() if utrans is not None else None
        h = C.c_void_p()
        _ffi.check(lib.nfa_runner_create(C.byref(h), specset.handle, ph, int(ncomp), int(bool(cold)),
                                         int(bool(lte))))
        self.handle = h

    def set_exp_mode(self, mode):
        """Pin the numerical mode of this runner ('table' / 'fast' or 0 / 2); None or -1
        = follow the process default (`nestfit_amd.set_exp_mode`) again."""
        mode = -1 if mode is None else {'table': 0, 'fast': 2}.get(mode, mode)
        _ffi.check(_ffi.load().nfa_runner_set_exp_mode(self.handle, int(mode)))

    def get_exp_mode(self):
        return _ffi.load().nfa_runner_get_exp_mode(self.handle)

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



class EngineSpectrumMixin:
    """Spectrum whose model values are computed by the engine: one private spectra set of
    one pixel, runner handles cached per (ncomp, cold, lte)."""
    MODEL = MODEL_AMMONIA

    def _attach(self, trans_id, rest_freq=None):
        self._ss = _SpecSet([self.xarr], [trans_id], self.data.reshape(1, -1),
                            np.array([[self.noise]]), model=self.MODEL,
                            rest_freqs=None if rest_freq is None else [rest_freq])
        self.null_lnZ = float(self._ss.null_lnZ()[0, 0])
        self._runners = {}

    def _runner(self, ncomp, cold=False, lte=False):
        key = (int(ncomp), bool(cold), bool(lte))
        if key not in self._runners:
            self._runners[key] = _RunnerHandle(self._ss, None, *key)
        return self._runners[key]

    def _predict(self, params, n_model, cold=False, lte=False):
        params = np.ascontiguousarray(params, dtype=np.float64)
        if params.ndim != 1 or params.shape[0] == 0 or params.shape[0] % n_model != 0:
            raise ValueError(f'Invalid parameter vector length: {params.shape}')
        run = self._runner(params.shape[0] // n_model, cold, lte)
        spec = np.empty((1, self.size))
        lnl = np.empty(1)
        _ffi.check(_ffi.load().nfa_runner_predict_batch(run.handle, None, _ffi.dptr(params), 1,
                                                        _ffi.dptr(spec), _ffi.dptr(lnl)))
        self._pred = spec[0]
        self._lnL = float(lnl[0])

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


class EngineRunner(Runner):
    """Runner attributes and methods common to the three models (reference:
    ammonia.pyx:369-447, diazenylium.pyx:161-231, gaussian.pyx:57-112)."""
    MODEL = MODEL_AMMONIA
    N_MODEL = 6

    def _setup(self, spectra, utrans, ncomp, cold=False, lte=False, rest_freqs=None):
        assert ncomp > 0
        self.n_model = self.N_MODEL
        self.utrans = utrans
        self.ncomp = int(ncomp)
        self.n_spec = len(spectra)
        self.n_params = self.n_model * self.ncomp
        self.ndim = self.n_params  # no nuisance parameters
        self.null_lnZ = 0.0
        self.n_chan_tot = 0
        for spec in spectra:
            self.null_lnZ += spec.null_lnZ
            self.n_chan_tot += spec.size
        self.run_lnZ = np.nan
        data = np.concatenate([s.data for s in spectra]).reshape(1, -1)
        noise = np.array([[s.noise for s in spectra]])
        self._ss = _SpecSet([s.xarr for s in spectra], [s.trans_id for s in spectra], data, noise,
                            model=self.MODEL, rest_freqs=rest_freqs)
        self._run = _RunnerHandle(self._ss, utrans, self.ncomp, cold, lte)

    def set_exp_mode(self, mode):
        """Numerical mode of this runner alone (None: the process default again): runners of different
        modes can then be used side by side, also from different threads."""
        self._run.set_exp_mode(mode)

    def loglikelihood(self, utheta):
        """lnL of one unit-cube point; `utheta` is overwritten with the physical
        parameters exactly like the reference (core.pyx:558-561)."""
        utheta = _as_inplace_vector(utheta)
        if utheta.shape[0] != self.ndim:
            raise ValueError(f'Invalid shape for ncomp={self.ncomp}: {utheta.shape[0]}')
        return float(self.loglikelihood_batch(utheta.reshape(1, -1))[0])

    def loglikelihood_batch(self, U, out=None):
        """lnL[B] for unit-cube rows U[B, ndim] (overwritten with parameters).  `out`: where lnL goes; with `U`
        and `out` from `nestfit_amd.pinned_empty` the kernels work on the caller's arrays directly (no copies)."""
        U = _as_inplace_matrix(U)
        if U.shape[1] != self.ndim:
            raise ValueError(f'Invalid shape for ncomp={self.ncomp}: {U.shape[1]}')
        lnL = np.empty(U.shape[0]) if out is None else out
        if lnL.shape != (U.shape[0],) or lnL.dtype != np.float64 or not lnL.flags.c_contiguous:
            raise ValueError('out must be a contiguous float64 array of one value per row')
        _ffi.check(_ffi.load().nfa_runner_loglike_batch(self._run.handle, None, _ffi.dptr(U),
                                                        _ffi.dptr(lnL), U.shape[0]))
        return lnL

    def _check_params(self, params):
        params = np.ascontiguousarray(params, dtype=np.float64)
        if params.shape[0] != self.ndim:
            ncomp = self.ncomp
            shape = params.shape[0]
            raise ValueError(f'Invalid shape for ncomp={ncomp}: {shape}')
        return params

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


def par_names(short, ncomp=None):
    if ncomp is not None:
        return [f'{label}{n}' for label in short for n in range(1, ncomp + 1)]
    return short
