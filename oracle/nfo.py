"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.  The product (``nestfit_amd``)
never does; it fails loudly when its HIP library is missing instead.

The classes below wrap ``oracle/nf_oracle.c`` with the reference's object names
(``AmmoniaSpectrum``, ``amm_predict``, ``AmmoniaRunner``) so parity tests read
like the reference's own usage (nestfit/models/ammonia.pyx:244-447).
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_long)


def build(native=False, quiet=True):
    """Compile the oracle with gcc (and oracle/_ref when /root/reference exists)."""
    targets = ['all'] + (['native'] if native else [])
    out = subprocess.run(['make', '-C', str(HERE)] + targets, capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError('oracle build failed:\n' + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


class _Dist(C.Structure):
    _fields_ = [('size', C.c_long), ('du', C.c_double), ('dx', C.c_double),
                ('xmin', C.c_double), ('xmax', C.c_double),
                ('xax', _dp), ('pdf', _dp), ('cdf', _dp), ('ppf', _dp)]


class _Prior(C.Structure):
    _fields_ = [('kind', C.c_int), ('p_ix', C.c_int), ('p_ix2', C.c_int),
                ('dist0', C.c_int), ('dist1', C.c_int), ('dist2', C.c_int),
                ('sub_kind', C.c_int), ('pad_', C.c_int),
                ('value', C.c_double), ('sep_scale', C.c_double)]


class _PriorSet(C.Structure):
    _fields_ = [('n_prior', C.c_int), ('n_dist', C.c_int),
                ('priors', C.POINTER(_Prior)), ('dists', C.POINTER(_Dist))]


def _load(name):
    path = HERE / '_build' / name
    if not path.exists():
        build(native='native' in name)
    lib = C.CDLL(str(path))
    lib.nfo_fastexp.restype = C.c_double
    lib.nfo_fastexp.argtypes = [C.c_float]
    lib.nfo_fastexp_indices.argtypes = [C.c_float] + [C.POINTER(C.c_int)] * 4
    lib.nfo_fastexp_many.argtypes = [C.POINTER(C.c_float), _dp, C.c_long]
    lib.nfo_fast_expn_many.argtypes = [_dp, _dp, C.c_long]
    lib.nfo_iemtex_many.argtypes = [_dp, _dp, C.c_long]
    lib.nfo_t0_xmin.restype = C.c_double
    lib.nfo_t0_xmax.restype = C.c_double
    lib.nfo_iemtex_set_table.argtypes = [_dp, _dp, C.c_long]
    lib.nfo_iemtex_index.restype = C.c_long
    lib.nfo_iemtex_index.argtypes = [C.c_double]
    lib.nfo_iemtex_interp.restype = C.c_double
    lib.nfo_iemtex_interp.argtypes = [C.c_double]
    lib.nfo_swift_convert.restype = C.c_double
    lib.nfo_swift_convert.argtypes = [C.c_double]
    lib.nfo_partition_level.restype = C.c_double
    lib.nfo_partition_level.argtypes = [C.c_long, C.c_double]
    lib.nfo_partition_func.restype = C.c_double
    lib.nfo_partition_func.argtypes = [C.c_int, C.c_double]
    lib.nfo_spectrum_new.restype = C.c_void_p
    lib.nfo_spectrum_new.argtypes = [_dp, _dp, C.c_long, C.c_double, C.c_int]
    lib.nfo_spectrum_new_model.restype = C.c_void_p
    lib.nfo_spectrum_new_model.argtypes = [_dp, _dp, C.c_long, C.c_double, C.c_int, C.c_int, C.c_double]
    lib.nfo_nnhp_predict.argtypes = [C.c_void_p, _dp, C.c_long]
    lib.nfo_gauss_predict.argtypes = [C.c_void_p, _dp, C.c_long]
    lib.nfo_n2hp_nhf.argtypes = [C.c_int]
    lib.nfo_n2hp_nu.restype = C.c_double
    lib.nfo_n2hp_nu.argtypes = [C.c_int]
    for f in ('nfo_n2hp_voff', 'nfo_n2hp_tau_wt'):
        getattr(lib, f).restype = C.c_double
        getattr(lib, f).argtypes = [C.c_int, C.c_int]
    lib.nfo_spectrum_free.argtypes = [C.c_void_p]
    lib.nfo_spectrum_size.restype = C.c_long
    lib.nfo_spectrum_size.argtypes = [C.c_void_p]
    lib.nfo_spectrum_null_lnZ.restype = C.c_double
    lib.nfo_spectrum_null_lnZ.argtypes = [C.c_void_p]
    for f in ('nfo_spectrum_pred', 'nfo_spectrum_tarr', 'nfo_spectrum_tbg'):
        getattr(lib, f).restype = _dp
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.nfo_spectrum_set_data.argtypes = [C.c_void_p, _dp]
    lib.nfo_spectrum_loglike.restype = C.c_double
    lib.nfo_spectrum_loglike.argtypes = [C.c_void_p]
    lib.nfo_hf_windows.argtypes = [C.c_void_p, C.c_double, C.c_double, _lp, _lp]
    lib.nfo_amm_predict.argtypes = [C.c_void_p, _dp, C.c_long, C.c_int, C.c_int]
    lib.nfo_transform.argtypes = [C.POINTER(_PriorSet), _dp, C.c_long]
    lib.nfo_dist_ppf_interp.restype = C.c_double
    lib.nfo_dist_ppf_interp.argtypes = [C.POINTER(_Dist), C.c_double]
    lib.nfo_dist_cdf_interp.restype = C.c_double
    lib.nfo_dist_cdf_interp.argtypes = [C.POINTER(_Dist), C.c_double]
    lib.nfo_runner_loglike.restype = C.c_double
    lib.nfo_runner_loglike.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(_PriorSet),
                                       _dp, C.c_long, C.c_int, C.c_int]
    lib.nfo_runner_loglike_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int,
                                             C.POINTER(_PriorSet), _dp, _dp, C.c_long,
                                             C.c_long, C.c_int, C.c_int]
    lib.nfo_trans_nhf.argtypes = [C.c_int]
    for f in ('nfo_trans_nu', 'nfo_trans_ea'):
        getattr(lib, f).restype = C.c_double
        getattr(lib, f).argtypes = [C.c_int]
    for f in ('nfo_trans_voff', 'nfo_trans_tau_wt'):
        getattr(lib, f).restype = C.c_double
        getattr(lib, f).argtypes = [C.c_int, C.c_int]
    # The reference fills its 1/(e^x-1) table with numpy at import
    # (nestfit/models/hyperfine.pyx:17-20); install the same arrays.
    t0_x, t0_y = iemtex_tables(lib)
    lib.nfo_iemtex_set_table(_p(t0_x), _p(t0_y), t0_x.size)
    lib.nfo_fastexp_init()
    return lib


def iemtex_tables(lib=None):
    """T0_X, T0_Y exactly as nestfit/models/hyperfine.pyx:12-20 builds them."""
    H, KB = 6.62607015e-27, 1.380649e-16
    xmin = (H * 23.0e9 / KB) / 8.0
    xmax = (H * 28.0e9 / KB) / 2.7
    t0_x = np.linspace(xmin, xmax, 1000)
    t0_y = 1.0 / (np.exp(t0_x) - 1.0)
    return np.ascontiguousarray(t0_x), np.ascontiguousarray(t0_y)


def _p(a):
    return a.ctypes.data_as(_dp)


_LIBS = {}


def lib(native=False):
    name = 'libnf_oracle_native.so' if native else 'libnf_oracle.so'
    if name not in _LIBS:
        _LIBS[name] = _load(name)
    return _LIBS[name]


def oldconst_lib():
    """The checker built with the reference's other constant set (__NEW_CONST = False,
    nestfit/models/ammonia.pyx:20-22): only for the reference's own partition-function known answers."""
    name = 'libnf_oracle_oldconst.so'
    if name not in _LIBS:
        _LIBS[name] = _load(name)
    return _LIBS[name]


def ref_fastexp_lib(fast_math=False):
    """The reference's own fastexp.c compiled as-is (oracle/_ref); None if absent."""
    path = HERE / '_ref' / ('libfastexp_ref_fm.so' if fast_math else 'libfastexp_ref.so')
    if not path.exists():
        if Path('/root/reference/nestfit/core/fastexp.c').exists():
            build()
        if not path.exists():
            return None
    r = C.CDLL(str(path))
    r.FastExp.restype = C.c_double
    r.FastExp.argtypes = [C.c_float]
    r.calcExpTableEntries.argtypes = [C.c_int, C.c_int]
    r.calcExpTableEntries(3, 8)          # includes/model_includes.pxi:12
    return r


# ---------------------------------------------------------------------------
def fastexp(x, native=False):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.float64)
    lib(native).nfo_fastexp_many(x.ctypes.data_as(C.POINTER(C.c_float)), _p(out), x.size)
    return out


def fast_expn(x):
    """FastExp of a double argument (narrowed to float like core/math.pxd:17)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().nfo_fast_expn_many(_p(x), _p(out), x.size)
    return out


def fastexp_indices(x):
    l, j0, j1, j2 = (C.c_int() for _ in range(4))
    lib().nfo_fastexp_indices(float(np.float32(x)), l, j0, j1, j2)
    return l.value, j0.value, j1.value, j2.value


def iemtex_interp(x):
    x = np.ascontiguousarray(np.atleast_1d(x), dtype=np.float64)
    out = np.empty_like(x)
    lib().nfo_iemtex_many(_p(x), _p(out), x.size)
    return out


def iemtex_index(x):
    return lib().nfo_iemtex_index(float(x))


def swift_convert(tkin):
    return lib().nfo_swift_convert(float(tkin))


def partition_level(j, trot):
    return lib().nfo_partition_level(int(j), float(trot))


def partition_func(para, trot):
    return lib().nfo_partition_func(int(bool(para)), float(trot))


MODEL_AMMONIA, MODEL_DIAZENYLIUM, MODEL_GAUSSIAN = 0, 1, 2


class AmmoniaSpectrum:
    """Oracle twin of nestfit.models.ammonia.AmmoniaSpectrum (ammonia.pyx:244-277)."""
    MODEL = MODEL_AMMONIA

    def __init__(self, xarr, data, noise, trans_id=1, native=False, rest_freq=0.0):
        self._lib = lib(native)
        xarr = np.ascontiguousarray(xarr, dtype=np.float64)
        data = np.ascontiguousarray(data, dtype=np.float64)
        assert xarr.shape == data.shape and xarr.ndim == 1
        self.size = xarr.size
        self.trans_id = int(trans_id)
        self.noise = float(noise)
        self.rest_freq = float(rest_freq)
        self._h = self._lib.nfo_spectrum_new_model(_p(xarr), _p(data), xarr.size, float(noise),
                                                   self.MODEL, int(trans_id), float(rest_freq))
        if not self._h:
            raise AssertionError('invalid spectrum arguments')

    def __del__(self):
        if getattr(self, '_h', None):
            self._lib.nfo_spectrum_free(self._h)
            self._h = None

    def _view(self, fn):
        return np.ctypeslib.as_array(fn(self._h), shape=(self.size,)).copy()

    def get_spec(self):
        return self._view(self._lib.nfo_spectrum_pred)

    @property
    def tarr(self):
        return self._view(self._lib.nfo_spectrum_tarr)

    @property
    def tbg_arr(self):
        return self._view(self._lib.nfo_spectrum_tbg)

    @property
    def null_lnZ(self):
        return self._lib.nfo_spectrum_null_lnZ(self._h)

    @property
    def loglikelihood(self):
        return self._lib.nfo_spectrum_loglike(self._h)

    @property
    def sum_spec(self):
        return float(np.nansum(self.get_spec()))

    @property
    def max_spec(self):
        return float(np.nanmax(self.get_spec()))

    def set_data(self, data):
        data = np.ascontiguousarray(data, dtype=np.float64)
        assert data.size == self.size
        self._lib.nfo_spectrum_set_data(self._h, _p(data))

    def hf_windows(self, voff, sigm):
        n = (self._lib.nfo_n2hp_nhf if self.MODEL == MODEL_DIAZENYLIUM
             else self._lib.nfo_trans_nhf)(self.trans_id)
        lo = np.zeros(n, dtype=np.int64)
        hi = np.zeros(n, dtype=np.int64)
        self._lib.nfo_hf_windows(self._h, float(voff), float(sigm),
                                 lo.ctypes.data_as(_lp), hi.ctypes.data_as(_lp))
        return lo, hi


def amm_predict(s, params, cold=False, lte=False):
    params = np.ascontiguousarray(params, dtype=np.float64)
    s._lib.nfo_amm_predict(s._h, _p(params), params.size, int(cold), int(lte))


class DiazenyliumSpectrum(AmmoniaSpectrum):
    """Oracle twin of nestfit.models.diazenylium.DiazenyliumSpectrum (diazenylium.pyx:108-136)."""
    MODEL = MODEL_DIAZENYLIUM


class Spectrum(AmmoniaSpectrum):
    """Oracle twin of nestfit.core.core.Spectrum as the Gaussian model uses it (core.pyx:486-530)."""
    MODEL = MODEL_GAUSSIAN

    def __init__(self, xarr, data, noise, rest_freq=None, trans_id=None, native=False):
        super().__init__(xarr, data, noise, trans_id=-1 if trans_id is None else trans_id,
                         native=native, rest_freq=0.0 if rest_freq is None else rest_freq)


def nnhp_predict(s, params):
    params = np.ascontiguousarray(params, dtype=np.float64)
    s._lib.nfo_nnhp_predict(s._h, _p(params), params.size)


def gauss_predict(s, params):
    params = np.ascontiguousarray(params, dtype=np.float64)
    s._lib.nfo_gauss_predict(s._h, _p(params), params.size)


class PriorSet:
    """C view of a lowered prior program (see nestfit_amd.core.PriorTransformer.lower)."""

    def __init__(self, program):
        self._keep = []
        dists = (_Dist * max(1, len(program['dists'])))()
        for k, d in enumerate(program['dists']):
            arrs = {n: np.ascontiguousarray(d[n], dtype=np.float64).copy()
                    for n in ('xax', 'pdf', 'cdf', 'ppf')}
            self._keep.append(arrs)
            dists[k].size = arrs['xax'].size
            for n in ('du', 'dx', 'xmin', 'xmax'):
                setattr(dists[k], n, float(d[n]))
            for n, a in arrs.items():
                setattr(dists[k], n, _p(a))
        priors = (_Prior * len(program['priors']))()
        for k, p in enumerate(program['priors']):
            for n in ('kind', 'p_ix', 'p_ix2', 'dist0', 'dist1', 'dist2', 'sub_kind'):
                setattr(priors[k], n, int(p[n]))
            priors[k].value = float(p['value'])
            priors[k].sep_scale = float(p['sep_scale'])
        self._dists, self._priors = dists, priors
        self.c = _PriorSet(len(program['priors']), len(program['dists']), priors, dists)
        self.n_param = int(program['n_param'])

    def transform(self, utheta, ncomp, native=False):
        assert utheta.dtype == np.float64 and utheta.flags.c_contiguous
        if self.n_param * ncomp != utheta.shape[0]:
            raise ValueError(f'Invalid shape for ncomp={ncomp}: {utheta.shape[0]}')
        lib(native).nfo_transform(C.byref(self.c), _p(utheta), int(ncomp))


class AmmoniaRunner:
    """Oracle twin of nestfit.models.ammonia.AmmoniaRunner (ammonia.pyx:369-447)."""
    N_MODEL = 6

    def __init__(self, spectra, priorset, ncomp=1, cold=False, lte=False, native=False):
        assert ncomp > 0
        self._lib = lib(native)
        self.spectra = list(spectra)
        self.priorset = priorset
        self.ncomp, self.cold, self.lte = int(ncomp), bool(cold), bool(lte)
        self.n_spec = len(self.spectra)
        self.ndim = self.n_params = self.N_MODEL * self.ncomp
        self.null_lnZ = sum(s.null_lnZ for s in self.spectra)
        self.n_chan_tot = sum(s.size for s in self.spectra)
        self._handles = (C.c_void_p * self.n_spec)(*[s._h for s in self.spectra])

    def loglikelihood(self, utheta):
        assert utheta.dtype == np.float64 and utheta.size == self.ndim
        ps = C.byref(self.priorset.c) if self.priorset is not None else None
        return self._lib.nfo_runner_loglike(self._handles, self.n_spec, ps, _p(utheta),
                                            self.ncomp, int(self.cold), int(self.lte))

    def loglikelihood_batch(self, U):
        """U[B, ndim] unit cube in, physical parameters out (in place); returns lnL[B]."""
        assert U.dtype == np.float64 and U.flags.c_contiguous and U.shape[1] == self.ndim
        lnL = np.empty(U.shape[0])
        ps = C.byref(self.priorset.c) if self.priorset is not None else None
        self._lib.nfo_runner_loglike_batch(self._handles, self.n_spec, ps, _p(U), _p(lnL),
                                           U.shape[0], self.ncomp, int(self.cold),
                                           int(self.lte))
        return lnL

    def predict(self, params):
        params = np.ascontiguousarray(params, dtype=np.float64)
        if params.shape[0] != self.ndim:
            raise ValueError(f'Invalid shape for ncomp={self.ncomp}: {params.shape[0]}')
        for s in self.spectra:
            self._predict_one(s, params)

    def _predict_one(self, s, params):
        amm_predict(s, params, self.cold, self.lte)


class DiazenyliumRunner(AmmoniaRunner):
    """Oracle twin of nestfit.models.diazenylium.DiazenyliumRunner (diazenylium.pyx:161-231)."""
    N_MODEL = 4

    def __init__(self, spectra, priorset, ncomp=1, native=False):
        super().__init__(spectra, priorset, ncomp=ncomp, native=native)

    def _predict_one(self, s, params):
        nnhp_predict(s, params)


class GaussianRunner(AmmoniaRunner):
    """Oracle twin of nestfit.models.gaussian.GaussianRunner (gaussian.pyx:57-112)."""
    N_MODEL = 3

    def __init__(self, spectrum, priorset, ncomp=1, native=False):
        super().__init__([spectrum], priorset, ncomp=ncomp, native=native)

    def _predict_one(self, s, params):
        gauss_predict(s, params)
