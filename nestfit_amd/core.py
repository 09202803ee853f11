"""Host-side mirror of ``nestfit.core.core`` for the hot path.

Same class names, constructor arguments and error behaviour as the reference
(nestfit/core/core.pyx:23-561), but the objects only *describe* the work: every
number that the reference computes inside ``Prior.interp`` /
``Spectrum.c_loglikelihood`` is computed by the HIP engine through the C ABI
(include/nestfit_amd.h).  ``Distribution.__init__`` is the one piece of host
arithmetic: like the reference it is one-time table construction with scipy.
"""
import ctypes as C

import numpy as np
from scipy import integrate, interpolate

from . import _ffi

FWHM = 2.3548200450309493   # core.pyx:20

# prior kinds, numbering of include/nestfit_amd.h
KIND_SIMPLE, KIND_DUPLICATE, KIND_CONSTANT, KIND_ORDERED = 0, 1, 2, 3
KIND_SPACED, KIND_CENSEP, KIND_RESOLVED_CENSEP, KIND_RESOLVED_PLACEMENT = 4, 5, 6, 7


class Distribution:
    """Tabulated pdf -> cdf/ppf tables (reference: core.pyx:23-45)."""

    def __init__(self, xax, pdf):
        xax = np.ascontiguousarray(xax, dtype=np.float64)
        pdf = np.ascontiguousarray(pdf, dtype=np.float64)
        assert xax[1] > xax[0]
        assert xax.shape == pdf.shape
        self.dx = float(xax[1] - xax[0])
        self.xax = xax
        self.pdf = pdf
        self.size = int(xax.shape[0])
        self.xmin = float(np.min(xax))
        self.xmax = float(np.max(xax))
        # scipy >= 1.14 renamed cumtrapz (core.pyx:34) to cumulative_trapezoid
        cdf = integrate.cumulative_trapezoid(pdf, xax, initial=0)
        cdf /= cdf.max()
        self.cdf = np.ascontiguousarray(cdf)
        # strictly ascending copy for the inverse interpolation (core.pyx:38-42)
        eps_cdf = cdf + np.arange(self.size) * 1e-16
        eps_cdf /= eps_cdf.max()
        inv_cdf = interpolate.UnivariateSpline(eps_cdf, xax, k=3, s=0)
        u = np.linspace(0, 1, self.size)
        self.du = float(u[1] - u[0])
        self.ppf = np.ascontiguousarray(inv_cdf(u), dtype=np.float64)

    def lower(self):
        return dict(size=self.size, du=self.du, dx=self.dx, xmin=self.xmin, xmax=self.xmax,
                    xax=self.xax, pdf=self.pdf, cdf=self.cdf, ppf=self.ppf)


class Prior:
    """Inverse-CDF sampling of one parameter (reference: core.pyx:169-197)."""
    kind = KIND_SIMPLE

    def __init__(self, dist, p_ix):
        assert p_ix >= 0
        self.dist = dist
        self.p_ix = int(p_ix)
        self.n_param = 1

    def _desc(self, add_dist):
        return dict(kind=self.kind, p_ix=self.p_ix, p_ix2=0, dist0=add_dist(self.dist), dist1=0,
                    dist2=0, sub_kind=0, value=0.0, sep_scale=0.0)


class DuplicatePrior(Prior):
    """One draw written to two parameter slots (reference: core.pyx:200-221)."""
    kind = KIND_DUPLICATE

    def __init__(self, dist, p_ix, p_ix_dup):
        assert p_ix >= 0
        assert p_ix_dup >= 0
        self.p_ix = int(p_ix)
        self.p_ix_dup = int(p_ix_dup)
        self.dist = dist
        self.n_param = 2

    def _desc(self, add_dist):
        d = Prior._desc(self, add_dist)
        d['p_ix2'] = self.p_ix_dup
        return d


class ConstantPrior(Prior):
    """Fixed value (reference: core.pyx:224-238)."""
    kind = KIND_CONSTANT

    def __init__(self, value, p_ix):
        self.value = float(value)
        self.p_ix = int(p_ix)
        self.dist = None
        self.n_param = 1

    def _desc(self, add_dist):
        return dict(kind=self.kind, p_ix=self.p_ix, p_ix2=0, dist0=0, dist1=0, dist2=0,
                    sub_kind=0, value=self.value, sep_scale=0.0)


class OrderedPrior(Prior):
    """Left-to-right ordered draws (reference: core.pyx:241-258)."""
    kind = KIND_ORDERED


class SpacedPrior(Prior):
    """First draw independent, later draws are offsets (reference: core.pyx:261-292)."""
    kind = KIND_SPACED

    def __init__(self, prior_indep, prior_depen):
        self.prior_indep = prior_indep
        self.prior_depen = prior_depen
        self.p_ix = self.prior_indep.p_ix
        self.n_param = 1

    def _desc(self, add_dist):
        return dict(kind=self.kind, p_ix=self.p_ix, p_ix2=0,
                    dist0=add_dist(self.prior_indep.dist), dist1=add_dist(self.prior_depen.dist),
                    dist2=0, sub_kind=0, value=0.0, sep_scale=0.0)


class CenSepPrior(Prior):
    """Centre + separation for <= 2 components (reference: core.pyx:295-318)."""
    kind = KIND_CENSEP

    def __init__(self, vcen_prior, vsep_prior):
        self.vcen_prior = vcen_prior
        self.vsep_prior = vsep_prior
        self.p_ix = self.vcen_prior.p_ix
        self.n_param = 1

    def _desc(self, add_dist):
        return dict(kind=self.kind, p_ix=self.p_ix, p_ix2=0,
                    dist0=add_dist(self.vcen_prior.dist), dist1=add_dist(self.vsep_prior.dist),
                    dist2=0, sub_kind=0, value=0.0, sep_scale=0.0)


def _sub_prior_fields(sigm_prior, add_dist):
    """`sigm_prior.interp(utheta, n)` is polymorphic in the reference; the engine
    runs the kinds its constructors use (Prior, ConstantPrior, OrderedPrior)."""
    if sigm_prior.kind not in (KIND_SIMPLE, KIND_CONSTANT, KIND_ORDERED):
        raise TypeError(f'unsupported sigm_prior type {type(sigm_prior).__name__}')
    if sigm_prior.kind == KIND_CONSTANT:
        return dict(p_ix2=sigm_prior.p_ix, dist2=0, sub_kind=KIND_CONSTANT, value=sigm_prior.value)
    return dict(p_ix2=sigm_prior.p_ix, dist2=add_dist(sigm_prior.dist), sub_kind=sigm_prior.kind,
                value=0.0)


class ResolvedCenSepPrior(Prior):
    """CenSep with a width-dependent minimum separation (reference: core.pyx:321-366)."""
    kind = KIND_RESOLVED_CENSEP

    def __init__(self, vcen_prior, vsep_prior, sigm_prior, scale=1.5):
        self.vcen_prior = vcen_prior
        self.vsep_prior = vsep_prior
        self.sigm_prior = sigm_prior
        self.scale = scale
        self.sep_scale = FWHM * scale
        self.p_ix = 0          # never assigned by the reference (cdef default)
        self.n_param = 2

    def _desc(self, add_dist):
        d = dict(kind=self.kind, p_ix=self.vcen_prior.p_ix, dist0=add_dist(self.vcen_prior.dist),
                 dist1=add_dist(self.vsep_prior.dist), sep_scale=self.sep_scale)
        d.update(_sub_prior_fields(self.sigm_prior, add_dist))
        return d


class ResolvedPlacementPrior(Prior):
    """Sequential placement of up to 10 resolved components (reference: core.pyx:369-435)."""
    kind = KIND_RESOLVED_PLACEMENT

    def __init__(self, vcen_prior, sigm_prior, scale=1.5):
        self.vcen_prior = vcen_prior
        self.sigm_prior = sigm_prior
        self.scale = scale
        self.sep_scale = FWHM * scale
        self.p_ix = 0
        self.n_param = 2

    def _desc(self, add_dist):
        d = dict(kind=self.kind, p_ix=self.vcen_prior.p_ix, dist0=add_dist(self.vcen_prior.dist),
                 dist1=0, sep_scale=self.sep_scale)
        d.update(_sub_prior_fields(self.sigm_prior, add_dist))
        return d


class PriorTransformer:
    """Unit cube -> physical parameters, in place (reference: core.pyx:438-483)."""

    def __init__(self, priors):
        priors = np.asarray(priors, dtype=object)
        n_prior = priors.shape[0]
        assert n_prior >= 1
        self.priors = priors
        self.n_prior = int(n_prior)
        self.n_param = int(sum(p.n_param for p in priors))
        self._handle = None
        self._keep = None

    def lower(self):
        """Flat prior program for the C ABI (and for the test oracle)."""
        dists, index = [], {}

        def add_dist(d):
            if id(d) not in index:
                index[id(d)] = len(dists)
                dists.append(d.lower())
            return index[id(d)]

        descs = [p._desc(add_dist) for p in self.priors]
        return dict(priors=descs, dists=dists, n_param=self.n_param)

    def free_mask(self, ncomp):
        """1 for every unit-cube slot (parameter-major, n_param * ncomp of them) the transformed
        parameters depend on, 0 for the dummies: the slot of a ConstantPrior, the second slot of a
        DuplicatePrior, the width slot of a placement prior whose width prior is constant.  A
        sampler may leave the dummies alone -- a uniform dimension the likelihood ignores integrates
        to one -- which is what the device sampler does."""
        mask = np.ones((self.n_param, ncomp), dtype=np.int32)
        for d in self.lower()['priors']:
            if d['kind'] == KIND_CONSTANT:
                mask[d['p_ix']] = 0
            elif d['kind'] == KIND_DUPLICATE:
                mask[d['p_ix2']] = 0
            elif d['kind'] in (KIND_CENSEP, KIND_RESOLVED_CENSEP, KIND_RESOLVED_PLACEMENT) \
                    and d['sub_kind'] == KIND_CONSTANT:
                mask[d['p_ix2']] = 0
        return np.ascontiguousarray(mask.reshape(-1))

    def _device_handle(self):
        if self._handle is not None:
            return self._handle
        lib = _ffi.engine()
        prog = self.lower()
        pd = (_ffi.PriorDesc * len(prog['priors']))()
        for k, p in enumerate(prog['priors']):
            for name in ('kind', 'p_ix', 'p_ix2', 'dist0', 'dist1', 'dist2', 'sub_kind'):
                setattr(pd[k], name, int(p[name]))
            pd[k].value = float(p['value'])
            pd[k].sep_scale = float(p['sep_scale'])
        dd = (_ffi.DistDesc * max(1, len(prog['dists'])))()
        for k, d in enumerate(prog['dists']):
            dd[k].size = d['size']
            for name in ('du', 'dx', 'xmin', 'xmax'):
                setattr(dd[k], name, float(d[name]))
            for name in ('xax', 'pdf', 'cdf', 'ppf'):
                setattr(dd[k], name, _ffi.dptr(d[name]))
        h = C.c_void_p()
        _ffi.check(lib.nfa_priors_create(C.byref(h), pd, len(prog['priors']), dd,
                                         len(prog['dists']), self.n_param))
        self._keep = prog
        self._handle = h
        return h

    def transform(self, utheta, ncomp):
        """In-place transform of one unit-cube vector (reference: core.pyx:478-483)."""
        utheta = _as_inplace_vector(utheta)
        if self.n_param * ncomp != utheta.shape[0]:
            shape = utheta.shape[0]
            raise ValueError(f'Invalid shape for ncomp={ncomp}: {shape}')
        self.transform_batch(utheta.reshape(1, -1), ncomp)

    def transform_batch(self, U, ncomp):
        """In-place transform of U[B, n_param*ncomp] on the device."""
        U = _as_inplace_matrix(U)
        if self.n_param * ncomp != U.shape[1]:
            raise ValueError(f'Invalid shape for ncomp={ncomp}: {U.shape[1]}')
        lib = _ffi.engine()
        _ffi.check(lib.nfa_priors_transform_batch(self._device_handle(), _ffi.dptr(U), U.shape[0],
                                                  int(ncomp), U.shape[1]))

    def __del__(self):
        if getattr(self, '_handle', None) is not None:
            try:
                _ffi.load().nfa_priors_destroy(self._handle)
            except Exception:
                pass
            self._handle = None


def _as_inplace_vector(a):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.ndim == 1
            and a.flags.c_contiguous and a.flags.writeable):
        raise ValueError('expected a writable, contiguous 1-D float64 array (double[::1])')
    return a


def _as_inplace_matrix(a):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.ndim == 2
            and a.flags.c_contiguous and a.flags.writeable):
        raise ValueError('expected a writable, C-contiguous 2-D float64 array')
    return a


class Spectrum:
    """Frequency axis + data + noise of one spectrum (reference: core.pyx:486-545).

    Like the reference, construction asserts ``noise > 0`` and an ascending
    axis.  Unlike the reference (which keeps memoryviews of the caller's
    arrays), the engine copies them to the device."""

    def __init__(self, xarr, data, noise, rest_freq=None, trans_id=None):
        xarr = np.ascontiguousarray(xarr, dtype=np.float64)
        data = np.ascontiguousarray(data, dtype=np.float64)
        assert noise > 0
        nu_chan = xarr[1] - xarr[0]
        assert nu_chan > 0
        assert xarr.shape == data.shape and xarr.ndim == 1
        self.xarr = xarr
        self.data = data
        self.noise = float(noise)
        self.size = int(xarr.shape[0])
        self.rest_freq = 0 if rest_freq is None else rest_freq
        self.trans_id = -1 if trans_id is None else int(trans_id)
        self.nu_chan = float(nu_chan)
        self.nu_min = float(xarr[0])
        self.nu_max = float(xarr[self.size - 1])
        self.prefactor = -self.size / 2 * np.log(2 * np.pi * noise**2)
        self._pred = np.zeros_like(data)
        self._lnL = None


class HyperfineSpectrum(Spectrum):
    pass


class Runner:
    """Base class keeping the reference attribute set (reference: core.pxd:63-73)."""
    n_model = 0
    ncomp = 0
    n_params = 0
    ndim = 0
    n_chan_tot = 0
    n_spec = 0
    null_lnZ = 0.0
    run_lnZ = float('nan')

    def loglikelihood(self, utheta):
        raise NotImplementedError
