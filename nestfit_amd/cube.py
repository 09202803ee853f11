"""Pixel sharding of a cube over GPUs (one process per GPU) and the device-
resident multi-pixel runner.

The reference fits map pixels independently and stripes them over processes with
``(lon_ix[i::nproc], lat_ix[i::nproc])`` (nestfit/main.py:565-571); each process
writes its own chunk file and nothing is exchanged while sampling
(nestfit/main.py:423-474, docs/store_spec.rst:12-32).  Here rank r owns the same
stripe, uploads its pixels once and evaluates batches of (pixel, unit-cube row)
items.  The only collective is the end-of-run gather of fixed-size per-pixel
records (`nestfit_amd.comm`: RCCL over xGMI through the engine's C ABI).
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._model import _RunnerHandle, _SpecSet
from .core import _as_inplace_matrix


def get_multiproc_indices(shape, nproc):
    """Same striping as the reference (nestfit/main.py:565-571)."""
    lon_ix, lat_ix = np.indices(shape)
    return [(lon_ix[i::nproc, ...].flatten(), lat_ix[i::nproc, ...].flatten()) for i in range(nproc)]


def shard_pixels(shape, rank, world):
    """(lon, lat) index arrays of the pixels owned by `rank` (i_lon % world == rank)."""
    if min(shape) < 1 or not (0 <= rank < world):
        raise ValueError('invalid shard request')
    return get_multiproc_indices(shape, world)[rank]


class CubeRunner:
    """AmmoniaRunner semantics for many pixels that share their frequency axes.

    Parameters
    ----------
    xarrs : list of 1-D frequency axes (Hz, ascending), one per transition
    trans_ids : list of int
    data : array [n_pix, sum(len(x) for x in xarrs)], K, spectra concatenated per pixel
    noise : array [n_pix, n_spec], K
    utrans : PriorTransformer
    """

    def __init__(self, xarrs, trans_ids, data, noise, utrans, ncomp=1, cold=False, lte=False,
                 model=0, rest_freqs=None):
        """model: 0 ammonia (default), 1 diazenylium, 2 gaussian (then `rest_freqs` = [Hz])."""
        assert ncomp > 0
        self._ss = _SpecSet(xarrs, trans_ids, data, noise, model=model, rest_freqs=rest_freqs)
        self._run = _RunnerHandle(self._ss, utrans, ncomp, cold, lte)
        self.utrans = utrans
        self.ncomp = int(ncomp)
        self.n_model = {0: 6, 1: 4, 2: 3}[int(model)]
        self.n_params = self.ndim = self.n_model * self.ncomp
        self.n_pix = self._ss.n_pix
        self.n_spec = self._ss.n_spec
        self.n_chan_tot = self._ss.chan_tot
        self.null_lnZ = self._ss.null_lnZ().sum(axis=1)      # per pixel

    def set_exp_mode(self, mode):
        """Numerical mode of this runner alone (None: the process default again)."""
        self._run.set_exp_mode(mode)

    def loglikelihood_batch(self, pix, U, out=None):
        """lnL[B] of unit-cube rows U[B, ndim] against pixels pix[B]; U is overwritten
        with the physical parameters (like Runner.loglikelihood, core.pyx:558-561).  `out`: where lnL goes;
        arrays from `nestfit_amd.pinned_empty` are used by the kernels in place (no copies)."""
        U = _as_inplace_matrix(U)
        if U.shape[1] != self.ndim:
            raise ValueError(f'Invalid shape for ncomp={self.ncomp}: {U.shape[1]}')
        pix = np.ascontiguousarray(pix, dtype=np.int32)
        if pix.shape != (U.shape[0],):
            raise ValueError('one pixel index per row is required')
        lnL = np.empty(U.shape[0]) if out is None else out
        if lnL.shape != (U.shape[0],) or lnL.dtype != np.float64 or not lnL.flags.c_contiguous:
            raise ValueError('out must be a contiguous float64 array of one value per row')
        _ffi.check(_ffi.load().nfa_runner_loglike_batch(self._run.handle, pix.ctypes.data_as(_ffi._ip),
                                                        _ffi.dptr(U), _ffi.dptr(lnL), U.shape[0]))
        return lnL


    def predict_batch(self, pix, theta, want_spectra=True, out=None):
        """Model spectra [B, n_chan_tot] and lnL[B] of physical parameter rows theta[B, ndim]
        against pixels pix[B]: `runner.predict` for many pixels at once (the spectra-out mode of
        the post-processing, nestfit/main.py:1106-1113, 1186).  `out`: where the spectra go; an array from
        `nestfit_amd.pinned_empty` is written by the kernel itself (no staging copy of B x n_chan_tot doubles)."""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        if theta.ndim != 2 or theta.shape[1] != self.ndim:
            raise ValueError(f'Invalid shape for ncomp={self.ncomp}: {theta.shape}')
        pix = np.ascontiguousarray(pix, dtype=np.int32)
        if pix.shape != (theta.shape[0],):
            raise ValueError('one pixel index per row is required')
        B = theta.shape[0]
        spec = (np.empty((B, self.n_chan_tot)) if out is None else out) if want_spectra else None
        if spec is not None and (spec.shape != (B, self.n_chan_tot) or spec.dtype != np.float64 or not spec.flags.c_contiguous):
            raise ValueError('out must be a contiguous float64 array [B, n_chan_tot]')
        lnl = np.empty(B)
        _ffi.check(_ffi.load().nfa_runner_predict_batch(
            self._run.handle, pix.ctypes.data_as(_ffi._ip), _ffi.dptr(theta), B,
            _ffi.dptr(spec) if want_spectra else None, _ffi.dptr(lnl)))
        return spec, lnl

    def peak_and_integrated(self, pix, theta):
        """max_spec and sum_spec of every spectrum for parameter rows (core.pyx:532-539 as
        `deblend_hf_intensity` uses them, main.py:1110-1113): two arrays [B, n_spec]."""
        if getattr(self, '_scratch_spec', None) is None or self._scratch_spec.shape[0] < theta.shape[0]:
            self._scratch_spec = _ffi.pinned_empty((max(theta.shape[0], 1), self.n_chan_tot))
        spec, _ = self.predict_batch(pix, theta, out=self._scratch_spec[:theta.shape[0]])
        off = self._ss.offsets
        peak = np.stack([np.nanmax(spec[:, off[k]:off[k + 1]], axis=1) for k in range(self.n_spec)], axis=1)
        tot = np.stack([np.nansum(spec[:, off[k]:off[k + 1]], axis=1) for k in range(self.n_spec)], axis=1)
        return peak, tot
