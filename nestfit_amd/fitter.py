"""Cube driver on the batched sampler: the reference's ``CubeFitter`` (nestfit/main.py:380-526)
with its per-pixel loop turned inside out.  The reference forks `nproc` processes, each walking
its longitude stripe pixel by pixel and component count by component count through serial
MultiNest runs.  Here a stripe belongs to one GPU; all its pixels with the same number of live
points are fitted together in lock-step (`sampler.fit_pixels`), first with one component, then
the pixels whose evidence gained at least `lnZ_thresh` with two, and so on -- the same decision
rule per pixel (main.py:452-469), the same groups and attributes in the store.
"""
import inspect
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import sampler
from .cube import get_multiproc_indices
from .store import HdfStore, StoreFile

_MODEL_ID = {'ammonia': 0, 'diazenylium': 1, 'gaussian': 2}


class _RunInfo:
    """What ``Dumper.dump`` reads from a runner (core.pyx:638-660)."""

    def __init__(self, ncomp, null_lnZ, n_chan_tot, n_params):
        self.ncomp, self.null_lnZ, self.n_chan_tot, self.n_params = ncomp, null_lnZ, n_chan_tot, n_params
        self.run_lnZ = np.nan


# run_multinest keywords a CubeFitter passes unless told otherwise (the reference's choice for cube fits:
# fewer live points and a looser tolerance than run_multinest's own defaults, main.py:381-387)
MN_CUBE_DEFAULTS = dict(nlive=100, tol=1.0, efr=0.3, updInt=2000)
# groups of pixels (one per number of live points) sampled side by side: a group of a few hundred pixels is bound by
# the latency of its rounds, not by the device, and the groups' rounds interleave (1 = one group after the other)
GROUP_WORKERS = 3
# device memory the sampler state of one lock-step group may take (dead points dominate: cap x (ndim + 2) doubles
# per pixel); larger groups are fitted in several passes
SAMPLER_MEMORY_BUDGET = 24 << 30
SAMPLER_MAX_NLIVE = 8192               # nfa_sampler_create's limit


def _runs_within_a_factor(nlive, factor):
    """Index sets of the pixels, cut so that inside a set the largest count is at most `factor` times the smallest
    (pixels sorted by count, sets grown greedily from the smallest; every set in ascending pixel order)."""
    order = np.argsort(nlive, kind='stable')
    runs, start = [], 0
    for k in range(1, order.size + 1):
        if k == order.size or nlive[order[k]] > factor * nlive[order[start]]:
            runs.append(np.sort(order[start:k]))
            start = k
    return runs


class CubeFitter:
    mn_default_kwargs = MN_CUBE_DEFAULTS

    def __init__(self, stack, utrans, runner_cls, runner_kwargs=None, lnZ_thresh=11, ncomp_max=2,
                 mn_kwargs=None, nlive_snr_fact=5, nlive_quantum=1, fit_backend=None):
        """Arguments of the reference's CubeFitter (main.py:388-420) plus two of this build:
        `nlive_quantum` -- the reference gives every pixel its own number of live points,
        nlive + int(nlive_snr_fact * snr); pixels are batched by that number rounded up to a multiple
        of `nlive_quantum`.  The default, 1, is exactly the reference's value (main.py:445-447); a larger quantum
        (20, say) makes fewer, larger lock-step groups -- a 20 x 20 cube fits several times faster -- at the price of
        up to quantum - 1 more live points than the reference would give a pixel; a store fitted that way says so
        in its `nlive_quantum` attribute;
        `fit_backend` -- None = the device sampler, otherwise a callable(fitter, lon, lat, ncomp,
        nlive, kw) -> (results, null_lnZ, n_chan_tot) that fits the given pixels some other way (the
        tests plug in the numpy twin of the sampler fed by the CPU oracle, so that the driver logic
        runs without a GPU)."""
        model = inspect.getmodule(runner_cls)
        self.model_id, self.n_model = _MODEL_ID[model.NAME], model.N
        self.stack, self.utrans, self.runner_cls = stack, utrans, runner_cls
        self.runner_kwargs = dict(runner_kwargs or {})
        self.mn_kwargs = {**MN_CUBE_DEFAULTS, **(mn_kwargs or {})}
        self.lnZ_thresh, self.ncomp_max, self.nlive_snr_fact = lnZ_thresh, ncomp_max, nlive_snr_fact
        self.nlive_quantum = max(1, int(nlive_quantum))
        # all pixels of a stripe in one lock-step run, each with its own number of live points (False: one run per
        # distinct count, side by side on `group_workers` threads -- round 2's scheme, kept for comparison)
        self.one_group = True
        self.fit_backend = fit_backend
        self.group_workers = GROUP_WORKERS
        self._tree_lock = threading.Lock()           # the store tree and runner creation: one thread at a time

    def _nlive(self, lon, lat):
        base = int(self.mn_kwargs['nlive'])
        out = np.empty(lon.size, dtype=np.int64)
        for k, (i, j) in enumerate(zip(lon, lat)):
            n = base + int(self.nlive_snr_fact * self.stack.get_max_snr(i, j))      # main.py:444-447
            out[k] = -(-n // self.nlive_quantum) * self.nlive_quantum
        return out

    def fit(self, *args):
        (all_lon, all_lat), chunk_path = args
        all_lon, all_lat = np.asarray(all_lon), np.asarray(all_lat)
        hdf = StoreFile(chunk_path, 'a')
        lon, lat = self.stack.good_pixels(all_lon, all_lat)
        good = set(zip(lon.tolist(), lat.tolist()))
        blind = set(zip(*(a.tolist() for a in self.stack.masked_beam_pixels(all_lon, all_lat))))
        for i_lon, i_lat in zip(all_lon.tolist(), all_lat.tolist()):
            if (i_lon, i_lat) in blind:
                # NaN-free data under an infinite noise (a masked primary-beam pixel of NoiseMap.from_pbimg): the
                # reference runs its sampler on the flat likelihood and stores nbest = 0 (main.py:437-472); the
                # outcome is known beforehand, so the group is written without sampling
                print(f'-- ({i_lon}, {i_lat}) infinite noise: nbest = 0 without sampling')
                group = hdf.require_group(f'/pix/{i_lon}/{i_lat}')
                group.attrs.update(i_lon=int(i_lon), i_lat=int(i_lat), nbest=0)
            elif (i_lon, i_lat) not in good:
                print(f'-- ({i_lon}, {i_lat}) SKIP: has NaN values')
        if lon.size:
            nlive = self._nlive(lon, lat)
            # ('precision': not one of MultiNest's arguments -- the built-in sampler's named setting, sampler.PRECISION)
            kw = {k: self.mn_kwargs[k] for k in ('tol', 'efr', 'seed', 'maxiter', 'precision') if k in self.mn_kwargs}
            if self.one_group:
                # every pixel keeps its own number of live points inside ONE lock-step run (the device sampler's
                # per-pixel counts, nfa_sampler_set_pixel_nlive): no group per count, no rounding of the counts
                # -- but not across more than a factor of two: the live arrays of a run are laid out for its largest
                # count (memory passes, the LDS staging of the refits and the several-ellipsoid bound are decided on
                # that stride), so ONE very bright pixel (nlive + 5 SNR) would size the run of a whole stripe.  Pixels
                # sorted by their counts are cut into runs whose largest count is at most twice the smallest (the real
                # test cube's 100 .. 180 stay one run); a count beyond the sampler's limit is clamped for that pixel alone.
                nlive = np.asarray(nlive)
                if np.ndim(nlive) and nlive.max() > SAMPLER_MAX_NLIVE:
                    over = np.flatnonzero(nlive > SAMPLER_MAX_NLIVE)
                    print(f'-- {over.size} pixel(s) ask for more than {SAMPLER_MAX_NLIVE} live points (up to {int(nlive.max())}): clamped')
                    nlive = np.minimum(nlive, SAMPLER_MAX_NLIVE)
                # (each run carries the counts of ITS pixels: _fit_group indexes them by the pixel's place in the run)
                groups = [(sel, nlive[sel]) for sel in _runs_within_a_factor(nlive, 2.0)] if np.ndim(nlive) else [(np.arange(lon.size), nlive)]
            else:
                groups = [(np.flatnonzero(nlive == nl), int(nl)) for nl in np.unique(nlive)]
            workers = min(len(groups), self.group_workers) if self.fit_backend is None else 1
            if workers > 1:
                # largest groups first; results do not depend on the company (the random streams are keyed by the
                # seed, the pixel's slot in its group and the candidate index)
                groups.sort(key=lambda g: -g[0].size)
                with ThreadPoolExecutor(max_workers=workers) as pool:
                    for f in [pool.submit(self._fit_group, hdf, lon[sel], lat[sel], nl, kw) for sel, nl in groups]:
                        f.result()
            else:
                for sel, nl in groups:
                    self._fit_group(hdf, lon[sel], lat[sel], nl, kw)
        hdf.close()                                  # saves the file (once)
        return hdf

    def _fit_on_device(self, lon, lat, ncomp, nlive, kw):
        with self._tree_lock:
            runner, rlon, rlat = self.stack.to_device(self.utrans, ncomp=ncomp, lon=lon, lat=lat,
                                                      model=self.model_id, **self.runner_kwargs)
        assert np.array_equal(rlon, lon) and np.array_equal(rlat, lat)
        # the sampler keeps every dead point of every pixel on the device: fit the group in passes that fit
        # the memory budget (a pass's pixels keep their slot numbers' random streams; the seed moves on)
        ndim = self.n_model * ncomp
        nl_max = int(np.max(nlive))
        cap = min(int(kw.get('maxiter', 10**6)), sampler.default_cap_iter(nl_max))
        # per pixel: the dead points, the live points, and the packed copy of both the read-back makes at the end of the run
        # (nfa_sampler_posterior_packed); per sampler, whatever the pixels: the proposal buffers, 32 x 262144 rows of the
        # unit-cube point, theta, lnL and three integers (1.8 GB in twelve dimensions)
        per_pixel = 8 * (2 * cap * (ndim + 2) + nl_max * (3 * ndim + 4))
        fixed = 32 * 262144 * (8 * (2 * ndim + 1) + 12)
        n_pass = max(1, (SAMPLER_MEMORY_BUDGET - fixed) // per_pixel)
        res = []
        for a in range(0, lon.size, n_pass):
            kw_pass = dict(kw)
            if a and kw.get('seed', -1) >= 0:
                kw_pass['seed'] = int(kw['seed']) + a
            b = min(lon.size, a + n_pass)
            res += sampler.fit_pixels(runner, np.arange(a, b), nlive=nlive if np.ndim(nlive) == 0 else np.asarray(nlive)[a:b],
                                      **kw_pass)
        return res, runner.null_lnZ.copy(), int(runner._ss.chan_tot)

    def _fit_group(self, hdf, lon, lat, nlive, kw):
        """`nlive`: the group's number of live points, or one per pixel of the group."""
        assert np.ndim(nlive) == 0 or np.shape(nlive) == (lon.size,), 'one live-point count per pixel of the group'
        old_lnZ = None
        nbest = np.zeros(lon.size, dtype=np.int64)
        alive = np.arange(lon.size)                      # pixels still adding components
        ncomp = 1
        while ncomp <= self.ncomp_max and alive.size:
            nl = nlive if np.ndim(nlive) == 0 else np.asarray(nlive)[alive]
            print(f'-- {alive.size} pixels, nlive = {nl if np.ndim(nl) == 0 else f"{int(np.min(nl))}..{int(np.max(nl))}"} -> N = {ncomp}')
            if self.fit_backend is not None:
                res, null_lnZ, n_chan_tot = self.fit_backend(self, lon[alive], lat[alive], ncomp, nl, kw)
            else:
                res, null_lnZ, n_chan_tot = self._fit_on_device(lon[alive], lat[alive], ncomp, nl, kw)
            if ncomp == 1:
                old_lnZ = np.array(null_lnZ, dtype=np.float64)
                assert np.isfinite(old_lnZ).all()
            gain = np.empty(alive.size)
            with self._tree_lock:
                for k, (p, r) in enumerate(zip(alive, res)):
                    group = hdf.require_group(f'/pix/{lon[p]}/{lat[p]}')
                    sub_group = group.create_group(f'{ncomp}')
                    info = _RunInfo(ncomp, float(null_lnZ[k]), n_chan_tot, self.n_model * ncomp)
                    sampler.Dumper(sub_group).dump(info, r)
                    assert np.isfinite(info.run_lnZ)
                    gain[k] = info.run_lnZ - old_lnZ[p]
            keep = gain >= self.lnZ_thresh               # main.py:464-469
            nbest[alive[keep]] = ncomp
            old_lnZ[alive[keep]] = np.array([r.lnZ for r in res])[keep]
            alive = alive[keep]
            ncomp += 1
        with self._tree_lock:
            for p in range(lon.size):
                hdf[f'/pix/{lon[p]}/{lat[p]}'].attrs.update(i_lon=int(lon[p]), i_lat=int(lat[p]), nbest=int(nbest[p]))

    def fit_cube(self, store_name='run/test_cube', nproc=1, rank=None, file_format=None):
        """Creates the store, fits every pixel and links the chunk files (main.py:476-526).
        `nproc` = number of stripes / chunk files (the reference's process count: one per GPU
        here).  With `rank` given (one process per GPU, e.g. under torch.distributed.run) this
        process fits only stripe `rank`; the caller links the files once all ranks are done
        (`HdfStore(store_name).link_files()`).  Without it all stripes are fitted in turn.
        `file_format`: 'hdf5' or 'npz' for a new store (default: HDF5 wherever libhdf5 loads,
        `store.store_format`); every rank has to pass the same."""
        n_lon = self.stack.spatial_shape[0]
        if nproc > n_lon:
            raise ValueError(
                f'The pixel width of the image in longitude ({n_lon}) '
                f'must be greater than or equal to the number of processes ({nproc}).')
        indices = get_multiproc_indices(self.stack.spatial_shape, nproc)
        if rank is not None and rank != 0:           # only rank 0 touches the table file
            from pathlib import Path
            from .store import FILE_SUFFIXES, check_ext, store_format
            store_dir = Path(check_ext(str(store_name), ext='store'))
            store_dir.mkdir(parents=True, exist_ok=True)
            suffix = FILE_SUFFIXES[file_format or store_format()]
            self.fit(indices[rank], store_dir / f'{HdfStore.chunk_prefix}{rank}{suffix}')
            return
        store = HdfStore(store_name, nchunks=nproc, file_format=file_format)
        if 'simple_header' not in store.hdf:
            store.insert_header(self.stack)
        store.insert_fitter_pars(self)
        store.insert_model_metadata(self.runner_cls)
        todo = range(store.nchunks) if rank is None else [0]
        written = {}
        for k in todo:
            written[store.chunk_paths[k]] = self.fit(indices[k], store.chunk_paths[k])
        if rank is None:
            store.link_files(loaded=written)           # the trees just written need not be read back
        store.close()
