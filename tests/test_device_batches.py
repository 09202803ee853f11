"""The device-pointer entry point (nfa_runner_loglike_batch_dev: what bench.py and a GPU-resident caller use) and
the engine's coalescing of batches that arrive back to back: every batch must come out bit for bit as the host
-pointer call gives it, whatever company it travelled in -- groups of four, a rest, shapes that change mid-stream,
batches that cannot be coalesced, a mode switch between two batches, cube runners with pixel arrays and single
-pixel runners without."""
import ctypes as C

import numpy as np
import pytest

from nestfit_amd.synth import freq_axis

pytestmark = pytest.mark.gpu


class _DeviceArrays:
    def __init__(self, lib, check):
        self.lib, self.check, self.ptrs = lib, check, []

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        self.check(self.lib.nfa_malloc(C.byref(p), a.nbytes))
        self.check(self.lib.nfa_memcpy_h2d(p, a.ctypes.data_as(C.c_void_p), a.nbytes))
        self.ptrs.append(p)
        return p

    def empty(self, nbytes):
        p = C.c_void_p()
        self.check(self.lib.nfa_malloc(C.byref(p), nbytes))
        self.ptrs.append(p)
        return p

    def download(self, p, like):
        out = np.empty_like(like)
        self.check(self.lib.nfa_memcpy_d2h(out.ctypes.data_as(C.c_void_p), p, out.nbytes))
        return out

    def free(self):
        for p in self.ptrs:
            self.lib.nfa_free(p)


def _run_on_device(_ffi, handle, batches, sync_after=()):
    """batches: list of (pix or None, U[B, ndim]); returns [(theta, lnL)] after one synchronise at the end (and
    after the batches whose index is in `sync_after`)."""
    lib = _ffi.load()
    dev = _DeviceArrays(lib, _ffi.check)
    try:
        slots = []
        for pix, U in batches:
            d_u = dev.upload(U)
            d_l = dev.empty(8 * U.shape[0])
            d_p = dev.upload(np.ascontiguousarray(pix, dtype=np.int32)) if pix is not None else None
            slots.append((d_p, d_u, d_l, U))
        for k, (d_p, d_u, d_l, U) in enumerate(slots):
            _ffi.check(lib.nfa_runner_loglike_batch_dev(handle, d_p, d_u, d_l, U.shape[0]))
            if k in sync_after:
                _ffi.check(lib.nfa_runner_synchronize(handle))
        _ffi.check(lib.nfa_runner_synchronize(handle))
        return [(dev.download(d_u, U), dev.download(d_l, np.empty(U.shape[0]))) for d_p, d_u, d_l, U in slots]
    finally:
        dev.free()


@pytest.mark.parametrize('mode', ['fast', 'table'])
def test_device_batches_coalesced_or_not_give_the_host_call_bits(engine, mode):
    from nestfit_amd import _ffi
    from nestfit_amd.cube import CubeRunner
    engine.set_exp_mode(mode)
    try:
        rng = np.random.default_rng(12)
        n, n_pix = 192, 6
        axes = [freq_axis(1, n), freq_axis(2, n)]
        data = rng.normal(0, 0.2, (n_pix, 2 * n))
        noise = rng.uniform(0.1, 0.3, (n_pix, 2))
        ut = engine.get_irdc_priors(size=200, vsys=0.0)
        # eleven batches of 256 (a group of eight and a rest of three), then the shape changes, then a batch whose size
        # is no multiple of 64 (never coalesced), then two more of the first shape
        sizes = [256] * 11 + [320] * 3 + [200] + [256] * 2
        batches = []
        for k, B in enumerate(sizes):
            pix = np.full(B, k % n_pix, dtype=np.int32) if k % 3 else rng.integers(0, n_pix, B).astype(np.int32)
            batches.append((pix, rng.uniform(size=(B, 12))))
        for coalesce in (8, 1, 3, 4):
            _ffi.set_option('coalesce', coalesce)
            rc = CubeRunner(axes, (1, 2), data, noise, ut, ncomp=2)
            want = []
            for pix, U in batches:
                Uh = U.copy()
                want.append((Uh, rc.loglikelihood_batch(pix, Uh)))          # host-pointer call: theta in place, lnL
            for sync_after in ((), (1, 9)):
                got = _run_on_device(_ffi, rc._run.handle, batches, sync_after)
                for k, ((th, ln), (wt, wl)) in enumerate(zip(got, want)):
                    assert np.array_equal(th, wt) and np.array_equal(ln, wl, equal_nan=True), (mode, coalesce, sync_after, k)
    finally:
        _ffi.set_option('coalesce', 8)
        engine.set_exp_mode('fast')


def test_single_pixel_runner_and_mode_switch_between_batches(engine):
    """No pixel array (an AmmoniaRunner); the runner's mode changes while two batches are held: they run in the
    mode they were enqueued under."""
    from nestfit_amd import _ffi
    rng = np.random.default_rng(4)
    spec_data = [[freq_axis(t, 256), rng.normal(0, 0.2, 256), 0.2, t] for t in (1, 2)]
    ut = engine.get_irdc_priors(size=200, vsys=0.0)
    run = engine.AmmoniaRunner.from_data(spec_data, ut, ncomp=1)
    U = [rng.uniform(size=(128, 6)) for _ in range(4)]
    want = {}
    for mode in ('fast', 'table'):
        run.set_exp_mode(mode)
        want[mode] = []
        for u in U:
            uh = u.copy()
            want[mode].append((uh, run.loglikelihood_batch(uh)))
    lib = _ffi.load()
    dev = _DeviceArrays(lib, _ffi.check)
    try:
        run.set_exp_mode('fast')
        bufs = [(dev.upload(u), dev.empty(8 * 128)) for u in U]
        for k, (d_u, d_l) in enumerate(bufs):
            if k == 2:
                run.set_exp_mode('table')                   # two batches are held at this point
            _ffi.check(lib.nfa_runner_loglike_batch_dev(run._run.handle, None, d_u, d_l, 128))
        _ffi.check(lib.nfa_device_synchronize())            # flushes what is held, too
        for k, (d_u, d_l) in enumerate(bufs):
            mode = 'fast' if k < 2 else 'table'
            assert np.array_equal(dev.download(d_u, U[k]), want[mode][k][0]), k
            assert np.array_equal(dev.download(d_l, np.empty(128)), want[mode][k][1]), k
    finally:
        dev.free()
        run.set_exp_mode(None)


@pytest.mark.parametrize('rows', [200, 4096, 20000])
def test_pinned_host_buffers_are_used_in_place(engine, rows):
    """Host buffers the device can address (nestfit_amd.pinned_empty -> nfa_host_alloc): the kernels read the unit
    cube and write theta and lnL there, no copies; bit for bit what pageable buffers give -- one chunk, several
    chunks over the lanes, pixels given or not, only some of the buffers pinned."""
    rng = np.random.default_rng(rows)
    spec_data = [[freq_axis(t, 256), rng.normal(0, 0.2, 256), 0.2, t] for t in (1, 2)]
    ut = engine.get_irdc_priors(size=200, vsys=0.0)
    run = engine.AmmoniaRunner.from_data(spec_data, ut, ncomp=2)
    U = rng.uniform(size=(rows, run.ndim))
    want_theta = U.copy()
    want = run.loglikelihood_batch(want_theta)
    pu, pl = engine.pinned_empty((rows, run.ndim)), engine.pinned_empty(rows)
    pu[...] = U
    pl[...] = np.nan
    got = run.loglikelihood_batch(pu, out=pl)
    assert got is pl and np.array_equal(pl, want) and np.array_equal(pu, want_theta)
    pu[...] = U                                            # pinned U, pageable lnL
    assert np.array_equal(run.loglikelihood_batch(pu), want) and np.array_equal(pu, want_theta)
    theta = U.copy()                                       # pageable U, pinned lnL
    pl[...] = np.nan
    run.loglikelihood_batch(theta, out=pl)
    assert np.array_equal(pl, want) and np.array_equal(theta, want_theta)
    with pytest.raises(ValueError, match='out must be'):
        run.loglikelihood_batch(U.copy(), out=np.empty(rows + 1))
    del pu, pl                                             # frees the pinned memory (weakref finaliser)


def test_predict_batch_writes_pinned_spectra_in_place(engine):
    """The spectra-out mode with an output buffer the device addresses itself: same spectra, same lnL."""
    from nestfit_amd.cube import CubeRunner
    rng = np.random.default_rng(12)
    xarrs = [freq_axis(t, 384) for t in (1, 2)]
    data = rng.normal(0, 0.2, (3, 768))
    run = CubeRunner(xarrs, [1, 2], data, np.full((3, 2), 0.2), None, ncomp=2)
    theta = np.column_stack([rng.uniform(-2, 2, 300), rng.uniform(-2, 2, 300), rng.uniform(8, 20, 300), rng.uniform(8, 20, 300),
                             rng.uniform(3, 8, 300), rng.uniform(3, 8, 300), rng.uniform(13.5, 15, 300), rng.uniform(13.5, 15, 300),
                             rng.uniform(0.2, 1, 300), rng.uniform(0.2, 1, 300), rng.uniform(0, 0.5, 300), rng.uniform(0, 0.5, 300)])
    pix = rng.integers(0, 3, 300).astype(np.int32)
    want_spec, want_lnl = run.predict_batch(pix, theta)
    out = engine.pinned_empty((300, 768))
    out[...] = np.nan
    spec, lnl = run.predict_batch(pix, theta, out=out)
    assert spec is out and np.array_equal(out, want_spec) and np.array_equal(lnl, want_lnl)
    peak, tot = run.peak_and_integrated(pix, theta)          # goes through a pinned scratch buffer of its own
    assert np.array_equal(peak[:, 0], np.nanmax(want_spec[:, :384], axis=1))
    np.testing.assert_allclose(tot[:, 1], np.nansum(want_spec[:, 384:], axis=1), rtol=1e-13)


@pytest.mark.parametrize('mode', ['fast', 'table'])
def test_predict_batch_dev_equals_the_host_call(engine, mode):
    """nfa_runner_predict_batch_dev (theta, spectra, lnL in HBM; asynchronous, rotating over the lanes): the host
    call's spectra and lnL bit for bit, theta untouched, with spectra only and with lnL only as well."""
    from nestfit_amd import _ffi
    from nestfit_amd.cube import CubeRunner
    engine.set_exp_mode(mode)
    rng = np.random.default_rng(21)
    xarrs = [freq_axis(t, 320) for t in (1, 2)]
    data = rng.normal(0, 0.2, (4, 640))
    run = CubeRunner(xarrs, [1, 2], data, np.full((4, 2), 0.2), None, ncomp=2)
    lib = _ffi.load()
    dev = _DeviceArrays(lib, _ffi.check)
    try:
        calls = []
        for rows in (256, 70, 1024):
            theta = np.column_stack([rng.uniform(-2, 2, rows), rng.uniform(-2, 2, rows), rng.uniform(8, 20, rows), rng.uniform(8, 20, rows),
                                     rng.uniform(3, 8, rows), rng.uniform(3, 8, rows), rng.uniform(13.5, 15, rows), rng.uniform(13.5, 15, rows),
                                     rng.uniform(0.2, 1, rows), rng.uniform(0.2, 1, rows), rng.uniform(0, 0.5, rows), rng.uniform(0, 0.5, rows)])
            pix = rng.integers(0, 4, rows).astype(np.int32)
            want_spec, want_lnl = run.predict_batch(pix, theta)
            d_t, d_p = dev.upload(theta), dev.upload(pix)
            d_s, d_l, d_s2, d_l2 = dev.empty(rows * 640 * 8), dev.empty(rows * 8), dev.empty(rows * 640 * 8), dev.empty(rows * 8)
            calls.append((rows, theta, want_spec, want_lnl, d_t, d_p, d_s, d_l, d_s2, d_l2))
        for rows, theta, want_spec, want_lnl, d_t, d_p, d_s, d_l, d_s2, d_l2 in calls:     # all in flight together
            _ffi.check(lib.nfa_runner_predict_batch_dev(run._run.handle, d_p, d_t, rows, d_s, d_l))
            _ffi.check(lib.nfa_runner_predict_batch_dev(run._run.handle, d_p, d_t, rows, d_s2, None))
            _ffi.check(lib.nfa_runner_predict_batch_dev(run._run.handle, d_p, d_t, rows, None, d_l2))
        _ffi.check(lib.nfa_runner_synchronize(run._run.handle))
        for rows, theta, want_spec, want_lnl, d_t, d_p, d_s, d_l, d_s2, d_l2 in calls:
            assert np.array_equal(dev.download(d_s, want_spec), want_spec)
            assert np.array_equal(dev.download(d_s2, want_spec), want_spec)
            assert np.array_equal(dev.download(d_l, want_lnl), want_lnl)
            assert np.array_equal(dev.download(d_l2, want_lnl), want_lnl)
            assert np.array_equal(dev.download(d_t, theta), theta)
        assert lib.nfa_runner_predict_batch_dev(run._run.handle, None, calls[0][4], 8, None, None) != 0     # nothing asked for
    finally:
        dev.free()
        engine.set_exp_mode('fast')


@pytest.mark.parametrize('mode', ['fast', 'table'])
def test_predict_batches_coalesced_or_not_give_the_host_call_bits(engine, mode):
    """Consecutive nfa_runner_predict_batch_dev calls of one shape travel as one launch (round 5: like the likelihood's
    batches), every batch writing its own spectra and lnL arrays: eleven batches (a group of eight and a rest), a
    likelihood batch in between (another kind: what is held is launched first), spectra-only batches behind it."""
    from nestfit_amd import _ffi
    from nestfit_amd.cube import CubeRunner
    engine.set_exp_mode(mode)
    rng = np.random.default_rng(33)
    xarrs = [freq_axis(t, 320) for t in (1, 2)]
    data = rng.normal(0, 0.2, (4, 640))
    ut = engine.get_irdc_priors(size=200, vsys=0.0)
    run = CubeRunner(xarrs, [1, 2], data, np.full((4, 2), 0.2), ut, ncomp=2)
    lib = _ffi.load()
    dev = _DeviceArrays(lib, _ffi.check)
    rows = 256

    def draw():
        theta = np.column_stack([rng.uniform(-2, 2, rows), rng.uniform(-2, 2, rows), rng.uniform(8, 20, rows), rng.uniform(8, 20, rows),
                                 rng.uniform(3, 8, rows), rng.uniform(3, 8, rows), rng.uniform(13.5, 15, rows), rng.uniform(13.5, 15, rows),
                                 rng.uniform(0.2, 1, rows), rng.uniform(0.2, 1, rows), rng.uniform(0, 0.5, rows), rng.uniform(0, 0.5, rows)])
        pix = rng.integers(0, 4, rows).astype(np.int32)
        return pix, theta
    try:
        batches = []
        for k in range(16):
            pix, theta = draw()
            want_spec, want_lnl = run.predict_batch(pix, theta)
            batches.append((pix, theta, want_spec, want_lnl))
        U = rng.uniform(size=(rows, 12))
        pix_l = rng.integers(0, 4, rows).astype(np.int32)
        Uh = U.copy()
        want_l = run.loglikelihood_batch(pix_l, Uh)
        for coalesce in (8, 1, 3):
            _ffi.set_option('coalesce', coalesce)
            got = []
            for k, (pix, theta, want_spec, want_lnl) in enumerate(batches):
                d_t, d_p = dev.upload(theta), dev.upload(pix)
                d_s = dev.empty(rows * 640 * 8)
                d_l = dev.empty(rows * 8) if k < 13 else None                     # the last three: spectra only
                if k == 11:                                                          # a likelihood batch cuts the run of predict batches
                    d_u, d_pl, d_ll = dev.upload(U), dev.upload(pix_l), dev.empty(rows * 8)
                    _ffi.check(lib.nfa_runner_loglike_batch_dev(run._run.handle, d_pl, d_u, d_ll, rows))
                _ffi.check(lib.nfa_runner_predict_batch_dev(run._run.handle, d_p, d_t, rows, d_s, d_l))
                got.append((d_s, d_l, d_t))
            _ffi.check(lib.nfa_runner_synchronize(run._run.handle))
            for (d_s, d_l, d_t), (pix, theta, want_spec, want_lnl) in zip(got, batches):
                assert np.array_equal(dev.download(d_s, want_spec), want_spec), (mode, coalesce)
                if d_l is not None:
                    assert np.array_equal(dev.download(d_l, want_lnl), want_lnl), (mode, coalesce)
                assert np.array_equal(dev.download(d_t, theta), theta)
            assert np.array_equal(dev.download(d_ll, want_l), want_l, equal_nan=True) and np.array_equal(dev.download(d_u, Uh), Uh)
    finally:
        _ffi.set_option('coalesce', 8)
        dev.free()
        engine.set_exp_mode('fast')
