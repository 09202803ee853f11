"""ctypes binding of include/nestfit_amd.h (the engine's C ABI).

There is deliberately no CPU fallback: if the HIP library is missing or no
gfx950 device is visible, every compute entry point raises `EngineError`.
"""
import os
import ctypes as C
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / 'lib' / 'libnestfit_amd.so'

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


class EngineError(RuntimeError):
    """The HIP engine reported an error (or is unavailable)."""


class DistDesc(C.Structure):
    _fields_ = [('size', C.c_int64), ('du', C.c_double), ('dx', C.c_double),
                ('xmin', C.c_double), ('xmax', C.c_double),
                ('xax', _dp), ('pdf', _dp), ('cdf', _dp), ('ppf', _dp)]


class PriorDesc(C.Structure):
    _fields_ = [('kind', C.c_int32), ('p_ix', C.c_int32), ('p_ix2', C.c_int32),
                ('dist0', C.c_int32), ('dist1', C.c_int32), ('dist2', C.c_int32),
                ('sub_kind', C.c_int32), ('pad_', C.c_int32),
                ('value', C.c_double), ('sep_scale', C.c_double)]


# name -> (restype, argtypes); exactly the symbols declared in the header
SIGNATURES = {
    'nfa_last_error': (C.c_char_p, []),
    'nfa_version': (C.c_int, []),
    'nfa_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'nfa_set_device': (C.c_int, [C.c_int]),
    'nfa_device_synchronize': (C.c_int, []),
    'nfa_device_name': (C.c_int, [C.c_char_p, C.c_int]),
    'nfa_device_uuid': (C.c_int, [C.c_char_p, C.c_int]),
    'nfa_set_exp_mode': (C.c_int, [C.c_int]),
    'nfa_get_exp_mode': (C.c_int, []),
    'nfa_set_option': (C.c_int, [C.c_char_p, C.c_int]),
    'nfa_set_iemtex_table': (C.c_int, [_dp, _dp, C.c_int64]),
    'nfa_specset_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_int, _lp, _ip,
                                     C.POINTER(_dp), C.c_int64, _dp, _dp]),
    'nfa_specset_create_model': (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, _lp, _ip, _dp,
                                           C.POINTER(_dp), C.c_int64, _dp, _dp]),
    'nfa_specset_destroy': (C.c_int, [C.c_void_p]),
    'nfa_specset_set_data': (C.c_int, [C.c_void_p, C.c_int64, _dp]),
    'nfa_specset_null_lnz': (C.c_int, [C.c_void_p, _dp]),
    'nfa_specset_tbg': (C.c_int, [C.c_void_p, _dp]),
    'nfa_specset_chan_tot': (C.c_int64, [C.c_void_p]),
    'nfa_priors_create': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(PriorDesc), C.c_int,
                                    C.POINTER(DistDesc), C.c_int, C.c_int]),
    'nfa_priors_destroy': (C.c_int, [C.c_void_p]),
    'nfa_priors_transform_batch': (C.c_int, [C.c_void_p, _dp, C.c_int64, C.c_int, C.c_int]),
    'nfa_runner_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int]),
    'nfa_runner_destroy': (C.c_int, [C.c_void_p]),
    'nfa_runner_ndim': (C.c_int, [C.c_void_p]),
    'nfa_runner_set_exp_mode': (C.c_int, [C.c_void_p, C.c_int]),
    'nfa_runner_get_exp_mode': (C.c_int, [C.c_void_p]),
    'nfa_runner_loglike_batch': (C.c_int, [C.c_void_p, _ip, _dp, _dp, C.c_int64]),
    'nfa_runner_predict_batch': (C.c_int, [C.c_void_p, _ip, _dp, C.c_int64, _dp, _dp]),
    'nfa_runner_predict_batch_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'nfa_runner_loglike_batch_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int64]),
    'nfa_runner_synchronize': (C.c_int, [C.c_void_p]),
    'nfa_runner_set_profiling': (C.c_int, [C.c_void_p, C.c_int]),
    'nfa_runner_get_profile': (C.c_int, [C.c_void_p, _dp, _lp]),
    'nfa_loglike_callback': (None, [_dp, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, C.c_void_p]),
    'nfa_host_alloc': (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    'nfa_host_free': (C.c_int, [C.c_void_p]),
    'nfa_broker_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int64, C.c_int]),
    'nfa_broker_destroy': (C.c_int, [C.c_void_p]),
    'nfa_broker_set_clients': (C.c_int, [C.c_void_p, C.c_int]),
    'nfa_broker_loglike': (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp]),
    'nfa_broker_callback': (None, [_dp, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, C.c_void_p]),
    'nfa_broker_stats': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    'nfa_ring_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_int]),
    'nfa_ring_create_multi': (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_int, C.c_int]),
    'nfa_ring_attach': (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_int]),
    'nfa_ring_close': (C.c_int, [C.c_void_p]),
    'nfa_ring_stop': (C.c_int, [C.c_void_p]),
    'nfa_ring_ndim': (C.c_int, [C.c_void_p]),
    'nfa_ring_slot': (C.c_int, [C.c_void_p]),
    'nfa_ring_max_points': (C.c_int, [C.c_void_p]),
    'nfa_ring_loglike': (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp]),
    'nfa_ring_loglike_many': (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp, C.c_int]),
    'nfa_ring_callback': (None, [_dp, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, C.c_void_p]),
    'nfa_ring_poll': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, _ip, _ip, _dp, C.POINTER(C.c_int),
                                C.POINTER(C.c_int)]),
    'nfa_ring_complete': (C.c_int, [C.c_void_p, C.c_int, _ip, _dp, _dp, C.c_int]),
    'nfa_ring_serve': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int]),
    'nfa_ring_serve_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    'nfa_ring_stats': (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    'nfa_sampler_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, _ip, C.c_int64, C.c_int, C.c_int,
                                     C.c_int64, C.c_int64, _ip]),
    'nfa_sampler_destroy': (C.c_int, [C.c_void_p]),
    'nfa_sampler_set_pixel_nlive': (C.c_int, [C.c_void_p, _ip, _lp, _ip]),
    'nfa_sampler_set_ellipsoids': (C.c_int, [C.c_void_p, C.c_int]),
    'nfa_sampler_set_boxes': (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    'nfa_sampler_set_shear': (C.c_int, [C.c_void_p, C.c_double]),
    'nfa_sampler_set_pairs': (C.c_int, [C.c_void_p, C.c_double]),
    'nfa_sampler_posterior_packed': (C.c_int, [C.c_void_p, _lp, _dp, _dp, _dp]),
    'nfa_sampler_run': (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int64, C.c_int64, C.c_int,
                                  C.c_double, C.c_int]),
    'nfa_sampler_begin': (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int64, C.c_int64, C.c_int,
                                    C.c_double, C.c_int, C.c_double, C.c_int, C.c_int]),
    'nfa_sampler_advance': (C.c_int, [C.c_void_p, C.c_int64, _lp]),
    'nfa_sampler_counts': (C.c_int, [C.c_void_p, _lp, _lp, _lp]),
    'nfa_sampler_dead': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _dp, _dp, _dp]),
    'nfa_sampler_dead_packed': (C.c_int, [C.c_void_p, _lp, _dp, _dp, _dp]),
    'nfa_sampler_live': (C.c_int, [C.c_void_p, _dp, _dp]),
    'nfa_comm_unique_id': (C.c_int, [C.POINTER(C.c_ubyte)]),
    'nfa_comm_create': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_ubyte), C.c_int, C.c_int]),
    'nfa_comm_destroy': (C.c_int, [C.c_void_p]),
    'nfa_comm_rank': (C.c_int, [C.c_void_p]),
    'nfa_comm_world': (C.c_int, [C.c_void_p]),
    'nfa_comm_allgather': (C.c_int, [C.c_void_p, _dp, C.c_int64, _dp]),
    'nfa_comm_allreduce': (C.c_int, [C.c_void_p, _dp, C.c_int64, C.c_int]),
    'nfa_comm_barrier': (C.c_int, [C.c_void_p]),
    'nfa_malloc': (C.c_int, [C.POINTER(C.c_void_p), C.c_int64]),
    'nfa_free': (C.c_int, [C.c_void_p]),
    'nfa_memcpy_h2d': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    'nfa_memcpy_d2h': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    'nfa_memcpy_d2d': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    'nfa_event_create': (C.c_int, [C.POINTER(C.c_void_p)]),
    'nfa_event_destroy': (C.c_int, [C.c_void_p]),
    'nfa_event_record': (C.c_int, [C.c_void_p, C.c_void_p]),
    'nfa_event_synchronize': (C.c_int, [C.c_void_p]),
    'nfa_event_elapsed_ms': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
}

# include/nestfit_amd_test.h: only in libnestfit_amd_test.so (the engine built with the unit-test hooks)
TEST_SIGNATURES = {
    'nfa_test_fastexp': (C.c_int, [_dp, _dp, C.c_int64, C.c_int]),
    'nfa_test_iemtex': (C.c_int, [_dp, _dp, C.c_int64]),
    'nfa_test_partition': (C.c_int, [_dp, _dp, _dp, C.c_int64]),
    'nfa_test_windows': (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, _ip, _ip]),
    'nfa_test_broker_storm': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _ip, _dp, _dp, _dp]),
    'nfa_test_callback_latency': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _dp, C.c_int, _dp, _dp]),
    'nfa_test_queue_trace': (C.c_int, [C.c_int]),
    'nfa_test_queue_trace_read': (C.c_int, [C.POINTER(C.c_ulonglong)]),
}
TEST_LIB_PATH = HERE / 'lib' / 'libnestfit_amd_test.so'

_lib = None
_test_lib = None
_tables_installed = False


def load():
    """dlopen the engine and bind every symbol of the header (no GPU needed)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise EngineError(
            f'{LIB_PATH} is missing: build it with `python -m nestfit_amd.build` '
            '(the engine has no CPU fallback)')
    # NFA_ENGINE_LIB: development override for A/B runs of two engine builds
    lib = C.CDLL(os.environ.get('NFA_ENGINE_LIB') or str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        raise EngineError((lib or load()).nfa_last_error().decode() or f'engine error {rc}')


def test_engine():
    """The test library (tests and measurement scripts only): a second instance of the engine with the
    unit-test hooks of include/nestfit_amd_test.h and the "ablate" option, its tables installed."""
    global _test_lib
    if _test_lib is None:
        if not TEST_LIB_PATH.exists():
            raise EngineError(f'{TEST_LIB_PATH} is missing: build it with `python -m nestfit_amd.build`')
        lib = C.CDLL(str(TEST_LIB_PATH))
        for name, (res, args) in {**SIGNATURES, **TEST_SIGNATURES}.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        t0_x, t0_y = iemtex_tables()
        check(lib.nfa_set_iemtex_table(dptr(t0_x), dptr(t0_y), t0_x.size), lib)
        _test_lib = lib
    return _test_lib


def test_check(rc):
    check(rc, _test_lib)


def broker_loglike_address():
    """Address of the product library's nfa_broker_loglike, for nfa_test_broker_storm."""
    return C.cast(load().nfa_broker_loglike, C.c_void_p)


def loglike_callback_address():
    """Address of the product library's nfa_loglike_callback (MultiNest's LogLike), for nfa_test_callback_latency."""
    return C.cast(load().nfa_loglike_callback, C.c_void_p)


def pinned_empty(shape, dtype=np.float64):
    """An uninitialised numpy array in pinned, device-addressable host memory (nfa_host_alloc): passed to
    `loglikelihood_batch` it is read and written by the kernels directly, without the copies in and out."""
    import weakref
    dtype = np.dtype(dtype)
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    nbytes = max(int(np.prod(shape)) * dtype.itemsize, 1)
    lib = load()
    p = C.c_void_p()
    check(lib.nfa_host_alloc(C.byref(p), nbytes))
    raw = (C.c_char * nbytes).from_address(p.value)
    weakref.finalize(raw, lib.nfa_host_free, C.c_void_p(p.value))
    return np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def dptr(a):
    return a.ctypes.data_as(_dp)


def iemtex_tables():
    """T0_X, T0_Y as the reference builds them (nestfit/models/hyperfine.pyx:12-20)."""
    H, KB = 6.62607015e-27, 1.380649e-16
    t0_xmin = (H * 23.0e9 / KB) / 8.0
    t0_xmax = (H * 28.0e9 / KB) / 2.7
    t0_x = np.linspace(t0_xmin, t0_xmax, 1000)
    t0_y = 1.0 / (np.exp(t0_x) - 1.0)
    return np.ascontiguousarray(t0_x), np.ascontiguousarray(t0_y)


def engine():
    """Loaded library with the device tables installed (requires a GPU)."""
    global _tables_installed
    lib = load()
    if not _tables_installed:
        t0_x, t0_y = iemtex_tables()
        check(lib.nfa_set_iemtex_table(dptr(t0_x), dptr(t0_y), t0_x.size))
        _tables_installed = True
    return lib


def set_device(index):
    check(load().nfa_set_device(int(index)))


def device_count():
    n = C.c_int(0)
    rc = load().nfa_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def set_exp_mode(mode):
    """0/'table': LDS product tables (reference FastExp, f64 throughout);
    2/'fast': fp32 exponentials on the fp64 float-narrowed arguments."""
    mode = {'table': 0, 'fast': 2}.get(mode, mode)
    check(load().nfa_set_exp_mode(int(mode)))


def set_option(key, value):
    check(load().nfa_set_option(key.encode(), int(value)))


def get_exp_mode():
    return load().nfa_get_exp_mode()
