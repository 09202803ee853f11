"""Callback-coalescing broker: concurrent LogLike-shaped calls are served in
batches and return exactly what a direct call returns."""
import ctypes as C
import threading

import numpy as np
import pytest

from nestfit_amd.synth import TRUTH_2COMP, freq_axis

pytestmark = pytest.mark.gpu


def _runner(engine, seed=3):
    rng = np.random.default_rng(seed)
    args = []
    for t in (1, 2):
        x = freq_axis(t, 512)
        s = engine.AmmoniaSpectrum(x, np.zeros(512), 0.2, t)
        engine.amm_predict(s, TRUTH_2COMP)
        args.append([x, s.get_spec() + rng.normal(0, 0.2, 512), 0.2, t])
    ut = engine.get_irdc_priors(size=300, vsys=0.0)
    return engine.AmmoniaRunner.from_data(args, ut, ncomp=2), args, ut


def test_threads_get_bitwise_the_direct_results(engine):
    from nestfit_amd.broker import LikelihoodBroker
    n_threads, n_calls = 48, 40
    run, args, ut = _runner(engine)
    twin = engine.AmmoniaRunner.from_data(args, ut, ncomp=2)      # direct path, same inputs
    U = np.random.default_rng(11).uniform(size=(n_threads, n_calls, run.ndim))
    want_theta = U.reshape(-1, run.ndim).copy()
    want_lnl = twin.loglikelihood_batch(want_theta).reshape(n_threads, n_calls)
    want_theta = want_theta.reshape(U.shape)
    got_lnl = np.zeros((n_threads, n_calls))
    got_theta = U.copy()
    broker = LikelihoodBroker(run, max_batch=256, max_wait_us=2000, n_clients=n_threads)
    errors = []

    def sampler(k):
        try:
            for j in range(n_calls):
                got_lnl[k, j] = broker.loglikelihood(got_theta[k, j])
        except Exception as e:                                    # pragma: no cover
            errors.append(e)
        finally:
            pass

    threads = [threading.Thread(target=sampler, args=(k,)) for k in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    assert np.array_equal(got_lnl, want_lnl)                      # bitwise
    assert np.array_equal(got_theta, want_theta)
    st = broker.stats()
    assert st['n_evals'] == n_threads * n_calls
    assert st['largest_batch'] <= n_threads
    assert st['mean_batch'] > 4, st                               # calls really were coalesced
    print('broker stats', st)
    broker.close()


def test_single_caller_times_out_into_batches_of_one(engine):
    from nestfit_amd.broker import LikelihoodBroker
    run, args, ut = _runner(engine, seed=4)
    twin = engine.AmmoniaRunner.from_data(args, ut, ncomp=2)
    broker = LikelihoodBroker(run, max_batch=64, max_wait_us=50, n_clients=0)
    U = np.random.default_rng(5).uniform(size=(20, run.ndim))
    for u in U:
        a, b = u.copy(), u.copy()
        assert broker.loglikelihood(a) == twin.loglikelihood(b)
        assert np.array_equal(a, b)
    st = broker.stats()
    assert st == dict(n_batches=20, n_evals=20, largest_batch=1, mean_batch=1.0)
    with pytest.raises(ValueError, match='Invalid shape for ncomp=2'):
        broker.loglikelihood(np.zeros(6))
    broker.close()
    broker.close()                                                # idempotent


def test_multinest_signature_with_pixels(engine):
    """nfa_broker_callback(Cube, ndim, npars, lnew, context) from several threads, each bound to
    its own pixel of a cube (context = broker + pixel)."""
    from nestfit_amd import _ffi
    from nestfit_amd.broker import LikelihoodBroker
    from nestfit_amd.cube import CubeRunner
    rng = np.random.default_rng(9)
    n_pix, n = 12, 256
    axes = [freq_axis(1, n), freq_axis(2, n)]
    data = rng.normal(0, 0.3, (n_pix, 2 * n))
    noise = np.full((n_pix, 2), 0.3)
    ut = engine.get_irdc_priors(size=200, vsys=0.0)
    cube = CubeRunner(axes, (1, 2), data, noise, ut, ncomp=1)
    twin = CubeRunner(axes, (1, 2), data, noise, ut, ncomp=1)
    broker = LikelihoodBroker(cube, max_batch=64, max_wait_us=2000, n_clients=n_pix)
    n_calls = 25
    U = rng.uniform(size=(n_pix, n_calls, cube.ndim))
    got = np.zeros((n_pix, n_calls))
    theta = U.copy()

    def sampler(p):
        fn, ctx = broker.client(pix=p)
        ndim, npars, lnew = C.c_int(cube.ndim), C.c_int(cube.ndim), C.c_double()
        for j in range(n_calls):
            fn(_ffi.dptr(theta[p, j]), C.byref(ndim), C.byref(npars), C.byref(lnew), C.byref(ctx))
            got[p, j] = lnew.value
        bad = C.c_int(cube.ndim + 1)                              # no error channel: NaN
        fn(_ffi.dptr(theta[p, 0].copy()), C.byref(bad), C.byref(npars), C.byref(lnew), C.byref(ctx))
        assert np.isnan(lnew.value)

    threads = [threading.Thread(target=sampler, args=(p,)) for p in range(n_pix)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    pix = np.repeat(np.arange(n_pix, dtype=np.int32), n_calls)
    want_theta = U.reshape(-1, cube.ndim).copy()
    want = twin.loglikelihood_batch(pix, want_theta).reshape(n_pix, n_calls)
    assert np.array_equal(got, want)
    assert np.array_equal(theta, want_theta.reshape(U.shape))
    assert broker.stats()['mean_batch'] > 2
    broker.close()
