"""Engine state on the GPU: numerical mode per runner (two runners of different modes used from two
threads give bit for bit what they give alone), the RCCL communicator through the C ABI, a real
(non-synthetic) frequency axis, and the option keys the engine accepts."""
import threading
from pathlib import Path

import numpy as np
import pytest

from nestfit_amd.synth import TRUTH_2COMP, freq_axis

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / 'golden'


def _pixel(engine, nfo, n=512, seed=5):
    rng = np.random.default_rng(seed)
    axes = [freq_axis(t, n) for t in (1, 2)]
    spec_data = []
    for t, x in zip((1, 2), axes):
        s = nfo.AmmoniaSpectrum(x, np.zeros(n), 0.2, t)
        nfo.amm_predict(s, TRUTH_2COMP)
        spec_data.append([x, s.get_spec() + rng.normal(0, 0.2, n), 0.2, t])
    return spec_data


def test_two_runners_in_different_modes_from_two_threads(engine, nfo):
    ut = engine.get_irdc_priors(size=500, vsys=0.0)
    spec_data = _pixel(engine, nfo)
    U = np.random.default_rng(7).uniform(size=(2048, 12))
    # serial references, each runner alone with the process-wide mode
    serial = {}
    for mode in ('table', 'fast'):
        engine.set_exp_mode(mode)
        r = engine.AmmoniaRunner.from_data(spec_data, ut, ncomp=2)
        Um = U.copy()
        serial[mode] = (r.loglikelihood_batch(Um), Um)
    engine.set_exp_mode('fast')                  # the pins must win over the process default
    runners = {m: engine.AmmoniaRunner.from_data(spec_data, ut, ncomp=2) for m in ('table', 'fast')}
    for m, r in runners.items():
        r.set_exp_mode(m)
    got, errors = {}, []

    def work(mode):
        try:
            out = []
            for _ in range(12):                  # interleaves with the other thread's launches
                Um = U.copy()
                out.append((runners[mode].loglikelihood_batch(Um), Um))
            got[mode] = out
        except Exception as exc:                 # pragma: no cover
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(m,)) for m in runners]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    for mode in runners:
        for lnl, theta in got[mode]:
            assert np.array_equal(lnl, serial[mode][0]) and np.array_equal(theta, serial[mode][1])
    assert not np.array_equal(serial['table'][0], serial['fast'][0])      # the modes do differ (1e-7)
    runners['table'].set_exp_mode(None)          # back to the process default
    Um = U.copy()
    engine.set_exp_mode('fast')
    assert np.array_equal(runners['table'].loglikelihood_batch(Um), serial['fast'][0])


def test_option_keys(engine):
    from nestfit_amd import _ffi
    for key, val in (('streams', 0), ('wpb', 1), ('wpb_table', 0), ('lnl_cap', 0), ('sampler_parts', 3), ('coalesce', 8),
                     ('sampler_ellipsoids', 0), ('sampler_walk_factor', 0), ('sampler_walkers', 0), ('sampler_refit_every', 4)):
        _ffi.set_option(key, val)
    for key, val in (('occ', 7), ('ablate', 1), ('no_such_option', 1), ('streams', 99), ('sampler_walkers', 100),
                     ('sampler_ellipsoids', 2), ('sampler_refit_every', 0)):
        with pytest.raises(engine.EngineError):
            _ffi.set_option(key, val)             # removed knob, test-build-only knob, unknown key, bad value


def test_rccl_communicator_through_the_c_abi(engine):
    """world = 1 on the one GPU: librccl.so is opened, a communicator is created from a unique id and
    the collectives are the identity (N ranks on N GPUs: the driver's scaling run; the CPU tests cover
    the N > 1 logic over gloo and sockets)."""
    import ctypes as C
    from nestfit_amd import _ffi
    from nestfit_amd.comm import RcclComm, gather_pixel_records
    buf = (C.c_ubyte * 128)()
    _ffi.check(_ffi.engine().nfa_comm_unique_id(buf))
    assert any(bytes(buf))
    comm = RcclComm(0, 1, bytes(buf))
    x = np.arange(1000, dtype=np.float64)
    assert np.array_equal(comm.allgather(x), x)
    assert np.array_equal(comm.allreduce(x, 'max'), x) and np.array_equal(comm.allreduce(x, 'sum'), x)
    comm.barrier()
    rec = np.arange(21.0).reshape(7, 3)
    assert np.array_equal(gather_pixel_records(rec, comm), rec)
    comm.close()


@pytest.mark.parametrize('mode', ['table', 'fast'])
def test_real_frequency_axis_from_the_reference_fixture(engine, nfo, mode):
    """The reference's test spectrum (380 channels, 12.498 kHz, an observed NH3 (1,1) profile) as data and
    as axis: model spectra and log-likelihoods of the engine against the oracle on a grid that no
    synthetic generator made."""
    from nestfit_amd.cubeio import read_spectrum
    x, d, hdr = read_spectrum(GOLD / 'test_spectrum_11.fits')
    engine.set_exp_mode(mode)
    noise = 0.002
    ut = engine.get_irdc_priors(size=500, vsys=float(299792.458 * (1 - x[x.size // 2] / float(hdr['RESTFRQ']))))
    for ncomp in (1, 2):
        gpu = engine.AmmoniaRunner.from_data([[x, d, noise, 1]], ut, ncomp=ncomp)
        cpu = nfo.AmmoniaRunner([nfo.AmmoniaSpectrum(x, d, noise, 1)], nfo.PriorSet(ut.lower()), ncomp=ncomp)
        U = np.random.default_rng(3 + ncomp).uniform(size=(300, 6 * ncomp))
        Ug, Uc = U.copy(), U.copy()
        lg, lc = gpu.loglikelihood_batch(Ug), cpu.loglikelihood_batch(Uc)
        np.testing.assert_allclose(Ug, Uc, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(lg, lc, rtol=1e-9 if mode == 'table' else 1e-6)
        spec, _ = gpu.predict_batch(Ug[:40])
        for k in range(40):
            s = nfo.AmmoniaSpectrum(x, d, noise, 1)
            nfo.amm_predict(s, Uc[k])
            ref = s.get_spec()
            assert np.array_equal(spec[k] == 0, ref == 0)                  # window support, channel for channel
            nz = np.abs(ref) > 1e-6
            np.testing.assert_allclose(spec[k][nz], ref[nz], rtol=1e-11 if mode == 'table' else 5e-7)
        assert gpu.null_lnZ == pytest.approx(cpu.null_lnZ, rel=1e-12)
    engine.set_exp_mode('fast')
