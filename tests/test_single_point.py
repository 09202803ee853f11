"""One point per call (MultiNest's callback): the one-launch point kernel against the batch kernels --
same theta and the same log-likelihood bit for bit, in every numerical mode, with and without a pixel
index, for 1 .. 4 components, 1 .. 6 spectra and 1, 2 or 4 waves per unit; and against the oracle."""
import numpy as np
import pytest

from nestfit_amd.synth import freq_axis

pytestmark = pytest.mark.gpu

LNL_RTOL = {'table': 1e-9, 'poly': 1e-9, 'fast': 1e-6}


def _spec_data(trans, n, seed=5):
    rng = np.random.default_rng(seed)
    return [[freq_axis(t, n), rng.normal(0, 0.2, n), 0.2, t] for t in trans]


@pytest.mark.parametrize('mode', ['table', 'poly', 'fast'])
def test_point_kernel_gives_the_bits_of_the_batch_kernels(engine, nfo, mode):
    from nestfit_amd import _ffi
    engine.set_exp_mode(mode)
    try:
        cases = [((1, 2), 1024, 2), ((1,), 300, 1), ((1, 2, 3), 200, 3), ((1, 2, 4, 5, 6, 2), 130, 2), ((2, 1), 96, 4)]
        for trans, n, ncomp in cases:
            spec_data = _spec_data(trans, n)
            ut = engine.get_irdc_priors(size=300, vsys=0.0)
            U = np.random.default_rng(11).uniform(size=(12, 6 * ncomp))
            for split in (1, 2, 4, 0):
                _ffi.set_option('lnl_split', split)
                run = engine.AmmoniaRunner.from_data(spec_data, ut, ncomp=ncomp)
                Ub = U.copy()
                _ffi.set_option('point', 0)
                want = run.loglikelihood_batch(Ub)              # the batch kernels; Ub = theta
                for point in (1, 0):
                    _ffi.set_option('point', point)
                    for k in range(U.shape[0]):
                        u = U[k].copy()
                        got = run.loglikelihood(u)
                        assert np.array_equal(u, Ub[k]), (trans, ncomp, split, point, k)
                        assert got == want[k] or (np.isnan(got) and np.isnan(want[k])), (trans, ncomp, split, point, k)
                # a broker's handful of points: one workgroup each in the same launch
                _ffi.set_option('point', 1)
                for a, b in ((0, 2), (3, 10), (0, 12)):
                    Uf = U[a:b].copy()
                    got = run.loglikelihood_batch(Uf)
                    assert np.array_equal(Uf, Ub[a:b]) and np.array_equal(got, want[a:b], equal_nan=True), (trans, ncomp, split, a, b)
            cpu = nfo.AmmoniaRunner([nfo.AmmoniaSpectrum(*sd) for sd in spec_data], nfo.PriorSet(ut.lower()), ncomp=ncomp)
            Uc = U.copy()
            np.testing.assert_allclose(want, cpu.loglikelihood_batch(Uc), rtol=LNL_RTOL[mode])
    finally:
        _ffi.set_option('lnl_split', 0)
        _ffi.set_option('point', 1)
        engine.set_exp_mode('fast')


def test_point_kernel_with_a_pixel_index_and_interleaved_batches(engine):
    """A cube runner: the pixel index travels in the kernel arguments; points and batches alternate on the
    same runner (the point kernel shares the lane's record buffers with the batch kernels)."""
    from nestfit_amd.cube import CubeRunner
    rng = np.random.default_rng(3)
    n, n_pix = 256, 9
    axes = [freq_axis(1, n), freq_axis(2, n)]
    data = rng.normal(0, 0.2, (n_pix, 2 * n))
    noise = rng.uniform(0.1, 0.3, (n_pix, 2))
    rc = CubeRunner(axes, (1, 2), data, noise, engine.get_irdc_priors(size=300, vsys=0.0), ncomp=2)
    U = rng.uniform(size=(40, 12))
    pix = rng.integers(0, n_pix, size=40).astype(np.int32)
    from nestfit_amd import _ffi
    Ub = U.copy()
    _ffi.set_option('point', 0)
    try:
        want = rc.loglikelihood_batch(pix, Ub)                  # the batch kernels
    finally:
        _ffi.set_option('point', 1)
    for k in range(40):
        u = U[k:k + 1].copy()
        assert rc.loglikelihood_batch(pix[k:k + 1], u)[0] == want[k]
        assert np.array_equal(u[0], Ub[k])
        if k % 7 == 0:
            Ub2 = U.copy()
            assert np.array_equal(rc.loglikelihood_batch(pix, Ub2), want)          # 40 points: one launch
            big = np.tile(U, (3, 1))                                               # 120 points: the batch kernels
            assert np.array_equal(rc.loglikelihood_batch(np.tile(pix, 3), big), np.tile(want, 3))
