"""One point per call (MultiNest's callback): the one-launch point kernel against the batch kernels --
same theta and the same log-likelihood bit for bit, in every numerical mode, with and without a pixel
index, for 1 .. 4 components, 1 .. 6 spectra and 1, 2 or 4 waves per unit; and against the oracle."""
import numpy as np
import pytest

from nestfit_amd.synth import freq_axis

pytestmark = pytest.mark.gpu

LNL_RTOL = {'table': 1e-9, 'fast': 1e-6}


def _spec_data(trans, n, seed=5):
    rng = np.random.default_rng(seed)
    return [[freq_axis(t, n), rng.normal(0, 0.2, n), 0.2, t] for t in trans]


@pytest.mark.parametrize('mode', ['table', 'fast'])
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


def test_every_prior_kind_through_the_fused_stage_and_the_point_kernel(engine):
    """PriorTransformer.transform_batch interprets the prior program prior after prior (prior_items_kernel);
    the set-up stage of a likelihood batch gives every prior its own wave when no two priors write the same
    parameter slot, and the point kernel does the same on its eight waves: the in-place theta of all three
    must be the same bits, for every Prior subclass (core.pyx:169-435) and 1 .. 4 components.  Two priors on
    one slot (the later one wins in the reference's sequence) must still come out in sequence."""
    from scipy import stats
    from nestfit_amd import _ffi
    na = engine
    u = np.linspace(0, 1, 300)
    d_v = na.Distribution(8 * u - 4, stats.beta(5, 5).pdf(u))
    d_sep = na.Distribution(3 * u + 0.7, stats.beta(1.5, 3.5).pdf(u))
    d_s = na.Distribution(2 * u + 0.067, stats.beta(1.5, 5).pdf(u))
    d_t = na.Distribution(23 * u + 7, stats.beta(3, 6.7).pdf(u))
    d_n = na.Distribution(3 * u + 13, stats.beta(2, 2).pdf(u))
    rest = [na.DuplicatePrior(d_t, 1, 2), na.Prior(d_n, 3), na.ConstantPrior(0.25, 5)]
    sets = {
        'ordered': [na.OrderedPrior(d_v, 0), na.Prior(d_s, 4)] + rest,
        'spaced': [na.SpacedPrior(na.Prior(d_v, 0), na.Prior(d_sep, 0)), na.Prior(d_s, 4)] + rest,
        'censep': [na.CenSepPrior(na.Prior(d_v, 0), na.Prior(d_sep, 0)), na.Prior(d_s, 4)] + rest,
        'rcensep': [na.ResolvedCenSepPrior(na.Prior(d_v, 0), na.Prior(d_sep, 0), na.Prior(d_s, 4))] + rest,
        'rplace': [na.ResolvedPlacementPrior(na.Prior(d_v, 0), na.Prior(d_s, 4), scale=1.2)] + rest,
        'rplace_const': [na.ResolvedPlacementPrior(na.Prior(d_v, 0), na.ConstantPrior(0.3, 4))] + rest,
        # slot 3 twice: the second prior overwrites what the first made of it
        # (slot 5 stays what the unit cube holds: six parameters are declared, and the count has to be six)
        'shared_slot': [na.Prior(d_v, 0), na.Prior(d_s, 4), na.Prior(d_t, 3), na.DuplicatePrior(d_t, 1, 2), na.Prior(d_n, 3)],
    }
    rng = np.random.default_rng(77)
    spec_data = _spec_data((1, 2), 128)
    for name, priors in sets.items():
        ut = na.PriorTransformer(np.array(priors, dtype=object))
        for ncomp in (1, 2, 3, 4):
            if name in ('censep', 'rcensep') and ncomp > 2:
                continue                                  # these leave n > 2 components untransformed: nothing to compare lnL on
            U = rng.uniform(size=(150, 6 * ncomp))
            want = U.copy()
            ut.transform_batch(want, ncomp)               # the sequential interpreter
            run = na.AmmoniaRunner.from_data(spec_data, ut, ncomp=ncomp)
            _ffi.set_option('point', 0)
            try:
                got = U.copy()
                lnl = run.loglikelihood_batch(got)        # set-up stage of the batch kernels
            finally:
                _ffi.set_option('point', 1)
            assert np.array_equal(got, want, equal_nan=True), (name, ncomp)     # (a shrunken placement can be NaN, as in the reference)
            few = U[:9].copy()
            lnl_few = run.loglikelihood_batch(few)        # point kernel, nine workgroups
            assert np.array_equal(few, want[:9], equal_nan=True) and np.array_equal(lnl_few, lnl[:9], equal_nan=True), (name, ncomp)
            one = U[11].copy()
            assert run.loglikelihood(one) == lnl[11] or np.isnan(lnl[11])
            assert np.array_equal(one, want[11], equal_nan=True), (name, ncomp)


def test_random_shapes_point_kernel_batch_kernels_oracle(engine, nfo):
    """Twenty random spectra sets -- 1 .. 4 transitions out of (1,1) .. (6,6), ragged channel counts from 40 to
    1500, 1 .. 3 components, cold / lte flags, fast and table mode: the point kernel gives the bits of the batch
    kernels, and both the oracle's log-likelihood within the mode's tolerance."""
    from nestfit_amd import _ffi
    rng = np.random.default_rng(2024)
    try:
        for case in range(20):
            mode = ('fast', 'table')[case % 2]
            engine.set_exp_mode(mode)
            n_spec = int(rng.integers(1, 5))
            trans = [int(t) for t in rng.choice(np.arange(1, 7), size=n_spec, replace=False)]
            sizes = [int(rng.integers(40, 1500)) for _ in trans]
            ncomp = int(rng.integers(1, 4))
            cold, lte = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            spec_data = [[freq_axis(t, n), rng.normal(0, 0.3, n), float(rng.uniform(0.1, 0.5)), t] for t, n in zip(trans, sizes)]
            ut = engine.get_irdc_priors(size=200, vsys=0.0)
            run = engine.AmmoniaRunner.from_data(spec_data, ut, ncomp=ncomp, cold=cold, lte=lte)
            cpu = nfo.AmmoniaRunner([nfo.AmmoniaSpectrum(*sd) for sd in spec_data], nfo.PriorSet(ut.lower()), ncomp=ncomp,
                                    cold=cold, lte=lte)
            U = rng.uniform(size=(33, 6 * ncomp))
            _ffi.set_option('point', 0)
            Ub = U.copy()
            want = run.loglikelihood_batch(Ub)
            _ffi.set_option('point', 1)
            Up = U.copy()
            got = run.loglikelihood_batch(Up)                        # 33 workgroups of the point kernel
            assert np.array_equal(got, want, equal_nan=True) and np.array_equal(Up, Ub), (case, trans, sizes, ncomp, cold, lte)
            one = U[5].copy()
            assert run.loglikelihood(one) == want[5]
            Uc = U.copy()
            np.testing.assert_allclose(want, cpu.loglikelihood_batch(Uc), rtol=LNL_RTOL[mode], err_msg=str((case, trans, sizes, ncomp)))
            np.testing.assert_allclose(Ub, Uc, rtol=1e-9, atol=1e-10)
    finally:
        _ffi.set_option('point', 1)
        engine.set_exp_mode('fast')


@pytest.mark.parametrize('mode', ['fast', 'table'])
def test_long_spectrum_many_row_groups(engine, nfo, mode):
    """20,000 channels = 313 rows of 64: five groups of 64 rows for the signal-free-row bookkeeping in each of the
    four row parts, a window test far from the origin; batch kernels, point kernel and oracle."""
    engine.set_exp_mode(mode)
    try:
        rng = np.random.default_rng(8)
        n = 20000
        spec_data = [[freq_axis(1, n, 400.0), rng.normal(0, 0.3, n), 0.3, 1], [freq_axis(2, 777, 60.0), rng.normal(0, 0.2, 777), 0.2, 2]]
        ut = engine.get_irdc_priors(size=200, vsys=0.0)
        run = engine.AmmoniaRunner.from_data(spec_data, ut, ncomp=2)
        cpu = nfo.AmmoniaRunner([nfo.AmmoniaSpectrum(*sd) for sd in spec_data], nfo.PriorSet(ut.lower()), ncomp=2)
        U = rng.uniform(size=(200, 12))
        Ub, Uc = U.copy(), U.copy()
        want = run.loglikelihood_batch(Ub)                          # 200 rows: the batch kernels
        np.testing.assert_allclose(want, cpu.loglikelihood_batch(Uc), rtol=LNL_RTOL[mode])
        few = U[:6].copy()
        assert np.array_equal(run.loglikelihood_batch(few), want[:6]) and np.array_equal(few, Ub[:6])     # point kernel
    finally:
        engine.set_exp_mode('fast')
