"""Prior sets of the reference (nestfit/prior_constructors.py:20-141), built on
the engine's `PriorTransformer`.  The distribution shapes and axes are the
reference's: they are the test inputs of the prior-transform kernel."""
import numpy as np
from scipy import stats

from .core import (ConstantPrior, Distribution, DuplicatePrior, Prior, PriorTransformer,
                   ResolvedCenSepPrior, ResolvedPlacementPrior)


def get_irdc_priors(size=500, vsys=0.0):
    """IRDC prior set (reference: prior_constructors.py:20-76).

    voff Beta(5,5) on [-4,4]+vsys km/s placed with `ResolvedPlacementPrior`
    (scale 1.2) using sigm Beta(1.5,5) on [0.067,2.067] km/s; trot Beta(3,6.7)
    on [7,30] K; tex Beta(1,2.5) on [2.8,12.06] K; ntot Beta(10,8.5) on
    [12.5,16.5]; orth fixed to 0."""
    u = np.linspace(0, 1, size)
    x_voff = 8.00 * u - 4.00 + vsys
    x_trot = 23.00 * u + 7.00
    x_tex = 9.26 * u + 2.80
    x_ntot = 4.00 * u + 12.50
    x_sigm = 2.00 * u + 0.067
    d_voff = Distribution(x_voff, stats.beta(5.0, 5.0).pdf(u))
    d_trot = Distribution(x_trot, stats.beta(3.0, 6.7).pdf(u))
    d_tex = Distribution(x_tex, stats.beta(1.0, 2.5).pdf(u))
    d_ntot = Distribution(x_ntot, stats.beta(10.0, 8.5).pdf(u))
    d_sigm = Distribution(x_sigm, stats.beta(1.5, 5.0).pdf(u))
    priors = np.array([
        ResolvedPlacementPrior(Prior(d_voff, 0), Prior(d_sigm, 4), scale=1.2),
        Prior(d_trot, 1),
        Prior(d_tex, 2),
        Prior(d_ntot, 3),
        ConstantPrior(0, 5),
    ])
    return PriorTransformer(priors)


def get_synth_priors(size=500):
    """Synthetic-test prior set after Keown et al. 2019 (reference:
    prior_constructors.py:79-141): uniform voff/vsep/tkin/ntot, log-normal sigm,
    `ResolvedCenSepPrior` with scale 1/FWHM, tex duplicated from tkin (use with
    cold=True, lte=True), orth fixed to 0."""
    u = np.linspace(0, 1, size)
    x_voff = 7.800 * u - 3.90
    x_vsep = 2.570 * u + 0.13
    x_tkin = 17.200 * u + 7.90
    x_ntot = 1.600 * u + 12.95
    x_sigm = 2.025 * u + 0.075
    flat = np.ones_like(u) / size
    d_voff = Distribution(x_voff, flat.copy())
    d_vsep = Distribution(x_vsep, flat.copy())
    d_tkin = Distribution(x_tkin, flat.copy())
    d_ntot = Distribution(x_ntot, flat.copy())
    d_sigm = Distribution(x_sigm, stats.lognorm(1.0, scale=0.136).pdf(u))
    fwhm = 2 * np.sqrt(2 * np.log(2))
    priors = np.array([
        ResolvedCenSepPrior(Prior(d_voff, 0), Prior(d_vsep, 0), Prior(d_sigm, 4), scale=1 / fwhm),
        DuplicatePrior(d_tkin, 1, 2),
        Prior(d_ntot, 3),
        ConstantPrior(0, 5),
    ])
    return PriorTransformer(priors)
