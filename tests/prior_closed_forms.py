"""The reference's prior transforms written out independently of the oracle, from nothing but a `Distribution`'s
tables (`xax`, `pdf`, `ppf`, `du`, `dx`): numpy on whole arrays -- `searchsorted` where the reference bisects,
`cumsum` where it keeps a running sum -- so that the oracle's expression-by-expression restatement (oracle/nf_oracle.c)
and these forms only agree if both read the reference text (nestfit/core/core.pyx) the same way.

    ppf_interp             core.pyx:47-63      linear interpolation of the ppf table on the uniform grid of u
    cdf_over_interval      core.pyx:109-161    the CDF restricted to [x_lo, x_hi] with weight (1 - t)^sfact
    cdf_interp             core.pyx:65-107     its inverse: the last index whose CDF lies below u, then a line
    OrderedPrior           core.pyx:241-258    SpacedPrior 261-292    CenSepPrior 295-318
    ResolvedCenSepPrior    core.pyx:321-366    ResolvedPlacementPrior 369-435
"""
import numpy as np

FWHM = 2.3548200450309493            # core.pyx:20


def ppf_interp(d, u):
    u = np.asarray(u, dtype=float)
    i_lo = ((d.size - 1) * u).astype(np.int64)
    i_hi = np.minimum(i_lo + 1, d.size - 1)           # u == 1 reads past the table in the reference; the builds clamp
    slope = (d.ppf[i_hi] - d.ppf[i_lo]) / d.du
    return slope * (u - i_lo * d.du) + d.ppf[i_lo]


def placement_draw(d, x_lo, x_hi, sfact, u):
    """cdf_over_interval(x_lo, x_hi, sfact) followed by cdf_interp(u), for one point."""
    if x_lo > x_hi:
        x_lo, x_hi = x_hi, x_lo
    size = d.size
    i_lo = min(max(int((x_lo - d.xmin) / d.dx), 0), size - 1) if (x_lo - d.xmin) / d.dx < size else size - 1
    i_hi = int((x_hi - d.xmin) / d.dx)
    if i_hi == i_lo:
        i_hi = i_lo + 1
    i_hi = size if i_hi > size else (1 if i_hi < 0 else i_hi)
    cdf = np.zeros(size)
    cdf[i_hi:] = 1.0
    if i_hi - i_lo == 1:
        cdf[i_lo] = 1.0
        csum = 0.0
    else:
        k = np.arange(i_lo + 1, i_hi)
        scale = (1.0 - (k - i_lo) * (1.0 / (i_hi - i_lo))) ** sfact if sfact != 0 else np.ones(k.size)
        cdf[i_lo + 1:i_hi] = np.cumsum(0.5 * (d.pdf[k] + d.pdf[k - 1]) * scale)
        csum = cdf[i_hi - 1]
    with np.errstate(divide='ignore', invalid='ignore'):
        cdf[i_lo:i_hi] /= csum
    if u <= cdf[0]:
        u = 1e-64
    j = int(np.searchsorted(cdf, u, side='left')) - 1     # the last index whose CDF lies below u
    j = min(max(j, 0), size - 1)
    j_hi = min(j + 1, size - 1)
    slope = (cdf[j_hi] - cdf[j]) / d.dx
    return 1 / slope * (u - cdf[j]) + d.xax[j]


class Degenerate(Exception):
    """The draw fell into the reference's own corner where its answer is an artefact of the bisection's path: the
    minimum separations overflow the velocity interval, the shrunk ones fill it exactly, the first interval has no
    width and the rewritten CDF holds an inf (core.pyx:139-140, 160-161: 1.0 / csum with csum == 0) -- not a
    monotonic table any more, so "the last index below u" is not defined."""


def transform(priors, u, n):
    """PriorTransformer.c_transform (core.pyx:459-476) on one unit-cube vector `u` (parameter-major, n components);
    `priors` are the package's prior objects (their `.dist` tables are the only inputs)."""
    th = np.array(u, dtype=float)
    for p in priors:
        kind = type(p).__name__
        ix = p.p_ix * n
        if kind == 'ConstantPrior':
            th[ix:ix + n] = p.value
        elif kind == 'Prior':
            th[ix:ix + n] = ppf_interp(p.dist, th[ix:ix + n])
        elif kind == 'DuplicatePrior':
            th[ix:ix + n] = ppf_interp(p.dist, th[ix:ix + n])
            th[p.p_ix_dup * n:p.p_ix_dup * n + n] = th[ix:ix + n]
        elif kind == 'OrderedPrior':
            # u'_i = 1 - prod_{k<=i} (1 - u_k): each draw conditional on the one before
            up = 1.0 - np.cumprod(1.0 - th[ix:ix + n])
            th[ix:ix + n] = ppf_interp(p.dist, up)
        elif kind == 'SpacedPrior':
            steps = np.concatenate([[ppf_interp(p.prior_indep.dist, th[ix])], ppf_interp(p.prior_depen.dist, th[ix + 1:ix + n])])
            th[ix:ix + n] = np.cumsum(steps)
        elif kind in ('CenSepPrior', 'ResolvedCenSepPrior'):
            if kind == 'ResolvedCenSepPrior':
                ix = p.vcen_prior.p_ix * n
                ixs = p.sigm_prior.p_ix * n
                th[ixs:ixs + n] = transform([p.sigm_prior], th, n)[ixs:ixs + n]
            vcen = ppf_interp(p.vcen_prior.dist, th[ix])
            if n == 1:
                th[ix] = vcen
            elif n == 2:
                vsep = ppf_interp(p.vsep_prior.dist, th[ix + 1])
                if kind == 'ResolvedCenSepPrior':
                    vsep = max(vsep, FWHM * p.scale * np.sqrt(th[ixs] * th[ixs + 1]))
                th[ix], th[ix + 1] = vcen - 0.5 * vsep, vcen + 0.5 * vsep
        elif kind == 'ResolvedPlacementPrior':
            if n > 10:
                continue
            d = p.vcen_prior.dist
            ix, ixs = p.vcen_prior.p_ix * n, p.sigm_prior.p_ix * n
            th[ixs:ixs + n] = transform([p.sigm_prior], th, n)[ixs:ixs + n]
            if n == 1:
                th[ix] = ppf_interp(d, th[ix])
                continue
            sig = th[ixs:ixs + n]
            seps = np.concatenate([[0.0], FWHM * p.scale * np.sqrt(sig[1:] * sig[:-1])])
            v_lo, v_hi = d.xmin, d.xmax
            if seps.sum() > v_hi - v_lo:
                raise Degenerate()
            v_hi -= seps.sum()
            for i in range(n):
                v_lo += seps[i]
                v_hi += seps[i]
                with np.errstate(invalid='ignore'):
                    v_lo = placement_draw(d, v_lo, v_hi, float(n - 1 - i), th[ix + i])
                if not np.isfinite(v_lo):
                    # an interval narrower than one table cell at the table's first cell: the rewritten CDF starts
                    # with 1 / 0 and the reference returns NaN, after which its `<long>` conversions of NaN are
                    # undefined behaviour -- nothing to compare from here on
                    raise Degenerate()
                th[ix + i] = v_lo
        else:
            raise NotImplementedError(kind)
    return th
