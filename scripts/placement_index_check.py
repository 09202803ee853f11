#!/usr/bin/env python3
"""How often does the engine's way of evaluating ResolvedPlacementPrior's CDF change the index the
reference's bisection finds?  

The reference rewrites a 500-entry CDF per component with a running trapezoid sum and bisects it
(nestfit/core/core.pyx:65-161); the engine bisects the same CDF evaluated on demand from prefix
moments (csrc/nfa_setup.h place_partial).  Both are restated here in numpy float64, operation for
operation, for the velocity distribution of get_irdc_priors and random intervals / draws; the
script counts bisection indices that differ and the size of the resulting difference in theta.

    python scripts/placement_index_check.py [n_draws] [out.json]
"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def sequential_cdf(pdf, ilo, ihi, p):
    """cdf_over_interval (core.pyx:109-161): running sum, then normalise."""
    size = pdf.size
    cdf = np.zeros(size)
    cdf[ihi:] = 1.0
    if ihi - ilo == 1:
        cdf[ilo] = 1.0
        return cdf
    inv = 1.0 / float(ihi - ilo)
    csum = 0.0
    for i in range(ilo + 1, ihi):
        s = 1.0 - float(i - ilo) * inv
        scale = 1.0 if p == 0 else s if p == 1 else s * s if p == 2 else s ** p
        csum += 0.5 * (pdf[i] + pdf[i - 1]) * scale
        cdf[i] = csum
    cdf[ilo:ihi] /= csum
    return cdf


def moment_cdf(pdf, m, ilo, ihi, p):
    """The engine's evaluation (nfa_setup.h place_partial / place_cdf_at), vectorised over the index."""
    size = pdf.size
    m0, m1, m2 = m
    cdf = np.zeros(size)
    cdf[ihi:] = 1.0
    if ihi - ilo == 1:
        cdf[ilo] = 1.0
        return cdf
    inv = 1.0 / float(ihi - ilo)
    dilo = float(ilo - size // 2)

    def partial(k):
        d0 = m0[k] - m0[ilo]
        if p == 0:
            return d0
        d1 = (m1[k] - m1[ilo]) - dilo * d0
        if p == 1:
            return d0 - d1 * inv
        d2 = (m2[k] - m2[ilo]) - 2.0 * dilo * (m1[k] - m1[ilo]) + dilo * dilo * d0
        return d0 - 2.0 * d1 * inv + d2 * inv * inv
    k = np.arange(ilo + 1, ihi)
    csum = partial(ihi - 1)
    cdf[ilo + 1:ihi] = partial(k) / csum
    cdf[ilo] = 0.0 / csum
    return cdf


def bisect(cdf, u):
    """cdf_interp's index search (core.pyx:83-96)."""
    if u <= cdf[0]:
        u = 1e-64
    lo, hi = 0, cdf.size
    i = hi // 2
    while i != lo:
        if u > cdf[i]:
            lo = i
        else:
            hi = i
        i = (hi + lo) // 2
    return min(i, cdf.size - 1), u


def interp(cdf, xax, dx, i, u):
    j = min(i + 1, cdf.size - 1)
    slope = (cdf[j] - cdf[i]) / dx
    return 1 / slope * (u - cdf[i]) + xax[i]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    import nestfit_amd as na
    ut = na.get_irdc_priors(size=500, vsys=0.0)
    prog = ut.lower()
    d = next(prog['dists'][p['dist0']] for p in prog['priors'] if p['kind'] == 7)       # the placement prior's vcen
    pdf, xax = np.asarray(d['pdf'], dtype=np.float64), np.asarray(d['xax'], dtype=np.float64)
    size, dx, xmin = pdf.size, float(d['dx']), float(d['xmin'])
    t = np.zeros(size, dtype=np.longdouble)
    t[1:] = 0.5 * (pdf[1:].astype(np.longdouble) + pdf[:-1].astype(np.longdouble))
    ic = (np.arange(size) - size // 2).astype(np.longdouble)
    m = tuple(np.cumsum(t * ic ** q).astype(np.float64) for q in (0, 1, 2))
    rng = np.random.default_rng(2)
    flips, worst, worst_rel, n_int = 0, 0.0, 0.0, 0
    per_interval = 2000
    while n_int * per_interval < n:
        a, b = np.sort(rng.uniform(xax[0], xax[-1], 2))
        ilo = int(min(max(int((a - xmin) / dx), 0), size - 1))
        ihi = int((b - xmin) / dx)
        ihi = ilo + 1 if ihi == ilo else ihi
        ihi = min(max(ihi, 1), size)
        p = int(rng.integers(0, 3))
        c_ref, c_eng = sequential_cdf(pdf, ilo, ihi, p), moment_cdf(pdf, m, ilo, ihi, p)
        for u in rng.uniform(size=per_interval):
            i_ref, u_ref = bisect(c_ref, u)
            i_eng, u_eng = bisect(c_eng, u)
            flips += i_ref != i_eng
            x_ref, x_eng = interp(c_ref, xax, dx, i_ref, u_ref), interp(c_eng, xax, dx, i_eng, u_eng)
            if np.isfinite(x_ref) and np.isfinite(x_eng):
                worst = max(worst, abs(x_ref - x_eng))
                worst_rel = max(worst_rel, abs(x_ref - x_eng) / (xax[-1] - xax[0]))
        n_int += 1
    out = {'draws': n_int * per_interval, 'intervals': n_int, 'index_flips': int(flips),
           'flip_rate': flips / (n_int * per_interval), 'max_abs_theta_diff_kms': worst,
           'max_theta_diff_over_prior_width': worst_rel,
           'note': 'numpy restatement of both CDF evaluations; a flipped index moves theta by the two '
                   'cells\' disagreement at their common edge, which is what max_abs_theta_diff bounds'}
    print(json.dumps(out))
    if len(sys.argv) > 2:
        Path(sys.argv[2]).write_text(json.dumps(out, indent=1) + '\n')


if __name__ == '__main__':
    main()
