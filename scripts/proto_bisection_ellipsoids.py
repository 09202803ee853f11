"""Prototype 2: recursive principal-axis bisection of the live points (no k-means), up to max_ell ellipsoids."""
import math, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')

def ln_vball(d): return 0.5 * d * math.log(math.pi) - math.lgamma(0.5 * d + 1.0)

def fit_one(Y, enlarge):
    n, d = Y.shape
    c = Y.sum(axis=0) / n
    D = Y - c
    cov = D.T @ D / (n - 1)
    cov = cov + 1e-12 * max(np.trace(cov), 1e-30) * np.eye(d)
    L = np.linalg.cholesky(cov)
    y = np.linalg.solve(L, D.T)
    r2 = float(np.max(np.sum(y * y, axis=0)))
    lnv = ln_vball(d) + 0.5 * d * math.log(r2) + float(np.log(np.diag(L)).sum()) + math.log(enlarge)
    return c, L, r2, lnv, cov

def bisect(Y, cov, c):
    d = Y.shape[1]
    v = np.ones(d)
    for _ in range(20):
        v = cov @ v; v /= np.linalg.norm(v)
    proj = (Y - c) @ v
    return proj >= 0.0

def fit_multi(Y, enlarge, max_ell=4, gain=0.7, min_pts=None):
    n, d = Y.shape
    min_pts = min_pts or 2 * (d + 2)
    clusters = [np.arange(n)]
    fits = [fit_one(Y, enlarge)]
    final = [False]
    while len(clusters) < max_ell:
        cand = [k for k in range(len(clusters)) if not final[k] and clusters[k].size >= 2 * min_pts]
        if not cand: break
        k = max(cand, key=lambda k: fits[k][3])
        idx = clusters[k]
        lab = bisect(Y[idx], fits[k][4], fits[k][0])
        if lab.sum() < min_pts or (~lab).sum() < min_pts:
            final[k] = True; continue
        fa, fb = fit_one(Y[idx[~lab]], enlarge), fit_one(Y[idx[lab]], enlarge)
        if np.logaddexp(fa[3], fb[3]) < fits[k][3] + math.log(gain):
            clusters[k:k + 1] = [idx[~lab], idx[lab]]; fits[k:k + 1] = [fa, fb]; final[k:k + 1] = [False, False]
        else:
            final[k] = True
    return [(c, L * math.sqrt(r2) * math.exp(math.log(enlarge) / d), lnv) for c, L, r2, lnv, _ in fits]

def draw(ells, rng, n):
    d = ells[0][0].size
    lnv = np.array([e[2] for e in ells]); p = np.exp(lnv - lnv.max()); p /= p.sum()
    out = np.empty((0, d)); raw = 0
    invs = [np.linalg.inv(e[1]) for e in ells]
    while out.shape[0] < n:
        m = 2 * (n - out.shape[0]) + 8
        k = rng.choice(len(ells), size=m, p=p)
        z = rng.normal(size=(m, d)); z /= np.linalg.norm(z, axis=1)[:, None]
        z *= rng.uniform(size=(m, 1)) ** (1.0 / d)
        x = np.stack([ells[ki][0] + ells[ki][1] @ zi for ki, zi in zip(k, z)])
        q = np.zeros(m)
        for e, inv in zip(ells, invs):
            y = (x - e[0]) @ inv.T
            q += (np.sum(y * y, axis=1) <= 1.0)
        keep = rng.uniform(size=m) < 1.0 / np.maximum(q, 1)
        out = np.concatenate([out, x[keep]]); raw += m
    return out[:n]

def nested(loglike, D, nlive=400, tol=0.5, efr=0.3, enlarge=1.5, seed=0, max_ell=4, upd=40, K=64):
    rng = np.random.default_rng(seed)
    U = rng.uniform(size=(nlive, D)); L = loglike(U)
    n_evals = nlive; it = 0; lnZ = -np.inf
    ln_shrink = math.log1p(-math.exp(-1.0 / nlive))
    since = upd; nell_hist = []
    while True:
        if since >= upd:
            ln_x = -it / nlive
            ells = fit_multi(U, enlarge, max_ell)
            tot = np.logaddexp.reduce([e[2] for e in ells])
            grow = max((ln_x - math.log(efr)) - tot, 0.0)
            if grow > 0:
                s = math.exp(grow / D); ells = [(c, A * s, lnv + grow) for c, A, lnv in ells]; tot += grow
            use_cube = tot >= 0.0; since = 0; nell_hist.append(len(ells))
        C = rng.uniform(size=(K, D)) if use_cube else draw(ells, rng, K)
        ok = np.all((C >= 0) & (C < 1), axis=1); C = C[ok]
        if C.shape[0] == 0: continue
        Lc = loglike(C); n_evals += C.shape[0]
        for j in range(C.shape[0]):
            w = int(np.argmin(L))
            if Lc[j] > L[w]:
                lnZ = np.logaddexp(lnZ, -it / nlive + ln_shrink + L[w])
                U[w], L[w] = C[j], Lc[j]; it += 1; since += 1
                remain = L.max() - it / nlive
                if np.logaddexp(lnZ, remain) - lnZ < tol:
                    lnZ = np.logaddexp(lnZ, np.logaddexp.reduce(L) - it / nlive - math.log(nlive))
                    return lnZ, it, n_evals, np.mean(nell_hist)

import nestfit_amd as na
from nestfit_amd.synth import freq_axis
from oracle import nfo
nfo.build(native=True)
n = 512; noise = 0.1
ut = na.get_irdc_priors(size=500, vsys=0.0)
ps = nfo.PriorSet(ut.lower())
axes = [freq_axis(1, n), freq_axis(2, n)]
mask = np.asarray(ut.free_mask(1)); fmap = np.flatnonzero(mask); D = fmap.size
for ntot in (14.6, 14.3, 14.0):
    truth = np.array([-0.5, 12.0, 5.0, ntot, 0.4, 0.0])
    rng = np.random.default_rng(0)
    specs = []
    for k, t in enumerate((1, 2)):
        s = nfo.AmmoniaSpectrum(axes[k], np.zeros(n), noise, t, native=True); nfo.amm_predict(s, truth)
        specs.append(nfo.AmmoniaSpectrum(axes[k], s.get_spec() + rng.normal(0, noise, n), noise, t, native=True))
    run = nfo.AmmoniaRunner(specs, ps, ncomp=1, native=True)
    def ll(Us):
        T = np.full((Us.shape[0], 6), 0.5); T[:, fmap] = Us
        out = run.loglikelihood_batch(T); out[~np.isfinite(out)] = -1e300
        return out
    for me in (1, 2, 4, 8):
        res = [nested(ll, D, seed=sd, max_ell=me) for sd in range(3)]
        print(f'ntot {ntot}: up to {me} ellipsoid(s): lnZ {np.mean([r[0] for r in res]):.2f} +- {np.std([r[0] for r in res]):.2f}, iterations {np.mean([r[1] for r in res]):.0f}, '
              f'evaluations {np.mean([r[2] for r in res]) / 1e3:.1f} k, mean ellipsoids {np.mean([r[3] for r in res]):.2f}', flush=True)
