"""Prototype: factorised multi-ellipsoid bound for nested sampling (numpy, one pixel)."""
import math, sys, time
import numpy as np

def ln_vball(d): return 0.5 * d * math.log(math.pi) - math.lgamma(0.5 * d + 1.0)

def fit_one(Y, enlarge):
    """bounding ellipsoid of points Y[n, d]: (centre, L scaled, lnvol)"""
    n, d = Y.shape
    c = Y.sum(axis=0) / n
    D = Y - c
    cov = D.T @ D / (n - 1)
    cov = cov + 1e-12 * max(np.trace(cov), 1e-30) * np.eye(d)
    L = np.linalg.cholesky(cov)
    y = np.linalg.solve(L, D.T)
    r2 = float(np.max(np.sum(y * y, axis=0)))
    lnv = ln_vball(d) + 0.5 * d * math.log(r2) + float(np.log(np.diag(L)).sum()) + math.log(enlarge)
    return c, L, r2, lnv

def kmeans2(Y, Lpar, iters=8):
    """split along the principal axis of the parent, Lloyd iterations in the parent's whitened coordinates,
    then reassignment by each cluster's own Mahalanobis distance (MultiNest style)."""
    n, d = Y.shape
    c = Y.mean(axis=0)
    W = np.linalg.solve(Lpar, (Y - c).T).T            # whitened
    cov = (Y - c).T @ (Y - c) / (n - 1)
    v = np.ones(d)
    for _ in range(30):
        v = cov @ v; v /= np.linalg.norm(v)
    proj = (Y - c) @ v
    lab = (proj >= np.median(proj)).astype(int)
    for _ in range(iters):
        if lab.sum() < d + 2 or (1 - lab).sum() < d + 2: return None
        m0, m1 = W[lab == 0].mean(axis=0), W[lab == 1].mean(axis=0)
        new = (np.sum((W - m1) ** 2, axis=1) < np.sum((W - m0) ** 2, axis=1)).astype(int)
        if np.array_equal(new, lab): break
        lab = new
    for _ in range(4):                                # Mahalanobis reassignment with volume weighting
        if lab.sum() < d + 2 or (1 - lab).sum() < d + 2: return None
        ds = []
        for k in (0, 1):
            ck, Lk, r2k, lnvk = fit_one(Y[lab == k], 1.0)
            y = np.linalg.solve(Lk, (Y - ck).T)
            nk = (lab == k).sum()
            ds.append(np.sum(y * y, axis=0) / r2k * math.exp((lnvk - math.log(nk)) * 1.0 / d) ** 0)   # plain scaled distance
        new = (ds[1] < ds[0]).astype(int)
        if np.array_equal(new, lab): break
        lab = new
    if lab.sum() < d + 2 or (1 - lab).sum() < d + 2: return None
    return lab

def fit_block(Y, enlarge, max_ell=4, gain=0.7):
    """list of (c, L_scaled(unit-ball map), lnv) covering points Y (one parameter block)"""
    n, d = Y.shape
    clusters = [np.arange(n)]
    fits = [fit_one(Y, enlarge)]
    changed = True
    while changed and len(clusters) < max_ell:
        changed = False
        order = np.argsort([-f[3] for f in fits])     # largest volume first
        for k in order:
            idx = clusters[k]
            if idx.size < 2 * (d + 2): continue
            lab = kmeans2(Y[idx], fits[k][1])
            if lab is None: continue
            fa, fb = fit_one(Y[idx[lab == 0]], enlarge), fit_one(Y[idx[lab == 1]], enlarge)
            if np.logaddexp(fa[3], fb[3]) < fits[k][3] + math.log(gain):
                clusters[k:k + 1] = [idx[lab == 0], idx[lab == 1]]
                fits[k:k + 1] = [fa, fb]
                changed = True
                break
    return [(c, L * math.sqrt(r2) * math.exp(math.log(enlarge) / d), lnv) for c, L, r2, lnv in fits]

def draw_block(ells, rng, n):
    """n uniform draws from the union of ellipsoids (rejection on the overlap count); returns points, and the number of raw draws"""
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
    return out[:n], raw

def nested(loglike, blocks, nlive=400, tol=0.5, efr=0.3, enlarge=1.5, seed=0, max_ell=4, upd=40, verbose=False):
    """blocks: list of index arrays partitioning the sampled dims. loglike(U[n, D]) -> L[n]"""
    rng = np.random.default_rng(seed)
    D = sum(len(b) for b in blocks)
    U = rng.uniform(size=(nlive, D)); L = loglike(U)
    n_evals = nlive; it = 0; lnZ = -np.inf
    ln_shrink = math.log1p(-math.exp(-1.0 / nlive))
    bound = None; since = upd
    while True:
        if since >= upd:
            ln_x = -it / nlive
            bound = [fit_block(U[:, b], enlarge, max_ell) for b in blocks]
            tot = sum(np.logaddexp.reduce([e[2] for e in ells]) for ells in bound)
            grow = max((ln_x - math.log(efr)) - tot, 0.0)
            if grow > 0:
                s = math.exp(grow / D)
                bound = [[(c, A * s, lnv + grow * len(b) / D) for c, A, lnv in ells] for ells, b in zip(bound, blocks)]
                tot += grow
            use_cube = tot >= 0.0
            since = 0
            if verbose: print(it, 'ln_x %.1f lnV %.1f' % (ln_x, tot), [len(e) for e in bound], n_evals)
        K = 64
        if use_cube:
            C = rng.uniform(size=(K, D))
        else:
            C = np.empty((K, D))
            for ells, b in zip(bound, blocks):
                C[:, b], _ = draw_block(ells, rng, K)
        ok = np.all((C >= 0) & (C < 1), axis=1)
        C = C[ok]
        if C.shape[0] == 0: continue
        Lc = loglike(C); n_evals += C.shape[0]
        for j in range(C.shape[0]):
            w = int(np.argmin(L))
            if Lc[j] > L[w]:
                lnZ = np.logaddexp(lnZ, -it / nlive + ln_shrink + L[w])
                U[w], L[w] = C[j], Lc[j]
                it += 1; since += 1
                remain = L.max() - it / nlive
                if np.logaddexp(lnZ, remain) - lnZ < tol:
                    lnZ = np.logaddexp(lnZ, np.logaddexp.reduce(L) - it / nlive - math.log(nlive))
                    return lnZ, it, n_evals


# ---------------------------------------------------------------------------------------------
# driver: python scripts/proto_multiellipsoid.py <ncomp> new|joint|auto|reject <n seeds>
import sys, math, time
sys.path.insert(0,'/root/repo')
import numpy as np
import nestfit_amd as na
from nestfit_amd import sampler
from nestfit_amd.synth import freq_axis
from oracle import nfo
nfo.build(native=True)
n=512; noise=0.1
ut = na.get_irdc_priors(size=500, vsys=0.0)
ps = nfo.PriorSet(ut.lower())
axes=[freq_axis(1,n),freq_axis(2,n)]
ncomp=int(sys.argv[1]); which=sys.argv[2]
truths={1:np.array([-0.5,12.0,5.0,14.4,0.4,0.0]), 2:np.array([-0.5,1.0, 12.0,15.0, 5.0,6.0, 14.4,14.6, 0.4,0.4, 0.0,0.0])}
rng=np.random.default_rng(0)
specs=[]
for k,t in enumerate((1,2)):
    s=nfo.AmmoniaSpectrum(axes[k],np.zeros(n),noise,t,native=True); nfo.amm_predict(s,truths[ncomp])
    specs.append(nfo.AmmoniaSpectrum(axes[k], s.get_spec()+rng.normal(0,noise,n), noise, t, native=True))
run=nfo.AmmoniaRunner(specs, ps, ncomp=ncomp, native=True)
mask=np.asarray(ut.free_mask(ncomp)); fmap=np.flatnonzero(mask); D=fmap.size; ndim=6*ncomp
def ll_sampled(Us):
    T=np.full((Us.shape[0], ndim),0.5); T[:,fmap]=Us
    out=run.loglikelihood_batch(T); out[~np.isfinite(out)]=-1e300
    return out
# blocks: sampled dims of each component (parameter-major layout: slot = par*ncomp + c)
comp_of=np.array([f % ncomp for f in fmap])
blocks=[np.flatnonzero(comp_of==c) for c in range(ncomp)]
t0=time.time()
if which=='new':
    for seed in range(int(sys.argv[3])):
        lnZ,it,ev=nested(ll_sampled, blocks, seed=seed, max_ell=4)
        print('new  blocks', [b.tolist() for b in blocks], 'lnZ %.2f iters %d evals %d  (%.0f s)'%(lnZ,it,ev,time.time()-t0), flush=True)
elif which=='joint':
    for seed in range(int(sys.argv[3])):
        lnZ,it,ev=nested(ll_sampled, [np.arange(D)], seed=seed, max_ell=4)
        print('new  joint lnZ %.2f iters %d evals %d  (%.0f s)'%(lnZ,it,ev,time.time()-t0), flush=True)
else:
    def loglike(px,U):
        return run.loglikelihood_batch(U)
    for seed in range(int(sys.argv[3])):
        r=sampler.run_nested(loglike, ndim, 1, nlive=400, tol=0.5, efr=0.3, seed=seed+1, free_mask=mask, method=which, batch_target=256)[0]
        print('old ', which, 'lnZ %.2f iters %d evals %d  (%.0f s)'%(r.lnZ,r.n_iter,r.n_evals,time.time()-t0), flush=True)
