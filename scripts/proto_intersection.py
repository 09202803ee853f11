"""Prototype (numpy, one pixel, CPU oracle as the likelihood): nested sampling by rejection from the INTERSECTION of
several cheap supersets of the live region.  A proposal is drawn uniformly from the bounding ellipsoid of all sampled
dimensions and thrown away -- without a likelihood evaluation -- when it lies outside any of the other bounds:

    box     the axis-aligned bounding box of the live points (a margin per side)
    pbox    their bounding box in the ellipsoid's principal axes
    blocks  per velocity component, the union of up to four ellipsoids around its five parameters (the one-component
            sampler's bound, projected): the live region lies inside the product of its projections
    pairs   per pair of sampled dimensions, the 2-D bounding ellipsoid of the projection

What remains is uniform over the intersection, which contains the region {L > L*} whenever every bound does.

    python scripts/proto_intersection.py <ncomp> <bounds: e.g. ell or ell+box+blocks> <n seeds> [ntot] [first seed] [margin]
"""
import math
import sys
import time

import numpy as np

sys.path.insert(0, '/root/repo')


def ln_vball(d):
    return 0.5 * d * math.log(math.pi) - math.lgamma(0.5 * d + 1.0)


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


def cut_fit(Y, enlarge, max_ell=4, gain=0.7):
    """Up to max_ell ellipsoids around Y: principal-axis cuts kept by the volume test (the shipped _fit_multi's rule)."""
    n, d = Y.shape
    minp = 2 * (d + 2)
    lab = np.zeros(n, dtype=int)
    fits, final = [fit_one(Y, enlarge)], [False]
    while len(fits) < max_ell:
        best = -1
        for k, f in enumerate(fits):
            if not final[k] and (lab == k).sum() >= 2 * minp and (best < 0 or f[3] > fits[best][3]):
                best = k
        if best < 0:
            break
        c, _, _, lnv, cov = fits[best]
        v = np.ones(d)
        for _ in range(20):
            w = cov @ v
            v = w / np.linalg.norm(w)
        idx = np.flatnonzero(lab == best)
        side = ((Y[idx] - c) @ v) >= 0
        ia, ib = idx[~side], idx[side]
        if ia.size < minp or ib.size < minp:
            final[best] = True
            continue
        fa, fb = fit_one(Y[ia], enlarge), fit_one(Y[ib], enlarge)
        if np.logaddexp(fa[3], fb[3]) < lnv + math.log(gain):
            lab[ib] = len(fits)
            fits[best] = fa
            fits.append(fb)
            final[best] = False
            final.append(False)
        else:
            final[best] = True
    out = []
    for c, L, r2, lnv, _ in fits:
        A = L * (math.sqrt(r2) * math.exp(math.log(enlarge) / d))
        out.append((c, np.linalg.inv(A), lnv))
    return out


def os_box(P, margin):
    """Bounding box of the columns of P with a margin per face.  margin > 0: that fraction of the range; margin = (k, c)
    (a tuple): c times the distance between the extreme and the k-th extreme point of that face -- small where the
    marginal ends abruptly (a box-like direction), large where it thins out (the projection of a round body)."""
    lo, hi = P.min(axis=0), P.max(axis=0)
    if isinstance(margin, tuple) and margin[0] == 's':
        # margin from the spread alone: c max(0.1 sigma, (extreme - mean) - a sigma).  A flat marginal ends near 1.73 sigma
        # (margin ~ 0.25 c sigma), the projection of a round ten-dimensional body near 2.9 sigma (margin ~ 1.4 c sigma):
        # the same adaptivity as the order statistics', from a maximum and a variance
        _, c, a = margin
        mu, sg = P.mean(axis=0), P.std(axis=0, ddof=1)
        return lo - c * np.maximum(0.1 * sg, (mu - lo) - a * sg), hi + c * np.maximum(0.1 * sg, (hi - mu) - a * sg)
    if isinstance(margin, tuple):
        k, c = margin
        S = np.sort(P, axis=0)
        return lo - c * (S[k - 1] - S[0]), hi + c * (S[-1] - S[-k])
    m = margin * (hi - lo)
    return lo - m, hi + m


def shear_features(X, i, kind, blocks):
    """Columns the coordinate i is regressed on: the earlier coordinates, their squares and (kind 'x') the products of
    pairs of them inside one velocity component / (kind 'X') all pairs.  An additive triangular map w_i = u_i - g_i(u_<i)
    has a unit Jacobian: uniform in w is uniform in u."""
    cols = [np.ones(X.shape[0])]
    for j in range(i):
        cols.append(X[:, j])
        cols.append(X[:, j] ** 2)
    if kind in ('x', 'X'):
        comp = np.zeros(X.shape[1], dtype=int)
        for c, b in enumerate(blocks):
            comp[b] = c
        for j in range(i):
            for k in range(j + 1, i):
                if kind == 'X' or comp[j] == comp[k]:
                    cols.append(X[:, j] * X[:, k])
    return np.stack(cols, axis=1)


class Shear:
    def __init__(self, U, kind, blocks, ridge=1e-6):
        n, D = U.shape
        self.loo = kind.endswith('L')                     # the fit's own points by their leave-one-out residuals
        kind = kind.rstrip('L')
        self.kind, self.blocks, self.coef = kind, blocks, [None] * D
        self.mu, self.sg = U.mean(axis=0), U.std(axis=0) + 1e-12
        Z = (U - self.mu) / self.sg                       # standardised: conditioning of the normal equations
        for i in range(1, D):
            F = shear_features(Z, i, kind, blocks)
            A = F.T @ F + ridge * n * np.eye(F.shape[1])
            self.coef[i] = np.linalg.solve(A, F.T @ Z[:, i])
            if self.loo:
                h = np.einsum('ij,ij->i', F @ np.linalg.inv(A), F)
                Wl = self.__dict__.setdefault('W_loo', Z.copy())
                Wl[:, i] = (Z[:, i] - F @ self.coef[i]) / (1.0 - h)

    def fwd(self, X):
        Z = (X - self.mu) / self.sg
        W = Z.copy()
        for i in range(1, Z.shape[1]):
            W[:, i] = Z[:, i] - shear_features(Z, i, self.kind, self.blocks) @ self.coef[i]
        return W                                           # standardised and sheared: a constant Jacobian (prod 1 / sg)

    def inv(self, W):
        Z = W.copy()
        for i in range(1, W.shape[1]):
            Z[:, i] = W[:, i] + shear_features(Z, i, self.kind, self.blocks) @ self.coef[i]
        return Z * self.sg + self.mu


class Bound:
    def __init__(self, U, blocks, which, enlarge, efr, ln_x, margin, enlarge_main=1.5):
        n, D = U.shape
        self.which = which
        self.shear = None
        self.lo, self.hi = os_box(U, margin)
        for tok in which.split('+'):
            if tok.startswith('shear'):
                self.shear = Shear(U, tok[5:], blocks)
        if self.shear is not None:
            self.ln_jac = float(np.log(self.shear.sg).sum())       # ln |du / dw|
            U = self.shear.W_loo if self.shear.loo else self.shear.fwd(U)
            ln_x = ln_x - self.ln_jac                               # the prior volume to hold, in w units
        enlarge, enlarge_f = enlarge_main, enlarge           # the sampling ellipsoid keeps its own factor; `enlarge` is the filters'
        c, L, r2, lnv, cov = fit_one(U, enlarge)
        grow = max((ln_x - math.log(efr)) - lnv, 0.0)
        self.c = c
        self.A = L * (math.sqrt(r2) * math.exp((grow + math.log(enlarge)) / D))
        self.lnv = lnv + grow
        self.use_cube = self.lnv + (self.ln_jac if self.shear is not None else 0.0) >= 0.0
        if 'pbox' in which:
            w, V = np.linalg.eigh(cov)
            self.V = V
            self.plo, self.phi = os_box((U - c) @ V, margin)
        if 'wbox' in which:                                  # the box in the Cholesky frame: z = L^-1 (u - c)
            self.Linv = np.linalg.inv(L)
            self.wlo, self.whi = os_box((U - c) @ self.Linv.T, margin)
        self.nrot = 0
        for tok in which.split('+'):
            if tok.startswith('rbox'):                       # boxes in K fixed rotations of the Cholesky frame
                self.nrot = int(tok[4:])
        if self.nrot:
            self.Linv = np.linalg.inv(L)
            Z = (U - c) @ self.Linv.T
            rr = np.random.default_rng(12345)
            self.Q = [np.linalg.qr(rr.normal(size=(D, D)))[0] for _ in range(self.nrot)]
            self.rb = [os_box(Z @ Q, margin) for Q in self.Q]
        if 'blocks' in which:
            self.blocks = blocks
            self.bell = [cut_fit(U[:, b], enlarge_f) for b in blocks]
        if 'pairs' in which:
            self.pairs = []
            for i in range(D):
                for j in range(i + 1, D):
                    c2, L2, r22, _, _ = fit_one(U[:, [i, j]], enlarge_f)
                    A2 = L2 * (math.sqrt(r22) * math.exp(math.log(enlarge_f) / 2))
                    self.pairs.append((i, j, c2, np.linalg.inv(A2)))
        if 'friends' in which:                               # union of balls around the live points, whitened metric
            self.Linv = np.linalg.inv(L)
            Z = (U - c) @ self.Linv.T
            d2 = ((Z[:, None, :] - Z[None, :, :]) ** 2).sum(axis=2)
            np.fill_diagonal(d2, np.inf)
            self.Z = Z
            self.r2f = d2.min(axis=1).max() * max(enlarge_f, 3.0) ** (2.0 / D)     # largest nearest-neighbour distance, its ball's volume x enlarge

    def member(self, X):
        """The free tests alone (not the sampling ellipsoid)."""
        K = X.shape[0]
        ok = np.ones(K, dtype=bool)
        if 'box' in self.which.split('+'):
            ok &= np.all((X >= self.lo) & (X <= self.hi), axis=1)
        if self.shear is not None:
            X = self.shear.fwd(X)
            y = np.linalg.solve(self.A, (X - self.c).T).T
            ok &= np.sum(y * y, axis=1) <= 1.0 + 1e-9
        if 'pbox' in self.which:
            P = (X - self.c) @ self.V
            ok &= np.all((P >= self.plo) & (P <= self.phi), axis=1)
        if 'blocks' in self.which:
            for b, ells in zip(self.blocks, self.bell):
                inside = np.zeros(K, dtype=bool)
                for c, Ainv, _ in ells:
                    y = (X[:, b] - c) @ Ainv.T
                    inside |= np.sum(y * y, axis=1) <= 1.0
                ok &= inside
        if 'pairs' in self.which:
            for i, j, c2, Ainv in self.pairs:
                y = (X[:, [i, j]] - c2) @ Ainv.T
                ok &= np.sum(y * y, axis=1) <= 1.0
        if 'wbox' in self.which:
            Zx = (X - self.c) @ self.Linv.T
            ok &= np.all((Zx >= self.wlo) & (Zx <= self.whi), axis=1)
        if self.nrot:
            Zx = (X - self.c) @ self.Linv.T
            for Q, (lo, hi) in zip(self.Q, self.rb):
                Pq = Zx @ Q
                ok &= np.all((Pq >= lo) & (Pq <= hi), axis=1)
        if 'friends' in self.which:
            idx = np.flatnonzero(ok)
            if idx.size:
                Zx = (X[idx] - self.c) @ self.Linv.T
                d2 = ((Zx[:, None, :] - self.Z[None, :, :]) ** 2).sum(axis=2).min(axis=1)
                ok[idx] = d2 <= self.r2f
        return ok

    def draw(self, rng, K):
        """K raw draws uniform in the ellipsoid (or the unit cube), the flags of those that pass every free test."""
        D = self.c.size
        if self.use_cube:
            X = rng.uniform(size=(K, D))
        else:
            z = rng.normal(size=(K, D))
            z *= (rng.uniform(size=(K, 1)) ** (1.0 / D)) / np.linalg.norm(z, axis=1)[:, None]
            X = self.c + z @ self.A.T
            if self.shear is not None:
                X = self.shear.inv(X)
        ok = np.all((X >= 0) & (X < 1), axis=1)
        ok &= self.member(X)
        return X, ok


def nested(loglike, blocks, which, nlive=400, tol=0.5, efr=0.3, enlarge=1.5, seed=0, upd=40, margin=0.04, trace=None, max_evals=3_000_000,
           audit=None, fenlarge=1.5):
    """audit: list of (name, which, margin, enlarge): bounds fitted beside the sampling one at every refit; every ACCEPTED
    point (a draw from the true region, as far as the sampling bound holds it) is tested against each: audit_out[name] =
    [accepted points, of which outside the audited bound] per third of the run."""
    rng = np.random.default_rng(seed)
    D = sum(len(b) for b in blocks)
    U = rng.uniform(size=(nlive, D))
    L = loglike(U)
    n_evals, n_raw, it, lnZ = nlive, 0, 0, -np.inf
    ln_shrink = math.log1p(-math.exp(-1.0 / nlive))
    since = upd
    ev_at = n_evals
    while True:
        if since >= upd:
            if isinstance(trace, list) and it % 800 < upd:
                trace.append((it, n_evals - ev_at))
            bound = Bound(U, blocks, which, fenlarge, efr, -it / nlive, margin, enlarge)
            if audit is not None:
                abounds = [(a[0], Bound(U, blocks, 'ell+' + a[1], a[3], efr, -it / nlive, a[2], a[4] if len(a) > 4 else enlarge), a[1]) for a in audit]
            since = 0
        X, ok = bound.draw(rng, 256)
        n_raw += X.shape[0]
        C = X[ok]
        if C.shape[0] == 0:
            continue
        C = C[:64]
        Lc = loglike(C)
        n_evals += C.shape[0]
        for j in range(C.shape[0]):
            w = int(np.argmin(L))
            if Lc[j] > L[w]:
                if audit is not None:
                    for name, ab, wch in abounds:
                        rec = trace.setdefault(name, np.zeros((4, 2)))
                        ph = min(3, it // 3500)
                        rec[ph, 0] += 1
                        rec[ph, 1] += 0 if ab.member(C[j:j + 1])[0] else 1
                lnZ = np.logaddexp(lnZ, -it / nlive + ln_shrink + L[w])
                U[w], L[w] = C[j], Lc[j]
                it += 1
                since += 1
                remain = L.max() - it / nlive
                if np.logaddexp(lnZ, remain) - lnZ < tol or n_evals > max_evals:
                    lnZ = np.logaddexp(lnZ, np.logaddexp.reduce(L) - it / nlive - math.log(nlive))
                    return lnZ, it, n_evals, n_raw


if __name__ == '__main__':
    import nestfit_amd as na
    from nestfit_amd.synth import freq_axis
    from oracle import nfo
    nfo.build(native=True)
    n, noise = 512, 0.1
    ncomp, which, n_seeds = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    ntot = float(sys.argv[4]) if len(sys.argv) > 4 else 14.4
    seed0 = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    margin = sys.argv[6] if len(sys.argv) > 6 else '0.04'
    margin = (int(margin[1:].split('x')[0]), float(margin.split('x')[1])) if margin.startswith('k') else ('s', float(margin[1:].split('x')[0]), float(margin.split('x')[1])) if margin.startswith('s') else float(margin)
    fenlarge = float(sys.argv[7]) if len(sys.argv) > 7 else 1.5
    ut = na.get_irdc_priors(size=500, vsys=0.0)
    ps = nfo.PriorSet(ut.lower())
    axes = [freq_axis(1, n), freq_axis(2, n)]
    truths = {1: np.array([-0.5, 12.0, 5.0, ntot, 0.4, 0.0]),
              2: np.array([-0.5, 1.0, 12.0, 15.0, 5.0, 6.0, ntot, ntot + 0.2, 0.4, 0.4, 0.0, 0.0])}
    rng = np.random.default_rng(0)
    specs = []
    for k, t in enumerate((1, 2)):
        s = nfo.AmmoniaSpectrum(axes[k], np.zeros(n), noise, t, native=True)
        nfo.amm_predict(s, truths[ncomp])
        specs.append(nfo.AmmoniaSpectrum(axes[k], s.get_spec() + rng.normal(0, noise, n), noise, t, native=True))
    run = nfo.AmmoniaRunner(specs, ps, ncomp=ncomp, native=True)
    mask = np.asarray(ut.free_mask(ncomp))
    fmap = np.flatnonzero(mask)
    ndim = 6 * ncomp

    def ll(Us):
        T = np.full((Us.shape[0], ndim), 0.5)
        T[:, fmap] = Us
        out = run.loglikelihood_batch(T)
        out[~np.isfinite(out)] = -1e300
        return out
    comp_of = np.array([f % ncomp for f in fmap])
    blocks = [np.flatnonzero(comp_of == c) for c in range(ncomp)]
    t0 = time.time()
    if which == 'audit_shear':
        audit = [(f'{k} e{e}', k, 0.04, 1.5, e) for k in ('shear', 'shearx', 'shearxL', 'shearX', 'shearXL') for e in (1.5, 2.5, 4.0)]
        for seed in range(seed0, seed0 + n_seeds):
            tr = {}
            lnZ, it, ev, raw = nested(ll, blocks, sys.argv[8] if len(sys.argv) > 8 else 'ell', seed=seed, trace=tr, audit=audit, enlarge=float(sys.argv[9]) if len(sys.argv) > 9 else 1.5)
            print(f'audit seed {seed}: lnZ {lnZ:.2f} iters {it} evals {ev}', flush=True)
            for name, rec in tr.items():
                print(f'   {name:14s} excluded / accepted by quarter of the run: ' + '  '.join(f'{int(b)}/{int(a)}' for a, b in rec) + f'   total {rec[:, 1].sum() / rec[:, 0].sum() * 100:.2f} %', flush=True)
        sys.exit(0)
    if which == 'audit':
        audit = [('box k20x1.5', 'box', (20, 1.5), 1.5), ('box s1x1.5', 'box', ('s', 1.0, 1.5), 1.5), ('box s1.25x1.5', 'box', ('s', 1.25, 1.5), 1.5),
                 ('wbox k20x1.5', 'wbox', (20, 1.5), 1.5), ('wbox s1x1.5', 'wbox', ('s', 1.0, 1.5), 1.5), ('wbox s1.25x1.5', 'wbox', ('s', 1.25, 1.5), 1.5),
                 ('rbox16 k20x1.5', 'rbox16', (20, 1.5), 1.5), ('rbox16 s1x1.5', 'rbox16', ('s', 1.0, 1.5), 1.5), ('rbox16 s1.25', 'rbox16', ('s', 1.25, 1.5), 1.5),
                 ('rbox64 k20x1.5', 'rbox64', (20, 1.5), 1.5), ('rbox64 k20x2', 'rbox64', (20, 2.0), 1.5), ('rbox64 s1x1.5', 'rbox64', ('s', 1.0, 1.5), 1.5),
                 ('rbox64 s1.25', 'rbox64', ('s', 1.25, 1.5), 1.5), ('rbox64 s1.5', 'rbox64', ('s', 1.5, 1.5), 1.5), ('rbox64 s2', 'rbox64', ('s', 2.0, 1.5), 1.5)]
        for seed in range(seed0, seed0 + n_seeds):
            tr = {}
            lnZ, it, ev, raw = nested(ll, blocks, 'ell', seed=seed, trace=tr, audit=audit)
            print(f'audit seed {seed}: lnZ {lnZ:.2f} iters {it} evals {ev}', flush=True)
            for name, rec in tr.items():
                print(f'   {name:14s} excluded / accepted by quarter of the run: ' + '  '.join(f'{int(b)}/{int(a)}' for a, b in rec) + f'   total {rec[:, 1].sum() / rec[:, 0].sum() * 100:.2f} %', flush=True)
        sys.exit(0)
    for seed in range(seed0, seed0 + n_seeds):
        tr = []
        lnZ, it, ev, raw = nested(ll, blocks, which, seed=seed, trace=tr, margin=margin, fenlarge=fenlarge, enlarge=float(sys.argv[8]) if len(sys.argv) > 8 else 1.5)
        print(f'{which:28s} ncomp {ncomp} ntot {ntot} seed {seed}: lnZ {lnZ:.2f} iters {it} evals {ev} ({ev / it:.1f} per iteration) '
              f'raw draws {raw} ({raw / ev:.1f} per evaluation)  [{time.time() - t0:.0f} s]', flush=True)
