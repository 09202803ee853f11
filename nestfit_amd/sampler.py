"""Built-in batched nested sampler: the stand-in for MultiNest when
``libmultinest`` is not available, built so that the GPU sees batches.

The reference drives one serial MultiNest instance per pixel
(``run_multinest``, nestfit/core/core.pyx:727-823; pixel loop nestfit/main.py:452-469), one
likelihood per callback.  Here every pixel of a cube is a nested-sampling run of its own, but all
runs advance in lock-step: each round proposes candidates per active pixel -- uniform in the
bounding ellipsoid(s) of its live points (MultiNest's ellipsoidal rejection sampling, Feroz et al. 2009;
up to four ellipsoids around clusters of live points where at most six dimensions are sampled),
or, where that has become hopeless, one
differential-evolution Metropolis step of each of 64 walkers inside the likelihood constraint --
all candidates of all pixels go to the device in ONE likelihood batch, and every pixel scans its
candidates in order: each one above the pixel's current threshold
replaces its worst live point (Skilling 2006 bookkeeping).  The outputs follow what the
reference's ``mn_dump`` stores (core.pyx:627-687): posterior rows ``[theta..., -2 lnL, weight]``,
``param_constr`` rows 2, 3 = best-fit and MAP, global lnZ and its error, max log-likelihood.

Two implementations of one algorithm live side by side: `run_nested` (numpy, any likelihood
callable) and `run_nested_device` (state and per-round logic on the GPU, csrc/nfa_sampler.h); they
share a counter-based random stream, so the same seed gives the same run.

This is not MultiNest: the random streams differ and the decomposition into ellipsoids is a simpler one
(principal-axis cuts kept by a volume test; none above six sampled dimensions, where constrained walks take
over), so evidences agree with a MultiNest run only within their sampling error.  What can be checked
bit-for-bit is the likelihood it is fed (tests drive the same sampler with the CPU oracle).
"""
import math

import numpy as np

LOG_ZERO = -1e100


class NestedResult:
    """Per-pixel outcome, named like the quantities MultiNest hands to ``mn_dump``."""

    def __init__(self, posterior, lnZ, lnZ_err, max_loglike, n_live, n_evals, n_iter, information):
        self.posterior = posterior                  # (n_samples, n_params + 2)
        self.n_samples = int(posterior.shape[0])
        self.n_params = int(posterior.shape[1] - 2)
        self.lnZ = float(lnZ)
        self.lnZ_err = float(lnZ_err)
        self.max_loglike = float(max_loglike)
        self.n_live = int(n_live)
        self.n_evals = int(n_evals)
        self.n_iter = int(n_iter)
        self.information = float(information)
        # True when the run was stopped by its iteration / dead-point cap before the evidence tolerance was
        # met: lnZ is then the evidence collected so far plus the live points' share, a lower-quality
        # estimate (set by run_nested / run_nested_device)
        self.truncated = False
        w = posterior[:, -1]
        th = posterior[:, :-2]
        best = th[np.argmin(posterior[:, -2])]      # max likelihood
        mapp = th[np.argmax(w)]                     # largest posterior mass
        # the weighted moments about the row of the largest weight, a point inside the posterior's bulk (raw second moments
        # cancel where |mean| >> sigma); one (n_samples, n_params) temporary, like the device's ns_finish_kernel
        d = th - mapp
        m1 = w @ d
        d *= d
        mean = m1 + mapp * w.sum()
        self.param_constr = np.stack([mean, np.sqrt(_var_about(w @ d, mean, mapp, w.sum())), best, mapp])    # (4, n_params)

    @classmethod
    def from_stats(cls, posterior, stats, n_live, n_evals, n_iter):
        """The same result from what the device has already formed of the table (nfa_sampler_posterior_packed with `stats`:
        lnZ, lnZ of the dead points, H, largest lnL, largest live lnL, sum of the weights, mean, second moment about the row of
        the largest weight, theta of the largest likelihood, theta of the largest weight): no pass over the table on the host."""
        self = cls.__new__(cls)
        nd = int(posterior.shape[1] - 2)
        self.posterior = posterior
        self.n_samples = int(posterior.shape[0])
        self.n_params = nd
        self.lnZ = float(stats[0])
        self.information = float(stats[2])
        self.lnZ_err = float(np.sqrt(max(self.information, 0.0) / n_live))
        self.max_loglike = float(stats[3])
        self.n_live, self.n_evals, self.n_iter = int(n_live), int(n_evals), int(n_iter)
        self.truncated = False
        mean, m2, mapp = stats[6:6 + nd], stats[6 + nd:6 + 2 * nd], stats[6 + 3 * nd:6 + 4 * nd]      # (m2: about the row of the largest weight)
        self.param_constr = np.stack([mean, np.sqrt(_var_about(m2, mean, mapp, stats[5])), stats[6 + 2 * nd:6 + 3 * nd], mapp])
        return self


def _var_about(s2, mean, c, wsum):
    """sum w (t - mean)^2 from s2 = sum w (t - c)^2, mean = sum w t and wsum = sum w (the weights add up to one to rounding)."""
    delta = mean - c
    return np.maximum(s2 - 2.0 * delta * (mean - c * wsum) + delta * delta * wsum, 0.0)


# ---- counter-based random numbers, shared bit for bit with csrc/nfa_sampler.h -------------
_U64 = np.uint64
_TAG_LIVE = _U64(1 << 62)
_B_RADIUS = _U64(255)


def _mix(x):
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over='ignore'):
        x = x + _U64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def _uniform(seed, p, a, b):
    """Uniform in (0, 1): a pure function of (seed, pixel, a, b) (ns_uniform on the device)."""
    with np.errstate(over='ignore'):
        h = _mix(_mix(_mix(_mix(np.asarray(seed, dtype=_U64)) + np.asarray(p, dtype=_U64))
                      + np.asarray(a, dtype=_U64)) + np.asarray(b, dtype=_U64))
    return ((h >> _U64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


_B_START = _U64(250)
_NS_W = 64


def _walkers_for(n):
    """Walkers of a pixel with n live points (ns_walkers_for): 64, 128 from 384 live points, 256 from 768."""
    return 256 if n >= 768 else 128 if n >= 384 else 64
_WALK_TARGET = 0.5              # acceptance the walk scale is tuned to (NS_WALK_TARGET on the device)
_WALK_LOWD, _WALK_FACTOR_LOWD, _WALK_FACTOR = 6, 64, 2      # NS_WALK_LOWD, NS_WALK_FACTOR_LOWD, NS_WALK_FACTOR


def _ball_points(seed, p, a, D):
    """Uniform points of the unit D-ball for stream indices a[K] of pixel p (the Box-Muller and
    radius draws of ns_propose_kernel)."""
    a = np.asarray(a, dtype=_U64)
    z = np.empty((a.size, D))
    for m in range(0, D, 2):
        u1 = _uniform(seed, p, a, _U64(m))
        u2 = _uniform(seed, p, a, _U64(m + 1))
        r = np.sqrt(-2.0 * np.log(u1))
        ang = 6.283185307179586 * u2
        z[:, m] = r * np.cos(ang)
        if m + 1 < D:
            z[:, m + 1] = r * np.sin(ang)
    ur = _uniform(seed, p, a, _B_RADIUS)
    return z * (np.exp(np.log(ur) / D) / np.sqrt((z * z).sum(axis=1)))[:, None]


def _candidates(seed, pix, base, K, centre, axes, use_cube, with_ball=False):
    """K candidates per pixel of `pix`, uniform in the bounding ellipsoids, or in the unit cube
    where `use_cube` says the ellipsoid is the larger of the two (ns_propose_kernel);
    base[n] = how many candidates each pixel has drawn before."""
    n, D = len(pix), centre.shape[1]
    p = np.asarray(pix, dtype=_U64)[:, None, None]
    a = (np.asarray(base, dtype=_U64)[:, None] + np.arange(K, dtype=_U64)[None, :])[:, :, None]
    z = np.empty((n, K, D))
    for m in range(0, D, 2):
        u1 = _uniform(seed, p, a, _U64(m))[..., 0]
        u2 = _uniform(seed, p, a, _U64(m + 1))[..., 0]
        r = np.sqrt(-2.0 * np.log(u1))
        ang = 6.283185307179586 * u2
        z[:, :, m] = r * np.cos(ang)
        if m + 1 < D:
            z[:, :, m + 1] = r * np.sin(ang)
    ur = _uniform(seed, p, a, _B_RADIUS)[..., 0]
    f = np.exp(np.log(ur) / D) / np.sqrt((z * z).sum(axis=2))
    zf = z * f[:, :, None]
    cand = centre[:, None, :] + np.einsum('pji,pki->pkj', axes, zf)
    if use_cube.any():
        cube = _uniform(seed, p, a, np.arange(D, dtype=_U64)[None, None, :])
        cand[use_cube] = cube[use_cube]
    if with_ball:
        return cand, zf
    return cand


def _fit_ellipsoids(U, efr, ln_x, enlarge=1.0):
    """Bounding ellipsoid of live points U[P, nlive, ndim] (ns_refit): centre c and lower-triangular
    A with {c + A z : |z| <= 1}: the covariance ellipsoid scaled until it encloses every live point,
    its volume times the safety factor `enlarge`, and, MultiNest's rule (Feroz et al. 2009, sec. 5.1.1), enlarged until its volume is at least
    the expected prior volume over the target efficiency, X / efr, with ln X = ln_x[P]."""
    P, nlive, ndim = U.shape
    c = U.sum(axis=1) / nlive
    d = U - c[:, None, :]
    cov = np.einsum('pni,pnj->pij', d, d) / (nlive - 1)
    tr = np.trace(cov, axis1=1, axis2=2)
    cov = cov + (1e-12 * np.maximum(tr, 1e-30))[:, None, None] * np.eye(ndim)[None]
    L = np.linalg.cholesky(cov)
    y = np.linalg.solve(L, d.transpose(0, 2, 1))               # (P, ndim, nlive)
    r2 = np.max(np.sum(y * y, axis=1), axis=1)                 # largest Mahalanobis distance^2
    ln_vball = 0.5 * ndim * np.log(np.pi) - math.lgamma(0.5 * ndim + 1.0)
    lnv = (ln_vball + 0.5 * ndim * np.log(r2) + np.log(np.diagonal(L, axis1=1, axis2=2)).sum(axis=1)
           + math.log(enlarge))
    grow = np.maximum((np.asarray(ln_x) - np.log(efr)) - lnv, 0.0)
    scale = np.sqrt(r2) * np.exp((grow + math.log(enlarge)) / ndim)
    lnv = lnv + grow
    # ln volume against ln 1 of the unit cube: a larger ellipsoid is no better than the prior itself
    return c, L * scale[:, None, None], lnv >= 0.0, lnv


# ---- several ellipsoids per pixel (ns_refit_multi / the rejection branch of ns_propose_kernel) ----------
_NS_ME, _NS_ME_MAXD, _NS_ME_GAIN = 4, 6, 0.7        # NS_ME, NS_ME_MAXD, NS_ME_GAIN
_B_ELL, _B_KEEP = _U64(253), _U64(254)


def _me_fit(Y, enlarge):
    """Mean, Cholesky factor, covariance, largest Mahalanobis distance^2, ln volume (safety factor included) and size
    of a cluster of live points Y[n, d] (ns_me_fit)."""
    n, d = Y.shape
    c = Y.sum(axis=0) / n
    dl = Y - c
    cov = dl.T @ dl / (n - 1)
    Lc = np.linalg.cholesky(cov + 1e-12 * max(float(np.trace(cov)), 1e-30) * np.eye(d))
    y = np.linalg.solve(Lc, dl.T)
    r2 = float(np.max(np.sum(y * y, axis=0)))
    ln_vball = 0.5 * d * math.log(math.pi) - math.lgamma(0.5 * d + 1.0)
    lnv = ln_vball + 0.5 * d * math.log(r2) + float(np.log(np.diag(Lc)).sum()) + math.log(enlarge)
    return c, Lc, cov, r2, lnv, n


def _fit_multi(U, efr, ln_x, enlarge=1.0, max_ell=4):
    """The bound of one pixel's live points U[nlive, d] as up to four ellipsoids (ns_refit_multi): the cluster with the
    largest ellipsoid is cut across its principal axis at its centre; the cut stays when the halves' ellipsoids together
    have less than 0.7 of its volume, else the cluster is final.  Then MultiNest's rule on the summed volume.
    Returns centres [4, d], axes [4, d, d], ln volumes [4], the number in use, ln of the summed volume, use_cube."""
    n, d = U.shape
    minp = 2 * (d + 2)
    lab = np.zeros(n, dtype=np.int64)
    fits, final = [_me_fit(U, enlarge)], [False]
    while len(fits) < max_ell:
        best = -1
        for k, f in enumerate(fits):
            if not final[k] and f[5] >= 2 * minp and (best < 0 or f[4] > fits[best][4]):
                best = k
        if best < 0:
            break
        c, _, cov, _, lnv, _ = fits[best]
        v = np.ones(d)
        for _ in range(20):                                      # power iteration from (1, ..., 1)
            w = cov @ v
            v = w * (1.0 / math.sqrt(float((w * w).sum())))
        idx = np.flatnonzero(lab == best)
        side = ((U[idx] - c) @ v) >= 0.0
        ia, ib = idx[~side], idx[side]
        if ia.size < minp or ib.size < minp:
            final[best] = True
            continue
        fa, fb = _me_fit(U[ia], enlarge), _me_fit(U[ib], enlarge)
        if np.logaddexp(fa[4], fb[4]) < lnv + math.log(_NS_ME_GAIN):
            lab[ib] = len(fits)
            fits[best] = fa
            fits.append(fb)
            final[best] = False
            final.append(False)
        else:
            final[best] = True
    tot = -np.inf
    for f in fits:
        tot = np.logaddexp(tot, f[4])
    grow = max((ln_x - math.log(efr)) - tot, 0.0)
    cs, As, lv = np.zeros((_NS_ME, d)), np.zeros((_NS_ME, d, d)), np.full(_NS_ME, -np.inf)
    for k, (c, Lc, _, r2, lnv, _) in enumerate(fits):
        cs[k], As[k], lv[k] = c, Lc * (math.sqrt(r2) * math.exp((grow + math.log(enlarge)) / d)), lnv + grow
    return cs, As, lv, len(fits), tot + grow, (tot + grow) >= 0.0


def _candidates_multi(seed, p, base, K, cs, As, lv, ne, lnvol):
    """K candidates of pixel p, uniform over the union of its `ne` ellipsoids: one is drawn by volume, and a point that
    lies in q of them is kept with probability 1 / q.  Returns the points and the keep flags."""
    D = cs.shape[1]
    a = np.asarray(base, dtype=_U64) + np.arange(K, dtype=_U64)
    usel = _uniform(seed, _U64(p), a, _B_ELL)
    cum = np.cumsum(np.exp(lv[:ne - 1] - lnvol))
    ke = np.searchsorted(cum, usel, side='right')
    z = _ball_points(seed, _U64(p), a, D)
    cand = cs[ke] + np.einsum('kji,ki->kj', As[ke], z)
    q = np.ones(K, dtype=np.int64)
    for k in range(ne):
        y = np.linalg.solve(As[k], (cand - cs[k]).T)
        q += ((np.sum(y * y, axis=0) <= 1.0) & (ke != k)).astype(np.int64)
    keep = (q == 1) | (_uniform(seed, _U64(p), a, _B_KEEP) * q < 1.0)
    return cand, keep


# ---- free rejections: boxes around the live points in several frames (ns_refit / ns_propose_kernel) ---------------
# Above six sampled dimensions no ellipsoid bounds the live region well (a two-component fit: its ten-dimensional live set
# is box-like in some directions, curved in others), and a proposal uniform in the bounding ellipsoid is rarely inside
# the region.  But every superset of the region may veto a proposal BEFORE its likelihood is evaluated, and what is left
# is still uniform over the intersection: the axis-aligned bounding box of the live points in the unit cube, their
# bounding box in the ellipsoid's own (Cholesky) frame, and their bounding boxes in `n_frames` fixed rotations of that
# frame.  A face sits beyond the extreme live point by c max(0.1 s, extreme - mean - 1.5 s), s the standard deviation
# along that direction: a marginal that ends abruptly (a flat, box-like direction: extreme near 1.7 s) gets a margin of
# a quarter of s, one that thins out (the projection of a round body: extreme near 2.9 s) one and a half s -- what the
# spacing of the extreme order statistics would give (scripts/proto_intersection.py: the fraction of the true region a
# bound cuts off, and the evaluations per iteration it saves).
_FRAME_SEED = _U64(0x5EEDF00D)
_NS_FRAMES, _NS_MARGIN_C, _NS_MARGIN_A, _NS_MARGIN_FLOOR = 32, 2.5, 1.5, 0.1      # NS_FRAMES, NS_MARGIN_C (round 4: 1.75 = precision='speed'), NS_MARGIN_A, NS_MARGIN_FLOOR
_NS_RATIO_MAX = 32                                                             # NS_RATIO_MAX


def _frames(D, K):
    """K fixed orthogonal D x D matrices (ns_make_frames): entries 2 u - 1 from the counter-based stream, columns
    orthonormalised one after the other (modified Gram-Schmidt).  Column b of frame k is the direction of coordinate b."""
    Q = np.zeros((K, D, D))
    for k in range(K):
        M = 2.0 * _uniform(_FRAME_SEED, _U64(k + 1), np.arange(D, dtype=_U64)[:, None], np.arange(D, dtype=_U64)[None, :]) - 1.0
        for b in range(D):
            v = M[:, b].copy()
            for q in range(b):
                dot = 0.0
                for a in range(D):
                    dot += Q[k, a, q] * v[a]
                v -= dot * Q[k, :, q]
            n2 = 0.0
            for a in range(D):
                n2 += v[a] * v[a]
            Q[k, :, b] = v / math.sqrt(n2)
    return Q


def _fit_boxes(U, c, A, frames, margin_c):
    """The boxes of one pixel (ns_refit's last part): `ubox` [D, 2] around U[n, D] in the unit cube's own axes, `fbox`
    [K + 1, D, 2] around zz = A^-1 (u - c) in the Cholesky frame (k = 0) and in frame k's rotation of it."""
    n, D = U.shape
    d = U - c
    cov_diag = np.einsum('ni,ni->i', d, d) / (n - 1)
    sg = np.sqrt(cov_diag)
    lo, hi = d.min(axis=0), d.max(axis=0)
    ubox = np.stack([c + lo - margin_c * np.maximum(_NS_MARGIN_FLOOR * sg, -lo - _NS_MARGIN_A * sg),
                     c + hi + margin_c * np.maximum(_NS_MARGIN_FLOOR * sg, hi - _NS_MARGIN_A * sg)], axis=1)
    zz = np.linalg.solve(A, d.T).T                                     # [n, D]; every direction of it has the same spread:
    sz = math.sqrt(float((zz * zz).sum()) / ((n - 1) * D))              # sqrt(trace of its covariance / D)
    K = frames.shape[0]
    fbox = np.empty((K + 1, D, 2))
    for k in range(K + 1):
        W = zz if k == 0 else zz @ frames[k - 1]
        wlo, whi = W.min(axis=0), W.max(axis=0)
        fbox[k, :, 0] = wlo - margin_c * np.maximum(_NS_MARGIN_FLOOR * sz, -wlo - _NS_MARGIN_A * sz)
        fbox[k, :, 1] = whi + margin_c * np.maximum(_NS_MARGIN_FLOOR * sz, whi - _NS_MARGIN_A * sz)
    return ubox, fbox


def _box_veto(cand, zz, ubox, fbox, frames):
    """Flags of the proposals cand[K, D] (zz[K, D] = their coordinates in the Cholesky frame) that lie inside every box."""
    ok = np.all((cand >= ubox[:, 0]) & (cand <= ubox[:, 1]), axis=1)
    for k in range(fbox.shape[0]):
        W = zz if k == 0 else zz @ frames[k - 1]
        ok &= np.all((W >= fbox[k, :, 0]) & (W <= fbox[k, :, 1]), axis=1)
    return ok


# ---- a volume-preserving shear in front of the one-ellipsoid bound (ns_shear_fit / ns_shear_inv on the device) --------
_NS_KP_START = 256                                                              # NS_KP_START
_NS_K_TARGET = 16                                                               # NS_K_TARGET
_NS_SHEAR_RIDGE = 1e-6                                                          # NS_SHEAR_RIDGE
_NS_SHEAR_PIVOT = 1e-9                                                          # NS_SHEAR_PIVOT
_NS_SHEAR_ENLARGE = 3.0                                                         # NS_SHEAR_ENLARGE (round 4: 2.5 = precision='speed')


def _shear_monomials(comp):
    """The monomials (a, b) -> z_a z_b (-1 = the factor 1) of the shear of sampled dimensions with velocity components
    comp[D], ordered by their largest coordinate: [1], then per coordinate j its own z_j, z_j^2 and z_k z_j for the earlier
    coordinates k of the same component.  start[j] = monomials before coordinate j's own = the features z_j is regressed
    on; mono[start[j]] is z_j itself.  (ns_shear_monomials on the host side of the device sampler.)"""
    D = len(comp)
    mono, start = [(-1, -1)], []
    for j in range(D):
        start.append(len(mono))
        mono.append((j, -1))
        mono.append((j, j))
        for k in range(j):
            if comp[k] == comp[j]:
                mono.append((k, j))
    return np.array(mono[:start[-1] + 1], dtype=np.int32), np.array(start, dtype=np.int32)


def _shear_phi(Z, mono, n):
    """The first n monomials of rows Z[K, D]."""
    F = np.ones((Z.shape[0], n))
    for m in range(1, n):
        a, b = mono[m]
        F[:, m] = Z[:, a] if b < 0 else Z[:, a] * Z[:, b]
    return F


def _fit_shear(U, mono, start):
    """The shear of one pixel's live points U[n, D]: standardise, z = (u - mu) / sg, then regress every coordinate on
    the monomials of the earlier ones (`_shear_monomials`): w_j = z_j - phi_j(z_<j) . beta_j.  An additive triangular map
    has a unit Jacobian: a point uniform in a w-ellipsoid is uniform in u over the ellipsoid's curved image, whose volume
    is the ellipsoid's times prod sg.  One Gram matrix of all monomials and ONE Cholesky factorisation of it serve every
    coordinate: the factor of a leading block is the leading block of the factor, and row start[j] of the factor is the
    forward substitution of coordinate j's normal equations.  Returns mu[D], sg[D], beta[D, M] (row j: start[j] numbers)."""
    n, D = U.shape
    M = mono.shape[0]
    mu = U.sum(axis=0) / n
    d = U - mu
    sg = np.sqrt(np.einsum('ni,ni->i', d, d) / (n - 1))
    sg = np.maximum(sg, 1e-300)
    Z = d / sg
    F = _shear_phi(Z, mono, M)
    G = F.T @ F
    G[np.diag_indices(M)] += _NS_SHEAR_RIDGE * n
    # Cholesky, column by column; a monomial whose pivot has drowned in rounding is dropped (ns_shear_fit): pivot = its own
    # norm, nothing below it, so that its coefficient comes out as zero
    Lc = np.zeros((M, M))
    for j in range(M):
        d = G[j, j] - Lc[j, :j] @ Lc[j, :j]
        keep = d > _NS_SHEAR_PIVOT * G[j, j]
        Lc[j, j] = math.sqrt(d) if keep else math.sqrt(max(G[j, j], 1e-300))
        if keep and j + 1 < M:
            Lc[j + 1:, j] = (G[j + 1:, j] - Lc[j + 1:, :j] @ Lc[j, :j]) / Lc[j, j]
    beta = np.zeros((D, M))
    for j in range(1, D):
        pj = int(start[j])
        y = Lc[pj, :pj]
        b = np.zeros(pj)
        for r in range(pj - 1, -1, -1):                                         # back substitution with the block's transpose
            b[r] = (y[r] - Lc[r + 1:pj, r] @ b[r + 1:]) / Lc[r, r]
        beta[j, :pj] = b
    return mu, sg, beta


def _shear_fwd(X, mu, sg, beta, mono, start):
    """w of rows X[K, D] of the unit cube."""
    Z = (X - mu) / sg
    W = Z.copy()
    for j in range(1, Z.shape[1]):
        pj = int(start[j])
        W[:, j] = Z[:, j] - _shear_phi(Z, mono, pj) @ beta[j, :pj]
    return W


def _shear_inv(W, mu, sg, beta, mono, start):
    """Unit-cube rows of rows W[K, D]: coordinate by coordinate, each from the ones before it."""
    Z = W.copy()
    # (a draw from the far end of an ellipsoid whose coefficients are large can run away through the products: it ends as
    # inf or nan, fails the unit cube's test like any other point outside, and is dropped -- on the device as here)
    with np.errstate(over='ignore', invalid='ignore'):
        for j in range(1, W.shape[1]):
            pj = int(start[j])
            Z[:, j] = W[:, j] + _shear_phi(Z, mono, pj) @ beta[j, :pj]
        return mu + sg * Z


_NS_PAIRS_ENLARGE = 2.0                                                         # NS_PAIRS_ENLARGE (round 4: 1.75 = precision='speed')


def _fit_pairs(W, enlarge):
    """The pair ellipses of one pixel (ns_refit): for every pair i < j of the sheared coordinates the ellipse around the live
    points' projection onto (w_i, w_j) -- covariance ellipse scaled to enclose every point, its area times `enlarge`.  The
    region lies inside the cylinder over every one of its projections: D (D - 1) / 2 more free vetoes, five numbers each:
    rows [c_i, c_j, 1 / L00, L10, 1 / L11] with {c + L y : |y| <= 1}."""
    n, D = W.shape
    c = W.sum(axis=0) / n
    d = W - c
    cov = (d.T @ d) / (n - 1)
    out = np.empty((D * (D - 1) // 2, 5))
    e = 0
    for j in range(1, D):
        for i in range(j):
            l00 = math.sqrt(max(cov[i, i], 1e-300))
            l10 = cov[j, i] / l00
            l11 = math.sqrt(max(cov[j, j] - l10 * l10, 1e-300))
            y0 = d[:, i] / l00
            y1 = (d[:, j] - l10 * y0) / l11
            s = math.sqrt(float(np.max(y0 * y0 + y1 * y1)) * enlarge)
            out[e] = (c[i], c[j], 1.0 / (l00 * s), l10 * s, 1.0 / (l11 * s))
            e += 1
    return out


def _pair_veto(W, pairs):
    """Flags of the rows W[K, D] (sheared coordinates) inside every pair ellipse."""
    D = W.shape[1]
    ok = np.ones(W.shape[0], dtype=bool)
    e = 0
    for j in range(1, D):
        for i in range(j):
            ci, cj, r00, l10, r11 = pairs[e]
            y0 = (W[:, i] - ci) * r00
            y1 = ((W[:, j] - cj) - l10 * y0) * r11
            ok &= (y0 * y0 + y1 * y1) <= 1.0
            e += 1
    return ok


def _assemble(ndim, nlive, n_iter, n_evals, dead, Tlive, Llive, tol=None):
    """NestedResult per pixel from dead points (theta, lnL, lnw per pixel) and final live points:
    every live point carries the mass X_final / nlive.  `nlive`: one number, or one per pixel (a pixel's live
    points are then the first nlive[p] of its slice of `Tlive` / `Llive`).  With `tol` given a run whose live
    points could still add more than `tol` to lnZ (the stop test it did not meet) is marked `truncated`."""
    nl_all = np.broadcast_to(np.asarray(nlive, dtype=np.int64), (len(n_iter),))
    def log_sum_exp(x):
        """ln sum exp(x) about the largest term (a sequential logaddexp.reduce costs five times as much: this
        loop runs once per pixel of a map)."""
        if x.size == 0:
            return -np.inf
        m = x.max()
        return m if not np.isfinite(m) else m + math.log(np.exp(x - m).sum())

    results = []
    for p in range(len(n_iter)):
        nlive = int(nl_all[p])
        ln_nlive = math.log(nlive)
        dT, dL, dlnw = dead[p]
        n_dead = dL.shape[0]
        post = np.empty((n_dead + nlive, ndim + 2))
        post[:n_dead, :ndim] = dT
        post[n_dead:, :ndim] = Tlive[p][:nlive]
        L = np.empty(n_dead + nlive)
        L[:n_dead] = dL
        L[n_dead:] = Llive[p][:nlive]
        lw = np.empty(n_dead + nlive)                           # ln(prior mass x likelihood)
        np.add(dlnw, dL, out=lw[:n_dead])
        np.add(Llive[p][:nlive], -n_iter[p] / nlive - ln_nlive, out=lw[n_dead:])
        lnZ_dead = log_sum_exp(lw[:n_dead])
        lnZ_tot = np.logaddexp(lnZ_dead, log_sum_exp(lw[n_dead:]))
        wt = np.exp(lw - lnZ_tot)
        # information H = sum w (lnL - lnZ), for the error estimate sqrt(H / nlive)
        with np.errstate(invalid='ignore'):
            Hp = float(np.sum(np.where(wt > 0, wt * (L - lnZ_tot), 0.0)))
        np.multiply(L, -2.0, out=post[:, ndim])
        post[:, ndim + 1] = wt
        results.append(NestedResult(post, lnZ_tot, np.sqrt(max(Hp, 0.0) / nlive), L.max(), nlive,
                                    n_evals[p], n_iter[p], Hp))
        if tol is not None:
            remain = Llive[p][:nlive].max() - n_iter[p] / nlive
            results[-1].truncated = bool(not (np.logaddexp(lnZ_dead, remain) - lnZ_dead < tol))
    return results


def _assemble_packed(ndim, nlive, n_iter, n_evals, n_dead, off, table, tol=None, stats=None):
    """`_assemble` for tables the device has laid out (nfa_sampler_posterior_packed): rows off[p] .. off[p + 1] of `table` are
    pixel p's dead points followed by its live points, columns theta, -2 lnL, ln(prior mass x likelihood); the last column
    becomes the weight here, in place -- or has become it on the device, which then also hands over `stats` (evidence,
    information, moments: `NestedResult.from_stats`)."""
    results = []
    for p in range(len(n_iter)):
        nl, nd = int(nlive[p]), int(n_dead[p])
        post = table[off[p]:off[p + 1]]
        if stats is not None:
            r = NestedResult.from_stats(post, stats[p], nl, n_evals[p], n_iter[p])
            if tol is not None:
                remain = stats[p, 4] - n_iter[p] / nl
                r.truncated = bool(not (np.logaddexp(stats[p, 1], remain) - stats[p, 1] < tol))
            results.append(r)
            continue
        lw = post[:, ndim + 1]
        L = -0.5 * post[:, ndim]

        def lse(x):
            if x.size == 0:
                return -np.inf
            m = x.max()
            return m if not np.isfinite(m) else m + math.log(np.exp(x - m).sum())
        lnZ_dead = lse(lw[:nd])
        lnZ_tot = np.logaddexp(lnZ_dead, lse(lw[nd:]))
        wt = np.exp(lw - lnZ_tot)
        with np.errstate(invalid='ignore'):
            Hp = float(np.sum(np.where(wt > 0, wt * (L - lnZ_tot), 0.0)))
        post[:, ndim + 1] = wt
        results.append(NestedResult(post, lnZ_tot, np.sqrt(max(Hp, 0.0) / nl), L.max(), nl, n_evals[p], n_iter[p], Hp))
        if tol is not None:
            remain = L[nd:].max() - n_iter[p] / nl
            results[-1].truncated = bool(not (np.logaddexp(lnZ_dead, remain) - lnZ_dead < tol))
    return results


def _resolve_seed(seed):
    if seed is None or seed < 0:                                # like MultiNest: from the system
        return int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0] >> np.uint64(1))
    return int(seed)


def default_cap_iter(nlive):
    """Dead-point slots per pixel when the caller names none: 60 nlive iterations reach ln X = -60, far
    past where any fit of this kind has collected its evidence; a run that does hit the cap is flagged
    `truncated`.  The same default on the host twin and on the device."""
    return 60 * int(nlive)


# Named settings of a sheared one-ellipsoid bound (two and three components) and its free rejections: the safety factor on
# the ellipsoid's volume and the margins of the boxes and the pair ellipses trade evaluations for a cut of prior mass that
# shows in the evidence.  Measured on 256 pixels of the two-component test cube against bound-free rejection
# (tests/test_sampler_bias.py pins them; tests/golden/sampler_bias_reference.json, scripts/sampler_bias_reference.py;
# a pixel's own lnZ_err is 0.25; profiles/r05/sampler_bias.txt):
#   'speed'     shear 2.5, margin 1.75, pairs 1.75 (round 4's default)         +0.075 in lnZ, 408 k evaluations per pixel
#   'default'   shear 3,   margin 2.5,  pairs 2.0                              +0.031, 699 k
#   'evidence'  shear 4,   margin 3.5,  pairs 2.5, rejection only (no walks)   +0.018, 1180 k
# (the sheared ellipsoid alone at 2.5, no boxes, no pairs: +0.029 at 5.6 M -- the factor on its volume, not the margins,
# carries the last 0.03)
PRECISION = {'speed': {'shear': 2.5, 'margin': 1.75, 'pairs': 1.75}, 'default': {},
             'evidence': {'shear': 4.0, 'margin': 3.5, 'pairs': 2.5, 'method': 'reject'}}


def resolve_precision(precision, margin=None, pairs=None, method='auto', shear=None):
    """(margin, pairs, method, shear) of the named setting `precision` (None = 'default'); values given explicitly win."""
    knobs = PRECISION[precision or 'default']
    return (knobs.get('margin') if margin is None else margin, knobs.get('pairs') if pairs is None else pairs,
            knobs.get('method', method) if method == 'auto' else method, knobs.get('shear') if shear is None else shear)


def run_nested(loglike, ndim, n_pix, nlive=400, tol=0.5, efr=0.3, seed=-1, maxiter=int(1e6),
               n_cand=None, upd_frac=0.1, log_zero=LOG_ZERO, chunk=1 << 18, cap_iter=None,
               check_every=32, batch_target=262144, enlarge=1.5, method='auto', n_steps=None, free_mask=None, walk_factor=None, ellipsoids=None, walkers=None,
               progress=None, frames=None, margin=None, refit_every=4, shear=None, kmax=None, k_target=None, ratio_max=None, pairs=None,
               precision=None):
    """Nested sampling of `n_pix` independent problems in lock-step: the host twin of the
    device-resident sampler (csrc/nfa_sampler.h), same random numbers, same decisions.

    Parameters
    ----------
    loglike : callable(pix[B] int32, U[B, ndim] float64) -> lnL[B]
        Evaluates unit-cube rows against pixels; must overwrite U with the physical parameters
        (the convention of Runner.loglikelihood, core.pyx:558-561).
    nlive, tol, efr, seed, maxiter : as in ``run_multinest`` (core.pyx:727-744): live points,
        evidence tolerance, target sampling efficiency (the bounding ellipsoid is enlarged until its
        volume reaches X / efr), RNG seed (-1 = from the OS), iteration cap per pixel.
    n_cand : candidates per pixel and round (default ceil(2 / efr)), at least: every
        `check_every` rounds the number is raised so that the round's batch stays near
        max(n_pix * n_cand, batch_target) proposals however few pixels are still running (at most
        65536 per pixel); only proposals inside the unit cube are evaluated.
        They are scanned in order and every one above the pixel's current threshold replaces its
        worst live point.
    upd_frac : the ellipsoids are refitted at the end of a round once this fraction of nlive
        replacements has accumulated.
    cap_iter : dead-point slots per pixel (default: min(maxiter, 60 nlive), `default_cap_iter`); a pixel
        that fills them before meeting `tol` is returned with `truncated = True`.
    method, n_steps : 'reject' = rejection sampling in the bounding ellipsoid only; 'auto' = a pixel
        whose rejection round accepted fewer than 1 in 2 `n_steps` of the evaluated candidates
        switches to constrained random walks (64 walkers from random live points, `n_steps`
        Metropolis steps inside {L > threshold}; a step is a scaled difference of two random live
        points -- differential evolution, ter Braak 2006 -- with the scale tuned to an acceptance
        of one half); 'walk' = walks from the start.  n_steps defaults to 10 x sampled dimensions:
        scripts/sampler_bias_check.py measures the lnZ bias of walks that are too short (10
        dimensions: +0.13 with 40 steps, +0.03 with 80, +0.014 with 120; the error per run is 0.25).
    free_mask : ndim flags, 0 for unit-cube slots the likelihood does not depend on (constant or
        duplicated parameters: `PriorTransformer.free_mask`).  They are not sampled -- a uniform dummy
        dimension integrates to one -- and stay at u = 0.5: fewer dimensions for the same evidence.
    precision : 'speed' / 'default' / 'evidence': named settings of `shear`, `margin`, `pairs` and `method` (`PRECISION`).
    frames, margin : the free rejections of a one-ellipsoid bound (`_fit_boxes`): `frames` rotated frames beside the unit
        cube's axes and the ellipsoid's own (-1: no boxes; None: 32 where the bound is sheared, none elsewhere), `margin` the factor c of a face's distance beyond the extreme live point (2.5).
        With boxes the proposals per round are scaled by the last rounds' ratio of drawn to evaluated proposals (at most 8).
    shear : 0 = off; a number >= 1 = the one-ellipsoid bound is fitted to the live points AFTER a volume-preserving
        polynomial shear (`_fit_shear`: every coordinate minus a quadratic function of the earlier ones, which straightens
        the curved tex / ntot ridges), with this safety factor on the enclosing volume instead of `enlarge`; None = the
        default, 3.  Proposals are drawn in the sheared frame and mapped back; boxes, if on, live in that frame.  Only
        where all five free parameters of two or three components are sampled (10 or 15 dimensions): elsewhere ignored.
    refit_every : rejection-mode pixels refit their bound in rounds that are multiples of this (the device's engine option
        `sampler_refit_every`).
    enlarge : safety factor on the volume of the ellipsoid that just encloses the live points
        (scripts/sampler_bias_check.py: 1.0 biases lnZ by +0.020, 1.25 by +0.011, 2.0 by nothing measurable; the error is 0.18).

    Returns a list of `NestedResult`, one per pixel.
    """
    P = int(n_pix)
    # live points per pixel: one number for everybody, or one per pixel (arrays are then laid out for the largest and a
    # pixel uses the first nl[p] slots, like the device sampler after nfa_sampler_set_pixel_nlive)
    nl = np.broadcast_to(np.asarray(nlive, dtype=np.int64), (P,)).copy()
    nlive = int(nl.max())
    assert ndim > 0 and nl.min() > ndim + 1 and tol > 0 and 0 < efr <= 1 and maxiter >= 0
    margin, pairs, method, shear = resolve_precision(precision, margin, pairs, method, shear)
    seed = _resolve_seed(seed)
    K = int(n_cand) if n_cand else int(np.ceil(2.0 / efr))
    if cap_iter is None:
        capp = np.array([min(maxiter, default_cap_iter(int(n))) for n in nl], dtype=np.int64)
    else:
        capp = np.full(P, min(cap_iter, maxiter) if maxiter > 0 else cap_iter, dtype=np.int64)
    all_pix = np.arange(P, dtype=np.int32)

    def evaluate(pix, U):
        out = np.empty(U.shape[0])
        for a in range(0, U.shape[0], chunk):
            out[a:a + chunk] = loglike(pix[a:a + chunk], U[a:a + chunk])
        return np.where(np.isfinite(out), out, log_zero)

    fmap = np.arange(ndim) if free_mask is None else np.flatnonzero(np.asarray(free_mask))
    assert fmap.size > 0 and fmap.max() < ndim
    nd = int(fmap.size)                                         # sampled dimensions

    def expand(U):
        """Rows of the full unit cube from rows of the sampled dimensions."""
        T = np.full(U.shape[:-1] + (ndim,), 0.5)
        T[..., fmap] = U
        return T

    # live points: unit-cube positions, physical parameters, log-likelihoods
    Ulive = _uniform(seed, all_pix[:, None, None], _TAG_LIVE + np.arange(nlive, dtype=_U64)[None, :, None],
                     np.arange(nd, dtype=_U64)[None, None, :])
    Tlive = expand(Ulive).reshape(-1, ndim)
    Llive = evaluate(np.repeat(all_pix, nlive), Tlive).reshape(P, nlive)
    Tlive = Tlive.reshape(P, nlive, ndim)
    n_evals = nl.copy()
    n_iter = np.zeros(P, dtype=np.int64)
    lnZ = np.full(P, -np.inf)
    ln_shrink = np.log1p(-np.exp(-1.0 / nl))                    # ln(X_i - X_{i+1}) - ln X_i, per pixel
    active = np.full(P, maxiter > 0)
    since_fit = np.zeros(P, dtype=np.int64)
    updp = np.maximum(1, (upd_frac * nl).astype(np.int64))
    # the bound: several ellipsoids per pixel where few dimensions are sampled (and the live points fit in the device's
    # LDS), one otherwise; [pixel][ellipsoid]
    max_ell = _NS_ME if not ellipsoids else int(ellipsoids)      # `ellipsoids`: None / 0 = the default, 1 = one, up to 4
    multi = nd <= _NS_ME_MAXD and nlive * nd * 8 <= 96 * 1024 and max_ell > 1
    centre, axes = np.zeros((P, _NS_ME, nd)), np.zeros((P, _NS_ME, nd, nd))
    elnv, nell = np.full((P, _NS_ME), -np.inf), np.ones(P, dtype=np.int64)
    use_cube, lnvol = np.empty(P, dtype=bool), np.empty(P)
    # free rejections (one-ellipsoid bounds only): boxes in the unit cube's axes, the ellipsoid's frame and n_frames rotations
    # (the device's shapes for the shear: all five free parameters of two or three components, slot % ncomp = dimension % ncomp;
    # there it is on by default, with NS_FRAMES box frames, like the device's -- shear=0 / frames=-1 turn them off)
    ncomp_s = max(1, nd // 5)
    shear = _NS_SHEAR_ENLARGE if shear is None else shear
    shear_on = (bool(shear) and (not multi) and nd in (10, 15) and ndim == 6 * ncomp_s and nlive * nd * 8 <= 96 * 1024
                and bool(np.all(fmap % ncomp_s == np.arange(nd) % ncomp_s)))
    n_frames = (_NS_FRAMES if shear_on else -1) if frames is None else int(frames)
    boxes = (not multi) and n_frames >= 0 and nlive * nd * 8 <= 96 * 1024
    margin_c = _NS_MARGIN_C if margin is None else float(margin)
    Qf = _frames(nd, max(n_frames, 0)) if boxes else None
    ubox = np.zeros((P, nd, 2))
    fbox = np.zeros((P, max(n_frames, 0) + 1, nd, 2))
    # the pair ellipses (`_fit_pairs`): with the shear and the boxes, unless pairs=0; `pairs` = the safety factor on their areas
    pairs_enl = _NS_PAIRS_ENLARGE if pairs is None else float(pairs)
    pairs_on = shear_on and boxes and pairs_enl >= 1.0
    pair_tab = np.zeros((P, nd * (nd - 1) // 2, 5)) if pairs_on else None
    if shear_on:
        assert float(shear) >= 1.0
        mono, mstart = _shear_monomials(fmap % ncomp_s)
        sh_mu, sh_sg, sh_beta = np.zeros((P, nd)), np.ones((P, nd)), np.zeros((P, nd, mono.shape[0]))

    def refit(p, ln_x):
        n = int(nl[p])
        if multi:
            centre[p], axes[p], elnv[p], nell[p], lnvol[p], use_cube[p] = _fit_multi(Ulive[p, :n], efr, ln_x, enlarge, max_ell)
        elif shear_on:
            sh_mu[p], sh_sg[p], sh_beta[p] = _fit_shear(Ulive[p, :n], mono, mstart)
            Wl = _shear_fwd(Ulive[p, :n], sh_mu[p], sh_sg[p], sh_beta[p], mono, mstart)
            ln_jac = float(np.log(sh_sg[p]).sum())                  # ln |du / dw|: volumes in w units are smaller by this
            c1, a1, _, v1 = _fit_ellipsoids(Wl[None], efr, np.array([ln_x - ln_jac]), float(shear))
            centre[p, 0], axes[p, 0], lnvol[p], nell[p] = c1[0], a1[0], v1[0] + ln_jac, 1
            use_cube[p], elnv[p, 0] = lnvol[p] >= 0.0, lnvol[p]
            if boxes:
                ubox[p], fbox[p] = _fit_boxes(Wl, c1[0], a1[0], Qf, margin_c)     # `ubox`: the box in the w axes
            if pairs_on:
                pair_tab[p] = _fit_pairs(Wl, pairs_enl)
        else:
            c1, a1, u1, v1 = _fit_ellipsoids(Ulive[p:p + 1, :n], efr, np.array([ln_x]), enlarge)
            centre[p, 0], axes[p, 0], use_cube[p], lnvol[p], elnv[p, 0], nell[p] = c1[0], a1[0], u1[0], v1[0], v1[0], 1
            if boxes:
                ubox[p], fbox[p] = _fit_boxes(Ulive[p, :n], c1[0], a1[0], Qf, margin_c)

    for p in range(P):
        refit(p, 0.0)
    dead_T, dead_L, dead_lnw, dead_pix = [], [], [], []
    rnd = 0
    b_target = max(P * K, int(batch_target))
    cand_base = np.zeros(P, dtype=np.int64)
    Kr = K
    method = {'reject': 0, 'auto': 1, 'walk': 2}[method] if isinstance(method, str) else int(method)
    n_steps = int(n_steps) if n_steps else 10 * nd
    # a pixel turns to walks when a rejection round accepts fewer than 1 in walk_factor * n_steps candidates (and back
    # above eight times that): 2 from seven sampled dimensions on, 64 below -- there a rejection round, one large batch,
    # beats a walk cycle of n_steps small ones down to very low acceptances (csrc/nfa_sampler.h: NS_WALK_FACTOR*,
    # profiles/r03/sweep_walk_factor.txt).  The device takes the same default (engine option `sampler_walk_factor`).
    walk_factor = int(walk_factor) if walk_factor else (_WALK_FACTOR_LOWD if nd <= _WALK_LOWD else _WALK_FACTOR)
    # constrained random walks (ns_update_kernel's walk branch): state per pixel and per walker
    walk = np.zeros(P, dtype=bool)
    wstep = np.zeros(P, dtype=np.int64)
    wW = np.zeros(P, dtype=np.int64)
    wscale = np.ones(P)
    wLthr = np.zeros(P)
    wacc_sum = np.zeros(P, dtype=np.int64)
    wtot_sum = np.zeros(P, dtype=np.int64)
    w_stride = int(walkers) if walkers else _walkers_for(nlive)     # walker slots per pixel (`walkers`: A/B knob, else by the live points)
    wU = np.zeros((P, w_stride, nd))
    wT = np.zeros((P, w_stride, ndim))
    wL = np.zeros((P, w_stride))
    wnacc = np.zeros((P, w_stride), dtype=np.int64)
    # what a pixel's rejection rounds did since the last decision point (every n_steps rounds): candidates scanned and
    # accepted, proposals drawn and evaluated; ln of the last window's evaluated / drawn (the boxes' share of the
    # ellipsoid: what the way back from the walks counts the bound's volume with)
    # a pixel's own share of a rejection round's proposals (ns_kp / NS_K_TARGET): halved after a round with more than twice
    # k_target replacements, doubled after one with fewer than half of it; 0 = the round's Kr
    k_target = _NS_K_TARGET if k_target is None else int(k_target)
    Kp = np.full(P, _NS_KP_START, dtype=np.int64)                # (a small share first, doubled while little is accepted)
    rj_scan, rj_acc = np.zeros(P, dtype=np.int64), np.zeros(P, dtype=np.int64)
    rj_raw, rj_val = np.zeros(P, dtype=np.int64), np.zeros(P, dtype=np.int64)
    ln_pass = np.zeros(P)

    def replace(p, cU, cT, Lk):
        """The worst live point of pixel p dies, the candidate takes its slot; True when p is done."""
        nlive, cap = int(nl[p]), int(capp[p])
        w = int(np.argmin(Llive[p, :nlive]))
        Lmin = Llive[p, w]
        lnw = -n_iter[p] / nlive + ln_shrink[p]
        lnZ[p] = np.logaddexp(lnZ[p], lnw + Lmin)
        if n_iter[p] < cap:
            dead_T.append(Tlive[p, w][None].copy()); dead_L.append(np.array([Lmin]))
            dead_lnw.append(np.array([lnw])); dead_pix.append(np.array([p]))
        Ulive[p, w], Tlive[p, w], Llive[p, w] = cU, cT, Lk
        n_iter[p] += 1
        since_fit[p] += 1
        remain = Llive[p, :nlive].max() - n_iter[p] / nlive
        return bool((np.logaddexp(lnZ[p], remain) - lnZ[p] < tol) or n_iter[p] >= maxiter or n_iter[p] >= cap)

    nlive_max = nlive
    raw_sum = val_sum = 0                                       # proposals drawn / evaluated since the last look
    while active.any():
        if rnd % check_every == 0:                              # the device compacts its pixel list here
            # with boxes most proposals are vetoed for free: draw so many more that a round still evaluates ~b_target
            ratio = min(int(ratio_max) if ratio_max else _NS_RATIO_MAX, max(1, (raw_sum + val_sum // 2) // max(val_sum, 1))) if boxes and raw_sum else 1
            n_chunk = int(active.sum())                          # the pixels the device's list holds until the next look
            Kr = int(min(kmax if kmax else 65536, max(K, (b_target * ratio) // n_chunk)))
            raw_sum = val_sum = 0
        raw_sum += Kr * n_chunk
        idx = np.flatnonzero(active)
        for p in idx:                                           # one wave per pixel on the device
            p = int(p)
            nlive = int(nl[p])
            done = False
            was_walking = bool(walk[p])
            k_used = Kr                                         # what the pixel's random stream advances by in this round
            if walk[p]:
                # one Metropolis step of every walker inside {L > threshold frozen at the cycle start}
                step = int(wstep[p])
                W = min(w_stride, int(walkers) if walkers else _walkers_for(nlive), Kr) if step == 0 else int(wW[p])
                a = _U64(cand_base[p]) + np.arange(W, dtype=_U64)
                if step == 0:
                    start = np.minimum(nlive - 1, (_uniform(seed, p, a, _B_START) * nlive).astype(np.int64))
                    wU[p, :W], wT[p, :W], wL[p, :W] = Ulive[p, start], Tlive[p, start], Llive[p, start]
                    wnacc[p, :W] = 0
                    wLthr[p] = Llive[p, :nlive].min()
                    wW[p] = W
                # differential-evolution move: a scaled difference of two random live points
                ia = np.minimum(nlive - 1, (_uniform(seed, p, a, _U64(251)) * nlive).astype(np.int64))
                ib = np.minimum(nlive - 2, (_uniform(seed, p, a, _U64(252)) * (nlive - 1)).astype(np.int64))
                ib = ib + (ib >= ia)
                gam = wscale[p] * 2.38 / math.sqrt(2.0 * nd)
                cand = wU[p, :W] + gam * (Ulive[p, ia] - Ulive[p, ib])
                valid = np.all((cand >= 0.0) & (cand < 1.0), axis=1)
                vi = np.flatnonzero(valid)
                val_sum += int(vi.size)
                if vi.size:
                    Tsub = expand(cand[vi])
                    Lsub = evaluate(np.full(vi.size, p, dtype=np.int32), Tsub)
                    n_evals[p] += vi.size
                    ok = Lsub > wLthr[p]
                    wU[p, vi[ok]], wT[p, vi[ok]], wL[p, vi[ok]] = cand[vi[ok]], Tsub[ok], Lsub[ok]
                    wnacc[p, vi[ok]] += 1
                    wacc_sum[p] += int(ok.sum())
                    wtot_sum[p] += vi.size
                wstep[p] = step + 1
                if wstep[p] >= n_steps:                         # cycle end: the walkers are the candidates
                    for k in range(W):
                        if done:
                            break
                        if wnacc[p, k] == 0 or not (wL[p, k] > Llive[p, :nlive].min()):
                            continue
                        done = replace(p, wU[p, k].copy(), wT[p, k].copy(), wL[p, k])
                    if wtot_sum[p] > 0:                         # acceptance near one half
                        wscale[p] = min(1.0, wscale[p] * math.exp((wacc_sum[p] / wtot_sum[p] - _WALK_TARGET)
                                                                  / (0.5 * math.sqrt(nd))))
                    wacc_sum[p] = wtot_sum[p] = wstep[p] = 0
                    # back to rejection once the bound promises clearly more than a walk delivers
                    if method == 1 and (-n_iter[p] / nlive - min(lnvol[p] + ln_pass[p], 0.0)) > math.log(8.0 / (walk_factor * n_steps)):
                        walk[p] = False
            else:
                k_used = int(min(Kp[p], Kr)) if (k_target > 0 and Kp[p] > 0) else Kr
                if nell[p] > 1 and not use_cube[p]:
                    cand, keep = _candidates_multi(seed, p, cand_base[p], k_used, centre[p], axes[p], elnv[p], int(nell[p]), lnvol[p])
                else:
                    cand, zf = _candidates(seed, [p], cand_base[p:p + 1], k_used, centre[p:p + 1, 0], axes[p:p + 1, 0],
                                           use_cube[p:p + 1], with_ball=True)
                    cand, keep = cand[0], True
                    if shear_on:
                        # the ellipsoid lives in the sheared frame: its draws are w, the unit cube's draws are u
                        if use_cube[p]:
                            wc = _shear_fwd(cand, sh_mu[p], sh_sg[p], sh_beta[p], mono, mstart)
                        else:
                            wc, cand = cand, _shear_inv(cand, sh_mu[p], sh_sg[p], sh_beta[p], mono, mstart)
                        if boxes:
                            zz = np.linalg.solve(axes[p, 0], (wc - centre[p, 0]).T).T if use_cube[p] else zf[0]
                            keep = _box_veto(wc, zz, ubox[p], fbox[p], Qf)
                            if pairs_on:
                                keep &= _pair_veto(wc, pair_tab[p])
                    elif boxes:
                        # the proposal's coordinates in the ellipsoid's frame: the unit-ball point it was made from, or
                        # (drawn from the unit cube) A^-1 (u - c)
                        zz = np.linalg.solve(axes[p, 0], (cand - centre[p, 0]).T).T if use_cube[p] else zf[0]
                        keep = _box_veto(cand, zz, ubox[p], fbox[p], Qf)
                valid = np.all((cand >= 0.0) & (cand < 1.0), axis=1) & keep   # outside the unit cube = outside the prior
                vi = np.flatnonzero(valid)
                scanned = accepted = 0
                val_sum += int(vi.size)
                if vi.size:
                    Tsub = expand(cand[vi])
                    Lsub = evaluate(np.full(vi.size, p, dtype=np.int32), Tsub)
                    for j in range(vi.size):                    # the wave's sequential scan
                        scanned += 1
                        n_evals[p] += 1
                        if Lsub[j] > Llive[p, :nlive].min():
                            accepted += 1
                            done = replace(p, cand[vi[j]].copy(), Tsub[j].copy(), Lsub[j])
                            if done:
                                break
                # walk cycles of all pixels stay in phase: they start at rounds that are multiples of n_steps.  The decision
                # looks at all rejection rounds since the last one (a single round of a few hundred candidates is noise)
                rj_scan[p] += scanned; rj_acc[p] += accepted; rj_raw[p] += k_used; rj_val[p] += int(vi.size)
                if k_target > 0:
                    if accepted > 2 * k_target:
                        Kp[p] = max(k_used // 2, K)
                    elif 2 * accepted < k_target:
                        Kp[p] = min(k_used * 2, 1 << 20)
                    else:
                        Kp[p] = k_used
                if (rnd + 1) % n_steps == 0:
                    if not done and (method == 2 or (method == 1 and (walk_factor * rj_acc[p] * n_steps < rj_scan[p] if rj_scan[p] >= 64 else rj_raw[p] >= 4096))):
                        walk[p], wstep[p], wscale[p], wacc_sum[p], wtot_sum[p] = True, 0, 1.0, 0, 0
                        ln_pass[p] = math.log(max(int(rj_val[p]), 1) / max(int(rj_raw[p]), 1)) if boxes else 0.0
                    rj_scan[p] = rj_acc[p] = rj_raw[p] = rj_val[p] = 0
            cand_base[p] += k_used
            if done:
                active[p] = False
            elif since_fit[p] >= updp[p] and (was_walking or (rnd + 1) % refit_every == 0):
                # (rejection-mode pixels refit only in every fourth round: on the device a refit makes the
                # whole launch wait, so they are taken together)
                refit(p, -n_iter[p] / nlive)
                since_fit[p] = 0
        rnd += 1
        if progress is not None:
            progress(int(active.sum()), int(n_iter.max()))
            if hasattr(progress, 'detail'):                     # (debugging aid: the round's state)
                progress.detail(dict(rnd=rnd, n_iter=n_iter, n_evals=n_evals, walk=walk, use_cube=use_cube, lnvol=lnvol, Kr=Kr,
                                     rj=(rj_scan, rj_acc, rj_raw, rj_val), ln_pass=ln_pass, Llive=Llive, Ulive=Ulive))

    dead_pix = np.concatenate(dead_pix) if dead_pix else np.zeros(0, dtype=np.int64)
    dead_T = np.concatenate(dead_T) if dead_T else np.zeros((0, ndim))
    dead_L = np.concatenate(dead_L) if dead_L else np.zeros(0)
    dead_lnw = np.concatenate(dead_lnw) if dead_lnw else np.zeros(0)
    order = np.argsort(dead_pix, kind='stable')
    bounds = np.searchsorted(dead_pix[order], np.arange(P + 1))
    dead = [(dead_T[order[bounds[p]:bounds[p + 1]]], dead_L[order[bounds[p]:bounds[p + 1]]],
             dead_lnw[order[bounds[p]:bounds[p + 1]]]) for p in range(P)]
    res = _assemble(ndim, nl, n_iter, n_evals, dead, Tlive, Llive, tol)
    for r in res:
        r.rounds = rnd
    return res


def run_nested_device(runner, pix, nlive=400, tol=0.5, efr=0.3, seed=-1, maxiter=int(1e6), n_cand=None,
                      upd_frac=0.1, log_zero=LOG_ZERO, cap_iter=None, check_every=32, batch_target=262144,
                      enlarge=1.5, method='auto', n_steps=None, free_mask=None, progress=None, time_limit=None,
                      ellipsoids=None, frames=None, margin=None, shear=None, pairs=None, precision=None):
    """The same algorithm with its whole state on the GPU (``nfa_sampler_*``): pixels `pix` of a
    `CubeRunner` (or pixel 0 of a single-pixel runner) in lock-step rounds, no per-round host
    work.  Options as `run_nested`; `cap_iter` defaults to min(maxiter, 60 nlive).  `progress`
    (callable(n_active, rounds)) is called about once a second; after `time_limit` seconds the
    pixels still running are stopped where they are (their results are then lower bounds)."""
    import time
    import ctypes as C
    from . import _ffi
    lib = _ffi.engine()
    pix = np.ascontiguousarray(pix, dtype=np.int32)
    P, ndim = int(pix.size), int(runner.ndim)
    # live points: one number, or one per pixel (nfa_sampler_set_pixel_nlive: one lock-step group all the same)
    nl = np.broadcast_to(np.asarray(nlive, dtype=np.int64), (P,)).copy()
    per_pixel = bool((nl != nl[0]).any())
    nlive = int(nl.max())
    assert nl.min() > ndim + 1 and tol > 0 and 0 < efr <= 1 and maxiter >= 0
    margin, pairs, method, shear = resolve_precision(precision, margin, pairs, method, shear)
    seed = _resolve_seed(seed)
    K = int(n_cand) if n_cand else int(np.ceil(2.0 / efr))
    capp = np.array([int(cap_iter) if cap_iter else int(max(1, min(maxiter, default_cap_iter(int(n))))) for n in nl], dtype=np.int64)
    cap = int(capp.max())
    fm = None if free_mask is None else np.ascontiguousarray(free_mask, dtype=np.int32)
    assert fm is None or fm.shape == (ndim,)
    nd = ndim if fm is None else int(np.count_nonzero(fm))
    h = C.c_void_p()
    _ffi.check(lib.nfa_sampler_create(C.byref(h), runner._run.handle, pix.ctypes.data_as(_ffi._ip), P,
                                      int(nlive), K, int(batch_target), cap,
                                      None if fm is None else fm.ctypes.data_as(_ffi._ip)))
    try:
        if ellipsoids:
            _ffi.check(lib.nfa_sampler_set_ellipsoids(h, int(ellipsoids)))
        if frames is not None or margin is not None:
            _ffi.check(lib.nfa_sampler_set_boxes(h, -2 if frames is None else int(frames), 0.0 if margin is None else float(margin)))
        if shear is not None:
            _ffi.check(lib.nfa_sampler_set_shear(h, float(shear)))
        if pairs is not None:
            _ffi.check(lib.nfa_sampler_set_pairs(h, float(pairs)))
        if per_pixel:
            nl32 = nl.astype(np.int32)
            upd32 = np.maximum(1, (upd_frac * nl).astype(np.int64)).astype(np.int32)
            _ffi.check(lib.nfa_sampler_set_pixel_nlive(h, nl32.ctypes.data_as(_ffi._ip), capp.ctypes.data_as(_ffi._lp),
                                                       upd32.ctypes.data_as(_ffi._ip)))
        _ffi.check(lib.nfa_sampler_begin(h, float(tol), float(efr), seed, int(maxiter),
                                         max(1, int(upd_frac * nlive)), float(log_zero), int(check_every),
                                         float(enlarge),
                                         {'reject': 0, 'auto': 1, 'walk': 2}[method] if isinstance(method, str)
                                         else int(method), int(n_steps) if n_steps else 10 * nd))
        n_active = C.c_int64(P)
        t0 = time.perf_counter()
        t_created = t0
        chunks = 16
        while True:
            t1 = time.perf_counter()
            _ffi.check(lib.nfa_sampler_advance(h, chunks, C.byref(n_active)))
            if n_active.value == 0:
                break
            dt = time.perf_counter() - t1
            chunks = int(min(4096, max(1, chunks * (1.0 / max(dt, 1e-3)))))     # about a second per call
            if progress is not None:
                progress(int(n_active.value), None)
                if hasattr(progress, 'counts'):                                  # (debugging aid: the counters, chunk by chunk)
                    ni, ne, rr = np.empty(P, dtype=np.int64), np.empty(P, dtype=np.int64), C.c_int64()
                    _ffi.check(lib.nfa_sampler_counts(h, ni.ctypes.data_as(_ffi._lp), ne.ctypes.data_as(_ffi._lp), C.byref(rr)))
                    chunks = progress.counts(ni, ne, int(rr.value)) or chunks
            if time_limit is not None and time.perf_counter() - t0 > time_limit:
                break
        t_rounds = time.perf_counter()
        n_iter = np.empty(P, dtype=np.int64)
        n_evals = np.empty(P, dtype=np.int64)
        rounds = C.c_int64()
        _ffi.check(lib.nfa_sampler_counts(h, n_iter.ctypes.data_as(_ffi._lp), n_evals.ctypes.data_as(_ffi._lp),
                                          C.byref(rounds)))
        # every pixel's table of posterior samples, laid out by the device (dead points, then live points; theta, -2 lnL,
        # ln(prior mass x likelihood)): one copy, and a pixel's table is a view into it
        n_dead = np.minimum(n_iter, capp)
        off = np.zeros(P + 1, dtype=np.int64)
        np.cumsum(n_dead + nl, out=off[1:])
        live_off = -n_iter / nl - np.log(nl)
        table = np.empty((int(off[-1]), ndim + 2))
        stats = np.empty((P, 6 + 4 * ndim))
        _ffi.check(lib.nfa_sampler_posterior_packed(h, off.ctypes.data_as(_ffi._lp), _ffi.dptr(live_off), _ffi.dptr(table), _ffi.dptr(stats)))
    finally:
        lib.nfa_sampler_destroy(h)
    t_read = time.perf_counter()
    res = _assemble_packed(ndim, nl, n_iter, n_evals, n_dead, off, table, tol, stats)
    for r in res:
        r.rounds = int(rounds.value)
    # where the call's time went (seconds): the rounds on the device, the read-back of live and dead points, the assembly of
    # the results on the host
    res[0].timings = {'rounds': t_rounds - t_created, 'read_back': t_read - t_rounds, 'assemble': time.perf_counter() - t_read}
    return res


# ---------------------------------------------------------------------------
#  reference-shaped front end: Dumper + run_multinest (core.pyx:564-823)
# ---------------------------------------------------------------------------
class MemoryGroup:
    """Minimal stand-in for an ``h5py.Group`` (attrs + datasets) when h5py is absent."""

    class _File:
        def flush(self):
            pass

    def __init__(self):
        self.attrs = {}
        self.datasets = {}
        self.file = MemoryGroup._File()

    def create_dataset(self, name, data=None):
        self.datasets[name] = np.array(data)
        return self.datasets[name]

    def __getitem__(self, name):
        return self.datasets[name]


# What a finished run leaves in its store group (docs/store_spec.rst:77-96; the reference's writer is
# mn_dump, core.pyx:627-687).  Every entry: name in the store <- function of (runner, result, dumper).
def information_criteria(n_chan, n_par, lnL):
    """(BIC, AIC, AICc) of a model with `n_par` parameters on `n_chan` channels at log-likelihood lnL."""
    n, k = float(n_chan), float(n_par)
    aic = 2.0 * k - 2.0 * lnL
    return np.log(n) * k - 2.0 * lnL, aic, aic + (2.0 * k * k + 2.0 * k) / (n - k - 1.0)


def _criteria(prefix, which_lnL):
    names = (prefix + 'BIC', prefix + 'AIC', prefix + 'AICc')

    def make(k):
        return lambda run, res, dmp: information_criteria(run.n_chan_tot, run.n_params, which_lnL(run, res))[k]
    return tuple((name, make(k)) for k, name in enumerate(names))


RUN_ATTRIBUTES = (
    ('ncomp', lambda run, res, dmp: run.ncomp),
    ('null_lnZ', lambda run, res, dmp: run.null_lnZ),
    ('n_chan_tot', lambda run, res, dmp: run.n_chan_tot),
    ('n_samples', lambda run, res, dmp: res.n_samples),
    ('n_live', lambda run, res, dmp: res.n_live),
    ('n_params', lambda run, res, dmp: res.n_params),
    ('global_lnZ', lambda run, res, dmp: res.lnZ),
    ('global_lnZ_err', lambda run, res, dmp: res.lnZ_err),
    ('max_loglike', lambda run, res, dmp: res.max_loglike),
    ('marg_cols', lambda run, res, dmp: dmp.marginal_cols),
    ('marg_quantiles', lambda run, res, dmp: dmp.quantiles),
) + _criteria('', lambda run, res: res.max_loglike) + _criteria('null_', lambda run, res: run.null_lnZ) + (
    ('truncated', lambda run, res, dmp: bool(getattr(res, 'truncated', False))),     # not in the reference: see NestedResult
)
RUN_DATASETS = (
    ('posteriors', lambda run, res, dmp: res.posterior.astype('float32')),           # (n_samples, n_params + 2)
    ('marginals', lambda run, res, dmp: dmp.calc_marginals(res.posterior)),          # (n_quantiles, n_params)
    ('bestfit_params', lambda run, res, dmp: res.param_constr[2]),
    ('map_params', lambda run, res, dmp: res.param_constr[3]),
)


def marginal_quantile_table():
    """(column names, quantiles) of the `marginals` dataset: extremes and percentiles, then the 1 / 2 / 3
    sigma credible intervals.  The reference stores the normal tail areas truncated to nine significant
    digits below and eight decimals above the median (its marg_quantiles attribute): reproduced here from
    the normal distribution, not typed in."""
    from scipy.stats import norm
    percent = (1, 10, 25, 50, 75, 90, 99)
    names = ['min'] + [f'p{q:02d}' for q in percent] + ['max']
    quant = [0.0] + [q / 100 for q in percent] + [1.0]
    for k in (1, 2, 3):
        names += [f'{k}s_lo', f'{k}s_hi']
        quant += [float(f'{norm.cdf(-k):.8e}'), float(f'{norm.cdf(k):.8f}')]
    return names, np.array(quant)


class Dumper:
    """Writer of one run's outputs into a store group (an h5py group, a `store.Group` or a
    `MemoryGroup`), with the reference's entry points (core.pyx:564-612)."""

    def __init__(self, group, no_dump=False):
        self.group, self.no_dump = group, no_dump
        self.n_calls, self.n_samples = 0, -1
        self.marginal_cols, self.quantiles = marginal_quantile_table()

    def calc_marginals(self, posteriors):
        """Quantiles of every parameter column (the two trailing columns are -2 lnL and the weights):
        `np.quantile(columns, quantiles, axis=0)` of the reference's Dumper (core.pyx:596-598), bit for bit, from
        one sort of the columns and numpy's own interpolation rule (a + (b - a) t, from the upper side for
        t >= 1/2) -- a map has tens of thousands of runs and the general routine costs six times as much per run."""
        n_par = posteriors.shape[1] - 2
        a = np.asarray(posteriors[:, :n_par], dtype=np.float64)
        n = a.shape[0]
        if n == 0 or np.isnan(a).any():
            return np.quantile(a, self.quantiles, axis=0)
        q = np.asarray(self.quantiles, dtype=np.float64)
        s = np.sort(a, axis=0)
        virtual = (n - 1) * q
        below = np.floor(virtual).astype(np.intp)
        above = np.minimum(below + 1, n - 1)
        t = (virtual - below)[:, None]
        lo, hi = s[below], s[above]
        step = hi - lo
        out = lo + step * t
        upper = (t >= 0.5)[:, 0]
        out[upper] = hi[upper] - step[upper] * (1 - t[upper])
        return out

    def flush(self):
        self.group.file.flush()

    def append_attributes(self, **attributes):
        self.group.attrs.update(attributes)

    def append_datasets(self, **datasets):
        for name in datasets:
            self.group.create_dataset(name, data=datasets[name])

    def dump(self, runner, res):
        """The final call of the sampler's dump callback: evidence into the runner, everything of
        RUN_ATTRIBUTES / RUN_DATASETS into the group."""
        self.n_calls += 1
        self.n_samples = res.n_samples
        runner.run_lnZ = res.lnZ
        if self.no_dump:
            return
        self.append_attributes(**{name: get(runner, res, self) for name, get in RUN_ATTRIBUTES})
        self.append_datasets(**{name: get(runner, res, self) for name, get in RUN_DATASETS})


def run_multinest(runner, dumper, IS=False, mmodal=True, ceff=False, nlive=400, tol=0.5, efr=0.3,
                  nClsPar=None, maxModes=100, updInt=10, Ztol=-1e90, root='results', seed=-1,
                  pWrap=None, fb=False, resume=False, initMPI=False, outfile=False, logZero=-1e100,
                  maxiter=int(1e6), precision=None):
    """Signature of the reference's ``run_multinest`` (core.pyx:727-823) on the built-in sampler,
    for one runner (one pixel).  `mmodal` / `maxModes`: clusters of live points get bounding ellipsoids of
    their own (at most min(maxModes, 4), and only where at most six dimensions are sampled; mmodal = False: one
    ellipsoid); the evidence is the global one either way (the reference's dumper stores no per-mode values).
    Options that concern importance sampling, constant efficiency, the clustering parameters or MultiNest's output
    files are accepted and ignored; the argument checks are the reference's.  `precision` (not MultiNest's): the
    built-in sampler's named setting, `PRECISION`."""
    assert runner.ndim > 0
    assert nlive > 0
    assert tol > 0
    assert 0 < efr <= 1
    assert maxModes > 0
    assert updInt > 0
    assert Ztol is not None and np.isfinite(Ztol)
    assert logZero is not None and np.isfinite(logZero)
    assert maxiter >= 0
    if nClsPar is None:
        nClsPar = runner.n_params
    if nClsPar > runner.n_params:
        raise ValueError('Number of clustering parameters must be less than total.')

    utrans = getattr(runner, 'utrans', None)
    free_mask = utrans.free_mask(runner.ncomp) if hasattr(utrans, 'free_mask') else None
    ellipsoids = min(int(maxModes), _NS_ME) if mmodal else 1
    if hasattr(runner, '_run'):          # engine runner: the whole run stays on the device
        res = run_nested_device(runner, np.zeros(1, dtype=np.int32), nlive=nlive, tol=tol, efr=efr, seed=seed,
                                maxiter=maxiter, log_zero=logZero, free_mask=free_mask, ellipsoids=ellipsoids, precision=precision)[0]
    else:                                # any object with loglikelihood_batch(U): the numpy twin
        def loglike(pix, U):
            return runner.loglikelihood_batch(U)

        res = run_nested(loglike, runner.ndim, 1, nlive=nlive, tol=tol, efr=efr, seed=seed,
                         maxiter=maxiter, log_zero=logZero, free_mask=free_mask, ellipsoids=ellipsoids, precision=precision)[0]
    dumper.dump(runner, res)
    return res


def fit_pixels(cube_runner, pix, nlive=400, tol=0.5, efr=0.3, seed=-1, maxiter=int(1e6), device=True,
               **kwargs):
    """All pixels `pix` of a `CubeRunner` in one lock-step run; returns a list of NestedResult.
    device=True keeps the sampler state on the GPU (`run_nested_device`); device=False runs the
    host twin and sends only the likelihood batches to the GPU."""
    pix = np.ascontiguousarray(pix, dtype=np.int32)
    if 'free_mask' not in kwargs and getattr(cube_runner, 'utrans', None) is not None:
        kwargs['free_mask'] = cube_runner.utrans.free_mask(cube_runner.ncomp)   # dummies are not sampled
    if device:
        return run_nested_device(cube_runner, pix, nlive=nlive, tol=tol, efr=efr, seed=seed,
                                 maxiter=maxiter, **kwargs)

    def loglike(k, U):
        return cube_runner.loglikelihood_batch(pix[k], U)

    return run_nested(loglike, cube_runner.ndim, pix.size, nlive=nlive, tol=tol, efr=efr, seed=seed,
                      maxiter=maxiter, **kwargs)
