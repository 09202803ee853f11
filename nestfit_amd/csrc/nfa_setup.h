// nfa_setup.h -- set-up stage of a likelihood batch: everything of
// AmmoniaRunner.c_loglikelihood that does not depend on the channel.
//
//   prior_transform_lane   unit cube -> theta      core/core.pyx:459-476 (all Prior kinds)
//   qsum_lane              partition sums          models/ammonia.pyx:289-315
//   derive_lane            tau_main, y(T0) model   models/ammonia.pyx:337-361
//
// This stage is scalar work per item (table look-ups, libm).  The mapping is one LANE per unit
// of scalar work, all three phases in one wave per 16 items (setup_kernel, one launch per batch):
//   * priors:         lane = item; the prior program is interpreted per lane, theta lives in
//                     LDS transposed ([slot][lane], conflict free);
//   * partition sums: lane = (item, component, quarter of the 51 J levels), the four partial
//                     sums meet through two DPP quad permutes;
//   * derive:         lane = (item, component, spectrum).
// theta and the partition sums pass from phase to phase through the wave's LDS.
// (prior_items_kernel = the first phase alone, for PriorTransformer.transform.)
#pragma once

// ---------------------------------------------------------------------------
//  prior program
// ---------------------------------------------------------------------------
struct DistDev {
    int     size, pad;
    double  du, dx, xmin, xmax;
    const double *xax, *pdf, *cdf, *ppf;
    // prefix moments of the trapezoid terms t_i = (pdf[i] + pdf[i-1]) / 2 (t_0 = 0) about the
    // table centre c0 = size / 2 (halves the cancellation when they are re-centred on i_lo):
    // m0[k] = sum_{i<=k} t_i, m1[k] = sum (i - c0) t_i, m2[k] = sum (i - c0)^2 t_i
    const double *m0, *m1, *m2;
};
#define MAXPRIOR 16
#define MAXDIST  16
// tables the set-up kernel copies into LDS before it interprets the program (a look-up is then
// an LDS access instead of a dependent trip to L2: the prior phase is a chain of ~35 of them)
#define MAXSTAGE (MAXDIST * 5)
enum { ST_XAX = 0, ST_PDF, ST_PPF, ST_M0, ST_M1, ST_M2 };
struct StageItem { int dist, field, n, off; };     // ds[dist].<field>: n doubles at LDS offset off (doubles)
struct PriorProg {
    int n_prior, n_dist, n_param, max_size;
    nfa_prior_desc pr[MAXPRIOR];
    DistDev        ds[MAXDIST];
    int            n_stage, stage_doubles;          // 0: the tables stay in global memory
    int            parallel, pad;                   // 1: no two priors share a parameter slot (one wave each)
    const double  *stage_image;                     // the staged tables back to back, in LDS order
    StageItem      stage[MAXSTAGE];
};

__device__ __forceinline__ double d_ppf_interp(const DistDev &d, double u) {   // core.pyx:47-63
    long i_lo = (long)((double)(d.size - 1) * u);
    long i_hi = i_lo + 1;
    i_lo = i_lo < 0 ? 0 : (i_lo > d.size - 1 ? d.size - 1 : i_lo);   // u==1 reads past the end
    i_hi = i_hi > d.size - 1 ? d.size - 1 : (i_hi < 0 ? 0 : i_hi);   // in the reference; clamp
    const double x_lo = (double)i_lo * d.du;
    const double y_lo = d.ppf[i_lo];
    const double y_hi = d.ppf[i_hi];
    const double slope = (y_hi - y_lo) / d.du;
    return slope * (u - x_lo) + y_lo;
}

// theta of this lane's item: slot k lives at u[k * 64 + lane]
#define TH(k) u[(k) * 64]

// `prior.interp(utheta, n)` of the simple kinds (core.pyx:192-197, 233-238, 242-258)
__device__ __forceinline__ void d_simple_interp(const PriorProg &pp, int kind, int dist, int p_ix,
                                                double value, double *u, int n) {
    const int ix = p_ix * n;
    if (kind == NFA_PRIOR_CONSTANT) {
        for (int i = 0; i < n; ++i) TH(ix + i) = value;
    } else if (kind == NFA_PRIOR_ORDERED) {
        double umin = 0.0;
        for (int i = 0; i < n; ++i) {
            const double uu = umin + (1 - umin) * TH(ix + i);
            umin = uu;
            TH(ix + i) = d_ppf_interp(pp.ds[dist], uu);
        }
    } else {
        for (int i = 0; i < n; ++i) TH(ix + i) = d_ppf_interp(pp.ds[dist], TH(ix + i));
    }
}

// The CDF that Distribution.cdf_over_interval (core.pyx:109-161) writes, evaluated at
// one index without materialising the table.  The running trapezoid sum
//     csum_k = sum_{i=ilo+1..k} t_i (1 - (i-ilo)/delta)^p
// is the difference of prefix moments for p = 0, 1, 2 (the only powers two- and
// three-component fits use) and a plain loop beyond.  Differs from the reference's
// sequential sum by rounding only (<= ~1e-13 of the interval's mass).
struct PlaceCtx {
    int ilo, ihi, size;
    double p, inv_delta, csum;
};

__device__ __forceinline__ double place_partial(const DistDev &d, const PlaceCtx &c, int k) {
    // k in (ilo, ihi)
    const double b0 = d.m0[c.ilo];
    const double d0 = d.m0[k] - b0;
    if (c.p == 0.0) return d0;
    const double dilo = (double)(c.ilo - d.size / 2);
    const double d1 = (d.m1[k] - d.m1[c.ilo]) - dilo * d0;                    // sum t_i (i - ilo)
    if (c.p == 1.0) return d0 - d1 * c.inv_delta;
    if (c.p == 2.0) {
        const double d2 = (d.m2[k] - d.m2[c.ilo]) - 2.0 * dilo * (d.m1[k] - d.m1[c.ilo]) + dilo * dilo * d0;
        return d0 - 2.0 * d1 * c.inv_delta + d2 * c.inv_delta * c.inv_delta;
    }
    double s = 0.0;
    for (int i = c.ilo + 1; i <= k; ++i)
        s += 0.5 * (d.pdf[i] + d.pdf[i - 1]) * pow(1.0 - (double)(i - c.ilo) * c.inv_delta, c.p);
    return s;
}

__device__ __forceinline__ double place_cdf_at(const DistDev &d, const PlaceCtx &c, int k) {
    if (k < c.ilo) return 0.0;                                  // core.pyx:133-134
    if (k >= c.ihi) return 1.0;                                 // core.pyx:135-136
    if (c.ihi - c.ilo == 1) return 1.0 / c.csum;                // core.pyx:139-140, 160-161 (csum == 0)
    if (k == c.ilo) return 0.0 / c.csum;                        // core.pyx:142, 160-161
    return place_partial(d, c, k) / c.csum;
}

// cdf_over_interval + cdf_interp (core.pyx:65-161) for one lane
__device__ double d_placement_draw(const DistDev &d, double x_lo, double x_hi, double sfact, double u) {
    if (x_lo > x_hi) { const double t = x_lo; x_lo = x_hi; x_hi = t; }     // core.pyx:116-117
    PlaceCtx c;
    c.size = d.size;
    long i_lo = (long)((x_lo - d.xmin) / d.dx);                            // core.pyx:120-131
    if (i_lo >= c.size) i_lo = c.size - 1; else if (i_lo < 0) i_lo = 0;
    long i_hi = (long)((x_hi - d.xmin) / d.dx);
    if (i_hi == i_lo) i_hi = i_lo + 1;
    if (i_hi > c.size) i_hi = c.size; else if (i_hi < 0) i_hi = 1;
    c.ilo = (int)i_lo; c.ihi = (int)i_hi;
    c.p = sfact;
    c.inv_delta = 1.0 / (double)(c.ihi - c.ilo);
    c.csum = (c.ihi - c.ilo > 1) ? place_partial(d, c, c.ihi - 1) : 0.0;
    // cdf_interp: the reference's bisection, probe by probe (core.pyx:83-96)
    if (u <= place_cdf_at(d, c, 0)) u = 1e-64;
    int lo = 0, hi = c.size, i = hi / 2;
    while (i != lo) {
        if (u > place_cdf_at(d, c, i)) lo = i; else hi = i;
        i = (hi + lo) / 2;
    }
    const int j_lo = i < c.size ? i : c.size - 1;
    int j_hi = j_lo + 1;
    if (j_hi > c.size - 1) j_hi = c.size - 1;                   // the reference reads cdf[size] here
    const double xl = d.xax[j_lo];
    const double y_lo = place_cdf_at(d, c, j_lo);
    const double y_hi = place_cdf_at(d, c, j_hi);
    const double slope = (y_hi - y_lo) / d.dx;                  // core.pyx:102-107
    return 1 / slope * (u - y_lo) + xl;
}

// prior k of PriorTransformer.c_transform (core.pyx:459-476) for the item of this lane
// (Inlined into its kernels since round 5: as a function of its own it was compiled to 248 vector registers -- a callee
// has no occupancy to aim for -- and every kernel that calls it is allotted its callees' maximum: the set-up kernel ran
// two waves per SIMD for a body that needs 165.  Inlined: three waves, and the launch stands less in the way of the
// likelihood launches of the neighbouring lanes -- 152.7 -> 157.4 M evaluations/s on the metric shape in the fast mode,
// config 5 (rounds 2-4's cube) 3.41 -> 3.38 s.  Held to 128 registers (four waves) it gains nothing more.)
__device__ __forceinline__ void prior_apply_lane(const PriorProg &pp, int k, double *u, int n) {
    {
        const nfa_prior_desc &p = pp.pr[k];
        const int ix = p.p_ix * n;
        switch (p.kind) {
        case NFA_PRIOR_SIMPLE:
        case NFA_PRIOR_CONSTANT:
        case NFA_PRIOR_ORDERED:
            d_simple_interp(pp, p.kind, p.dist0, p.p_ix, p.value, u, n);
            break;
        case NFA_PRIOR_DUPLICATE:                             // core.pyx:211-221
            for (int i = 0; i < n; ++i) {
                const double v = d_ppf_interp(pp.ds[p.dist0], TH(ix + i));
                TH(ix + i) = v;
                TH(p.p_ix2 * n + i) = v;
            }
            break;
        case NFA_PRIOR_SPACED: {                              // core.pyx:280-292
            double v = d_ppf_interp(pp.ds[p.dist0], TH(ix));
            TH(ix) = v;
            for (int i = 1; i < n; ++i) {
                v = v + d_ppf_interp(pp.ds[p.dist1], TH(ix + i));
                TH(ix + i) = v;
            }
        } break;
        case NFA_PRIOR_CENSEP: {                              // core.pyx:305-318
            const double vcen = d_ppf_interp(pp.ds[p.dist0], TH(ix));
            if (n == 1) TH(ix) = vcen;
            else if (n == 2) {
                const double vsep = d_ppf_interp(pp.ds[p.dist1], TH(ix + 1));
                TH(ix)     = vcen - 0.5 * vsep;
                TH(ix + 1) = vcen + 0.5 * vsep;
            }
        } break;
        case NFA_PRIOR_RESOLVED_CENSEP: {                     // core.pyx:347-366
            const int ix_s = p.p_ix2 * n;
            d_simple_interp(pp, p.sub_kind, p.dist2, p.p_ix2, p.value, u, n);
            const double vcen = d_ppf_interp(pp.ds[p.dist0], TH(ix));
            if (n == 1) TH(ix) = vcen;
            else if (n == 2) {
                double vsep = d_ppf_interp(pp.ds[p.dist1], TH(ix + 1));
                const double min_sep = p.sep_scale * sqrt(TH(ix_s) * TH(ix_s + 1));
                if (min_sep > vsep) vsep = min_sep;
                TH(ix)     = vcen - 0.5 * vsep;
                TH(ix + 1) = vcen + 0.5 * vsep;
            }
        } break;
        case NFA_PRIOR_RESOLVED_PLACEMENT: {                  // core.pyx:391-435
            if (n > MAXCOMP) break;
            const DistDev &vd = pp.ds[p.dist0];
            const int ix_s = p.p_ix2 * n;
            double v_lo = vd.xmin, v_hi = vd.xmax;
            d_simple_interp(pp, p.sub_kind, p.dist2, p.p_ix2, p.value, u, n);
            if (n == 1) { TH(ix) = d_ppf_interp(vd, TH(ix)); break; }
            double sep_tot = 0.0;                             // core.pyx:409-415
            for (int i = 1; i < n; ++i) sep_tot += p.sep_scale * sqrt(TH(ix_s + i) * TH(ix_s + i - 1));
            double overf = 1.0;
            const bool shrink = sep_tot > v_hi - v_lo;        // core.pyx:418-423
            if (shrink) {
                overf = (v_hi - v_lo) / sep_tot;
                sep_tot = 0.0;
                for (int i = 1; i < n; ++i)
                    sep_tot += (p.sep_scale * sqrt(TH(ix_s + i) * TH(ix_s + i - 1))) * overf;
            }
            v_hi -= sep_tot;
            for (int i = 0; i < n; ++i) {                     // core.pyx:427-435
                double sep = (i == 0) ? 0.0 : p.sep_scale * sqrt(TH(ix_s + i) * TH(ix_s + i - 1));
                if (shrink) sep *= overf;
                v_lo += sep;
                v_hi += sep;
                v_lo = d_placement_draw(vd, v_lo, v_hi, (double)(n - 1 - i), TH(ix + i));
                TH(ix + i) = v_lo;
            }
        } break;
        default: break;
        }
    }
}

// PriorTransformer.c_transform (core.pyx:459-476) for the item of this lane
__device__ __forceinline__ void prior_transform_lane(const PriorProg &pp, double *u, int n) {
    for (int k = 0; k < pp.n_prior; ++k) prior_apply_lane(pp, k, u, n);
}

// prior_items_kernel: one lane per item.  LDS: theta transposed, [ndim][64] doubles.
__global__ void __launch_bounds__(64) prior_items_kernel(const PriorProg *__restrict__ ppp,
                                                         double *__restrict__ U, long B, int n) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __builtin_amdgcn_s_setprio(3);     // few waves, long dependent chains: do not queue behind the likelihood waves
    const PriorProg &pp = *ppp;
    const int lane = threadIdx.x;
    const int ndim = pp.n_param * n;
    const long b = (long)blockIdx.x * 64 + lane;
    if (b >= B) return;
    double *u = smem + lane;
    for (int k = 0; k < ndim; ++k) TH(k) = U[b * ndim + k];
    prior_transform_lane(pp, u, n);
    for (int k = 0; k < ndim; ++k) U[b * ndim + k] = TH(k);
}
#undef TH

// ---------------------------------------------------------------------------
//  partition sums (ammonia.pyx:289-315): lane = (item, component, J mod 16)
//  qrec[{0: qpara, 1: qorth, 2: trot', 3..11: (2J+1) FastExp(E_J/kT), J = 1..9}] per (item, component)
//  Sixteen lanes, one level each per pass of sixteen levels; the partial sums meet through four DPP steps.
//  FastExp is exactly 0 from an argument of 32 on (fastexp.c:272-273) and the level energies grow with J, so a
//  pass in which no lane of the wave has a level below that adds zeros, and so does every pass after it: it is
//  not run.  Below ~100 K that leaves the first pass -- the sum of a lone item is then ONE exponential deep
//  (it was seven with eight lanes of seven levels), and a batch runs a quarter of the exponentials.
// ---------------------------------------------------------------------------
#define QREC 12
#define QSUM_LANES 16
template <int MODE>
__device__ __forceinline__ void qsum_lane(bool on, double trot_in, int cold, int j0, double *qrec,
                                          const double *sm) {
    double trot = trot_in;
    if (cold) trot = nf_swift(trot);                          // ammonia.pyx:344-345
    double qp = 0.0, qo = 0.0;
    for (int base = 0; base < NFA_NPART; base += QSUM_LANES) {
        const int j = base + j0;
        const bool lev_on = on && j < NFA_NPART;
        const double arg = nf_partition_arg(lev_on ? j : 0, trot);
        // (a NaN or a negative argument is "below 32": those sums run to the end like the reference's)
        if (base > 0 && __builtin_amdgcn_ballot_w64(lev_on && !((float)arg >= 32.0f)) == 0ull) break;
        double lev = (double)(2 * j + 1) * nf_fastexp<MODE>(arg, sm);
        if (!lev_on) lev = 0.0;
        if (j % 3 == 0) qo += 2 * lev; else qp += lev;
        if (lev_on && j >= 1 && j <= NFA_N_LEVELS) qrec[2 + j] = lev;
    }
    // the sixteen lanes of one (item, component) are one DPP row: pairs, quads, half rows, the row
    qp += dpp_move<0xB1>(qp); qp += dpp_move<0x4E>(qp); qp += dpp_move<0x141>(qp); qp += dpp_move<0x140>(qp);
    qo += dpp_move<0xB1>(qo); qo += dpp_move<0x4E>(qo); qo += dpp_move<0x141>(qo); qo += dpp_move<0x140>(qo);
    if (on && j0 == 0) { qrec[0] = qp; qrec[1] = qo; qrec[2] = trot; }
}

// y(T0) = 1/(e^(T0/tex) - 1) over the band of spectrum s, as the fast mode evaluates it:
// kind 1 / 2 = the reference's table cell(s) (hyperfine.pyx:30-45) written as a line in T0,
// kind 3 = band entirely outside the table (exact function there): quadratic about the band
// centre, kind 0 = anything else (per-lane nf_iemtex).
// FAST: the record is for the fast mode's kernel only (the exact modes' cell is left out).
template <bool FAST>
__device__ __forceinline__ void write_y_model(double *dk, const SpecDev &S, int s, double tex,
                                              const double *__restrict__ g_tabs) {
    const double T0a = S.t0[S.off[s]], T0b = S.t0[S.off[s] + S.size[s] - 1];
    const double inv_tex = 1.0 / tex;
    const double xa = T0a * inv_tex, xb = T0b * inv_tex;
    const bool ina = S.t0_xmin < xa && xa < S.t0_xmax, inb = S.t0_xmin < xb && xb < S.t0_xmax;
    double kind = 0.0, A0 = 0.0, B0 = 0.0, A1 = 0.0, B1 = 0.0, split = INFINITY, m = 0.0, q = 0.0;
    const double *t0x = g_tabs + SM_T0X, *t0y = g_tabs + SM_T0Y;
    if (ina && inb) {
        const long ia = (long)((xa - S.t0_xmin) * S.t0_inv_dx);
        const long ib = (long)((xb - S.t0_xmin) * S.t0_inv_dx);
        if (ia >= 0 && ib <= T0_SIZE - 2 && ib - ia <= 1) {
            const double sl0 = (t0y[ia + 1] - t0y[ia]) * S.t0_inv_dx;
            A0 = t0y[ia] - sl0 * t0x[ia];
            B0 = sl0 * inv_tex;
            A1 = A0; B1 = B0;
            kind = 1.0;
            if (ib != ia) {
                const double sl1 = (t0y[ib + 1] - t0y[ib]) * S.t0_inv_dx;
                A1 = t0y[ib] - sl1 * t0x[ib];
                B1 = sl1 * inv_tex;
                split = t0x[ib] * tex;
                kind = 2.0;
            }
        }
    } else if (!ina && !inb && ((xa <= S.t0_xmin && xb <= S.t0_xmin) || (xa >= S.t0_xmax && xb >= S.t0_xmax))) {
        m = 0.5 * (T0a + T0b);
        const double y = 1.0 / expm1(m * inv_tex);
        A0 = y;
        B0 = -y * (1.0 + y) * inv_tex;
        q = 0.5 * (1.0 + 2.0 * y) * y * (1.0 + y) * inv_tex * inv_tex;
        A1 = A0; B1 = B0;
        kind = 3.0;
    }
    dk[DK_KIND] = kind; dk[DK_A0] = A0; dk[DK_B0] = B0; dk[DK_A1] = A1; dk[DK_B1] = B1;
    dk[DK_SPLIT] = split; dk[DK_M] = m; dk[DK_Q] = q;
    const double kappa = NFA_H / NFA_KB;                       // T0 = kappa x (hyperfine.pyx:106)
    dk[DK_A0X] = A0 * kappa; dk[DK_B0X] = B0 * (kappa * kappa);
    // the exact modes' cell (x by division, the cell found exactly as nf_iemtex finds it)
    double xkind = 0.0, xs = 0.0, xlo = 0.0, ylo = 0.0;
    const double xad = FAST ? 0.0 : T0a / tex, xbd = FAST ? 0.0 : T0b / tex;
    if (!FAST && S.t0_xmin < xad && xad < S.t0_xmax && S.t0_xmin < xbd && xbd < S.t0_xmax) {
        long ia = (long)((xad - S.t0_xmin) * S.t0_inv_dx), ib = (long)((xbd - S.t0_xmin) * S.t0_inv_dx);
        ia = ia > T0_SIZE - 2 ? T0_SIZE - 2 : ia;
        ib = ib > T0_SIZE - 2 ? T0_SIZE - 2 : ib;
        if (ia == ib && ia >= 0) {
            xkind = 1.0; xlo = t0x[ia]; ylo = t0y[ia];
            xs = (t0y[ia + 1] - ylo) * S.t0_inv_dx;
        }
    }
    dk[DK_XKIND] = xkind; dk[DK_XS] = xs; dk[DK_XLO] = xlo; dk[DK_YLO] = ylo; dk[15] = 0.0;
}

// ---------------------------------------------------------------------------
//  derive: lane = (item, component, spectrum); ammonia.pyx:337-361.  TH(k) = parameter slot k of the
//  lane's item (parameter-major, ammonia.pyx:338-343), qrec = its partition record.
// ---------------------------------------------------------------------------
#define TH(k) th[(k) * 64]
// FAST (the fast mode, 1e-6 on Tb): 10^ntot by exp10, and tau_main as it stands -- the reference hands it on as
// log10(tau_main) and c_hf_predict raises 10 to that (ammonia.pyx:361, hyperfine.pyx:63), a round trip of a few 1e-16
// that the exact modes make as well.
template <bool FAST>
__device__ __forceinline__ void derive_lane(const SpecDev &S, const double *th, const double *qrec, double *Db,
                                            int c, int s, const double *__restrict__ g_tabs) {
    const int ncomp = S.ncomp, nspec = S.n_spec;
    const int t = S.trans[s] - 1;
    const double nu0 = c_nu[t];
    const bool para = ((t + 1) % 3) != 0;
    const double trot = qrec[2];
    double tex = TH(2 * ncomp + c);
    if (S.lte) tex = trot;                                    // ammonia.pyx:346
    const double ntot = TH(3 * ncomp + c);
    const double sigm = TH(4 * ncomp + c);
    const double orth = TH(5 * ncomp + c);
    const double zlev = qrec[2 + (t + 1)];
    const double qtot = para ? qrec[0] : qrec[1];
    const double species_frac = para ? 1.0 - orth : orth;
    // (10^x by exp10 in every mode: a fifth of pow's instructions in a chain the set-up launch waits for; both are within an
    // ulp of the reference's libm, neither is its bits)
    const double pop_rotstate = exp10(ntot) * species_frac * zlev / qtot;
    const double ex = exp(-NFA_H * nu0 / (NFA_KB * tex));
    const double expterm = (1.0 - ex) / (1.0 + ex);
    const double fracterm = (NFA_CCMS * NFA_CCMS) * c_ea[t] / (8 * M_PI * (nu0 * nu0));
    const double widthterm = NFA_CKMS / (sigm * nu0 * sqrt(2 * M_PI));
    const double tau_main = pop_rotstate * fracterm * expterm * widthterm;
    if (s == 0) {
        double *d = Db + c * 4;
        d[0] = tex;
        d[1] = sigm / NFA_CKMS;                               // hyperfine.pyx:72
        d[2] = TH(c) / NFA_CKMS;                              // hyperfine.pyx:73
        d[3] = 1.0 / tex;
    }
    double *dk = Db + 4 * ncomp + (c * nspec + s) * DREC_CS;
    dk[DK_TMAIN] = FAST ? tau_main : exp10(log10(tau_main));          // ammonia.pyx:361, hyperfine.pyx:63
    write_y_model<FAST>(dk, S, s, tex, g_tabs);
}

//  The sibling models hand c_hf_predict its arguments directly.
//  N2H+ (diazenylium.pyx:138-154): voff, tex, ltau, sigm -> tau_main = 10**ltau (hyperfine.pyx:63)
//  Gaussian (gaussian.pyx:17-35): voff, sigm, peak -> one line of weight `peak`, no Tb pass
template <bool FAST>
__device__ __forceinline__ void derive_simple_lane(const SpecDev &S, const double *th, double *Db, int c, int s,
                                                   const double *__restrict__ g_tabs) {
    const int ncomp = S.ncomp, nspec = S.n_spec;
    const bool gauss = S.model == NFA_MODEL_GAUSSIAN;
    const double voff = TH(c);
    const double tex  = gauss ? 1.0 : TH(ncomp + c);
    const double sigm = gauss ? TH(ncomp + c) : TH(3 * ncomp + c);
    const double amp  = gauss ? TH(2 * ncomp + c) : pow(10.0, TH(2 * ncomp + c));
    if (s == 0) {
        double *d = Db + c * 4;
        d[0] = tex;
        d[1] = sigm / NFA_CKMS;
        d[2] = voff / NFA_CKMS;
        d[3] = 1.0 / tex;
    }
    double *dk = Db + 4 * ncomp + (c * nspec + s) * DREC_CS;
    dk[DK_TMAIN] = amp;
    if (gauss) {
        for (int q = 1; q < DREC_CS; ++q) dk[q] = 0.0;
    } else {
        write_y_model<FAST>(dk, S, s, tex, g_tabs);
    }
}

// ---------------------------------------------------------------------------
//  setup_kernel: the whole set-up stage of SETUP_TI = 64 items in one workgroup of four waves,
//  one launch per batch:
//      all threads                          prior tables -> LDS (StageItem list of the program)
//      wave 0, lanes = items                unit cube -> theta (written back to U in place)
//      lanes = (item, component, J mod 16)  partition sums -> LDS
//      lanes = (item, component, spectrum)  derived record D of the item
//  theta and the partition records stay in LDS between the phases (the three-kernel version of
//  this stage paid three launch gaps and two round trips through global memory per batch, and
//  its prior phase was a chain of ~35 dependent trips to L2: the latency of the stage, not its
//  work, set the pace of a 4096-row batch).  The waves raise their priority: there are few of
//  them, each a long dependent chain, beside thousands of likelihood waves of the neighbouring
//  stream lanes.
//  LDS: [exp tables][theta: ndim x 64][Q: 64 x ncomp x QREC][PriorProg copy][staged prior tables]
// ---------------------------------------------------------------------------
#define SETUP_TI 64
#define SETUP_THREADS 256
// the set-up stage of the 64 items of workgroup `block_id`, by the blockDim.x threads of the workgroup (`sm` =
// the staged exponential tables, n_shared doubles at the start of smem)
// STAGED: the prior program and its tables are in LDS already (setup_stage_priors, once per resident workgroup)
template <int MODE, bool FAST = false, int NSUB = 1, bool STAGED = false>
__device__ __forceinline__ void setup_body(const PriorProg *__restrict__ ppp, const SpecDev &S,
                                           double *__restrict__ U, double *__restrict__ D, long B, int has_prior,
                                           const double *__restrict__ g_tabs, int ablate_in, double *smem,
                                           const double *sm, int n_shared, unsigned block_id, int ti = SETUP_TI) {
    constexpr int nsub = NSUB;
#ifdef NFA_ABLATE
    const int ablate = ablate_in;      // timing experiments: 16 skip the priors, 32 the partition sums, 64 the derive phase
#else
    const int ablate = 0;
#endif
    // nsub > 1: the workgroup is `nsub` groups of threads, each with `ti` items of its own (theta and partition records
    // side by side in LDS) behind ONE copy of the exponential tables, the prior program and its tables -- the table
    // mode's 92 KB of tables are then staged once per 128 items, and the device holds the whole batch in one round of
    // workgroups instead of two
    const int tid_wg = threadIdx.x, nthr_wg = blockDim.x;
    const int nthr = nthr_wg / nsub, sub = tid_wg / nthr, tid = tid_wg - sub * nthr;
    const int ncomp = S.ncomp, nspec = S.n_spec, ndim = S.npar * ncomp;
    const int drec = drec_size(ncomp, nspec);
    const int per_sub = 64 * ndim + SETUP_TI * ncomp * QREC;
    double *th_all = smem + n_shared + sub * per_sub;          // theta of item `it`: th_all[k * 64 + it]
    double *q_all = th_all + 64 * ndim;
    PriorProg *lp = (PriorProg *)(smem + n_shared + nsub * per_sub);
    double *tab = (double *)(lp + 1);
    const long b0 = ((long)block_id * nsub + sub) * ti;        // ti <= SETUP_TI items per group (the LDS layout is SETUP_TI's)
    const int n_it = (int)(B - b0 < ti ? (B - b0 > 0 ? B - b0 : 0) : ti);
    const bool do_prior = has_prior && !(ablate & 16);
    // ---- phase 0: the program and its tables -> LDS (flat copies: all loads in flight at once); theta -> LDS
    if (do_prior && !STAGED) {
        const int nw = (int)(sizeof(PriorProg) / sizeof(int));
        for (int k = tid_wg; k < nw; k += nthr_wg) ((int *)lp)[k] = ((const int *)ppp)[k];
        const double *image = ppp->stage_image;
        const int n_tab = ppp->stage_doubles;
        for (int k = tid_wg; k < n_tab; k += nthr_wg) tab[k] = image[k];
    }
    for (int q = tid; q < n_it * ndim; q += nthr) {            // coalesced; LDS holds it transposed
        const int it = q / ndim, k = q - it * ndim;
        th_all[k * 64 + it] = U[b0 * ndim + q];
    }
    __syncthreads();
    if (!STAGED) {
        if (do_prior && tid_wg < lp->n_stage) {                // the LDS copy of the program points at the LDS tables
            const StageItem it = lp->stage[tid_wg];
            DistDev &d = lp->ds[it.dist];
            const double *p = tab + it.off;
            if (it.field == ST_XAX) d.xax = p; else if (it.field == ST_PDF) d.pdf = p; else if (it.field == ST_PPF) d.ppf = p;
            else if (it.field == ST_M0) d.m0 = p; else if (it.field == ST_M1) d.m1 = p; else d.m2 = p;
        }
        __syncthreads();
    }
    // ---- phase 1: lanes = items (core.pyx:459-476); priors that share no parameter slot take a wave each
    // (a wave interprets ONE prior for its 64 items: no divergence, and the longest prior sets the time)
    if (do_prior) {
        const int lane = tid & 63, wave = tid >> 6, n_waves = nthr >> 6;
        if (lp->parallel) {
            for (int k = wave; k < lp->n_prior; k += n_waves)
                if (lane < n_it) prior_apply_lane(*lp, k, th_all + lane, ncomp);
        } else if (tid < n_it) {
            prior_transform_lane(*lp, th_all + tid, ncomp);
        }
    }
    __syncthreads();
    if (do_prior)
        for (int q = tid; q < n_it * ndim; q += nthr) {
            const int it = q / ndim, k = q - it * ndim;
            U[b0 * ndim + q] = th_all[k * 64 + it];
        }
    const bool ammonia = S.model == NFA_MODEL_AMMONIA;
    // ---- phase 2: lanes = (item, component, J mod 16)
    if (ammonia && !(ablate & 32)) {
        const int n_task = n_it * ncomp * QSUM_LANES;           // whole groups of sixteen are on or off
        for (int q0 = 0; q0 < n_task; q0 += nthr) {
            const int q = q0 + tid;
            const int pair = q / QSUM_LANES, chunk = q % QSUM_LANES;
            const int it = pair / ncomp, c = pair - it * ncomp;
            const bool on = q < n_task;
            const double trot = on ? th_all[(ncomp + c) * 64 + it] : 1.0;
            qsum_lane<MODE>(on, trot, S.cold, chunk, q_all + (on ? pair : 0) * QREC, sm);
        }
    }
    __syncthreads();
    // ---- phase 3: lanes = (item, component, spectrum)
    const int per_item = ncomp * nspec;
    for (int q = tid; q < n_it * per_item && !(ablate & 64); q += nthr) {
        const int it = q / per_item, k = q - it * per_item;
        const int c = k / nspec, s = k - c * nspec;
        double *Db = D + (b0 + it) * drec;
        if (ammonia) derive_lane<FAST>(S, th_all + it, q_all + (it * ncomp + c) * QREC, Db, c, s, g_tabs);
        else derive_simple_lane<FAST>(S, th_all + it, Db, c, s, g_tabs);
    }
}

// MODE: the exponential of the partition sums (0 the reference's tables, 1 the polynomial); FAST: the record of the fast mode
// The prior program and its tables -> LDS, where setup_body<..., STAGED = true> expects them (one group of items): for a
// resident workgroup that serves point after point with the same priors (the ring's kernel: 1.7 us of every point's ~23
// went into this copy and its two barriers).  Doubles of LDS from `smem + n_shared` up to the end of the tables: the
// caller keeps everything else (line tables) behind that.
__device__ __forceinline__ int setup_stage_priors(const PriorProg *__restrict__ ppp, const SpecDev &S, double *smem, int n_shared) {
    const int ncomp = S.ncomp, ndim = S.npar * ncomp;
    const int per_sub = 64 * ndim + SETUP_TI * ncomp * QREC;
    PriorProg *lp = (PriorProg *)(smem + n_shared + per_sub);
    double *tab = (double *)(lp + 1);
    const int nw = (int)(sizeof(PriorProg) / sizeof(int));
    for (int k = threadIdx.x; k < nw; k += blockDim.x) ((int *)lp)[k] = ((const int *)ppp)[k];
    const double *image = ppp->stage_image;
    const int n_tab = ppp->stage_doubles;
    for (int k = threadIdx.x; k < n_tab; k += blockDim.x) tab[k] = image[k];
    __syncthreads();
    if ((int)threadIdx.x < lp->n_stage) {
        const StageItem it = lp->stage[threadIdx.x];
        DistDev &d = lp->ds[it.dist];
        const double *p = tab + it.off;
        if (it.field == ST_XAX) d.xax = p; else if (it.field == ST_PDF) d.pdf = p; else if (it.field == ST_PPF) d.ppf = p;
        else if (it.field == ST_M0) d.m0 = p; else if (it.field == ST_M1) d.m1 = p; else d.m2 = p;
    }
    __syncthreads();
    return per_sub + (int)(sizeof(PriorProg) / sizeof(double)) + 1 + n_tab;
}

// NSUB = 2: sixteen waves, two groups of eight with `ti` items each (setup_body)
template <int MODE, bool FAST = false, int NSUB = 1>
__global__ void __launch_bounds__(512 * NSUB) __attribute__((amdgpu_waves_per_eu(3))) setup_kernel(const PriorProg *__restrict__ ppp, SpecDev S,
                                                    BatchGroup grp, double *__restrict__ D,
                                                    long B, int has_prior,
                                                    const double *__restrict__ g_tabs, int ablate_in, int ti) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // the workgroup's items belong to one batch of the group (the host sees to it that `each` is a multiple of ti):
    // its unit-cube array, shifted so that the body can go on indexing it with the item's number in the launch
    const int c = group_of(grp, (long)blockIdx.x * ti * NSUB);
    double *U = grp.U[c] - (long)c * grp.each * (S.npar * S.ncomp);
    __builtin_amdgcn_s_setprio(3);
    int n_shared;
    const double *sm = stage_exp_tables<MODE>(smem, g_tabs, &n_shared);
    setup_body<MODE, FAST, NSUB>(ppp, S, U, D, B, has_prior, g_tabs, ablate_in, smem, sm, n_shared, blockIdx.x, ti);
}

// ---------------------------------------------------------------------------
//  One point per call (MultiNest's callback, core.pyx:513-531 through ammonia.pyx:405-432) and the few points
//  a broker gathers from concurrent callers: the whole path in ONE launch, one workgroup per point -- the set-up
//  stage, the likelihood waves and the sum follow each other across workgroup barriers, and theta, lnL and a
//  sequence number are written straight into a mapped host buffer the caller spins on.  A single point arrives in
//  the kernel arguments, several are read from the mapped buffer.  No copy commands, no second and third launch,
//  no stream synchronisation: what is left is one dispatch and the dependent chain of the arithmetic itself.
//  Same device functions as the batch kernels, so a point gives the same bits either way.
//  Mapped buffer (doubles): [theta out: n x ndim][lnL out: n][sequence number][unit cube in: n x ndim][pixel in: n ints]
// ---------------------------------------------------------------------------
#define NFA_POINT_MAXDIM 24
#define NFA_POINT_MAXB 128
#define POINT_THREADS 512
#define POINT_WAVES (POINT_THREADS / 64)
struct PointIn {
    double u[NFA_POINT_MAXDIM];        // n == 1: the point
    unsigned long long seq;            // written to the host buffer last
    int n;                             // points of this launch = workgroups
    int pix;                           // n == 1: its pixel (< 0: the runner has one pixel); n > 1: pixels given or not
    int n_blocks;                      // likelihood workgroups the one workgroup stands in for
    int pad;
};
template <int MODE, int NCOMP>
__global__ void __launch_bounds__(POINT_THREADS) point_kernel(const PriorProg *__restrict__ ppp, SpecDev S, PointIn in,
                                                              int *__restrict__ d_pix, double *__restrict__ U_all,
                                                              double *__restrict__ D_all, double *__restrict__ part_all,
                                                              double *__restrict__ host, unsigned *__restrict__ n_done,
                                                              LnlGeom G, const double *__restrict__ g_tabs) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int SMODE = MODE == 0 ? 0 : 1;                   // fast mode: the set-up stage runs its polynomial form
    int n_shared;
    const double *sm = stage_exp_tables<SMODE>(smem, g_tabs, &n_shared);
    const int tid = threadIdx.x;
    const int ndim = S.npar * S.ncomp, n = in.n;
    const long b = blockIdx.x;
    double *U = U_all + b * ndim, *D = D_all + b * drec_size(S.ncomp, S.n_spec), *part = part_all + b * S.n_spec;
    double *out_theta = host + b * ndim, *out_lnl = host + (long)n * ndim + b;
    unsigned long long *out_seq = (unsigned long long *)(host + (long)n * (ndim + 1));
    const double *in_u = host + (long)n * (ndim + 1) + 1;
    const int *in_pix = (const int *)(in_u + (long)n * ndim);
    int my_pix = in.pix;
    if (n == 1) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < NFA_POINT_MAXDIM; ++k) v = tid == k ? in.u[k] : v;      // kernel arguments: no dynamic index
        if (tid < ndim) U[tid] = v;
    } else {
        if (tid < ndim) U[tid] = in_u[b * ndim + tid];           // one trip over PCIe per workgroup
        if (in.pix >= 0) my_pix = in_pix[b];
    }
    if (tid == 0 && in.pix >= 0) d_pix[b] = my_pix;
    __syncthreads();
    setup_body<SMODE, MODE == 2>(ppp, S, U, D, 1, 1, g_tabs, 0, smem, sm, n_shared, 0u);
    __threadfence();                                            // theta in U, the derived record in D: at L2 ...
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();                            // ... where the scalar loads of the record find them
    if (tid < ndim) out_theta[tid] = U[tid];
    const int *pix = in.pix >= 0 ? d_pix + b : nullptr;
    for (int blk = 0; blk < in.n_blocks; ++blk) {
        if (blk) __syncthreads();                               // the line tables are reused
        lnl_body<MODE, false, false, NCOMP>(S, pix, D, part, nullptr, 1, G, g_tabs, smem, MODE == 2 ? smem : sm,
                                            MODE == 2 ? 0 : n_shared, (unsigned)blk);
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        *out_lnl = lnl_of_item(part, S.noise, in.pix >= 0 ? (long)my_pix : 0, 0, S.n_spec);
        __threadfence_system();
        // the last workgroup to get here publishes the sequence number (and leaves the counter at zero)
        bool last = true;
        if (n > 1) {
            const unsigned seen = __hip_atomic_fetch_add(n_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            last = seen == (unsigned)n - 1u;
            if (last) __hip_atomic_store(n_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (last) __hip_atomic_store(out_seq, in.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#undef TH
