"""Diagnostic (GPU box): where does the fast mode deviate most from the oracle?"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import nestfit_amd as na
from nestfit_amd.synth import freq_axis
from oracle import nfo

na.set_exp_mode(sys.argv[1] if len(sys.argv) > 1 else 'fast')
rng = np.random.default_rng(31)
rows = []
for trans in (1, 2, 3, 4, 9):
    for n in (256, 1024, 2048):
        x = freq_axis(trans, n, 40.0 if n == 2048 else 30.0)
        data = rng.normal(0, 0.3, n)
        sg = na.AmmoniaSpectrum(x, data, 0.3, trans)
        sc = nfo.AmmoniaSpectrum(x, data, 0.3, trans)
        for ncomp in (1, 2, 3):
            for _ in range(6):
                th = np.concatenate([rng.uniform(-4, 4, ncomp), rng.uniform(7, 30, ncomp),
                                     rng.uniform(2.8, 12, ncomp), rng.uniform(12.5, 16.5, ncomp),
                                     rng.uniform(0.067, 2.067, ncomp), rng.uniform(0, 0.5, ncomp)])
                na.amm_predict(sg, th)
                nfo.amm_predict(sc, th)
                pg, pc = sg.get_spec(), sc.get_spec()
                zp = np.array_equal(pg == 0, pc == 0)
                nz = pc != 0
                if not nz.any():
                    continue
                rel = np.abs(pg[nz] - pc[nz]) / np.abs(pc[nz])
                k = np.argmax(rel)
                idx = np.flatnonzero(nz)[k]
                rows.append((rel[k], trans, n, ncomp, idx, pc[idx], pg[idx], np.abs(pc).max(), zp,
                             abs(sg.loglikelihood - sc.loglikelihood) / abs(sc.loglikelihood)))
rows.sort(key=lambda r: -r[0])
for r in rows[:15]:
    print('rel %.3e trans %d n %d ncomp %d ch %d ref %.6e got %.6e peak %.3e zero-pattern %s lnL rel %.2e' % r)
rel = np.array([r[0] for r in rows])
print('cases', len(rows), 'median worst-rel', np.median(rel), 'p90', np.quantile(rel, 0.9), 'max', rel.max())
