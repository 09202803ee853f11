import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import nestfit_amd as na
from nestfit_amd.synth import freq_axis
n = 1024
rng = np.random.default_rng(0)
ut = na.get_irdc_priors()
axes = [freq_axis(t, n) for t in (1, 2)]
run = na.AmmoniaRunner.from_data([[axes[0], rng.normal(0, .2, n), 0.2, 1], [axes[1], rng.normal(0, .2, n), 0.2, 2]], ut, ncomp=2)
U = rng.uniform(size=(4096, 12))
for _ in range(5):
    V = U.copy(); ut.transform_batch(V, 2)          # prior_kernel alone
for _ in range(5):
    run.predict_batch(V, want_spectra=False)        # setup<.,false> + lnl
for _ in range(5):
    run.loglikelihood_batch(U.copy())               # setup<.,true> + lnl
