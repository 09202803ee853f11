#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (run under `rocprofv3 --pmc FETCH_SIZE` and,
separately, `--pmc WRITE_SIZE`):

  1. calibration: null_lnz_kernel streams a 256 MiB cube exactly once with the engine's
     own access pattern (8 B per lane, coalesced rows) -> known byte count;
  2. the benchmark batch (config C2, B = 4096) a few times.

profiles/traffic_summary.py turns the two counter CSVs into profiles/pmc_traffic.json.
"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import nestfit_amd as na
from nestfit_amd.cube import CubeRunner
from nestfit_amd.synth import TRUTH_2COMP, freq_axis

mode = sys.argv[1] if len(sys.argv) > 1 else 'fast'
na.set_exp_mode(mode)
n = 1024
axes = [freq_axis(t, n) for t in (1, 2)]
rng = np.random.default_rng(0)
# 1. calibration stream: 16384 pixels x 2048 channels x 8 B = 256 MiB
n_pix = 16384
data = rng.normal(0, 0.2, (n_pix, 2 * n))
ut = na.get_irdc_priors()
cube = CubeRunner(axes, (1, 2), data, np.full((n_pix, 2), 0.2), ut, ncomp=2)
print('calibration bytes', data.nbytes, 'null_lnZ checksum', float(cube.null_lnZ.sum()))
# 2. benchmark batch, one pixel (every item re-reads the same 16 KB of data)
run = na.AmmoniaRunner.from_data([[axes[0], data[0, :n], 0.2, 1], [axes[1], data[0, n:], 0.2, 2]], ut, ncomp=2)
U = rng.uniform(size=(4096, 12))
for _ in range(5):
    run.loglikelihood_batch(U.copy())
# 3. the same batch spread over 4096 different pixels (every item reads its own data)
pix = np.arange(4096, dtype=np.int32)
for _ in range(5):
    cube.loglikelihood_batch(pix, U.copy())
# 4. the spectra-out mode (predict_batch): 4096 rows of physical parameters against one pixel, 4096 x 2048 doubles written
theta = U.copy()
ut.transform_batch(theta, 2)
for _ in range(5):
    cube.predict_batch(np.zeros(4096, dtype=np.int32), theta)
print('done')
