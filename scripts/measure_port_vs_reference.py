#!/usr/bin/env python3
"""the speed of the CPU restatement (oracle/nf_oracle.c, reference compile flags) beside the
real reference's, on the same CPU.

The reference's extensions cannot be rebuilt in this repository's rounds, so its side of the
ratio is the number the survey measured in this same build container (Intel Xeon @ 2.10 GHz
KVM guest, reference flags, 1024 channels, 2 components, NH3 (1,1)+(2,2), `Runner.loglikelihood(u)` in a Python
loop: 12,653-12,783 lnL/s on one core, 102,147 on eight; of the 78 us per evaluation about 6 are the Python call).
This script times the port here on the same shape with the same prior set and uniform unit-cube draws, on one core,
and writes profiles/r03/port_vs_reference.json; bench.py scales its `cpu_baseline` by that ratio
(`cpu_baseline.ratio_to_reference`, `reference_equivalent`).

    python scripts/measure_port_vs_reference.py            (build container, no GPU)
"""
import json
import os
import platform
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

REFERENCE_ONE_CORE = (12653.0 + 12783.0) / 2      # SURVEY.md section 6, 1024 ch, 2 comp, (1,1)+(2,2), 1 process
REFERENCE_EIGHT_CORES = 102147.0


def main():
    import nestfit_amd as na
    from nestfit_amd.synth import TRUTH_2COMP, freq_axis
    from oracle import nfo
    nfo.build(native=True)
    n = 1024
    rng = np.random.default_rng(5)
    spectra = []
    for t in (1, 2):
        x = freq_axis(t, n)
        s = nfo.AmmoniaSpectrum(x, np.zeros(n), 0.2, t, native=True)
        nfo.amm_predict(s, TRUTH_2COMP)
        spectra.append(nfo.AmmoniaSpectrum(x, s.get_spec() + rng.normal(0, 0.2, n), 0.2, t, native=True))
    ut = na.get_irdc_priors(size=500, vsys=0.0)
    run = nfo.AmmoniaRunner(spectra, nfo.PriorSet(ut.lower()), ncomp=2, native=True)
    U = np.random.default_rng(7).uniform(size=(4096, 12))
    run.loglikelihood_batch(U[:256].copy())
    rates = []
    for _ in range(5):
        Uc = U.copy()
        t0 = time.perf_counter()
        run.loglikelihood_batch(Uc)
        rates.append(U.shape[0] / (time.perf_counter() - t0))
    port = float(np.median(rates))
    # one evaluation per call, like MultiNest's callback and like the survey's loop over Runner.loglikelihood
    Uc = U[:2000].copy()
    t0 = time.perf_counter()
    for k in range(Uc.shape[0]):
        run.loglikelihood(Uc[k])
    port_per_call = Uc.shape[0] / (time.perf_counter() - t0)
    cpu = ''
    try:
        cpu = [ln.split(':', 1)[1].strip() for ln in open('/proc/cpuinfo') if ln.startswith('model name')][0]
    except Exception:
        pass
    out = {
        'shape': 'C2: 1024 channels x NH3 (1,1)+(2,2), 2 components, get_irdc_priors(size=500), uniform unit-cube rows',
        'cpu': cpu, 'machine': platform.machine(), 'cores_visible': len(os.sched_getaffinity(0)),
        'port_evals_per_s_one_core': port,
        'port_evals_per_s_one_core_spread': [float(min(rates)), float(max(rates))],
        'port_evals_per_s_one_call_per_point': port_per_call,
        'reference_evals_per_s_one_core': REFERENCE_ONE_CORE,
        'reference_evals_per_s_eight_cores': REFERENCE_EIGHT_CORES,
        'reference_source': 'SURVEY.md section 6 [measured in the survey container: the reference\'s own Cython build, '
                            'reference flags, Python call included]',
        'ratio_to_reference': port / REFERENCE_ONE_CORE,
        'note': 'ratio > 1: the port is faster than the reference (batched C loop, no Python call per point); '
                'bench.py divides its cpu_baseline by this ratio to quote the speed-up against the reference itself',
    }
    dest = ROOT / 'profiles' / 'r03' / 'port_vs_reference.json'
    dest.parent.mkdir(parents=True, exist_ok=True)
    dest.write_text(json.dumps(out, indent=1) + '\n')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
