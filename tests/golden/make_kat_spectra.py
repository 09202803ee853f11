#!/usr/bin/env python3
"""Per-channel model spectra of the eight known-answer cases of tests/golden/survey_kat.json (max, sum,
support count and lnL recorded from the compiled reference), as the CPU oracle computes them -- written only after the
oracle has met every recorded summary of the case to 1e-12.  The vectors freeze the oracle's channel-by-channel
output in the repository (a later change of the oracle cannot drift unnoticed) and give the device tests a
per-channel target whose summaries are the reference's own numbers.

    python tests/golden/make_kat_spectra.py        ->  tests/golden/kat_spectra.npz
"""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))


def main():
    from nestfit_amd.synth import freq_axis
    from oracle import nfo
    kat = json.loads((HERE / 'survey_kat.json').read_text())
    out = {}
    for k, c in enumerate(kat['spectra']):
        x = freq_axis(c['trans_id'], c['n_chan'], c['vhalf'])
        s = nfo.AmmoniaSpectrum(x, np.zeros(c['n_chan']), c['noise'], c['trans_id'])
        nfo.amm_predict(s, np.array(c['params'], dtype=float))
        p = s.get_spec().copy()
        assert int((p != 0).sum()) == c['nnz'] and p[-1] == 0.0
        assert abs(p.max() / c['max'] - 1) < 1e-12 and abs(s.loglikelihood / c['lnL'] - 1) < 1e-12
        if 'sum' in c:
            assert abs(p.sum() / c['sum'] - 1) < 1e-12
        out[f'pred_{k}'] = p
        out[f'tau_{k}'] = np.array(s.tarr).copy()                # optical depth of the LAST component (hyperfine.pyx:66-96)
    np.savez_compressed(HERE / 'kat_spectra.npz', **out)
    print(f'wrote {HERE / "kat_spectra.npz"}: {len(kat["spectra"])} cases')


if __name__ == '__main__':
    main()
