#!/usr/bin/env python3
"""A long randomised parity run of the HIP path against the CPU oracle (test infrastructure, oracle/): many random
spectra sets -- 1..4 transitions out of all nine, ragged channel counts, velocity spans from narrow to wide, 1..4
components, cold / lte, both reference prior sets -- each with a batch of unit-cube draws through
`loglikelihood_batch` (theta and lnL) and `predict_batch` (model spectra), in the three numerical modes.
Reports, per mode, the largest deviations seen and checks them against the bars of DESIGN section 5:
support (zero pattern) exact; theta 1e-9 (the placement prior's CDF comes from prefix moments); Tb 1e-6 relative on
channels above 1e-6 K (exact modes 1e-11); lnL at the oracle's own theta (the predict path) 1e-6 (exact modes 1e-11);
lnL end to end, i.e. through the prior transform with its 1e-9, 1e-6 (exact modes 1e-8).
usage: parity_soak.py [n_sets=600] [seed=1]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import nestfit_amd as na                                   # noqa: E402
from nestfit_amd.synth import freq_axis                    # noqa: E402
from oracle import nfo                                     # noqa: E402

BARS = {'table': (1e-11, 1e-11, 1e-8), 'fast': (1e-6, 1e-6, 1e-6)}     # Tb, lnL at theta, lnL end to end


def main():
    n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    worst = {m: dict(tb=0.0, lnl=0.0, lnl_theta=0.0, theta=0.0, support=0, evals=0, channels=0) for m in BARS}
    t0 = time.perf_counter()
    for case in range(n_sets):
        n_spec = int(rng.integers(1, 5))
        trans = [int(t) for t in rng.choice(np.arange(1, 10), size=n_spec, replace=False)]
        sizes = [int(rng.integers(33, 2500)) for _ in trans]
        vhalf = float(rng.choice([8.0, 30.0, 60.0, 200.0]))
        ncomp = int(rng.integers(1, 5))
        cold, lte = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        synth = bool(rng.integers(0, 2))
        ut = na.get_synth_priors(size=300) if synth else na.get_irdc_priors(size=300, vsys=float(rng.uniform(-3, 3)))
        if synth:                                            # the synth set: centre / separation prior (two components at most),
            ncomp, cold, lte = min(ncomp, 2), True, True     # tex tied to tkin (prior_constructors.py:79-141)
        spec_data = [[freq_axis(t, n, vhalf), rng.normal(0, 0.3, n), float(rng.uniform(0.05, 0.6)), t] for t, n in zip(trans, sizes)]
        cpu = nfo.AmmoniaRunner([nfo.AmmoniaSpectrum(*sd) for sd in spec_data], nfo.PriorSet(ut.lower()), ncomp=ncomp,
                                cold=cold, lte=lte)
        gpu = na.AmmoniaRunner.from_data(spec_data, ut, ncomp=ncomp, cold=cold, lte=lte)
        B = 48
        U = rng.uniform(size=(B, 6 * ncomp))
        Uc = U.copy()
        want_lnl = cpu.loglikelihood_batch(Uc)
        want_spec = []
        for b in range(8):                                   # theta rows -> model spectra
            cpu.predict(Uc[b])
            want_spec.append(np.concatenate([s.get_spec() for s in cpu.spectra]))
        want_spec = np.stack(want_spec)
        want_lnl_theta = want_lnl[:8]                        # the same rows: lnL of the oracle at its own theta
        for mode in BARS:
            gpu.set_exp_mode(mode)
            Ug = U.copy()
            got_lnl = gpu.loglikelihood_batch(Ug)
            got_spec, got_lnl_theta = gpu.predict_batch(Uc[:8])
            w = worst[mode]
            ok = np.isfinite(want_lnl)
            assert np.array_equal(ok, np.isfinite(got_lnl)), (case, mode)
            w['lnl'] = max(w['lnl'], float(np.max(np.abs(got_lnl[ok] - want_lnl[ok]) / np.abs(want_lnl[ok]), initial=0.0)))
            ok8 = ok[:8]
            w['lnl_theta'] = max(w['lnl_theta'], float(np.max(np.abs(got_lnl_theta[ok8] - want_lnl_theta[ok8]) / np.abs(want_lnl_theta[ok8]),
                                                              initial=0.0)))
            w['theta'] = max(w['theta'], float(np.nanmax(np.abs(Ug - Uc) / np.maximum(np.abs(Uc), 1e-3))))
            w['support'] += int(np.sum((got_spec == 0) != (want_spec == 0)))
            big = np.abs(want_spec) > 1e-6
            if big.any():
                w['tb'] = max(w['tb'], float(np.max(np.abs(got_spec[big] - want_spec[big]) / np.abs(want_spec[big]))))
            w['evals'] += B
            w['channels'] += int(big.sum())
        gpu.set_exp_mode(None)
        if case % 50 == 49:
            print(f'... {case + 1} sets, {time.perf_counter() - t0:.0f} s', flush=True)
    print(f'{n_sets} random spectra sets (1-4 of the nine transitions, 33-2500 channels, 1-4 components, cold / lte, irdc and synth priors)')
    failed = False
    for mode, w in worst.items():
        tb_bar, lnl_theta_bar, lnl_bar = BARS[mode]
        verdict = w['support'] == 0 and w['theta'] <= 1e-9 and w['tb'] <= tb_bar and w['lnl_theta'] <= lnl_theta_bar and w['lnl'] <= lnl_bar
        failed |= not verdict
        print(f'{mode:5s}: {w["evals"]} evaluations, {w["channels"]} channels above 1e-6 K | support mismatches {w["support"]} | '
              f'theta {w["theta"]:.2e} | Tb {w["tb"]:.2e} (bar {tb_bar:g}) | lnL at theta {w["lnl_theta"]:.2e} (bar {lnl_theta_bar:g}) | '
              f'lnL end to end {w["lnl"]:.2e} (bar {lnl_bar:g}) | '
              f'{"within the bars" if verdict else "OUTSIDE THE BARS"}')
    raise SystemExit(1 if failed else 0)


if __name__ == '__main__':
    main()
