"""Synthetic inputs of SURVEY.md section 8(d): frequency axes and truth
parameters of the benchmark configurations.  Pure numpy, no model arithmetic
(synthetic *data* are made by the engine itself, see bench.py)."""
import numpy as np

CKMS = 299792.458
NU0 = {1: 23.6944955e9, 2: 23.722633335e9, 3: 23.8701296e9, 4: 24.1394169e9, 5: 24.53299e9,
       6: 25.05603e9, 7: 25.71518e9, 8: 26.51898e9, 9: 27.477943e9}

# get_test_spectra(kind=0) truth (reference: nestfit/synth_spectra.py:251-258)
TRUTH_2COMP = np.array([-1.0, 1.5, 10.0, 15.0, 4.0, 6.0, 14.5, 15.0, 0.3, 0.6, 0.0, 0.0])
TRUTH_1COMP = np.array([-1.0, 10.0, 4.0, 14.5, 0.3, 0.0])
# cold + hot 3-component model of config 4
TRUTH_3COMP = np.array([-2, .5, 3, 12, 25, 60, 5, 8, 20, 14.5, 14.8, 14.2, .3, .5, 1, .3, .3, .5])


def freq_axis(trans_id, n_chan, vhalf=30.0):
    """Ascending, uniform frequency axis nu0 (1 - v/c), v = linspace(+vh, -vh, N)."""
    v = np.linspace(vhalf, -vhalf, n_chan)
    return NU0[trans_id] * (1.0 - v / CKMS)


def param_sampler_draw(rng, vsep=(0.16, 3), trot=(3, 30), tex=(2.8, 12), ntot=(13, 16),
                       sigm=(0.15, 2), orth=(0, 0)):
    """Two-component truth drawn like ParamSampler.draw
    (reference: nestfit/synth_spectra.py:165-192)."""
    v = rng.uniform(*vsep)
    return np.concatenate([
        [0.0, v], rng.uniform(*trot, size=2), rng.uniform(*tex, size=2),
        rng.uniform(*ntot, size=2), rng.uniform(*sigm, size=2), rng.uniform(*orth, size=2)])
