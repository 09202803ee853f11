"""Timeline of one table-mode launch of the metric's shape through the unit queue (lnl_kernel_queue): when the waves
start and stop, how many are at work over the launch, how long a unit takes by its place in the order.
Test library only (nfa_test_queue_trace).  usage: python scripts/queue_timeline.py [rows]"""
import ctypes as C
import os
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault('NFA_ENGINE_LIB', str(ROOT / 'nestfit_amd' / 'lib' / 'libnestfit_amd_test.so'))
import numpy as np
import nestfit_amd as na
from nestfit_amd import _ffi
from nestfit_amd.synth import freq_axis

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
lib = _ffi.load()
na.set_exp_mode('table')
rng = np.random.default_rng(5)
n = 1024
spec_data = [[freq_axis(t, n), rng.normal(0, 0.2, n), 0.2, t] for t in (1, 2)]
run = na.AmmoniaRunner.from_data(spec_data, na.get_irdc_priors(size=500, vsys=0.0), ncomp=2)
U = np.random.default_rng(7).uniform(size=(B, 12))
run.loglikelihood_batch(U.copy())
assert lib.nfa_test_queue_trace(1) == 0
run.loglikelihood_batch(U.copy())
buf = np.zeros(8192 * 8 * 4, dtype=np.uint64)
assert lib.nfa_test_queue_trace_read(buf.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
lib.nfa_test_queue_trace(0)
rec = buf.reshape(8192, 8, 4).astype(np.int64)
used = rec[:, :, 1] > 0
t0 = rec[:, :, 0][used].min()
start = (rec[:, :, 0] - t0) * 0.01
end = (rec[:, :, 1] - t0) * 0.01                      # microseconds
dur = (end - start)[used]
print(f'{B} rows x 2 spectra = {2 * B} units on {used.any(1).sum()} waves; units recorded {used.sum()}')
print(f'launch (first start to last end) {end[used].max():.1f} us; unit duration mean {dur.mean():.1f} us, sd {dur.std():.1f}, min {dur.min():.1f}, max {dur.max():.1f}')
first = np.where(used.any(1), start[:, 0], np.nan)
print(f'waves start {np.nanmin(first):.1f} .. {np.nanpercentile(first, 50):.1f} (median) .. {np.nanmax(first):.1f} us')
last = np.where(used, end, 0).max(1)
lw = last[used.any(1)]
print('waves leave: ' + ', '.join(f'p{q} {np.percentile(lw, q):.1f}' for q in (1, 10, 50, 90, 99, 100)) + ' us')
T = end[used].max()
edges = np.linspace(0, T, 21)
s_, e_ = start[used], end[used]
print('waves at work (of 8192) in twenty slices of the launch:')
print('  ' + ' '.join(f'{int(np.sum(np.clip(np.minimum(e_, b) - np.maximum(s_, a), 0, None)) / (b - a)):5d}' for a, b in zip(edges[:-1], edges[1:])))
pos = rec[:, :, 3][used]
for a, b in ((0, 8192), (8192, 16384), (16384, 24576), (24576, 1 << 30)):
    m = (pos >= a) & (pos < b)
    if m.any():
        print(f'positions {a:6d}..{min(b, 2 * B):6d}: {m.sum():6d} units, duration mean {dur[m].mean():6.1f} us, start mean {s_[m].mean():6.1f} us')
# who is slow: the first unit's duration by the wave's place in its workgroup, and by the workgroup's number
w_in_wg = np.arange(8192) % 16
d0 = np.where(used[:, 0], end[:, 0] - start[:, 0], np.nan)
print('first unit, mean duration by wave of the workgroup: ' + ' '.join(f'{np.nanmean(d0[w_in_wg == k]):5.0f}' for k in range(16)))
wg = np.arange(8192) // 16
print('first unit, mean duration by workgroup number (eighths of the grid): ' + ' '.join(f'{np.nanmean(d0[(wg >= a) & (wg < a + 64)]):5.0f}' for a in range(0, 512, 64)))
n_units = used.sum(1)
print('units per wave by wave of the workgroup: ' + ' '.join(f'{n_units[w_in_wg == k].mean():4.1f}' for k in range(16)))
