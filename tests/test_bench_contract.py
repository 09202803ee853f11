"""bench.py as the driver runs it (short): ONE JSON line with the contract's keys, the two modes, the
roofline and -- with two ranks sharing the one GPU (`NFA_BENCH_SAME_GPU`, the rehearsal of `--gpus N`) --
the whole-job value of two stripes."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
SHORT = ['--steps', '6', '--warmup', '2', '--blocks', '3', '--side', '16', '--no-cpu-baseline']


def _run(args, env=None):
    res = subprocess.run([sys.executable, str(ROOT / 'bench.py')] + args, capture_output=True, text=True, timeout=600,
                         env={**os.environ, **(env or {})}, cwd=str(ROOT))
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_default_line_has_the_contract_keys():
    d = _run(SHORT)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'modes', 'spread'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 6 and d['warmup'] == 2 and d['unit'] == 'evals/s'
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    # the headline is the table mode: the reference's own arithmetic (fastexp.c:234-283, hyperfine.pyx:93-96)
    assert set(d['modes']) == {'table', 'fast'} and d['modes']['table']['value'] == d['value'] and d['dtype'].startswith('f64 (reference FastExp')
    r = d['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in r, key
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak']) and 0 < r['frac'] < 1
    # value = evaluations of the timed block / its time
    assert d['value'] == pytest.approx(4096 / (d['ms_per_step'] * 1e-3), rel=1e-6)
    assert d['modes']['fast']['value'] > d['value']
    # ... its figures lead `roofline`, the fast mode's follow as fast_* inside the first twenty keys (what reads the line
    # keeps that many scalars)
    keys = list(r)
    assert r['mode'] == 'table' and r['kernel'].startswith('void lnl_kernel')
    for key in ('avg_launch_us', 'pipeline_frac', 'fast_value', 'fast_frac', 'fast_ms_per_step', 'fast_avg_launch_us'):
        assert key in keys[:20], (key, keys[:24])
    assert r['fast_value'] == d['modes']['fast']['value'] and 0 < r['frac'] < r['fast_frac'] < 1
    rp = r['reference_precision']
    assert rp['mode'] == 'table' and rp['dtype'].startswith('f64') and rp['value'] == d['value']
    assert r['reference_precision_value'] == rp['value'] and r['reference_precision_frac'] == pytest.approx(r['frac'])
    assert d['config']['reference_precision_value'] == rp['value']
    # the spectra-out mode and the short C4 block of the default command
    so = d['spectra_out']
    assert so['algorithmic_bytes_per_eval'] == 16584 + 16384 and so['value'] > 0 and 0 < so['frac'] < 1
    assert r['spectra_out_value'] == so['value']
    c4 = d['configs']['C4']
    assert c4['algorithmic_bytes_per_eval'] == 49448 and c4['table']['value'] < c4['fast']['value']
    assert r['C4_fast_value'] == c4['fast']['value'] and r['C4_table_frac'] == c4['table']['roofline_frac']
    # BASELINE config 5 as specified, through the cube driver
    c5 = d['configs']['C5']
    assert 'ncomp_max 2' in c5['workload'] and c5['pixels'] == 1024 and c5['seconds'] > 0
    assert sum(c5['nbest_histogram'].values()) == 1024 and c5['nbest_histogram']['2'] > 300
    assert c5['evals_per_pixel'] > 1e5 and 0.1 < c5['mean_lnZ_err'] < 0.5 and r['C5_seconds'] == c5['seconds']


def test_two_ranks_on_one_gpu_report_the_whole_job():
    d = _run(['--gpus', '2', '--modes', 'one'] + SHORT, env={'NFA_BENCH_SAME_GPU': '1'})
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak'
    assert d['value'] == pytest.approx(2 * 4096 / (d['ms_per_step'] * 1e-3), rel=1e-6)
    assert 'i_lon % 2' in d['config']['workload'] or 'stripe' in d['config']['workload']
    r = d['spread']['rank_ms_per_step']
    assert 0 < r['fastest'] <= r['slowest'] == pytest.approx(d['ms_per_step'])


def test_gpus_flag_must_match_the_world_size():
    res = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '2'] + SHORT, capture_output=True, text=True,
                         timeout=120, env={**os.environ, 'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0'}, cwd=str(ROOT))
    assert res.returncode != 0 and 'WORLD_SIZE' in (res.stderr + res.stdout)
