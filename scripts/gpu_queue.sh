#!/bin/bash
# usage (GPU box): scripts/gpu_queue.sh <rounds> <lnl_queue value> ...  -- table mode, A/B of the unit queue (option lnl_queue:
# 0 one unit per wave, 1 the queue kernel) on one box; extra bench arguments in NFA_BENCH_ARGS
rounds=$1; shift
out=gpurun_out/${NFA_ROUND:-r05}/queue; mkdir -p $out
for r in $(seq $rounds); do
  for q in "$@"; do
    python bench.py --no-cpu-baseline --skip-single-step --modes one --spectra-out off --configs off --exp-mode table --steps 20 --warmup 5 --blocks 15 --lnl-queue $q $NFA_BENCH_ARGS > $out/k${q}_$r.json 2>> $out/err.log || { tail -3 $out/err.log; exit 1; }
  done
done
python - "$@" <<'P'
import json, sys, glob, os, statistics as st
for k in sys.argv[1:]:
    v, ker = [], []
    for f in sorted(glob.glob(f"gpurun_out/{os.environ.get('NFA_ROUND', 'r05')}/queue/k{k}_[0-9].json")):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        m = d['modes'][d['config']['exp_mode']]
        v.append(d['value'] / 1e6); ker.append(m.get('lnl_kernel_us', 0))
    print(f'lnl_queue {k:>2s}: value {st.median(v):7.2f} M (min {min(v):.2f} max {max(v):.2f})   one-lane lnl_kernel {st.median(ker):7.2f} us per launch')
P
