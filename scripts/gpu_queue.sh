#!/bin/bash
# usage (GPU box): scripts/gpu_queue.sh <rounds> <queue,order> ...  -- table mode, A/B of the unit queue (option lnl_queue) and of its
# order by cost class (option lnl_order) on one box; extra bench arguments in NFA_BENCH_ARGS
rounds=$1; shift
out=gpurun_out/${NFA_ROUND:-r05}/queue; mkdir -p $out
for r in $(seq $rounds); do
  for k in "$@"; do
    q=${k%,*}; o=${k#*,}
    python bench.py --no-cpu-baseline --skip-single-step --modes one --spectra-out off --configs off --exp-mode table --steps 20 --warmup 5 --blocks 15 --lnl-queue $q --lnl-order $o $NFA_BENCH_ARGS > $out/k${k}_$r.json 2>> $out/err.log || { tail -3 $out/err.log; exit 1; }
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
    print(f'lnl_queue,lnl_order {k:>4s}: value {st.median(v):7.2f} M (min {min(v):.2f} max {max(v):.2f})   one-lane lnl_kernel {st.median(ker):7.2f} us per launch')
P
