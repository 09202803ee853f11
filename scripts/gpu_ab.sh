#!/bin/bash
# usage (GPU box): scripts/gpu_ab.sh <rounds> <bench args ...> -- <libA> <libB> ...
# alternates the engine builds (NFA_ENGINE_LIB) over <rounds> rounds of the same bench command and prints, per build,
# the median rate and the median one-lane lnl_kernel time: box and clock drift hit all builds alike
rounds=$1; shift
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
out=gpurun_out/${NFA_ROUND:-r05}/ab; mkdir -p $out
for r in $(seq $rounds); do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    NFA_ENGINE_LIB=$lib python bench.py --no-cpu-baseline --skip-single-step --modes one --spectra-out off --configs off "${args[@]}" > $out/${tag}_$r.json 2>> $out/err.log || { tail -3 $out/err.log; exit 1; }
  done
done
python - "$@" <<'P'
import json, sys, glob, os, statistics as st
for lib in sys.argv[1:]:
    tag = lib.split('/')[-1][:-3]
    v, k, s = [], [], []
    for f in sorted(glob.glob(f"gpurun_out/{os.environ.get('NFA_ROUND', 'r05')}/ab/{tag}_[0-9].json")):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        m = d['modes'][d['config']['exp_mode']]
        v.append(d['value'] / 1e6); k.append(m.get('lnl_kernel_us', 0)); s.append(m.get('setup_kernel_us', 0))
    print(f'{tag:24s} value {st.median(v):7.2f} M (min {min(v):.2f} max {max(v):.2f})   lnl_kernel {st.median(k):7.2f} us (min {min(k):.2f})   setup {st.median(s):.2f} us')
P
