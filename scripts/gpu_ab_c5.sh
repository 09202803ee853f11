#!/bin/bash
# usage (GPU box): scripts/gpu_ab_c5.sh <rounds> <bench args ...> -- <libA> <libB> ...   -- the sampler's kernels A/B: bench.py --workload C5r4
# (rounds 2-4's cube, precision speed) with each engine build in turn
rounds=$1; shift
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
out=gpurun_out/${NFA_ROUND:-r05}/ab_c5; mkdir -p $out
for r in $(seq $rounds); do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    NFA_ENGINE_LIB=$lib python bench.py --workload C5r4 --exp-mode fast "${args[@]}" > $out/${tag}_$r.json 2>> $out/err.log || { tail -3 $out/err.log; exit 1; }
  done
done
python - "$@" <<'P'
import json, sys, glob, os, statistics as st
for lib in sys.argv[1:]:
    tag = lib.split('/')[-1][:-3]
    one, two, ev = [], [], []
    for f in sorted(glob.glob(f"gpurun_out/{os.environ.get('NFA_ROUND', 'r05')}/ab_c5/{tag}_[0-9].json")):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        one.append(d['one_component']['seconds']); two.append(d['two_components']['seconds']); ev.append(d['two_components']['evals_per_pixel'])
    print(f'{tag:16s} one component {st.median(one):6.3f} s   two components {st.median(two):6.3f} s (min {min(two):.3f})   {st.median(ev) / 1e3:6.1f} k evaluations per pixel')
P
