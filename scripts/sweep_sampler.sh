#!/bin/bash
# usage (GPU box): scripts/sweep_sampler.sh <flag> "<values ...>" [more bench flags]  -- config 5 with one knob of the device sampler swept
fmt='import json,sys; d=json.loads(sys.stdin.read()); a,b=d["one_component"],d["two_components"]; print(sys.argv[1], "| 1 comp %.2f s %.0f k evals/pixel lnZ err %.3f | 2 comp %.2f s %.0f k evals/pixel lnZ err %.3f" % (a["seconds"], a["evals_per_pixel"]/1e3, a["mean_lnZ_err"], b["seconds"], b["evals_per_pixel"]/1e3, b["mean_lnZ_err"]))'
flag=$1; vals=$2; shift 2
for v in $vals; do
  python bench.py --workload C5 $flag $v "$@" 2>/dev/null | python -c "$fmt" "$flag $v $*"
done
