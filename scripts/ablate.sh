#!/bin/bash
# usage: ablate.sh <exp-mode> <mask> ...   -- the test library (-DNFA_ABLATE) with parts of the kernels switched off
# (timing only): 1 no Tb pass, 2 no line loop, 4 no rows, 8 no line set-up (masks add)
mode=$1; shift
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[2], "ablate", sys.argv[1], "| us/step", round(d["ms_per_step"]*1e3,1), "| one-lane lnl_kernel per launch of", d["config"].get("steps_per_launch"), "steps:", round(r.get("avg_launch_us") or 0,1), "us")'
for a in "$@"; do
  NFA_ENGINE_LIB=$PWD/nestfit_amd/lib/libnestfit_amd_test.so python bench.py --no-cpu-baseline --modes one --exp-mode $mode --skip-single-step --blocks 5 --ablate $a 2>/dev/null | python -c "$fmt" $a $mode
done
