#!/bin/bash
# usage: ab2.sh <reps> [bench args]  -- interleaved A/B on one box: ab_old/ (previous build, its own bench.py) vs the tree
reps=$1; shift
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "Mev/s", round(d["value"]/1e6,2), "us/step", round(d["ms_per_step"]*1e3,1), "one-lane lnl", round(r.get("avg_launch_us") or 0,1))'
for i in $(seq $reps); do
  (cd ab_old && python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "$fmt" old)
  python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "$fmt" new
done
