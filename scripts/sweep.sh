#!/bin/bash
# usage: sweep.sh "<args A>" "<args B>" ...   -- the tree's bench with different arguments, interleaved twice
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "| Mev/s", round(d["value"]/1e6,2), "us/step", round(d["ms_per_step"]*1e3,1), "one-lane lnl", round(r.get("avg_launch_us") or 0,1), "setup", round(r.get("setup_kernel_us") or 0,1))'
for rep in 1 2; do
  for a in "$@"; do
    python bench.py --no-cpu-baseline $a 2>/dev/null | python -c "$fmt" "$a"
  done
done
