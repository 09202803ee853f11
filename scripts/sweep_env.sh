#!/bin/bash
# usage: sweep_env.sh "<ENV=.. args>" ...  -- like sweep.sh; a leading VAR=value in each item is exported for that run
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], "| Mev/s", round(d["value"]/1e6,2), "us/step", round(d["ms_per_step"]*1e3,1), "one-lane lnl", round(r.get("avg_launch_us") or 0,1), "setup", round(r.get("setup_kernel_us") or 0,1))'
for rep in 1 2; do
  for a in "$@"; do
    env ${a%% *} python bench.py --no-cpu-baseline ${a#* } 2>/dev/null | python -c "$fmt" "$a"
  done
done
