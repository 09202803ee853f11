#!/bin/bash
# usage: ablate.sh <mask> ...   -- the test library (-DNFA_ABLATE) with parts of the kernels switched off (timing only)
fmt='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ablate", sys.argv[1], "| us/step", round(d["ms_per_step"]*1e3,1), "lnl alone", round(r["alone_launch_us"],1), "setup alone", round(r["alone_setup_us"],1))'
for a in "$@"; do
  NFA_ENGINE_LIB=$PWD/nestfit_amd/lib/libnestfit_amd_test.so python bench.py --no-cpu-baseline --ablate $a 2>/dev/null | python -c "$fmt" $a
done
