#!/bin/bash
# usage: ab.sh <libA> <libB> [bench args]   -- interleaved A/B of two engine builds on one box
a=$1; b=$2; shift 2
for i in 1 2 3; do
  for l in "$a" "$b"; do
    NFA_ENGINE_LIB=$PWD/$l python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l', round(d['value']/1e6,2), round(d['ms_per_step']*1e3,1), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3))"
  done
done
