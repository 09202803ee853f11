#!/bin/bash
# usage (GPU box): scripts/gpu_group8.sh <mode> [bench args]  -- steps per launch: the default build (at most four device-pointer batches
# travel as one launch) against a -DNFA_GROUP_MAX=8 build (nestfit_amd/lib/ab_g8.so) with coalesce 6 and 8
mode=${1:-table}; shift
mkdir -p gpurun_out/r05/g8
B="--no-cpu-baseline --skip-single-step --modes one --spectra-out off --configs off --exp-mode $mode --steps 20 --warmup 5 --blocks 15 $*"
for r in 1 2 3; do
  python bench.py $B > gpurun_out/r05/g8/g4_$r.json 2>>gpurun_out/r05/g8/err.log || exit 1
  NFA_ENGINE_LIB=nestfit_amd/lib/ab_g8.so NFA_GROUP_MAX=8 python bench.py $B --coalesce 8 > gpurun_out/r05/g8/g8_$r.json 2>>gpurun_out/r05/g8/err.log || exit 1
  NFA_ENGINE_LIB=nestfit_amd/lib/ab_g8.so NFA_GROUP_MAX=8 python bench.py $B --coalesce 6 > gpurun_out/r05/g8/g6_$r.json 2>>gpurun_out/r05/g8/err.log || exit 1
done
python - $mode <<'P'
import json,glob,sys,statistics as st
for k in ('g4','g6','g8'):
    v=[];ker=[];spl=[];fr=[]
    for f in sorted(glob.glob(f'gpurun_out/r05/g8/{k}_*.json')):
        d=json.loads(open(f).read().strip().splitlines()[-1]); m=d['modes'][sys.argv[1]]
        v.append(d['value']/1e6); ker.append(m.get('lnl_kernel_us',0)); spl.append(d['roofline'].get('steps_per_launch')); fr.append(d['roofline']['frac'])
    print(sys.argv[1], k, 'value %.2f M'%st.median(v), 'kernel %.1f us'%st.median(ker), 'steps/launch', spl[0], 'frac %.3f'%st.median(fr))
P
