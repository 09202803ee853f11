#!/bin/bash
# usage (GPU box): scripts/gpu_setup_shape.sh <mode>  -- the set-up kernel's workgroup shape (options setup_ti / setup_threads) with the
# default build and a build that holds the kernel to 128 registers (nestfit_amd/lib/ab_w4.so, -DNFA_SETUP_W4)
mode=${1:-fast}
out=gpurun_out/r05/setup_shape; mkdir -p $out
B="--no-cpu-baseline --skip-single-step --modes one --spectra-out off --configs off --exp-mode $mode --steps 20 --warmup 5 --blocks 15"
for r in 1 2; do
  for lib in libnestfit_amd ab_w4; do
    for shape in "0 0" "32 256" "64 512" "32 512" "16 256"; do
      set -- $shape
      NFA_ENGINE_LIB=nestfit_amd/lib/$lib.so python bench.py $B --setup-ti $1 --setup-threads $2 > $out/${lib}_$1_$2_$r.json 2>>$out/err.log || { tail -3 $out/err.log; exit 1; }
    done
  done
done
python - $mode <<'P'
import json,glob,sys,statistics as st
for lib in ('libnestfit_amd','ab_w4'):
    for shape in ('0_0','32_256','64_512','32_512','16_256'):
        v=[];su=[]
        for f in sorted(glob.glob(f'gpurun_out/r05/setup_shape/{lib}_{shape}_*.json')):
            d=json.loads(open(f).read().strip().splitlines()[-1]); m=d['modes'][sys.argv[1]]
            v.append(d['value']/1e6); su.append(m.get('setup_kernel_us',0))
        print(f'{sys.argv[1]} {lib:16s} ti_threads {shape:8s} value {st.median(v):7.2f} M   setup kernel {st.median(su):6.1f} us')
P
