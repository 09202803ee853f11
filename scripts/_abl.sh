out=gpurun_out/r05/setup_abl; mkdir -p $out
for mode in table; do
for a in 0 16 32 64 48 96 80 112; do
  NFA_ENGINE_LIB=nestfit_amd/lib/ab_abl.so python bench.py --no-cpu-baseline --skip-single-step --modes one --spectra-out off --configs off --exp-mode $mode --steps 16 --warmup 8 --blocks 3 --ablate $a > $out/${mode}_$a.json 2>>$out/err.log || { tail -3 $out/err.log; exit 1; }
done; done
python - <<'P'
import json
for mode in ('table',):
    for a in (0,16,32,64,48,96,80,112):
        d=json.loads(open(f'gpurun_out/r05/setup_abl/{mode}_{a}.json').read().strip().splitlines()[-1]); m=d['modes'][mode]
        print(mode, 'ablate', a, 'setup kernel %.1f us'%m.get('setup_kernel_us',0), 'value %.1f M'%(d['value']/1e6))
P
