#!/bin/bash
# usage (GPU box): scripts/gpu_pmc_ablate.sh <mode> <ablate masks ...>  -- instruction counts of the likelihood kernel with parts
# switched off (test library; 1 no Tb pass, 2 no line loop, 4 no rows, 8 no line set-up): where the instructions are
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mode=$1; shift
out=gpurun_out/${NFA_ROUND:-r05}/pmc_ablate; mkdir -p $out
for a in "$@"; do
  NFA_ENGINE_LIB=$PWD/nestfit_amd/lib/libnestfit_amd_test.so timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $out/a$a -- python bench.py --steps 8 --warmup 2 --blocks 2 --no-cpu-baseline --streams 1 --modes one --exp-mode $mode --skip-single-step --spectra-out off --configs off --ablate $a > $out/a$a.log 2>&1 || echo "ablate $a failed"
done
python - "$@" <<'P'
import csv, glob, os, sys, collections
base = f"gpurun_out/{os.environ.get('NFA_ROUND', 'r05')}/pmc_ablate"
for a in sys.argv[1:]:
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f'{base}/a{a}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if 'lnl_kernel' not in row['Kernel_Name'] or int(row['Grid_Size']) < 500000: continue
            tot[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
    c = {k: tot[k] / n[k] for k in tot}
    if not c: print('ablate', a, 'no rows'); continue
    print(f"ablate {a:>2s}: per evaluation VALU {c['SQ_INSTS_VALU'] / 16384:7.0f}  SALU {c['SQ_INSTS_SALU'] / 16384:6.0f}  LDS {c['SQ_INSTS_LDS'] / 16384:6.0f}  VMEM {c['SQ_INSTS_VMEM'] / 16384:5.0f}  SMEM {c['SQ_INSTS_SMEM'] / 16384:5.0f}  branch {c['SQ_INSTS_BRANCH'] / 16384:6.0f}   (waves {c['SQ_WAVES']:.0f})")
P
