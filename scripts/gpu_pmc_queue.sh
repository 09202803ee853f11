#!/bin/bash
# usage (GPU box): scripts/gpu_pmc_queue.sh <queue values ...>  -- SQ counters of the table-mode likelihood kernel, one-lane launches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${NFA_ROUND:-r05}/pmc_queue; mkdir -p $out
for k in "$@"; do
  i=0
  for p in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_BRANCH SQ_INSTS_LDS_ATOMIC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $p --output-format csv -d $out/k${k}_$i -- python bench.py --steps 10 --warmup 2 --blocks 3 --no-cpu-baseline --streams 1 --modes one --exp-mode table --skip-single-step --spectra-out off --configs off --lnl-queue $k > $out/k${k}_$i.log 2>&1 || echo "pmc pass $i of queue $k failed"
  done
done
python - "$@" <<'P'
import csv, glob, os, sys, collections
base = f"gpurun_out/{os.environ.get('NFA_ROUND', 'r05')}/pmc_queue"
for k in sys.argv[1:]:
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f'{base}/k{k}_*/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            name, grid = row['Kernel_Name'], int(row['Grid_Size'])
            # the table mode's launches of four steps: the queue form (resident grid), or one wave per unit
            if not ('lnl_kernel_queue<false' in name or ('lnl_kernel<0, false' in name and grid >= 2000000)): continue
            tot[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
    c = {a: tot[a] / n[a] for a in tot}
    if not c: print('queue', k, 'no rows'); continue
    el = c['GRBM_GUI_ACTIVE'] / 8
    print(f"queue {k:>3s}: elapsed {el:9.0f} cyc  waves {c['SQ_WAVES']:.0f}  avg waves/SIMD {c['SQ_WAVE_CYCLES'] * 4 / 1024 / el:.2f}  VALU busy {c['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / el:.3f}  "
          f"LDS idx active/CU {c['SQ_LDS_IDX_ACTIVE'] / 256 / el:.3f}  bank conflict share {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}  "
          f"VALU {c['SQ_INSTS_VALU'] / 16384:.0f} SALU {c['SQ_INSTS_SALU'] / 16384:.0f} LDS {c['SQ_INSTS_LDS'] / 16384:.0f} per eval  "
          f"wait_any {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.3f} wait_inst {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f} active {c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']:.3f}")
P
