#!/bin/bash
# usage: collect_r03.sh <part>     (on the GPU box, from the repo root; parts keep each gpurun call short)
#   bench    default bench line + C4 / C1 / one-evaluation-per-pixel lines
#   rocprof  rocprofv3 --kernel-trace --stats of the one-lane command (the roofline's kernel time) and of the default command
#   pmc      SQ counter passes of the one-lane command -> pmc_lnl_fast.json
#   pmc_table  LDS counters of the table-mode kernel -> pmc_lnl_table.json
#   traffic  FETCH_SIZE / WRITE_SIZE passes -> pmc_traffic.json
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03; mkdir -p $out
ONE="--streams 1 --modes one --no-cpu-baseline --skip-single-step --blocks 3 --steps 60 --warmup 12"     # one lane; 72 steps = 18 launches of four steps: one launch shape
case "$1" in
bench)
  python bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
  python bench.py --pixels-per-step B --no-cpu-baseline > $out/bench_pixel_per_row.json 2>> $out/bench_default.err || exit 1
  python bench.py --workload C4 --no-cpu-baseline --side 32 > $out/bench_C4.json 2>> $out/bench_default.err || exit 1
  python bench.py --workload C1 --no-cpu-baseline --side 32 > $out/bench_C1.json 2>> $out/bench_default.err || exit 1
  python bench.py --batch 16384 --steps 50 --no-cpu-baseline --modes one > $out/bench_B16384.json 2>> $out/bench_default.err || exit 1
  python bench.py --workload C5 > $out/bench_C5.json 2>> $out/bench_default.err || exit 1
  ;;
rocprof)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/onelane -- python bench.py $ONE > $out/bench_onelane.json 2> $out/onelane.err || exit 2
  cp $out/onelane/*/*kernel_stats.csv $out/onelane_kernel_stats.csv
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/default -- python bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/default.err || exit 2
  cp $out/default/*/*kernel_stats.csv $out/default_kernel_stats.csv
  ;;
pmc)
  i=0
  for p in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_BRANCH SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $p --output-format csv -d $out/pmc_$i -- python bench.py $ONE > $out/pmc_$i.log 2>&1 || echo "pmc pass $i failed"
  done
  python profiles/pmc_to_json.py "lnl_kernel<2, false" 16384 $out/pmc_lnl_fast.json $out/pmc_*/*/*counter_collection.csv > $out/pmc_lnl_fast.txt
  python profiles/pmc_to_json.py "setup_kernel" 16384 $out/pmc_setup.json $out/pmc_*/*/*counter_collection.csv > $out/pmc_setup.txt
  ;;
pmc_table)
  # table mode: is the likelihood kernel bound by its LDS gathers (three product-table reads per line x row step)?
  i=0
  for p in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_CVT"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $p --output-format csv -d $out/pmct_$i -- python bench.py $ONE --exp-mode table > $out/pmct_$i.log 2>&1 || echo "pmc_table pass $i failed"
  done
  python profiles/pmc_to_json.py "lnl_kernel<0, false" 16384 $out/pmc_lnl_table.json $out/pmct_*/*/*counter_collection.csv > $out/pmc_lnl_table.txt
  ;;
traffic)
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/traffic_$c -- python profiles/traffic_probe.py fast > $out/traffic_$c.log 2>&1 || exit 3
    cp $out/traffic_$c/*/*counter_collection.csv $out/traffic_fast_$c.csv
  done
  python profiles/traffic_summary.py fast $out/traffic_fast_FETCH_SIZE.csv $out/traffic_fast_WRITE_SIZE.csv $out/pmc_traffic.json > $out/traffic_summary.txt
  ;;
esac
echo "collect_r03 $1 done"
