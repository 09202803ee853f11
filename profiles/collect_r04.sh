#!/bin/bash
# usage: collect_r04.sh <part>     (on the GPU box, from the repo root; parts keep each gpurun call short)
#   bench      the default bench line (headline + reference precision + spectra-out + C4) and the other shapes
#   rocprof    rocprofv3 --kernel-trace --stats: one lane fast, one lane table (the rooflines' kernel times), four lanes fast
#              only (sum of lnl_kernel against the wall time of the timed blocks), spectra-out on one lane
#   pmc        SQ counter passes of the one-lane command, fast mode -> pmc_lnl_fast.json, pmc_setup.json
#   pmc_table  the same for the table mode (LDS bank conflicts) -> pmc_lnl_table.json
#   traffic    FETCH_SIZE / WRITE_SIZE passes: fast, table, spectra-out -> pmc_traffic.json, pmc_traffic_spectra_out.json
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04; mkdir -p $out
ONE="--streams 1 --modes one --no-cpu-baseline --skip-single-step --spectra-out off --configs off --blocks 3 --steps 60 --warmup 12"     # one lane; 72 steps = 18 launches of four steps: one launch shape
case "$1" in
bench)
  python3 bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err || exit 1
  python3 bench.py --pixels-per-step B --no-cpu-baseline > $out/bench_pixel_per_row.json 2>> $out/bench_default.err || exit 1
  python3 bench.py --workload C4 --no-cpu-baseline --side 32 > $out/bench_C4.json 2>> $out/bench_default.err || exit 1
  python3 bench.py --workload C1 --no-cpu-baseline --side 32 > $out/bench_C1.json 2>> $out/bench_default.err || exit 1
  python3 bench.py --batch 16384 --steps 50 --no-cpu-baseline --modes one > $out/bench_B16384.json 2>> $out/bench_default.err || exit 1
  ;;
rocprof)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/onelane -- python3 bench.py $ONE --exp-mode fast > $out/bench_onelane.json 2> $out/onelane.err || exit 2
  cp $out/onelane/*/*kernel_stats.csv $out/onelane_kernel_stats.csv
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/onelane_table -- python3 bench.py $ONE --exp-mode table > $out/bench_onelane_table.json 2> $out/onelane_table.err || exit 2
  cp $out/onelane_table/*/*kernel_stats.csv $out/onelane_kernel_stats_table.csv
  # four lanes, the fast mode alone, the headline's own timing: what the kernels add up to against the wall time
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fourlane -- python3 bench.py --steps 20 --warmup 5 --modes one --exp-mode fast --no-cpu-baseline --skip-single-step --spectra-out off --configs off > $out/bench_fourlane_fast.json 2> $out/fourlane.err || exit 2
  cp $out/fourlane/*/*kernel_stats.csv $out/fourlane_fast_kernel_stats.csv
  python3 profiles/fourlane_summary.py $out/fourlane/*/*kernel_trace.csv $out/bench_fourlane_fast.json > $out/fourlane_fast_summary.txt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/spectra -- python3 bench.py --steps 20 --warmup 5 --streams 1 --spectra-out only --blocks 5 > $out/bench_spectra_out_onelane.json 2> $out/spectra.err || exit 2
  cp $out/spectra/*/*kernel_stats.csv $out/onelane_kernel_stats_spectra_out.csv
  ;;
pmc)
  i=0
  for p in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_BRANCH SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $p --output-format csv -d $out/pmc_$i -- python3 bench.py $ONE --exp-mode fast > $out/pmc_$i.log 2>&1 || echo "pmc pass $i failed"
  done
  python3 profiles/pmc_to_json.py "lnl_kernel<2, false" 16384 $out/pmc_lnl_fast.json $out/pmc_*/*/*counter_collection.csv > $out/pmc_lnl_fast.txt
  python3 profiles/pmc_to_json.py "setup_kernel" 16384 $out/pmc_setup.json $out/pmc_*/*/*counter_collection.csv > $out/pmc_setup.txt
  ;;
pmc_table)
  i=0
  for p in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_CVT"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $p --output-format csv -d $out/pmct_$i -- python3 bench.py $ONE --exp-mode table > $out/pmct_$i.log 2>&1 || echo "pmc_table pass $i failed"
  done
  python3 profiles/pmc_to_json.py "lnl_kernel<0, false" 16384 $out/pmc_lnl_table.json $out/pmct_*/*/*counter_collection.csv > $out/pmc_lnl_table.txt
  ;;
traffic)
  for m in fast table; do
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/traffic_${m}_$c -- python3 profiles/traffic_probe.py $m > $out/traffic_${m}_$c.log 2>&1 || exit 3
      cp $out/traffic_${m}_$c/*/*counter_collection.csv $out/traffic_${m}_$c.csv
    done
    python3 profiles/traffic_summary.py $m $out/traffic_${m}_FETCH_SIZE.csv $out/traffic_${m}_WRITE_SIZE.csv $out/pmc_traffic.json $out/pmc_traffic_spectra_out.json > $out/traffic_summary_$m.txt
  done
  ;;
esac
echo "collect_r04 $1 done"
