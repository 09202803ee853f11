#!/bin/bash
# usage: collect_r05.sh <part>     (on the GPU box, from the repo root; parts keep each gpurun call short)
#   bench      the default bench line (table-mode headline + fast mode + spectra-out + C4 + C5) and the other shapes
#   rocprof    rocprofv3 --kernel-trace --stats: one lane table (the headline's roofline kernel time), one lane fast, four lanes
#              table and fast (sum of the likelihood launches against the wall time of the timed blocks), spectra-out on one lane
#   pmc        SQ counter passes of the one-lane command, table mode -> pmc_lnl_table.json (lnl_kernel_queue), pmc_setup.json
#   pmc_fast   the same for the fast mode -> pmc_lnl_fast.json
#   traffic    FETCH_SIZE / WRITE_SIZE passes: table, fast, spectra-out -> pmc_traffic.json, pmc_traffic_spectra_out.json
#   c5         the sampler: bench.py --workload C5 (as specified, both modes), C5r4 (rounds 2-4's cube), kernel stats of its two-component run
#   ring       scripts/measure_ring.py native
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05; mkdir -p $out
ONE="--streams 1 --modes one --no-cpu-baseline --skip-single-step --spectra-out off --configs off --blocks 3 --steps 64 --warmup 16"     # one lane; blocks of 64 steps = launches of eight steps: one launch shape
case "$1" in
bench)
  python3 bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err || exit 1
  python3 bench.py --pixels-per-step B --no-cpu-baseline --configs off --spectra-out off > $out/bench_pixel_per_row.json 2>> $out/bench_default.err || exit 1
  python3 bench.py --workload C4 --no-cpu-baseline --side 32 > $out/bench_C4.json 2>> $out/bench_default.err || exit 1
  python3 bench.py --workload C1 --no-cpu-baseline --side 32 > $out/bench_C1.json 2>> $out/bench_default.err || exit 1
  python3 bench.py --batch 16384 --steps 50 --no-cpu-baseline --modes one --configs off --spectra-out off > $out/bench_B16384.json 2>> $out/bench_default.err || exit 1
  ;;
rocprof)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/onelane_table -- python3 bench.py $ONE --exp-mode table > $out/bench_onelane_table.json 2> $out/onelane_table.err || exit 2
  cp $out/onelane_table/*/*kernel_stats.csv $out/onelane_kernel_stats_table.csv
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/onelane -- python3 bench.py $ONE --exp-mode fast > $out/bench_onelane.json 2> $out/onelane.err || exit 2
  cp $out/onelane/*/*kernel_stats.csv $out/onelane_kernel_stats.csv
  for m in table fast; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fourlane_$m -- python3 bench.py --steps 20 --warmup 5 --blocks 15 --modes one --exp-mode $m --no-cpu-baseline --skip-single-step --spectra-out off --configs off > $out/bench_fourlane_$m.json 2> $out/fourlane_$m.err || exit 2
    cp $out/fourlane_$m/*/*kernel_stats.csv $out/fourlane_${m}_kernel_stats.csv
    python3 profiles/fourlane_summary.py $out/fourlane_$m/*/*kernel_trace.csv $out/bench_fourlane_$m.json > $out/fourlane_${m}_summary.txt
    rm -rf $out/fourlane_$m
  done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/spectra -- python3 bench.py --steps 24 --warmup 8 --streams 1 --spectra-out only --exp-mode fast --blocks 5 > $out/bench_spectra_out_onelane.json 2> $out/spectra.err || exit 2
  cp $out/spectra/*/*kernel_stats.csv $out/onelane_kernel_stats_spectra_out.csv
  rm -rf $out/onelane $out/onelane_table $out/spectra
  ;;
pmc|pmc_fast)
  mode=table; kern="lnl_kernel_queue<false"; tag=table
  [ "$1" = pmc_fast ] && { mode=fast; kern="lnl_kernel<2, false"; tag=fast; }
  i=0
  for p in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_BRANCH SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $p --output-format csv -d $out/pmc_${tag}_$i -- python3 bench.py $ONE --exp-mode $mode > $out/pmc_${tag}_$i.log 2>&1 || echo "pmc pass $i failed"
  done
  python3 profiles/pmc_to_json.py "$kern" 32768 $out/pmc_lnl_$tag.json $out/pmc_${tag}_*/*/*counter_collection.csv > $out/pmc_lnl_$tag.txt
  [ "$1" = pmc ] && python3 profiles/pmc_to_json.py "setup_kernel" 32768 $out/pmc_setup.json $out/pmc_${tag}_*/*/*counter_collection.csv > $out/pmc_setup.txt
  rm -rf $out/pmc_${tag}_[0-9]
  ;;
traffic)
  for m in table fast; do
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/traffic_${m}_$c -- python3 profiles/traffic_probe.py $m > $out/traffic_${m}_$c.log 2>&1 || exit 3
      cp $out/traffic_${m}_$c/*/*counter_collection.csv $out/traffic_${m}_$c.csv
      rm -rf $out/traffic_${m}_$c
    done
    python3 profiles/traffic_summary.py $m $out/traffic_${m}_FETCH_SIZE.csv $out/traffic_${m}_WRITE_SIZE.csv $out/pmc_traffic.json $out/pmc_traffic_spectra_out.json > $out/traffic_summary_$m.txt
  done
  ;;
c5)
  python3 bench.py --workload C5 > $out/bench_C5.json 2> $out/bench_C5.err || exit 4
  python3 bench.py --workload C5r4 --exp-mode fast > $out/bench_C5r4.json 2>> $out/bench_C5.err || exit 4
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5r4_stats -- python3 bench.py --workload C5r4 --exp-mode fast --c5-ncomp 2 > $out/c5r4_under_rocprof.json 2> $out/c5r4_rocprof.err || exit 4
  cp $out/c5r4_stats/*/*kernel_stats.csv $out/sampler_kernel_stats.csv; rm -rf $out/c5r4_stats
  ;;
ring)
  python3 scripts/measure_ring.py native 1:dev 4:dev 8:dev 14:dev 16:dev 14:1:1 > $out/ring.txt 2> $out/ring.err || exit 5
  ;;
esac
echo "collect_r05 $1 done"
