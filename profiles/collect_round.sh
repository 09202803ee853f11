#!/bin/bash
# usage: collect_round.sh <tag>      (on the GPU box, from the repo root)
# Default bench line, rocprofv3 kernel stats of the same command, and the HBM-traffic passes.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
python bench.py > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats -- python bench.py --no-cpu-baseline > gpurun_out/$tag/bench_under_rocprof.json 2> gpurun_out/$tag/rocprof.err || exit 2
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/$tag/traffic_$c -- python profiles/traffic_probe.py fast > gpurun_out/$tag/traffic_$c.log 2>&1 || exit 3
done
python profiles/traffic_summary.py fast gpurun_out/$tag/traffic_FETCH_SIZE/*/*counter_collection.csv gpurun_out/$tag/traffic_WRITE_SIZE/*/*counter_collection.csv gpurun_out/$tag/pmc_traffic.json > gpurun_out/$tag/traffic_summary.txt
cat gpurun_out/$tag/bench_default.json
