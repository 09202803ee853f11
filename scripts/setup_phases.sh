#!/bin/bash
# usage (GPU box): scripts/setup_phases.sh   -- vector instructions of setup_kernel per evaluation with its phases
# switched off one at a time (the -DNFA_ABLATE test library: 16 no priors, 32 no partition sums, 64 no derive phase)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03/setup_phases; mkdir -p $out
export NFA_ENGINE_LIB=$PWD/nestfit_amd/lib/libnestfit_amd_test.so
for a in 0 16 32 64 112; do
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $out/p$a -- python bench.py --streams 1 --modes one --no-cpu-baseline --skip-single-step --blocks 2 --steps 40 --warmup 8 --ablate $a > $out/p$a.log 2>&1 || { echo "pass $a failed"; tail -3 $out/p$a.log; }
  python profiles/pmc_to_json.py "setup_kernel" 16384 $out/p$a.json $out/p$a/*/*counter_collection.csv > /dev/null
  python -c "
import json,sys; d=json.load(open(sys.argv[1])); print('ablate', sys.argv[2], d['instructions_per_eval'])" $out/p$a.json $a
done
