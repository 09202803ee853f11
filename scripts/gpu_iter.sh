#!/bin/bash
# usage (on the GPU box, through gpurun): scripts/gpu_iter.sh <tag> [tests|notests] [bench args...]
# one iteration of kernel work: the parity tests of the likelihood path, then the bench line of the headline mode
set -o pipefail
tag=$1; shift
what=${1:-tests}; shift
out=gpurun_out/r03; mkdir -p $out
if [ "$what" = tests ]; then
  python -m pytest tests/test_gpu_parity.py tests/test_single_point.py tests/test_row_split.py tests/test_device_batches.py \
      tests/test_sibling_models.py tests/test_engine_state.py -m gpu -x -q > $out/${tag}_tests.log 2>&1
  rc=$?; tail -4 $out/${tag}_tests.log
  [ $rc -ne 0 ] && exit $rc
fi
python bench.py --no-cpu-baseline --skip-single-step "$@" > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { tail -5 $out/${tag}_bench.err; exit 1; }
python - <<P
import json
d = json.loads(open('$out/${tag}_bench.json').read().strip().splitlines()[-1])
print('$tag', 'value %.1f M' % (d['value'] / 1e6), 'ms/step %.4f' % d['ms_per_step'],
      {m: (round(v['value'] / 1e6, 1), round(v.get('lnl_kernel_us', 0), 1)) for m, v in d['modes'].items()},
      'frac', d['roofline'].get('frac'))
P
