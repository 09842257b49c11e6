#!/bin/bash
# GPU box: quick parity check + bench lines for library variants built by tools/variants.sh.
#   tools/ab.sh "<configs>" name...      (CHECK=1: also run the K3 parity test under each variant; TRACE=1: per-wave trace needs a -DDIRT_TRACE build)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
CONFIGS=$1; shift
for v in "$@"; do
  export DIRT_AMD_LIBRARY=$PWD/tools/_bin/$v.so
  if [ -n "$CHECK" ]; then
    timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -q -x -m gpu -k "${CHECK_K:-K3 or hostile or c5 or square or shapes}" 2>&1 | tail -3
  fi
  for cfg in $CONFIGS; do
    python bench.py $BENCH_ARGS --config $cfg --steps ${STEPS:-200} --warmup 50 --no-cpu-baseline --traffic off --launch eager 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-14s' % '$v', d['config']['workload'][:8], 'step %.1f us' % (d['ms_per_step'] * 1e3), {k: round(v.get('avg_us_corrected', v['avg_us']), 1) for k, v in d['kernels'].items()})
    elif 'rror' in l: print(l.rstrip())"
  done
done 2>&1 | tee -a gpurun_out/ab.log
