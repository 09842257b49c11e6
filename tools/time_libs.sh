#!/bin/bash
# Times bench configurations under several builds of the library (tools/_bin/<name>.so).  usage: time_libs.sh "<configs>" name...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
CONFIGS=$1; shift
for v in "$@"; do
  for cfg in $CONFIGS; do
    DIRT_AMD_LIBRARY=$PWD/tools/_bin/$v.so python bench.py $BENCH_ARGS --config $cfg --steps 200 --warmup 50 --no-cpu-baseline --launch eager 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-8s' % '$v', d['config']['workload'][:8], 'step %.1f us' % (d['ms_per_step'] * 1e3), {k: round(v.get('avg_us_corrected', v['avg_us']), 1) for k, v in d['kernels'].items()})
    elif 'rror' in l: print(l.rstrip())"
  done
done 2>&1 | tee -a gpurun_out/time_libs.log
