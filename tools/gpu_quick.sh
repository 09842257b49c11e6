#!/bin/bash
# GPU session: reduction micro-test, parity tests, per-phase trace of grad_kernel, bench lines.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  timeout 60 tools/_bin/reduce_test
  ( time timeout 900 python -m pytest tests -q -m gpu -x --timeout=600 2>&1 | tail -${TAIL:-15} ) 2>&1
  for c in ${TRACE_CONFIGS:-K3}; do
    DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so python tools/trace_grad.py $c 2>&1 | grep -v amdgpu.ids | head -${TRACE_HEAD:-14}
    [ -n "$TRACE_RASTER" ] && DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so python tools/trace_raster.py $c 2>&1 | grep -v amdgpu.ids
  done
  for cfg in ${CONFIGS:-K3 K3-256 K3-2048 K5}; do
    python bench.py --config $cfg --steps 200 --warmup 50 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['config']['workload'][:8], 'step %.1f us' % (d['ms_per_step'] * 1e3), {k: round(v.get('avg_us_corrected', v['avg_us']), 1) for k, v in d['kernels'].items()})
    elif 'rror' in l: print(l.rstrip())"
  done
} > gpurun_out/t.log 2>&1
cat gpurun_out/t.log
