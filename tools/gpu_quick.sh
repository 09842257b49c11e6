#!/bin/bash
# GPU session: test suite (stop at first failure), per-phase traces of grad_kernel, short bench lines.  -> gpurun_out/t.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  timeout 900 python -m pytest tests -q -m gpu -x --timeout=600 2>&1 | tail -${TAIL:-25}
  if [ -f tools/_bin/libdirt_hip_trace.so ]; then
    for c in K3 K3-2048 K3-256; do DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so python tools/trace_grad.py $c 2>&1 | grep -v amdgpu.ids; done
  fi
  for cfg in ${CONFIGS:-K3 K3-256 K3-2048 K5}; do
    python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['config']['workload'][:8], 'step %.1f us' % (d['ms_per_step'] * 1e3), {k: round(v['avg_us'], 1) for k, v in d['kernels'].items()})
    elif 'rror' in l: print(l.rstrip())"
  done
} > gpurun_out/t.log 2>&1
cat gpurun_out/t.log
