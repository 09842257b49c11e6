#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  for v in base spair sfree; do
    export DIRT_AMD_LIBRARY=$PWD/tools/_bin/$v.so
    timeout 600 python tools/quick_ab.py "K3 K3-2048" "0" dense 200 2>&1 | grep -v amdgpu.ids
    SCENES=8 timeout 600 python tools/quick_ab.py "K3" "0" dense 50 2>&1 | grep -v amdgpu.ids
  done
  unset DIRT_AMD_LIBRARY
  echo "== bench.py as the driver runs it (x3)"
  for i in 1 2 3; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --traffic off 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step %.2f us  events median %.2f  eager(calib) %.2f graph %.2f  autograd %.1f  kernels %s' % (d['ms_per_step']*1e3, d['ms_per_step_events_median']*1e3, d['ms_per_step_eager']*1e3, d['ms_per_step_graph']*1e3, d['ms_per_step_autograd']*1e3, {k: round(v['avg_us'],1) for k,v in d['kernels'].items()}))"; done
} > gpurun_out/r6_call11.log 2>&1
cat gpurun_out/r6_call11.log
