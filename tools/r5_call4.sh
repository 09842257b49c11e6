#!/bin/bash
# Round 5, GPU call 4: full GPU suite (graphed step, re-entrant state outputs, tight tolerances), the default bench line, N = 2 dry run.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== reduce_test"; timeout 60 tools/_bin/reduce_test
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -q -m gpu --timeout=600 2>&1 | tail -25
echo "== bench (driver's flags)"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/call4_bench.json 2> gpurun_out/call4_bench.err; tail -3 gpurun_out/call4_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/call4_bench.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_events_median','ms_per_step_autograd','ms_per_step_autograd_engine_on_calling_thread','ms_per_step_autograd_graphed','ms_per_step_eager','ms_per_step_graph')})
print(d['config']); print(d['other_configs']); print({k:(v['avg_us'],v['avg_us_corrected']) for k,v in d['kernels'].items()}); print(d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_all_kernels'])
PY
echo "== N=2 dry run on one GPU (gloo)"; DIRT_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --traffic off 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print({k: d[k] for k in ('value', 'n_gpus', 'ranks_seen', 'ms_per_step')}, d['other_configs'], d['scaling_reference'])"
} > gpurun_out/call4.log 2>&1
tail -70 gpurun_out/call4.log
