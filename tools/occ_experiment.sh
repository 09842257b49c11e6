for d in 0 12000 24000; do
  echo "dyn lds $d"
  DIRT_TRACE_DYN_LDS=$d DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so python bench.py --config K3 --steps 100 --warmup 20 --no-cpu-baseline --launch eager 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['config']['workload'][:8], 'step %.1f us' % (d['ms_per_step'] * 1e3), {k: round(v['avg_us'], 1) for k, v in d['kernels'].items()})
    elif 'rror' in l: print(l.rstrip())"
done
