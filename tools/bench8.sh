python bench.py --scenes-per-gpu 8 --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['config']['workload'][:30], d['config']['launch'][:12], 'step %.1f us' % (d['ms_per_step'] * 1e3), {k: round(v['avg_us'], 1) for k, v in d['kernels'].items()}, round(d['value']))
    elif 'rror' in l: print(l.rstrip())"
