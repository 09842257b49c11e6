#!/bin/bash
# Quick bench lines on the GPU box: tools/bq.sh "<bench args>" ["<bench args>" ...]  -> step time and per-kernel times
cd "$(dirname "$0")/.."
for a in "$@"; do
  python bench.py --steps 200 --warmup 50 --no-cpu-baseline --traffic off $a 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-40s step %.1f us (eager %.1f graph %.1f)' % (sys.argv[1], d['ms_per_step'] * 1e3, d['ms_per_step_eager'] * 1e3, d['ms_per_step_graph'] * 1e3), {k: round(v.get('avg_us_corrected', v['avg_us']), 1) for k, v in d['kernels'].items()}, 'pair %.1f' % d['roofline']['event_pair_us'])
" "$a"
done
