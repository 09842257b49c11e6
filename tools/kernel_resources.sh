#!/bin/bash
# Per-kernel register / LDS / spill summary of the library's kernels (hipcc -Rpass-analysis=kernel-resource-usage),
# every source with the flags dirt_amd/build.py gives it.  Extra flags are appended: tools/kernel_resources.sh -DX=1
cd "$(dirname "$0")/.."
python - "$@" <<'PY'
import os, re, subprocess, sys
sys.path.insert(0, os.getcwd())
from dirt_amd import build as b
rows = {}
for src in b.SOURCES:
    cmd = [b.hipcc_path()] + [f for f in b.HIPCC_FLAGS if f != '-shared'] + b.PER_SOURCE_FLAGS.get(src, []) + sys.argv[1:] + \
          ['-c', os.path.join(b.CSRC, src), '-o', '/tmp/_res.o', '-Rpass-analysis=kernel-resource-usage']
    cur = None
    for l in subprocess.run(cmd, capture_output=True, text=True).stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", l)
        if not m:
            if ' error' in l: print(l.rstrip())
            continue
        t = m.group(1)
        if t.startswith('Function Name:'):
            cur = t.split(':', 1)[1].strip(); rows[cur] = {}
        elif cur and ':' in t:
            k, v = t.split(':', 1); rows[cur][k.strip()] = v.strip()
for k, r in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip().split('(')[0]
    print('%-48s vgpr=%-4s sgpr=%-4s scratch=%-4s occ=%s lds=%s' % (name, r.get('VGPRs'), r.get('TotalSGPRs'), r.get('ScratchSize [bytes/lane]'), r.get('Occupancy [waves/SIMD]'), r.get('LDS Size [bytes/block]')))
PY
