#!/bin/bash
# Per-kernel register / LDS / spill summary of the library's kernels (hipcc -Rpass-analysis=kernel-resource-usage: dirt_amd/build.py::
# kernel_resources), every source with the flags dirt_amd/build.py gives it.  Extra flags are appended: tools/kernel_resources.sh -DX=1
cd "$(dirname "$0")/.."
python - "$@" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from dirt_amd import build as b
try:
    res = b.kernel_resources(sys.argv[1:])
except RuntimeError as e:
    print(str(e)[-1500:]); sys.exit(1)
for name, r in res.items():
    print('%-48s vgpr=%-4s sgpr=%-4s scratch=%-4s occ=%s lds=%s' % (name, r['vgpr'], r['sgpr'], r['scratch'], r['occupancy'], r['lds']))
PY
