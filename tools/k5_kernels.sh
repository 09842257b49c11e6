#!/bin/bash
# rocprofv3 kernel averages of the K5 step under several builds.  usage: k5_kernels.sh name...
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf /tmp/k5p; DIRT_AMD_LIBRARY=$PWD/tools/_bin/$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5p -o t -- python tools/prof_run.py K5 4 > /dev/null 2>&1
  echo "== $v"; python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/k5p/**/t_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dirt' in r['Name']: print('   %-60s calls %3s avg %8.1f us' % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
