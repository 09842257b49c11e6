#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
R=$PWD
{
  echo "== new tests"; timeout 1500 python -m pytest tests -q -m gpu -x --timeout=900 -k "nccl_group or fastest_kernel_shape or texture" -s 2>&1 | grep -v "^$" | tail -25
  echo "== texture bench (baseline)"; timeout 600 python tools/bench_texture.py gpurun_out/r6_texture_before.json 2>&1 | grep -v amdgpu.ids
  echo "== deferred"; timeout 600 python tools/bench_deferred.py K5 10 2>&1 | grep -v amdgpu.ids
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pd && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pd -o x --output-format csv -- python $R/tools/prof_deferred.py K5 5 > /tmp/pd.log 2>&1; tail -2 /tmp/pd.log
  python $R/tools/pmc_summary.py /tmp/pd | head -40
  cp $(find /tmp/pd -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r6_deferred_K5_kernel_stats_before.csv
} > gpurun_out/r6_call8.log 2>&1
cat gpurun_out/r6_call8.log
