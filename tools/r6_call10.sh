#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  echo "== forward parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x --timeout=900 2>&1 | tail -3
  timeout 300 python tests/fuzz_parity.py 60 5 2>&1 | tail -2
  echo "== A/B (cull)"; timeout 600 python tools/quick_ab.py "K3 K3-2048 K3-768" "0" dense 200 2>&1 | grep -v amdgpu.ids
  SCENES=8 timeout 600 python tools/quick_ab.py "K3" "0" dense 50 2>&1 | grep -v amdgpu.ids
  echo "== A/B (no cull)"; export DIRT_AMD_LIBRARY=$PWD/tools/_bin/nocull.so
  timeout 600 python tools/quick_ab.py "K3 K3-2048 K3-768" "0" dense 200 2>&1 | grep -v amdgpu.ids
  SCENES=8 timeout 600 python tools/quick_ab.py "K3" "0" dense 50 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r6_call10.log 2>&1
cat gpurun_out/r6_call10.log
