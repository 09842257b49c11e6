#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  for v in nopf pf nopf pf; do
    export DIRT_AMD_LIBRARY=$PWD/tools/_bin/$v.so
    timeout 600 python tools/quick_ab.py "K3 K3-2048" "0" dense 200 2>&1 | grep -v amdgpu.ids
    SCENES=8 timeout 600 python tools/quick_ab.py "K3" "0" dense 50 2>&1 | grep -v amdgpu.ids
  done
  export DIRT_AMD_LIBRARY=$PWD/tools/_bin/pf.so
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x --timeout=900 2>&1 | tail -2
} > gpurun_out/r6_call14.log 2>&1
cat gpurun_out/r6_call14.log
