#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  for v in base v2w5; do
    export DIRT_AMD_LIBRARY=$PWD/tools/_bin/$v.so
    timeout 600 python tools/quick_ab.py "K3 K3-2048" "0" dense 200 2>&1 | grep -v amdgpu.ids
    SCENES=8 timeout 600 python tools/quick_ab.py "K3" "0" dense 50 2>&1 | grep -v amdgpu.ids
  done
} > gpurun_out/r6_call13.log 2>&1
cat gpurun_out/r6_call13.log
