#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  for v in sm0 sm1 sm2 sm1r; do
    export DIRT_AMD_LIBRARY=$PWD/tools/_bin/$v.so
    timeout 200 python tools/check_stream.py 15 2 2>&1 | tail -2
    timeout 600 python tools/quick_ab.py "K3 K3-2048" "0x20000" dense 200 2>&1 | grep -v amdgpu.ids
    SCENES=8 timeout 600 python tools/quick_ab.py "K3" "0x20000" dense 50 2>&1 | grep -v amdgpu.ids
  done
  export DIRT_AMD_LIBRARY=$PWD/tools/_bin/sm0.so
  timeout 600 python tools/quick_ab.py "K3 K3-2048" "0x10000" dense 200 2>&1 | grep -v amdgpu.ids
  SCENES=8 timeout 600 python tools/quick_ab.py "K3" "0x10000" dense 50 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r6_call3.log 2>&1
cat gpurun_out/r6_call3.log
