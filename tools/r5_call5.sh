#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for v in $VARIANTS; do DIRT_AMD_LIBRARY=$PWD/tools/_bin/$v.so python tools/quick_ab.py "${CONFIGS:-K3}" "${FLAGS:-0}" ${MODE:-dense} 2>&1 | grep -v amdgpu.ids; done
python tools/quick_ab.py "${CONFIGS:-K3}" "${FLAGS:-0}" ${MODE:-dense} 2>&1 | grep -v amdgpu.ids
} > gpurun_out/call5.log 2>&1
cat gpurun_out/call5.log
