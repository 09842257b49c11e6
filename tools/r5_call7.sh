#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
DIRT_AMD_LIBRARY=$PWD/tools/_bin/px2occ8.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "px2" --timeout=200 2>&1 | tail -3
DIRT_AMD_LIBRARY=$PWD/tools/_bin/px2occ8.so timeout 200 python tools/quick_ab.py "K3-3ch K3-1ch" "0x8000" dense 2>&1 | grep -v amdgpu.ids
timeout 200 python tools/quick_ab.py "K3-3ch K3-1ch" "0x8000 0x10000" dense 2>&1 | grep -v amdgpu.ids
echo "== trace occ8 3ch"; python -m dirt_amd.build --quiet --out /tmp/trace8.so --flags "-DDIRT_TRACE -DDIRT_PX2_XVS=36 -DDIRT_PX2_WAVES=8" > /dev/null 2>&1
} > gpurun_out/call7.log 2>&1
cat gpurun_out/call7.log
