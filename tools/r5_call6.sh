#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== K5 parity under alias4"; DIRT_AMD_LIBRARY=$PWD/tools/_bin/alias4.so timeout 300 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -q -m gpu -x -k "K5 or c16 or channel_group or c5" --timeout=200 2>&1 | tail -4
for v in alias4 alias3; do DIRT_AMD_LIBRARY=$PWD/tools/_bin/$v.so timeout 200 python tools/quick_ab.py "K5" "0" dense 50 2>&1 | grep -v amdgpu.ids; done
timeout 200 python tools/quick_ab.py "K5" "0" dense 50 2>&1 | grep -v amdgpu.ids
DIRT_AMD_LIBRARY=$PWD/tools/_bin/taps.so timeout 200 python tools/quick_ab.py "K3 K3-2048" "0x10000" dense 2>&1 | grep -v amdgpu.ids
timeout 200 python tools/quick_ab.py "K3 K3-2048" "0x10000" dense 2>&1 | grep -v amdgpu.ids
} > gpurun_out/call6.log 2>&1
cat gpurun_out/call6.log
