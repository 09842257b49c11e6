#!/bin/bash
# Round 5, GPU call 3: the persistent two-tile px2 kernel (prefetch of the second half-tile): parity + timing.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== pytest parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x --timeout=600 2>&1 | tail -8
echo "== fuzz 40 s"; timeout 200 python tests/fuzz_parity.py 40 12 2>&1 | tail -4
echo "== A/B"; python tools/quick_ab.py "K3" "0 0x10000" both
python tools/quick_ab.py "K3-2048 K3-3ch K3-1ch K3-768" "0 0x10000" dense 100
for v in $VARIANTS; do DIRT_AMD_LIBRARY=$PWD/tools/_bin/$v.so python tools/quick_ab.py "K3 K3-2048" "0" dense; done
echo "== trace px2 K3"; DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so python tools/trace_px2.py K3 2>&1 | grep -v amdgpu.ids
} > gpurun_out/call3.log 2>&1
tail -60 gpurun_out/call3.log
