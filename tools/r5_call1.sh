#!/bin/bash
# Round 5, GPU call 1: parity of the two-pixels-per-lane gradient kernel + A/B timing against the four-pixel kernel.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== pytest parity (all gradient shapes)"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu --timeout=600 2>&1 | tail -30
echo "== fuzz 60 s"; timeout 200 python tests/fuzz_parity.py 60 11 2>&1 | tail -8
echo "== A/B product build"; python tools/quick_ab.py "K3" "0 0x10000" both
python tools/quick_ab.py "K3-2048 K3-3ch K3-1ch K3-768 K3-512" "0 0x10000" dense 100
echo "== A/B px2 at 6 waves per SIMD"; DIRT_AMD_LIBRARY=$PWD/tools/_bin/px2w6.so python tools/quick_ab.py "K3 K3-2048" "0" dense
echo "== rocprofv3 kernel stats, K3 dense, px2 then px4"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c1_trace; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c1_trace/px2 -o t -- python tools/prof_run.py K3 100 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c1_trace/px4 -o t -- python tools/prof_run.py K3 100 0x10000 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/c1_trace/px2 gpurun_out/c1_trace/px4
} > gpurun_out/call1.log 2>&1
tail -120 gpurun_out/call1.log
