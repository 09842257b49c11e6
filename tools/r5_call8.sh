#!/bin/bash
# Round 5, GPU call 8: the masked directory (setup_kernel_masked + the raster scan over face masks): parity, then A/B against the old directory.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x --timeout=300 2>&1 | tail -4
echo "== fuzz 60 s"; timeout 200 python tests/fuzz_parity.py 60 51 2>&1 | tail -3
echo "== A/B (new, then -DDIRT_NO_MASKED_DIR)"
timeout 200 python tools/quick_ab.py "K3 K3-256 K3-768 K3-2048" "0" dense 2>&1 | grep -v amdgpu.ids
DIRT_AMD_LIBRARY=$PWD/tools/_bin/nomask.so timeout 200 python tools/quick_ab.py "K3 K3-256 K3-768 K3-2048" "0" dense 2>&1 | grep -v amdgpu.ids
echo "== set-up trace"; DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so timeout 100 python tools/trace_setup.py K3 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c8_trace; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c8_trace/new -o t -- python tools/prof_run.py K3 100 > /dev/null 2>&1
DIRT_AMD_LIBRARY=$PWD/tools/_bin/nomask.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c8_trace/old -o t -- python tools/prof_run.py K3 100 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/c8_trace/new gpurun_out/c8_trace/old
} > gpurun_out/call8.log 2>&1
tail -60 gpurun_out/call8.log
