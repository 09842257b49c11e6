#!/bin/bash
# round 6, GPU session 1: streaming gradient kernel -- parity, then A/B timing against the 4-pixel kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  echo "== check_stream"; timeout 300 python tools/check_stream.py 90 1 2>&1 | tail -25
  echo "== K3 parity (pinned)"; timeout 600 python -m pytest tests/test_gpu_configs.py -q -x -m gpu -k "baseline_config and (K3 or 2048)" 2>&1 | tail -5
  echo "== A/B"; timeout 600 python tools/quick_ab.py "K3 K3-2048" "0x10000 0x20000" dense 200 2>&1 | grep -v amdgpu.ids
  SCENES=8 timeout 600 python tools/quick_ab.py "K3" "0x10000 0x20000" dense 50 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r6_call1.log 2>&1
cat gpurun_out/r6_call1.log
