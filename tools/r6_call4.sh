#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu -x --timeout=900 2>&1 | tail -15
  echo "== A/B"; timeout 600 python tools/quick_ab.py "K3 K3-2048 K3-256 K3-3ch" "0" dense 200 2>&1 | grep -v amdgpu.ids
  SCENES=8 timeout 600 python tools/quick_ab.py "K3" "0" dense 50 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r6_call4.log 2>&1
cat gpurun_out/r6_call4.log
