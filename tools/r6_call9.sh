#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  echo "== texture tests"; timeout 900 python -m pytest tests/test_texture.py tests/test_gpu_fullsize.py -q -m gpu -x --timeout=900 -k "texture or textured or lookup or patch" 2>&1 | tail -5
  echo "== texture bench (after)"; timeout 600 python tools/bench_texture.py gpurun_out/r6_texture_after.json 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r6_call9.log 2>&1
cat gpurun_out/r6_call9.log
