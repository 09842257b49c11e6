#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{ timeout 300 python tools/step_ramp2.py 20 2>&1 | grep -v amdgpu.ids; timeout 300 python tools/step_ramp2.py 200 2>&1 | grep -v amdgpu.ids | tail -3; } > gpurun_out/r6_call12.log 2>&1
cat gpurun_out/r6_call12.log
