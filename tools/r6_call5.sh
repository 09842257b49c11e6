#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so timeout 300 python tools/trace_forward.py K3 2>&1 | grep -v amdgpu.ids
  DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so timeout 300 python tools/trace_forward.py K3-2048 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r6_call5.log 2>&1
cat gpurun_out/r6_call5.log
