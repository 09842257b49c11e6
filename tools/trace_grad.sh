#!/bin/bash
# Builds the tracing variant of the library next to the normal one and runs tools/trace_grad.py with it.
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -Wno-unused-function -DDIRT_TRACE \
  dirt_amd/csrc/dirt_capi.hip dirt_amd/csrc/dirt_raster.hip dirt_amd/csrc/dirt_grad.hip dirt_amd/csrc/dirt_texture.hip -o tools/_bin/libdirt_hip_trace.so
