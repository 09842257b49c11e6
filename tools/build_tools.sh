#!/bin/bash
# Builds what the GPU-side tools need under tools/_bin/ (hipcc cross-compiles without a GPU; tools/_bin travels with gpurun):
# the tracing variant of the library (-DDIRT_TRACE: tools/trace_grad.py, tools/trace_raster.py) and the micro-benchmarks.
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -Wno-unused-function"
python -m dirt_amd.build --quiet --out tools/_bin/libdirt_hip_trace.so --flags "-DDIRT_TRACE" &
for t in reduce_test atomic_bench lds_atomic_bench; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -munsafe-fp-atomics -Wno-unused-value -Wno-unused-result tools/$t.hip -o tools/_bin/$t 2>&1 | grep -E " error" &
done
wait
ls tools/_bin
