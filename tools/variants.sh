#!/bin/bash
# Builds variants of the library under tools/_bin/ for A/B runs on the GPU box (DIRT_AMD_LIBRARY picks one).
#   tools/variants.sh name1="-DFLAG=1 ..." name2="..."      ->  tools/_bin/<name>.so   (in parallel)
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -Wno-unused-function"
SRC="dirt_amd/csrc/dirt_capi.hip dirt_amd/csrc/dirt_raster.hip dirt_amd/csrc/dirt_grad.hip dirt_amd/csrc/dirt_grad_small.hip dirt_amd/csrc/dirt_texture.hip"
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"; [ "$flags" = "$spec" ] && flags=""
  ( /opt/rocm/bin/hipcc $FL $flags $SRC -o tools/_bin/$name.so 2>&1 | grep -E "error|warning: (?!argument)" ; echo "built $name [$flags]" ) &
done
wait
