#!/bin/bash
# Builds variants of the library under tools/_bin/ for A/B runs on the GPU box (DIRT_AMD_LIBRARY picks one; tools/ab.sh).
#   tools/variants.sh name1="-DFLAG=1 ..." name2="..."      ->  tools/_bin/<name>.so   (in parallel, the product's own flags)
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"; [ "$flags" = "$spec" ] && flags=""
  ( python -m dirt_amd.build --quiet --out tools/_bin/$name.so --flags "$flags" > /dev/null 2> tools/_bin/$name.log; grep -E " error" tools/_bin/$name.log; echo "built $name [$flags]" ) &
done
wait
