#!/bin/bash
# Per-kernel register / LDS / spill summary of the library's kernels (hipcc -Rpass-analysis=kernel-resource-usage).
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -Wall -Wno-unused-function "$@" \
  dirt_amd/csrc/dirt_capi.hip dirt_amd/csrc/dirt_raster.hip dirt_amd/csrc/dirt_grad.hip dirt_amd/csrc/dirt_grad_small.hip dirt_amd/csrc/dirt_texture.hip -o /tmp/_res.so \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re,subprocess
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r"remark:\s+(.*?)\s*\[-Rpass",l)
    if not m:
        if "error" in l or "warning" in l: print(l.rstrip())
        continue
    t=m.group(1)
    if t.startswith("Function Name:"):
        cur=t.split(":",1)[1].strip(); rows[cur]={}
    elif cur and ":" in t:
        k,v=t.split(":",1); rows[cur][k.strip()]=v.strip()
for k,r in rows.items():
    name=subprocess.run(["c++filt",k],capture_output=True,text=True).stdout.strip().split("(")[0]
    print("%-48s vgpr=%-4s sgpr=%-4s scratch=%-4s occ=%s lds=%s"%(name,r.get("VGPRs"),r.get("TotalSGPRs"),r.get("ScratchSize [bytes/lane]"),r.get("Occupancy [waves/SIMD]"),r.get("LDS Size [bytes/block]")))
'
