#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
  echo "== trace stream K3"; DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so timeout 300 python tools/trace_stream.py K3 2>&1 | grep -v amdgpu.ids
  echo "== trace px4 K3"; DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so timeout 300 python tools/trace_grad.py K3 2>&1 | grep -v amdgpu.ids | head -16
  echo "== pmc"; cd /tmp && export TMPDIR=/tmp
  for pm in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
    rm -rf /tmp/pmc_o; timeout 300 rocprofv3 --pmc $pm -d /tmp/pmc_o -o x --output-format csv -- python $GRAFT_REPO_ROOT/tools/quick_ab.py K3 "0x10000 0x20000" dense 3 > /dev/null 2>&1
    python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_o 2>&1 | grep -A12 "grad_kernel" | head -40
  done
} > gpurun_out/r6_call2.log 2>&1
cat gpurun_out/r6_call2.log
