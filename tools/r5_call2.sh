#!/bin/bash
# Round 5, GPU call 2: why the two-pixels-per-lane kernel is no faster -- per-wave trace + SQ counters, px2 against px4.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== trace px2 K3"; DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so python tools/trace_px2.py K3 2>&1 | grep -v amdgpu.ids
echo "== trace px4 K3"; DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so python tools/trace_grad.py K3 2>&1 | grep -v amdgpu.ids | head -16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in px2:0x8000 px4:0x10000; do
  name=${v%%:*}; fl=${v##*:}
  OUT=gpurun_out/c2_pmc_$name; rm -rf $OUT; mkdir -p $OUT
  RUN="python tools/prof_run.py K3 5 $fl"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o pmc -- $RUN > /dev/null 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_act -o pmc -- $RUN > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -o pmc -- $RUN > /dev/null 2>&1
  echo "== PMC $name"; python tools/pmc_summary.py $OUT | grep -A 12 "grad_kernel"
done
} > gpurun_out/call2.log 2>&1
tail -150 gpurun_out/call2.log
