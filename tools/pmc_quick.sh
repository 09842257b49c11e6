#!/bin/bash
# Runs on the GPU box: SQ / LDS counter passes of the K3 step (rocprofv3 --pmc only), summarised per kernel.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_quick; rm -rf $OUT; mkdir -p $OUT
RUN="python tools/prof_run.py ${1:-K3} 5"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_act -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -o pmc -- $RUN > /dev/null 2>&1
python tools/pmc_summary.py $OUT > gpurun_out/pmc_quick.txt 2>&1
cat gpurun_out/pmc_quick.txt
