#!/bin/bash
# Runs on the GPU box: memory-path counter passes (vector L1 = TCP, texture addresser = TA, L2 = TCC) of a step,
# rocprofv3 --pmc only, summarised per kernel.   tools/pmc_mem.sh [config] [steps]      (DIRT_AMD_LIBRARY picks a build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG=${1:-K5}; OUT=gpurun_out/pmc_mem_$CFG; rm -rf $OUT; mkdir -p $OUT
RUN="python tools/prof_run.py $CFG ${2:-3}"
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum --output-format csv -d $OUT/p1 -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum --output-format csv -d $OUT/p2 -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum --output-format csv -d $OUT/p3 -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/p4 -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_BUSY_avr --output-format csv -d $OUT/p5 -o pmc -- $RUN > /dev/null 2>&1
python tools/pmc_summary.py $OUT > gpurun_out/pmc_mem_$CFG.txt 2>&1
cat gpurun_out/pmc_mem_$CFG.txt
