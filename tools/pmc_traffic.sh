#!/bin/bash
# Runs on the GPU box: HBM traffic counters (separate rocprofv3 --pmc passes, no tracing) of a configuration's step.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG=${1:-K5}
OUT=gpurun_out/pmc_traffic_$CFG; rm -rf $OUT; mkdir -p $OUT
RUN="python tools/prof_run.py $CFG 3"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o pmc -- $RUN > /dev/null 2>&1
python tools/pmc_summary.py $OUT > gpurun_out/pmc_traffic_$CFG.txt 2>&1
cat gpurun_out/pmc_traffic_$CFG.txt
