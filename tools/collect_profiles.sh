#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace stats of bench.py, then separate --pmc passes
# (never combined with tracing options) for instruction mix, LDS and HBM traffic.  Output: gpurun_out/prof_$1/
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 50 --warmup 10 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
RUN="python tools/prof_run.py K3 5"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_lds -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_act -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o pmc -- $RUN > /dev/null 2>&1
python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.log
for cfg in K3-256 K3-2048 K5; do
  python bench.py --config $cfg --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_$cfg.json 2>> $OUT/bench.log
done
python bench.py --scenes-per-gpu 8 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_K3x8.json 2>> $OUT/bench.log
# the K5 gradient pass is the traffic-bound one: its own counters
RUN5="python tools/prof_run.py K5 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_K5 -o trace -- $RUN5 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_K5 -o pmc -- $RUN5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_K5 -o pmc -- $RUN5 > /dev/null 2>&1
ls -R $OUT | head -40
