#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace stats of bench.py, then separate --pmc passes
# (never combined with tracing options) for instruction mix, LDS and HBM traffic.  Output: gpurun_out/prof_$1/
set -u
TAG=${1:-r04}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs --traffic off"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
RUN="python tools/prof_run.py K3 5"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_lds -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_act -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $RUN > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o pmc -- $RUN > /dev/null 2>&1
# the default bench line (in-run traffic measurement, CPU baseline), then the other configurations
python bench.py > $OUT/bench.json 2> $OUT/bench.log
for cfg in K3-256 K3-2048 K5; do
  python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline > $OUT/bench_$cfg.json 2>> $OUT/bench.log
done
python bench.py --scenes-per-gpu 8 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_K3x8.json 2>> $OUT/bench.log
for cfg in K3-3ch K3-1ch; do python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline --traffic off > $OUT/bench_$cfg.json 2>> $OUT/bench.log; done
python bench.py --scenes-per-gpu 64 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_K4x1.json 2>> $OUT/bench.log
# kernel trace of the small-frame step (the one-pixel-per-lane gradient kernel) and of K5
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_K3-256 -o trace -- python tools/prof_run.py K3-256 50 > /dev/null 2>&1
RUN5="python tools/prof_run.py K5 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_K5 -o trace -- $RUN5 > /dev/null 2>&1
# HBM traffic per launch, stamped with the kernel sources' hash (profiles/pmc_traffic.json is rewritten in place)
python tools/measure_traffic.py K3 K3-256 K5 > $OUT/traffic.log 2>&1
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
timeout 300 python tools/soak.py 10000 > $OUT/soak.log 2>&1
# round 6 extras: the texture kernels, the deferred step's kernel list, per-wave traces of the forward kernels and of the
# streaming gradient kernel (tracing build), the driver's own command three times, a fuzz sweep on this very build
timeout 300 python tools/bench_texture.py $OUT/texture.json > $OUT/texture.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_deferred_K5 -o trace -- python tools/prof_deferred.py K5 5 > /dev/null 2>&1
timeout 300 python tools/bench_deferred.py K5 10 > $OUT/bench_K5_deferred.json 2>> $OUT/bench.log
if [ -f tools/_bin/libdirt_hip_trace.so ]; then
  { DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so timeout 300 python tools/trace_forward.py K3
    DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so timeout 300 python tools/trace_stream.py K3
    DIRT_AMD_LIBRARY=$PWD/tools/_bin/libdirt_hip_trace.so timeout 300 python tools/trace_grad.py K3 | head -24; } > $OUT/traces.log 2>&1
fi
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --traffic off 2>/dev/null | grep '^{' > $OUT/bench_driver_cmd_$i.json; done
timeout 300 python tools/step_ramp2.py 20 > $OUT/step_ramp.log 2>&1
{ timeout 500 python tests/fuzz_parity.py 400 601; timeout 300 python tests/fuzz_parity.py 240 602 hostile; timeout 200 python tools/check_stream.py 150 603; } > $OUT/fuzz.log 2>&1
ls -R $OUT | head -80
