#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   pass 1: --kernel-trace --stats          -> per-kernel durations
#   pass 2: --pmc FETCH_SIZE                -> HBM read traffic per dispatch   (own pass, TCC slots)
#   pass 3: --pmc WRITE_SIZE                -> HBM write traffic per dispatch
#   pass 4: --pmc SQ_* wave / stall counters
# Outputs land in gpurun_out/prof_<tag>/ ; tools/summarize_profile.py condenses them into profiles/.
set -u
TAG=${1:-r1}
ARGS=${2:-"--steps 60 --warmup 10 --no-extras --no-cpu-baseline"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $ROOT/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o fetch -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/fetch.err
timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o write -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/write.err
timeout 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT -o sq -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/sq.err
cd $ROOT
ls $OUT
python tools/summarize_profile.py $OUT
# the per-dispatch traces are tens of MB (gpurun merges at most 64 MiB back): the stats, counters and the summary are what is kept
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*.db" -delete

