#!/bin/bash
# Kernel trace of the lock-step driver (tools/lockstep_driver.cpp) on the case files tools/lockstep_sweep.py left under $TMPDIR/ov2_lockstep_cases:
# per-kernel durations and the gaps between the dispatches of a frame step.  Run through gpurun AFTER lockstep_sweep.py in the same call.
#   tools/lockstep_prof.sh <tag> [nob|case]      (nob: sequences without localBA windows: the SLAM + mapper streams alone)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r5}; KIND=${2:-nob}; OUT=$ROOT/gpurun_out/prof_lockstep_$TAG; mkdir -p $OUT; export TMPDIR=${TMPDIR:-/tmp}
D=$TMPDIR/ov2_lockstep_cases
CASES=$(ls $D/${KIND}*.bin | tr '\n' ',' | sed 's/,$//')
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- $D/lockstep_driver $CASES newest 0 4 0 1 3 > $OUT/driver.json 2> $OUT/err
tail -1 $OUT/driver.json
python $ROOT/tools/step_gaps.py $OUT | head -24 | tee $OUT/gaps.txt
