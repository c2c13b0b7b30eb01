#!/bin/bash
# Round-4 evidence pass: rocprofv3 stats + PMC passes of the bench step (tools/profile.sh), the KITTI step's kernel stats, a fuzz campaign.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r4f; mkdir -p $O
bash tools/profile.sh r4 > $O/profile.log 2>&1; tail -5 $O/profile.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kitti_ks -o t -- python $ROOT/bench.py --workload kitti --steps 40 --warmup 10 --no-extras --no-cpu-baseline > $O/kitti_ks_bench.json 2> $O/kitti_ks.err
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete
cd $ROOT
timeout 900 python tools/fuzz_parity.py ${1:-3000} 2024 > $O/fuzz.log 2>&1; tail -4 $O/fuzz.log
