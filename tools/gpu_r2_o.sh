#!/bin/bash
# round-2 session o: kernel timeline of the detector calls (single image, pyramid-resident and host-image forms)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/r2o; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r2o
python tools/detect_latency.py > $OUT/latency.txt 2>&1; cat $OUT/latency.txt
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python $ROOT/tools/detect_latency.py > $OUT/t.log 2>&1)
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-130 "$f" | head -12
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
