#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of an arbitrary command, top rows printed.  Usage: kstat_cmd.sh <tag> <command...>
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/ks_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- "$@" > $OUT/out.txt 2> $OUT/err.txt
tail -5 $OUT/out.txt
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print("%-60s calls %6s avg %10.1f us  min %9.1f  max %9.1f  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
