#!/bin/bash
# round-2 session r: per-kernel split of the batched detector (bench.py's detect_batch section under rocprofv3)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/r2r; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r2r
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/t.err)
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("mineig", "grid_select", "subpix", "fast_cells")):
        print(r["Name"][:60], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "max us", round(float(r["MaxNs"]) / 1e3, 1), "total ms", round(float(r["TotalDurationNs"]) / 1e6, 2))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
