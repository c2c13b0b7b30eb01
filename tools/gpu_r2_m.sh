#!/bin/bash
# round-2 session m: kernel timeline of one LM iteration of the config-4 local BA (start / duration / gap per kernel)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/r2m; export TMPDIR=/tmp
cat > /tmp/t.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ov2slam_amd
from ov2slam_amd import optimizer, synth
ctx = ov2slam_amd.Context(0)
pb = synth.make_ba_problem(50, 10000, 30, stereo=False, seed=42)
rp = optimizer.ResidentProblem(ctx, pb)
for _ in range(4): g = rp.solve()
print(g["iterations"], g["solve_ms"])
PY
OUT=$ROOT/gpurun_out/r2m
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python /tmp/t.py > $OUT/t.log 2>&1)
tail -1 $OUT/t.log
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last solve: find the last k_ba_init
idx = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_ba_init"))
prev_end = None
out = []
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append("%-46s dur %7.1f us  gap %6.1f us" % (r["Kernel_Name"][:46], (e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3))
    prev_end = e
open(sys.argv[1].rsplit("/", 1)[0] + "/../timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:60]))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
