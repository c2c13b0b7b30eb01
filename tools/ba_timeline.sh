#!/bin/bash
# Timeline of one resident local-BA solve: kernel-by-kernel start / duration / gap to the previous kernel (rocprofv3 --kernel-trace).
#   tools/ba_timeline.sh [config4_mono|config4_stereo|window] [det: OV2_OPT_BA_DETERMINISTIC]
CFG=${1:-config4_mono}; DET=${2:-0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/batl_$CFG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/ba_tl.py <<PY
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ov2slam_amd
from ov2slam_amd import optimizer, synth
ctx = ov2slam_amd.Context(0)
if "$DET" == "det": ctx.set_option(ov2slam_amd._lib.OV2_OPT_BA_DETERMINISTIC, 1)
pb = {"config4_mono": lambda: synth.make_ba_problem(50, 10000, 30, stereo=False, seed=42),
      "config4_stereo": lambda: synth.make_ba_problem(50, 10000, 30, stereo=True, seed=42),
      "window": lambda: synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7)}["$CFG"]()
rp = optimizer.ResidentProblem(ctx, pb)
for i in range(4):
    r = rp.solve()
print("$CFG", r["iterations"], r["solve_ms"])
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python /tmp/ba_tl.py > $OUT/b.txt 2> $OUT/err
cat $OUT/b.txt
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last solve: from the last k_ba_init on
idx = max(i for i, r in enumerate(rows) if "k_ba_init" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"]); prev_end = None
lines = []
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    lines.append("%9.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, r["Kernel_Name"][:50]))
    prev_end = e
open("$OUT/timeline.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:70]))
PY
find $OUT -name "*.db" -delete
