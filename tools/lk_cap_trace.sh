#!/bin/bash
# per-dispatch durations of the two passes of the cap-and-defer LK kernel (tools/lk_micro.py workload)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/lkcap -o t -- python $ROOT/tools/lk_micro.py ${1:-2048} > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("$ROOT/gpurun_out/lkcap/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_fb_klt3" in r["Kernel_Name"] or "k_lk3_plan" in r["Kernel_Name"]]
for r in rows[:14]:
    print(r["Kernel_Name"][:28], r.get("Grid_Size_X", r.get("Grid_Size")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
rm -rf $ROOT/gpurun_out/lkcap
