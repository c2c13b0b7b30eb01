#!/bin/bash
# Per-kernel durations of resident local-BA solves (rocprofv3 --kernel-trace --stats); run through gpurun.
#   tools/ba_kstats.sh [config4_mono|config4_stereo|window] [out-dir-tag]
CFG=${1:-config4_mono}; TAG=${2:-ksba_$CFG}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/ba_run_$CFG.py <<PY
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ov2slam_amd
from ov2slam_amd import optimizer, synth
ctx = ov2slam_amd.Context(0)
cfg = "$CFG"
pb = {"config4_mono": lambda: synth.make_ba_problem(50, 10000, 30, stereo=False, seed=42),
      "config4_stereo": lambda: synth.make_ba_problem(50, 10000, 30, stereo=True, seed=42),
      "window": lambda: synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7)}[cfg]()
rp = optimizer.ResidentProblem(ctx, pb)
its = 0; ms = 0.0
for i in range(6):
    r = rp.solve()
    if i: its += r["iterations"]; ms += r["solve_ms"]
print(cfg, "iterations per solve", r["iterations"], "solve_ms", r["solve_ms"], "us per LM iteration %.1f" % (ms / its * 1e3))
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python /tmp/ba_run_$CFG.py > $OUT/b.txt 2> $OUT/err
cat $OUT/b.txt
python - <<PY
import csv,glob,os
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.reader(open(f)))[1:18]: print(r[0][:44], r[1], r[2], r[3])
PY
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
