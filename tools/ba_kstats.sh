#!/bin/bash
# Per-kernel durations of six config-4 local-BA solves (rocprofv3 --kernel-trace --stats); run through gpurun.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/ksba; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/ba_run.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ov2slam_amd
from ov2slam_amd import optimizer, synth
ctx = ov2slam_amd.Context(0)
pb = synth.make_ba_problem(50, 10000, 30, stereo=False, seed=42)
rp = optimizer.ResidentProblem(ctx, pb)
for _ in range(6):
    r = rp.solve()
print(r["iterations"], r["solve_ms"])
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python /tmp/ba_run.py > $OUT/b.txt 2> $OUT/err
cat $OUT/b.txt
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/ksba/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.reader(open(f)))[1:16]: print(r[0][:40], r[1], r[2], r[3])
PY
