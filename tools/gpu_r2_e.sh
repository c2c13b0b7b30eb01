#!/bin/bash
# round-2 GPU session E: new BA tests, KITTI profile (config 3), BA kernel profile (config 4 + variants)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2e; mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ba.py tests/test_gpu_xyz_ba.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kitti -o t -- python $ROOT/bench.py --workload kitti --steps 40 --warmup 10 --no-extras --no-cpu-baseline > $OUT/kitti_bench.json 2> $OUT/kitti.err
cat > /tmp/ba_prof.py <<'PY'
import sys, os, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, ov2slam_amd
from ov2slam_amd import optimizer, synth
ctx = ov2slam_amd.Context(0)
out = {}
for label, pb in (("config4_mono_50kf_10k_30obs", synth.make_ba_problem(50, 10000, 30, stereo=False, seed=42)),
                  ("config4_stereo", synth.make_ba_problem(50, 10000, 30, stereo=True, seed=42)),
                  ("window_25kf_3k_12obs_stereo", synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7))):
    rp = optimizer.ResidentProblem(ctx, pb); rp.solve()
    its = ms = 0
    for _ in range(5):
        r = rp.solve(); its += r["iterations"]; ms += r["solve_ms"]
    out[label] = {"iterations_per_solve": its / 5, "solve_ms": ms / 5, "us_per_iteration": ms / its * 1e3}
    rp.close()
pbx = synth.make_xyz_ba_problem(25, 3000, 10, stereo=True, seed=4)
optimizer.solve_xyz(ctx, pbx)
r = optimizer.solve_xyz(ctx, pbx)
out["xyz_25kf_3k_10obs_stereo"] = {"iterations_per_solve": r["iterations"], "solve_ms": r["solve_ms"], "us_per_iteration": r["solve_ms"] / max(1, r["iterations"]) * 1e3}
print(json.dumps(out))
PY
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ba -o t -- python /tmp/ba_prof.py > $OUT/ba_bench.json 2> $OUT/ba.err
cat $OUT/ba_bench.json
cd $ROOT
python - <<'PY'
import csv,glob,os
root=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for tag in ("kitti","ba"):
    for f in glob.glob(root+"/gpurun_out/r2e/%s/**/t_kernel_stats.csv"%tag,recursive=True):
        print(tag)
        for r in list(csv.reader(open(f)))[1:16]: print("  ", r[0][:48], r[1], r[3], r[4])
PY
tail -c 700 $OUT/kitti_bench.json
