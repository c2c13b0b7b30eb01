#!/bin/bash
# round-2 session k: per-kernel profile of the large-problem BA path
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/r2k; export TMPDIR=/tmp
cat > /tmp/t.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ov2slam_amd
from ov2slam_amd import optimizer, synth
ctx = ov2slam_amd.Context(0)
n_kf, n_lm, obs = [int(x) for x in sys.argv[1:4]]
pb = synth.make_ba_problem(n_kf, n_lm, obs, stereo=True, seed=1)
rp = optimizer.ResidentProblem(ctx, pb)
o = optimizer.default_options(ctx.lib, max_iter=5)
for _ in range(3): g = rp.solve(o)
print(g["iterations"], g["solve_ms"])
PY
OUT=$ROOT/gpurun_out/r2k
(cd /tmp && OV2_BA_BIG=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p50 -o p50 -- python /tmp/t.py 50 10000 30 > $OUT/p50.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p300 -o p300 -- python /tmp/t.py 300 30000 20 > $OUT/p300.log 2>&1)
for n in p50 p300; do tail -2 $OUT/$n.log; f=$(find $OUT/$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-150; done
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete
