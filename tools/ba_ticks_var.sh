#!/bin/bash
# phase clocks of k_ba_cholesky for A/B builds (tools/build_variant.sh): tools/ba_ticks_var.sh <variant> ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for v in "$@"; do echo "== $v"; OV2SLAM_HIP_LIB=$ROOT/build_var/$v/libov2slam_hip.so OV2_DEBUG=1 python - <<PY 2>&1 | tail -2
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ov2slam_amd
from ov2slam_amd import optimizer, synth
ctx = ov2slam_amd.Context(0)
pb = synth.make_ba_problem(50, 10000, 30, stereo=False, seed=42)
rp = optimizer.ResidentProblem(ctx, pb)
for _ in range(4):
    r = rp.solve()
print(r["iterations"], r["solve_ms"])
PY
done
