#!/bin/bash
# BA GPU tests + the in-kernel phase clocks of k_ba_cholesky (OV2_DEBUG=1) + solve time of config 4; run through gpurun.
python -m pytest tests/test_gpu_ba.py -x -q 2>&1 | tail -1; OV2_DEBUG=1 python /dev/stdin <<PY 2>&1 | tail -2
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
