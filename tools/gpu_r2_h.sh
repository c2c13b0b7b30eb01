#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2h; mkdir -p $OUT
cd $ROOT
cat > /tmp/ba_t.py <<'PY'
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ov2slam_amd
from ov2slam_amd import optimizer, synth
ctx = ov2slam_amd.Context(0)
out = {}
for label, pb in (("config4_mono", synth.make_ba_problem(50, 10000, 30, stereo=False, seed=42)),):
    rp = optimizer.ResidentProblem(ctx, pb); rp.solve()
    its = ms = 0
    for _ in range(8):
        r = rp.solve(); its += r["iterations"]; ms += r["solve_ms"]
    out[label] = {"us_per_it": round(ms / its * 1e3, 1)}
    rp.close()
print(os.environ.get("OV2SLAM_HIP_LIB", "default g4 s1"), json.dumps(out))
PY
OV2_BA_DEBUG=1 python /tmp/ba_t.py 2> $OUT/e0 | tail -1; grep ticks $OUT/e0 | tail -1
for v in g8_s1 g4_s0 g8_s0; do OV2_BA_DEBUG=1 OV2SLAM_HIP_LIB=$ROOT/ko/lib_$v.so python /tmp/ba_t.py 2> $OUT/e_$v | tail -1; grep ticks $OUT/e_$v | tail -1; done
timeout 600 python -m pytest tests/test_gpu_ba.py tests/test_gpu_xyz_ba.py -x -q -m gpu 2>&1 | tail -2
