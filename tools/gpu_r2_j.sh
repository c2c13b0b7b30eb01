#!/bin/bash
# round-2 session j: timing of the large-problem BA path (BADev::big) against the LDS-resident path and the oracle
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/r2j
cat > /tmp/t.py <<'PY'
import sys, os, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, ov2slam_amd
from ov2slam_amd import optimizer, synth
from oracle import oracle
ctx = ov2slam_amd.Context(0)
out = []
for n_kf, n_lm, obs, big in ((50, 10000, 30, 0), (50, 10000, 30, 1), (90, 10000, 30, 0), (90, 10000, 30, 1), (150, 20000, 30, None), (300, 30000, 20, None), (340, 30000, 20, None)):
    if big is None: os.environ.pop("OV2_BA_BIG", None)
    else: os.environ["OV2_BA_BIG"] = str(big)
    pb = synth.make_ba_problem(n_kf, n_lm, obs, stereo=True, seed=1)
    o = optimizer.default_options(ctx.lib, max_iter=5)
    rp = optimizer.ResidentProblem(ctx, pb)
    g = rp.solve(o); ts = []
    for _ in range(3):
        g = rp.solve(o); ts.append(g["solve_ms"] * 1e-3)
    rp.close()
    row = dict(n_kf=n_kf, n_lm=n_lm, obs=obs, forced=big, iters=int(g["iterations"]), ms=min(ts) * 1e3, ms_per_iter=min(ts) * 1e3 / max(1, int(g["iterations"])),
               cost0=float(g["initial_cost"]), cost=float(g["final_cost"]))
    if n_kf in (150, 300):
        t0 = time.perf_counter(); r = oracle.ba_solve(pb, oracle.ba_default_options(max_iter=5)); row["oracle_s"] = time.perf_counter() - t0
        row["pose_maxdiff"] = float(np.abs(np.asarray(g["poses"]) - np.asarray(r["poses"])).max()); row["cost_rel"] = abs(g["final_cost"] - r["final_cost"]) / r["final_cost"]
    print(row, flush=True); out.append(row)
json.dump(out, open("gpurun_out/r2j/ba_big_timing.json", "w"), indent=1)
PY
python /tmp/t.py 2>&1 | grep -v Warning | tail -12
