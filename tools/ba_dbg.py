import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["OV2_BA_DEBUG"] = "1"
import ov2slam_amd
from ov2slam_amd import synth, optimizer
ctx = ov2slam_amd.Context(0)
pb = synth.make_ba_problem(50, 10000, 30, stereo=False, seed=42)
rp = optimizer.ResidentProblem(ctx, pb)
for _ in range(3):
    r = rp.solve(); print("solve_ms", r["solve_ms"], r["iterations"])
