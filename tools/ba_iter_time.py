"""us per LM iteration of the resident config-4 problems (mono / stereo) and the 25-KF window -- the bench's `ba` figures alone."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ov2slam_amd
from ov2slam_amd import optimizer, synth
ctx = ov2slam_amd.Context(0)
for name, pb in (("config4 mono", synth.make_ba_problem(50, 10000, 30, stereo=False, seed=42)),
                 ("config4 stereo", synth.make_ba_problem(50, 10000, 30, stereo=True, seed=42)),
                 ("window 25x3000x12 stereo", synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7))):
    rp = optimizer.ResidentProblem(ctx, pb)
    rp.solve()
    its = ms = 0
    for _ in range(5):
        r = rp.solve(); its += r["iterations"]; ms += r["solve_ms"]
    print("%-26s %.1f us per iteration (%d iterations per solve, %.3f ms)" % (name, ms / its * 1e3, its // 5, ms / 5))
    rp.close()
