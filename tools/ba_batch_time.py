"""Wall clock and device time of ov2_local_ba_batch on n copies of the 25-KF local-BA window (different seeds) against n ov2_local_ba calls.
    python tools/ba_batch_time.py [n=11] [reps=20]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ov2slam_amd
from ov2slam_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 11
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = ov2slam_amd.Context(0)
pbs = [synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7 + i) for i in range(n)]
opt = ov2slam_amd.Optimizer(ctx)
for _ in range(3):
    res, nb = opt.localBA_batch(pbs, want_chi2=False)
t0 = time.perf_counter()
for _ in range(reps):
    res, nb = opt.localBA_batch(pbs, want_chi2=False)
tb = (time.perf_counter() - t0) / reps
dev_b = sum(res[0]["solve_ms"])
for pb in pbs[:2]:
    opt.localBA(pb, want_chi2=False)
t0 = time.perf_counter()
for _ in range(max(1, reps // 4)):
    singles = [opt.localBA(pb, want_chi2=False) for pb in pbs]
ts = (time.perf_counter() - t0) / max(1, reps // 4)
dev_s = sum(sum(r["solve_ms"]) for r in singles)
print("batch of %d: %.3f ms wall (python packing included), %.3f ms device (both passes); %d one-problem calls: %.3f ms wall, %.3f ms device; iterations %s vs %s"
      % (n, tb * 1e3, dev_b, n, ts * 1e3, dev_s, [r["iterations"] for r in res][:3], [r["iterations"] for r in singles][:3]))
