"""Where the wall-clock of the two-pass localBA protocol goes (host buffers in and out): packing, ov2_ba_solve wall vs device time."""
import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ov2slam_amd
from ov2slam_amd import synth, optimizer
ctx = ov2slam_amd.Context(0)
for name, pb in (("config4 stereo 50x10000x30", synth.make_ba_problem(50, 10000, 30, stereo=True, seed=42)),
                 ("window 25x3000x12 stereo", synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7))):
    opt = optimizer.Optimizer(ctx)
    opt.localBA(pb, want_chi2=False)
    t0 = time.perf_counter(); r1 = opt.localBA(pb, want_chi2=False); wall1 = time.perf_counter() - t0
    print("%s: ov2_local_ba (one call, resident) wall %.2f ms, device %.2f + %.2f ms" % (name, wall1 * 1e3, r1["solve_ms"][0], r1["solve_ms"][1]))
    opt.localBA_two_calls(pb)
    t0 = time.perf_counter(); r = opt.localBA_two_calls(pb); wall = time.perf_counter() - t0
    o = optimizer.default_options(ctx.lib, max_iter=5, function_tolerance=1e-3)
    t0 = time.perf_counter(); P, keep = optimizer.pack_problem(pb); t_pack = time.perf_counter() - t0
    t0 = time.perf_counter(); g = optimizer.solve(ctx, pb, o); t_solve = time.perf_counter() - t0
    rp = optimizer.ResidentProblem(ctx, pb)
    t0 = time.perf_counter(); rp2 = optimizer.ResidentProblem(ctx, pb); t_create = time.perf_counter() - t0
    t0 = time.perf_counter(); g2 = rp.solve(o); t_res = time.perf_counter() - t0
    print("%s: n_res %d | two-call localBA wall %.2f ms (device %.2f + %.2f) | pack %.2f ms | one ov2_ba_solve wall %.2f ms (device %.2f) | ov2_ba_create %.2f ms | resident solve wall %.2f ms"
          % (name, pb["n_res"], wall * 1e3, r["pass1"]["solve_ms"], r["pass2"]["solve_ms"] if r["l2_done"] else 0, t_pack * 1e3, t_solve * 1e3, g["solve_ms"], t_create * 1e3, t_res * 1e3))
