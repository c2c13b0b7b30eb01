"""Latency of MultiViewGeometry::ceresPnP (src/multi_view_geometry.cpp:492-586) through the C ABI: robust pass + outlier removal +
L2 pass on one pose and n fixed world points (host buffers in and out), next to the oracle on one host core."""
import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ov2slam_amd
from ov2slam_amd import synth, optimizer
ctx = ov2slam_amd.Context(0)
for n in (100, 300, 1000):
    pb = synth.make_pnp_problem(n, seed=9)
    K = pb["calib_l"]
    args = (pb["res_uv"], pb["res_xyz"], np.zeros(n), pb["poses"][0], 5, 5.9915, True, True) + tuple(K)
    mvg = ov2slam_amd.MultiViewGeometry(ctx)
    mvg.ceresPnP(*args)
    t0 = time.perf_counter()
    for _ in range(50): ok, T, out = mvg.ceresPnP(*args)
    ms = (time.perf_counter() - t0) / 50 * 1e3
    o = optimizer.default_options(ctx.lib, max_iter=5, function_tolerance=1e-3)
    g = optimizer.solve(ctx, pb, o)
    t0 = time.perf_counter()
    for _ in range(50): g = optimizer.solve(ctx, pb, o)
    ms1 = (time.perf_counter() - t0) / 50 * 1e3
    line = "n=%4d  ceresPnP (2 passes) %.3f ms   one solve %.3f ms (%d iterations, device %.3f ms)" % (n, ms, ms1, g["iterations"], g["solve_ms"])
    try:
        from oracle import oracle as O
        t0 = time.perf_counter()
        for _ in range(5): r = O.ba_solve(pb, O.ba_default_options(max_iter=5, function_tolerance=1e-3))
        line += "   oracle one solve %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3)
    except Exception as e:
        line += "   (oracle: %s)" % e
    print(line)
