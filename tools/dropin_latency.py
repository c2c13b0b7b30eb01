import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ov2slam_amd, bench
views, kps, pri = bench.make_inputs(1, 1234)
W,H,WIN,LEVELS = bench.W, bench.H, bench.WIN, bench.LEVELS
ctx1 = ov2slam_amd.Context(0)
trk = ov2slam_amd.FeatureTracker(ctx1, 30, 0.01)
P0 = ov2slam_amd.Pyramid(ctx1, W, H, WIN, LEVELS).build(views[0]); ctx1.sync()
P1 = ov2slam_amd.Pyramid(ctx1, W, H, WIN, LEVELS)
cl = ov2slam_amd.CLAHE(ctx1, bench.CLAHE_CLIP, bench.CLAHE_TILES)
NA = bench.N_PASS_A
def t(f, n=100):
    f(); t0=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t0)/n*1e3
eq = cl.apply(views[1])
print("clahe_h       %.3f ms" % t(lambda: cl.apply(views[1])))
print("pyr build_h   %.3f ms" % t(lambda: (P1.build(eq), ctx1.sync())))
print("preprocess (clahe+pyr fused, host image) %.3f ms" % t(lambda: (P1.build_clahe(views[1], bench.CLAHE_CLIP, bench.CLAHE_TILES[0], bench.CLAHE_TILES[1]), ctx1.sync())))
print("fbklt A       %.3f ms" % t(lambda: trk.fbKltTracking(P0, P1, WIN, 1, 30., 0.5, kps[0,0][:NA], pri[0,0][:NA])))
print("fbklt B       %.3f ms" % t(lambda: trk.fbKltTracking(P0, P1, WIN, LEVELS, 30., 0.5, kps[0,0][NA:], pri[0,0][NA:])))
