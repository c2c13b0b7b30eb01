import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ov2slam_amd, bench
views, kps, pri = bench.make_inputs(1, 1234)
ctx = ov2slam_amd.Context(0)
fx = ov2slam_amd.FeatureExtractor(ctx, dmaxquality=0.001)
roi = (5, 5, bench.W - 10, bench.H - 10)
def t(f, n=30):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
e = np.zeros((0, 2), np.float32)
print("detectSingleScale (no current kps)   %.3f ms, %d pts" % (t(lambda: fx.detectSingleScale(views[0], bench.CELL, e, roi)), len(fx.detectSingleScale(views[0], bench.CELL, e, roi))))
print("detectSingleScale (150 current kps)  %.3f ms" % t(lambda: fx.detectSingleScale(views[0], bench.CELL, kps[0, 0][:150], roi)))
print("detectSingleScale no subpix          %.3f ms" % t(lambda: fx.detectSingleScale(views[0], bench.CELL, e, roi, subpix=False)))
fg = ov2slam_amd.FeatureExtractor(ctx, nfast_th=10)
print("detectGridFAST                       %.3f ms" % t(lambda: fg.detectGridFAST(views[0], bench.CELL, e)))
