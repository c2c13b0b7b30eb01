import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ov2slam_amd
from ov2slam_amd import synth
ctx = ov2slam_amd.Context(0)
for (w, h) in ((752, 480), (1241, 376)):
    a, b, flow = synth.frame_pair(w, h, seed=5, shift=(2.0, -1.0))
    rng = np.random.default_rng(1)
    for n in (50, 300, 1000, 2000):
        kps = np.stack([rng.uniform(10, w - 10, n), rng.uniform(10, h - 10, n)], 1).astype(np.float32)
        hp = (rng.uniform(size=n) < 0.7).astype(np.uint8)
        pri = np.where(hp[:, None] > 0, flow(kps) + rng.normal(0, 1, kps.shape), kps).astype(np.float32)
        trk = ov2slam_amd.VisualFrontEndTracker(ctx, w, h, use_clahe=True, fclahe_val=3.0, nbmaxkps=max(512, n))
        trk.trackFrame(a, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), None)
        trk.trackFrame(b, kps, pri, hp)
        t0 = time.perf_counter()
        for i in range(40): out, st, _ = trk.trackFrame(b if i % 2 else a, kps, pri, hp)
        ms = (time.perf_counter() - t0) / 40 * 1e3
        print("%dx%d n=%4d  %.3f ms/frame  tracked %.2f" % (w, h, n, ms, (st & 1).mean()))
        trk.close()
