"""Per-call wall-clock latency of the host-buffer (drop-in) entry points at EuRoC and KITTI sizes -- a sweep for anomalies
(an entry that is an order of magnitude slower at one size than at the other: the 2.8 ms KITTI upload was found this way)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ov2slam_amd
from ov2slam_amd import synth, stereo, optimizer
ctx = ov2slam_amd.Context(0)

def t(f, n=20):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3

for name, (w, h) in (("EuRoC 752x480", (752, 480)), ("KITTI 1241x376", (1241, 376)), ("odd 641x479", (641, 479))):
    a, b, flow = synth.frame_pair(w, h, seed=5, shift=(2.0, -1.0))
    rng = np.random.default_rng(1)
    kps = synth.grid_keypoints(w, h, 35, rng)[:300]
    pri = (flow(kps) + rng.normal(0, 1, kps.shape)).astype(np.float32)
    P0 = ov2slam_amd.Pyramid(ctx, w, h, 9, 3); P1 = ov2slam_amd.Pyramid(ctx, w, h, 9, 3)
    P0.build(a); P1.build(b); ctx.sync()
    trk = ov2slam_amd.FeatureTracker(ctx, 30, 0.01)
    fx = ov2slam_amd.FeatureExtractor(ctx, nfast_th=10, dmaxquality=0.001)
    cal = ov2slam_amd.CameraCalibration(ctx, "pinhole", 458.654, 457.296, w / 2, h / 2, D=[-0.28, 0.07, 1e-4, 2e-5])
    cl = ov2slam_amd.CLAHE(ctx, 3.0, (w // 50, h // 50))
    roi = (5, 5, w - 10, h - 10); e = np.zeros((0, 2), np.float32)
    rows = [
        ("Pyramid.build + sync", lambda: (P0.build(a), ctx.sync())),
        ("Pyramid.build_clahe + sync", lambda: (P0.build_clahe(a, 3.0, w // 50, h // 50), ctx.sync())),
        ("CLAHE.apply (host in/out)", lambda: cl.apply(a)),
        ("fbKltTracking 300 kps, 3 lvls", lambda: trk.fbKltTracking(P0, P1, 9, 3, 30., 0.5, kps, pri)),
        ("getLineMinSAD 300 pts lvl 3", lambda: trk.getLineMinSAD(P0, P1, 3, kps / 8, 7, True)),
        ("detectSingleScale (host image)", lambda: fx.detectSingleScale(a, 35, e, roi)),
        ("detectGridFAST (host image)", lambda: fx.detectGridFAST(a, 35, e)),
        ("detectSingleScalePyr", lambda: fx.detectSingleScalePyr(P0, 35, e, roi)),
        ("computeKeypoints 300 (rad-tan)", lambda: cal.computeKeypoints(kps)),
        ("stereo_matching_fused 300", lambda: stereo.stereo_matching_fused(trk, P0, P1, kps, kps, cal, rect=True)),
        ("Pyramid.download level 0", lambda: P0.download(0)),
    ]
    print("== " + name)
    for label, f in rows:
        print("   %-34s %8.3f ms" % (label, t(f)))
pb = synth.make_structure_problem(12, 2000, 5, stereo=True, seed=3)
print("== BA-side calls")
print("   %-34s %8.3f ms" % ("structureOnlyBA 2000 pts", t(lambda: ov2slam_amd.Optimizer(ctx).structureOnlyBA(pb), 10)))
pbw = synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7)
print("   %-34s %8.3f ms" % ("localBA 25 KF x 3000 x 12 stereo", t(lambda: ov2slam_amd.Optimizer(ctx).localBA(pbw), 5)))
print("   %-34s %8.3f ms" % ("looseBA same", t(lambda: ov2slam_amd.Optimizer(ctx).looseBA(pbw), 5)))
