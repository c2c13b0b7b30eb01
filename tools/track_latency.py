"""Single-sequence per-frame latency (host wall-clock incl. PCIe and the sync) of the tracker entry points.
Run through gpurun:  python tools/track_latency.py [n_frames]"""
import sys, os, time, json, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ov2slam_amd, bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
views, kps, pri = bench.make_inputs(1, 1234)
W, H, WIN, LEVELS, NA, NF = bench.W, bench.H, bench.WIN, bench.LEVELS, bench.N_PASS_A, bench.NF
hp = np.zeros(bench.NKPS, np.uint8); hp[:NA] = 1
out = {}
def run(label, use_graph, pinned, split=False):
    ctx = ov2slam_amd.Context(0)
    trk = ov2slam_amd.VisualFrontEndTracker(ctx, W, H, use_clahe=True, fclahe_val=bench.CLAHE_CLIP, nbmaxkps=512, use_graph=use_graph)
    trk.trackFrame(views[0], kps[0, 0][:0], kps[0, 0][:0], None)
    ts = []
    for i in range(N + 20):
        f = i % NF
        if f == 0:
            trk.trackFrame(views[0], kps[0, 0][:0], kps[0, 0][:0], None)
        k = kps[f, 0]; p = np.where(hp[:, None] > 0, pri[f, 0], k)
        if pinned:
            trk.image_buffer[:, :W] = views[f + 1]; img = trk.image_buffer
        else:
            img = views[f + 1]
        t0 = time.perf_counter()
        if split:
            trk.preprocessImage(img); o, st, _ = trk.kltTracking(k, p, hp)
        else:
            o, st, _ = trk.trackFrame(img, k, p, hp)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[20:]) * 1e3
    out[label] = {"median_ms": float(np.median(ts)), "mean_ms": float(ts.mean()), "p90_ms": float(np.percentile(ts, 90)),
                  "tracked": float((st & 1).mean()), "retried": int((st & 2).astype(bool).sum()), "graph": trk.uses_graph}
    trk.close(); ctx.close()
run("track_frame_graph_pinned", True, True)
run("track_frame_graph_pageable", True, False)
run("track_frame_plain_pinned", False, True)
run("split_plain_pageable", False, False, split=True)
print(json.dumps(out, indent=1))
