#!/usr/bin/env python3
"""Pre-processing of S resident frames (ov2_pyr_build_clahe_d: CLAHE + 4-level pyramid), HIP-event time per call, the bench's 64 image contents.
Usage: pre_time.py [S] [reps] [euroc | kitti]   (A/B builds through OV2SLAM_HIP_LIB)"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ov2slam_amd
import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
if len(sys.argv) > 3 and sys.argv[3] == "kitti": bench.W, bench.H = 1241, 376
view_sets, _, _ = bench.make_inputs(S, 1234)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
ctx = ov2slam_amd.Context(0, stream=stream.cuda_stream)
W, H = bench.W, bench.H
sets_d = torch.from_numpy(np.ascontiguousarray(view_sets[:, 0])).to(dev)
fr = sets_d[torch.arange(S, device=dev) % view_sets.shape[0]].contiguous()
P = ov2slam_amd.Pyramid(ctx, W, H, 9, 3, batch=S)
ts = []
for r in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    P.build_clahe_from_device(fr.data_ptr(), 3.0, W // 50, H // 50)
    e1.record(stream); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = sorted(ts[1:])
print("S=%d %dx%d: min %.1f us  median %.1f us per call" % (S, W, H, t[0] * 1e3, t[len(t) // 2] * 1e3))
