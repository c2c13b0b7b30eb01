#!/usr/bin/env python3
"""Latency of the keyframe detector where it is latency-bound: ONE image (the single-camera drop-in: detectSingleScale on level 0 of the tracker's
pyramid, one sync) and a lock-step batch of 11 (BASELINE configs[4] on one GPU), empty frames and topping up 154 tracked keypoints.
Usage: detect_single_time.py [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ov2slam_amd
import bench

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
S = 11
view_sets, kps, _ = bench.make_inputs(S, 1234)
dev = torch.device("cuda", 0)
ctx = ov2slam_amd.Context(0)
W, H, CELL = bench.W, bench.H, bench.CELL
roi = (5, 5, W - 10, H - 10)
img = np.ascontiguousarray(view_sets[0, 0])
P1 = ov2slam_amd.Pyramid(ctx, W, H, 9, 3).build_clahe(img, 3.0, W // 50, H // 50)
cur = np.ascontiguousarray(kps[0][0][::2]).astype(np.float32)
e = np.zeros((0, 2), np.float32)
for name, c in (("empty", e), ("top-up of %d keypoints" % len(cur), cur)):
    for subpix in (True, False):
        fx = ov2slam_amd.FeatureExtractor(ctx, dmaxquality=0.001)
        d = fx.detectSingleScalePyr(P1, CELL, c, roi, subpix=subpix)
        t = time.perf_counter()
        for _ in range(REPS):
            fx.dmaxquality_ = 0.001
            d = fx.detectSingleScalePyr(P1, CELL, c, roi, subpix=subpix)
        print("1 image, %s, subpix=%d: %.1f us per call, %d points" % (name, subpix, (time.perf_counter() - t) / REPS * 1e6, len(d)), flush=True)
sets_d = torch.from_numpy(np.ascontiguousarray(view_sets[:, 0])).to(dev)
fr = sets_d[torch.arange(S, device=dev) % view_sets.shape[0]].contiguous()
PB = ov2slam_amd.Pyramid(ctx, W, H, 9, 3, batch=S).build_clahe_from_device(fr.data_ptr(), 3.0, W // 50, H // 50)
ncells = (W // CELL) * (H // CELL); cap = 2 * ncells
out = torch.zeros((S, cap, 2), dtype=torch.float32, device=dev)
cur_d = torch.from_numpy(np.ascontiguousarray(kps[0][:, ::2])).to(dev); n_half = int(cur_d.shape[1])
ncur_d = torch.full((S,), n_half, dtype=torch.int32, device=dev)
for name, (cp, cc, npz) in (("empty", (0, 0, 0)), ("top-up of %d keypoints" % n_half, (cur_d.data_ptr(), n_half, ncur_d.data_ptr()))):
    qual = np.full(S, 1e-3)
    torch.cuda.synchronize()
    nd = ov2slam_amd.FeatureExtractor.detectSingleScaleBatch(ctx, PB, CELL, cp, cc, npz, roi, qual, out.data_ptr(), cap)
    t = time.perf_counter()
    for _ in range(REPS):
        qual[:] = 1e-3
        nd = ov2slam_amd.FeatureExtractor.detectSingleScaleBatch(ctx, PB, CELL, cp, cc, npz, roi, qual, out.data_ptr(), cap)
    print("batch of %d, %s: %.1f us per call, %.1f points per image" % (S, name, (time.perf_counter() - t) / REPS * 1e6, nd.mean()), flush=True)
