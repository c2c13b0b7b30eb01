#!/usr/bin/env python3
"""Batched keyframe detector on the GPU box in isolation: ov2_detect_singlescale_batch_d on level 0 of an S-image pyramid
(bench.py's detect_batch section without the rest of the bench; the bench's distinct image contents: sequence s shows view set s % 64).
Both keyframe cases: no current keypoints, and topping up the tracked keypoints of every sequence (half of the cells occupied).
Usage: detect_batch_time.py [S] [reps]."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ov2slam_amd
import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
view_sets, kps, _ = bench.make_inputs(S, 1234)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
ctx = ov2slam_amd.Context(0, stream=stream.cuda_stream)
W, H, CELL = bench.W, bench.H, bench.CELL
sets_d = torch.from_numpy(np.ascontiguousarray(view_sets[:, 0])).to(dev)
fr = sets_d[torch.arange(S, device=dev) % view_sets.shape[0]].contiguous()
P = ov2slam_amd.Pyramid(ctx, W, H, 9, 3, batch=S).build_clahe_from_device(fr.data_ptr(), 3.0, W // 50, H // 50)
ncells = (W // CELL) * (H // CELL); cap = 2 * ncells
out = torch.zeros((S, cap, 2), dtype=torch.float32, device=dev)
roi = (5, 5, W - 10, H - 10)
cur_half = torch.from_numpy(np.ascontiguousarray(kps[0][:, ::2])).to(dev)
n_half = int(cur_half.shape[1])
ncur_d = torch.full((S,), n_half, dtype=torch.int32, device=dev)
for name, (cp, cc, npz) in (("empty", (0, 0, 0)), ("top-up of %d keypoints" % n_half, (cur_half.data_ptr(), n_half, ncur_d.data_ptr()))):
    for subpix in (True, False):
        qual = np.full(S, 1e-3)
        torch.cuda.synchronize()
        nd = ov2slam_amd.FeatureExtractor.detectSingleScaleBatch(ctx, P, CELL, cp, cc, npz, roi, qual, out.data_ptr(), cap, subpix=subpix)
        t = time.perf_counter()
        for _ in range(REPS):
            qual[:] = 1e-3
            nd = ov2slam_amd.FeatureExtractor.detectSingleScaleBatch(ctx, P, CELL, cp, cc, npz, roi, qual, out.data_ptr(), cap, subpix=subpix)
        ms = (time.perf_counter() - t) / REPS * 1e3
        print("S=%d %s subpix=%d: %.2f ms per call = %.2f us per image, %.1f points per image" % (S, name, subpix, ms, ms * 1e3 / S, nd.mean()), flush=True)
