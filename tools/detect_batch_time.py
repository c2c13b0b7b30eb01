#!/usr/bin/env python3
"""Batched keyframe detector on the GPU box in isolation: ov2_detect_singlescale_batch_d on level 0 of an S-image pyramid
(bench.py's detect_batch section without the rest of the bench).  Usage: detect_batch_time.py [S]."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ov2slam_amd
import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
views, kps, _ = bench.make_inputs(S, 1234)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
ctx = ov2slam_amd.Context(0, stream=stream.cuda_stream)
W, H, CELL = bench.W, bench.H, bench.CELL
fr = torch.from_numpy(views[0]).to(dev)[None].expand(S, H, W).contiguous()
P = ov2slam_amd.Pyramid(ctx, W, H, 9, 3, batch=S).build_clahe_from_device(fr.data_ptr(), 3.0, W // 50, H // 50)
ncells = (W // CELL) * (H // CELL); cap = 2 * ncells
out = torch.zeros((S, cap, 2), dtype=torch.float32, device=dev)
roi = (5, 5, W - 10, H - 10)
for subpix in (True, False):
    qual = np.full(S, 1e-3)
    torch.cuda.synchronize()
    nd = ov2slam_amd.FeatureExtractor.detectSingleScaleBatch(ctx, P, CELL, 0, 0, 0, roi, qual, out.data_ptr(), cap, subpix=subpix)
    t = time.perf_counter()
    for _ in range(3):
        qual[:] = 1e-3
        nd = ov2slam_amd.FeatureExtractor.detectSingleScaleBatch(ctx, P, CELL, 0, 0, 0, roi, qual, out.data_ptr(), cap, subpix=subpix)
    ms = (time.perf_counter() - t) / 3 * 1e3
    print("S=%d subpix=%d: %.2f ms per call, %.1f points per image" % (S, subpix, ms, nd.mean()))
