#!/usr/bin/env python3
"""The bench's tracking step in isolation (preprocessImage of the new frame + fbKltTracking pass A: 216 keypoints, 2 levels + pass B: 92 keypoints,
4 levels; S sequences, 64 image contents) with the upper pyramid levels on the context's stream (OV2_OPT_PYR_ASYNC_LEVELS = 0) and on its
auxiliary stream beside pass A (= 1), alternating; wall clock per step over K steps, and the HIP-event time of the two LK launches.
Usage: step_time.py [S] [steps] [rounds]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ov2slam_amd
from ov2slam_amd import _lib as L
import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
view_sets, kps, pri = bench.make_inputs(S, 1234)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
ctx = ov2slam_amd.Context(0, stream=stream.cuda_stream)
lib = ctx.lib
W, H, NK = bench.W, bench.H, bench.NKPS
NA = 216
nviews = view_sets.shape[1]
sets_d = torch.from_numpy(np.ascontiguousarray(view_sets)).to(dev)                       # (sets, views, H, W)
idx = torch.arange(S, device=dev) % view_sets.shape[0]
frames = [sets_d[:, v][idx].contiguous() for v in range(nviews)]
vp = lambda t: C.c_void_p(t.data_ptr())
P = [ov2slam_amd.Pyramid(ctx, W, H, 9, 3, batch=S) for _ in range(2)]
NF = kps.shape[0]
kA = [torch.from_numpy(np.ascontiguousarray(kps[f][:, :NA])).to(dev) for f in range(NF)]
pA = [torch.from_numpy(np.ascontiguousarray(pri[f][:, :NA])).to(dev) for f in range(NF)]
kB = [torch.from_numpy(np.ascontiguousarray(kps[f][:, NA:])).to(dev) for f in range(NF)]
pB = [torch.from_numpy(np.ascontiguousarray(pri[f][:, NA:])).to(dev) for f in range(NF)]
stA = torch.zeros((S, NA), dtype=torch.uint8, device=dev); stB = torch.zeros((S, NK - NA), dtype=torch.uint8, device=dev)
wA = [t.clone() for t in pA]; wB = [t.clone() for t in pB]

def pre(p, v):
    L.check(lib.ov2_pyr_build_clahe_d(ctx.h, p.h_pyr, vp(frames[v]), W, W * H, C.c_double(3.0), W // 50, H // 50))

def step(i, ev=None):
    f = i % NF; v = (i + 1) % nviews
    prev, cur = P[i & 1], P[(i + 1) & 1]
    pre(cur, v)
    wA[f].copy_(pA[f]); wB[f].copy_(pB[f])
    if ev: ev[0].record(stream)
    L.check(lib.ov2_fb_klt_d(ctx.h, prev.h_pyr, cur.h_pyr, 9, 1, 30, 0.01, 30.0, 0.5, vp(kA[f]), vp(wA[f]), NA, None, vp(stA), None))
    if ev: ev[1].record(stream)
    L.check(lib.ov2_fb_klt_d(ctx.h, prev.h_pyr, cur.h_pyr, 9, 3, 30, 0.01, 30.0, 0.5, vp(kB[f]), vp(wB[f]), NK - NA, None, vp(stB), None))
    if ev: ev[2].record(stream)

res = {0: [], 1: []}
for r in range(ROUNDS):
    for mode in ((0, 1) if hasattr(L, "OV2_OPT_PYR_ASYNC_LEVELS") else (0,)):      # (the option exists only with profiles/r6_async_pyramid_levels.patch applied)
        if hasattr(L, "OV2_OPT_PYR_ASYNC_LEVELS"): ctx.set_option(L.OV2_OPT_PYR_ASYNC_LEVELS, mode)
        pre(P[0], 0)
        for i in range(5): step(i)
        torch.cuda.synchronize()
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
        t = time.perf_counter()
        for i in range(K): step(5 + i, evs[i])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / K * 1e3
        a = np.mean([e[0].elapsed_time(e[1]) for e in evs]); b = np.mean([e[1].elapsed_time(e[2]) for e in evs])
        res[mode].append(ms)
        print("async_levels=%d: %.3f ms per step (%.0f frames/s); LK pass A %.3f ms, pass B %.3f ms; tracked %.3f / %.3f" %
              (mode, ms, S / ms * 1e3, a, b, stA.float().mean().item(), stB.float().mean().item()), flush=True)
print("median ms per step: serial %.3f%s" % (float(np.median(res[0])), ", levels beside pass A %.3f" % float(np.median(res[1])) if res[1] else ""))
