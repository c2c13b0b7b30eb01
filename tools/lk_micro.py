#!/usr/bin/env python3
"""LK kernel cost model on the GPU box: time ov2_fb_klt_d for different (levels, max_iter) at fixed batch."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ov2slam_amd
from ov2slam_amd import _lib as L
import bench
if os.environ.get("OV2_LK_MICRO_LIB"):          # knock-out experiments: time a variant build of the library
    L.LIB_PATH = os.environ["OV2_LK_MICRO_LIB"]

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
view_sets, kps, pri = bench.make_inputs(S, 1234)      # (sets, NF+1, H, W): sequence s shows view set s % sets
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
ctx = ov2slam_amd.Context(0, stream=stream.cuda_stream)
lib = ctx.lib
W, H, NK = bench.W, bench.H, bench.NKPS
_sets = torch.from_numpy(np.ascontiguousarray(view_sets[:, :2])).to(dev)                     # the first two views of every set
_idx = torch.arange(S, device=dev) % view_sets.shape[0]
fr = torch.stack([_sets[:, 0][_idx], _sets[:, 1][_idx]])
P0 = ov2slam_amd.Pyramid(ctx, W, H, 9, 3, batch=S); P1 = ov2slam_amd.Pyramid(ctx, W, H, 9, 3, batch=S)
vp = lambda t: C.c_void_p(t.data_ptr())
# the bench's pre-processing (CLAHE + pyramid), like the step the kernel runs in
clahe = lib.ov2_pyr_build_clahe_d
L.check(clahe(ctx.h, P0.h_pyr, vp(fr[0]), W, W * H, C.c_double(3.0), W // 50, H // 50)); L.check(clahe(ctx.h, P1.h_pyr, vp(fr[1]), W, W * H, C.c_double(3.0), W // 50, H // 50))
k = torch.from_numpy(kps[0]).to(dev); p0 = torch.from_numpy(pri[0]).to(dev); p = p0.clone()
st = torch.zeros((S, NK), dtype=torch.uint8, device=dev); stats = torch.zeros(2, dtype=torch.int64, device=dev)
print("S=%d points=%d" % (S, S * NK))
cases = ((3, 30), (3, 1), (3, 0), (0, 30), (0, 1), (0, 0), (1, 30))
if len(sys.argv) > 2 and sys.argv[2] == "cap":          # Gauss-Newton trip cap sweep: the upper bound of what "cap in-wave, finish the stragglers in a second launch" can gain
    cases = tuple((lv, mi) for lv in (3, 1) for mi in (30, 12, 8, 6, 5, 4, 3, 2, 1, 0))
for lvl, mi in cases:
    ts = []
    for rep in range(6):
        p.copy_(p0); stats.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        L.check(lib.ov2_fb_klt_d(ctx.h, P0.h_pyr, P1.h_pyr, 9, lvl, mi, 0.01, 30.0, 0.5, vp(k), vp(p), NK, None, vp(st), None if os.environ.get('OV2_LK_MICRO_NOSTATS') else vp(stats)))
        e1.record(stream); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    it, vis = stats.tolist()
    t = min(ts[1:])
    print("nbpyrlvl=%d max_iter=%2d : %8.1f us  iters=%9d visits=%8d tracked=%.3f  -> %.2f ns/pt" % (lvl, mi, t * 1e3, it, vis, st.float().mean().item(), t * 1e6 / (S * NK)))
