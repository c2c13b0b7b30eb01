#!/usr/bin/env python3
"""Pre-processing step on the GPU box in isolation: ov2_pyr_build_clahe_d (CLAHE LUT + apply [+ level 1] + pyramid levels) on S
resident 752x480 (or, 4th argument "kitti", 1241x376) frames, HIP-event time per call.  Usage: pre_micro.py [S] [reps] [strips: -1 auto | 0 | 1 | 2] [euroc | kitti] [entropy];
under rocprofv3 for per-kernel counters.  With a 5th argument "entropy" the same call is timed on inputs of decreasing histogram entropy
(the histogram half of the fused kernel is bound by LDS atomics: equal grey levels inside a wavefront serialise on one bin) and one JSON
line is printed: {"noise": ms, "levels16": ms, "dark": ms, "saturated": ms, "constant": ms}."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ov2slam_amd
from ov2slam_amd import _lib as L
import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
strips = int(sys.argv[3]) if len(sys.argv) > 3 else -1
if len(sys.argv) > 4 and sys.argv[4] == "kitti": bench.W, bench.H = 1241, 376        # configs[2]
views, _, _ = bench.make_inputs(S, 1234)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
ctx = ov2slam_amd.Context(0, stream=stream.cuda_stream)
lib = ctx.lib
ctx.set_option(L.OV2_OPT_CLAHE_STRIPS, strips)
W, H = bench.W, bench.H
fr = torch.from_numpy(views[:2]).to(dev)[:, None].expand(-1, S, H, W).contiguous()
P = ov2slam_amd.Pyramid(ctx, W, H, 9, 3, batch=S)
ts = []
for r in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    L.check(lib.ov2_pyr_build_clahe_d(ctx.h, P.h_pyr, C.c_void_p(fr[r & 1].data_ptr()), W, W * H, C.c_double(3.0), W // 50, H // 50))
    e1.record(stream); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("S=%d  pre-processing per call: min %.1f us  median %.1f us   (%s)" % (S, min(ts[1:]) * 1e3, float(np.median(ts[1:])) * 1e3,
      "OV2_OPT_CLAHE_STRIPS=%d" % strips))

if len(sys.argv) > 5 and sys.argv[5] == "entropy":
    import json
    rng = np.random.default_rng(5)
    base = views[:2].astype(np.int32)
    sel = rng.uniform(size=base.shape) < 0.8
    variants = {
        "noise": base,                                                           # band-limited noise stretched to 0..255: the bench input
        "levels16": np.where(sel, (base // 16) * 16 + 8, base),                  # 80 % of the pixels on 16 grey levels
        "dark": np.where(sel, base // 32, base // 4),                            # an under-exposed frame: 80 % of the pixels in 8 levels near 0
        "saturated": np.where(sel, 255, 255 - base // 8),                        # an over-exposed frame: 80 % of the pixels exactly 255
        "constant": np.full_like(base, 128),                                     # every pixel equal: all 64 adds of a wavefront on one bin
    }
    out = {}
    for name, v in variants.items():
        frv = torch.from_numpy(np.clip(v, 0, 255).astype(np.uint8)).to(dev)[:, None].expand(-1, S, H, W).contiguous()
        tv = []
        for r in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            L.check(lib.ov2_pyr_build_clahe_d(ctx.h, P.h_pyr, C.c_void_p(frv[r & 1].data_ptr()), W, W * H, C.c_double(3.0), W // 50, H // 50))
            e1.record(stream); torch.cuda.synchronize()
            tv.append(e0.elapsed_time(e1))
        out[name] = float(np.median(tv[1:]))
        del frv
    out["slowdown_worst"] = max(out.values()) / out["noise"]
    print(json.dumps({"pre_entropy_ms_per_call": out, "S": S, "strips": strips}))
