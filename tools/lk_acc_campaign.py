#!/usr/bin/env python3
"""How far is the canonical int64-accumulator LK (what the HIP kernels compute bit-exactly) from the float-accumulator
variants stock OpenCV builds execute?  CPU only (oracle): fbKltTracking on the synthetic EuRoC / KITTI sets -- raw, CLAHE'd,
and a worst case of hard-edged binary blocks (the strongest gradients an 8-bit image can have) -- under every accumulator
mode of oracle/frontend.c; status flips and position deltas against the int64 variant go to profiles/archive/r3_lk_acc_modes.json.
The float orders are restated from the public lkpyramid.cpp (3.4 SSE2 intrinsics, 4.x universal intrinsics, scalar): no
OpenCV exists in this image or on the GPU box (gpurun_out/r3probe/probe.txt), so this bounds the deviation, it does not pin it."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from ov2slam_amd import synth           # noqa: E402


def merge(tot, r):
    if tot is None:
        return r
    tot["points"] += r["points"]; tot["tracked_int64"] += r["tracked_int64"]
    for k, v in r["modes"].items():
        t = tot["modes"][k]
        for f in ("status_flips", "bit_identical_positions", "above_0.01px"):
            t[f] += v[f]
        for f in ("max_abs_dpx", "p99_abs_dpx"):
            t[f] = max(t[f], v[f])
    return tot


def blocks_pair(w, h, seed):
    rng = np.random.default_rng(100 + seed)
    tex = np.kron((rng.integers(0, 2, (260, 260)) * 255).astype(np.uint8), np.ones((7, 7), np.uint8))
    sx, sy = 3 + seed % 5, 2
    prev = tex[100:100 + h, 100:100 + w].copy(); cur = tex[100 - sy:100 - sy + h, 100 - sx:100 - sx + w].copy()
    return prev, cur, (lambda p: p + np.array([sx, sy], np.float32))


def run(seeds=12):
    out = {}
    for (w, h, tag) in ((752, 480, "euroc"), (1241, 376, "kitti")):
        for kind in ("raw", "clahe", "binary_blocks"):
            tot = None
            for seed in range(seeds):
                if kind == "binary_blocks":
                    prev, cur, flow = blocks_pair(w, h, seed)
                else:
                    prev, cur, flow = synth.frame_pair(w, h, seed=seed, shift=(3.1 + 0.7 * (seed % 6), -2.2 + 0.5 * (seed % 5)), theta=0.004)
                    if kind == "clahe":
                        prev = O.clahe(prev, 3.0, w // 50, h // 50); cur = O.clahe(cur, 3.0, w // 50, h // 50)
                rng = np.random.default_rng(seed)
                kps = synth.grid_keypoints(w, h, 35, rng)
                pri = (flow(kps) + rng.normal(0, 1.5, kps.shape)).astype(np.float32)
                P, Q = O.Pyramid(prev, 9, 3), O.Pyramid(cur, 9, 3)
                for lvl in (1, 3):                                   # both fbKltTracking calls of a frame
                    tot = merge(tot, O.lk_acc_mode_report(P, Q, kps, pri, nbpyrlvl=lvl))
            out["%s_%s" % (tag, kind)] = tot
    return out


if __name__ == "__main__":
    res = run(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
    worst = {"status_flips": 0, "max_abs_dpx": 0.0, "points": 0}
    for v in res.values():
        worst["points"] += v["points"]
        for m in v["modes"].values():
            worst["status_flips"] = max(worst["status_flips"], m["status_flips"]); worst["max_abs_dpx"] = max(worst["max_abs_dpx"], m["max_abs_dpx"])
    res["summary"] = worst
    res["note"] = __doc__
    path = os.path.join(ROOT, "profiles", "r3_lk_acc_modes.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(worst))
