#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_digests.json.

There are NO reference-provided golden vectors for this path (the reference ships no tests and cannot
be built here); the only external known answers are the Ceres unit-test values
restated in tests/test_oracle_ba.py.  These digests therefore pin the ORACLE AGAINST ITSELF: they
catch an accidental change of the CPU restatement (compiler flags, refactors), nothing more.
Inputs are the deterministic generators of ov2slam_amd/synth.py."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from ov2slam_amd import synth           # noqa: E402


def dig(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:32]


def compute():
    out = {}
    prev, cur, flow = synth.frame_pair(376, 240, seed=21, shift=(2.7, -1.9), theta=0.005)
    P, Cq = O.Pyramid(prev, 9, 3), O.Pyramid(cur, 9, 3)
    out["pyramid"] = dig(*[a for l in range(P.levels) for a in P.level(l, padded=True)])
    rng = np.random.default_rng(5)
    kps = synth.grid_keypoints(376, 240, 35, rng)
    pri = (flow(kps) + rng.normal(0, 1.5, kps.shape)).astype(np.float32)
    p, st, stats = O.fb_klt(P, Cq, 9, 3, 30., 0.5, kps, pri)
    out["fb_klt"] = dig(p, st.astype(np.uint8), np.array(stats))
    out["clahe"] = dig(O.clahe(prev, 3.0, 7, 4))
    pts, th = O.detect_grid_fast(prev, 50, kps[::5], 10, O.MASK_AS_EXECUTED, True)
    out["detect_grid_fast"] = dig(pts, np.array([th]))
    pts, q = O.detect_singlescale(prev, 35, kps[::5], (5, 5, 366, 230), 0.001, True)
    out["detect_singlescale"] = dig(pts, np.array([q]))
    # the min-eigenvalue map under both Sobel-dy evaluation orders (oracle/detect.c: default = OpenCV's row-filter order)
    out["cell_mineig_opencv_rowfilter"] = dig(O.cell_mineig(prev, 105, 70, 35))
    O.set_sobel_dy_order(O.SOBEL_DY_EXACT_SUM)
    out["cell_mineig_exact_sum"] = dig(O.cell_mineig(prev, 105, 70, 35))
    O.set_sobel_dy_order(O.SOBEL_DY_OPENCV_ROWFILTER)
    pb = synth.make_ba_problem(8, 120, 5, stereo=True, seed=3)
    r = O.ba_solve(pb)
    # BA is fp64 with libm calls: digest a rounded view (12 significant digits) to stay libm-version tolerant
    rnd = lambda a: np.array([float("%.10e" % v) for v in np.ravel(a)])
    out["ba_solve"] = dig(rnd(r["poses"]), rnd(r["invdepth"]), np.array([r["iterations"], r["termination"]]))
    # round-1 additions: undistortion + bearing, stereo SAD scan / epipolar gate, structure-only BA
    K = (458.654, 457.296, 367.215, 248.375); D = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)
    iK = np.linalg.inv(np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1.0]]))
    u, b = O.compute_keypoints(O.CAM_PINHOLE, K, D, iK, kps)
    out["compute_keypoints_pinhole"] = dig(u, b)
    right = np.roll(prev, -6, axis=1)
    xp, err = O.line_min_sad(P.level(2)[0], O.Pyramid(right, 9, 3).level(2)[0], kps * np.float32(0.25), 7, True)
    out["line_min_sad"] = dig(xp, err)
    Frl = np.array([[0, 0, 0], [0, 0, -1e-2], [0, 1e-2, 0]])
    rk = (kps + np.array([-6.0, 0.7], np.float32)).astype(np.float32)
    o1 = O.stereo_epipolar_check(False, Frl, O.CAM_PINHOLE, K, D, u, rk)
    out["epipolar_sampson"] = dig(o1[0], o1[1], o1[2], o1[3].astype(np.uint8))
    sp = synth.make_structure_problem(8, 80, 4, seed=9)
    r = O.structure_ba(sp)
    out["structure_ba"] = dig(rnd(r["xyz"]), np.array([r["iterations"], r["termination"]]))
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_digests.json")
    json.dump(compute(), open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
