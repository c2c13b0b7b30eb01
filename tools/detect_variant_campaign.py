#!/usr/bin/env python3
"""How far are the keypoint sets of detectSingleScale (src/feature_extractor.cpp:288-440) under the arithmetic variants a different
OpenCV build would run from the canonical ones the HIP kernels implement bit for bit?  CPU only (oracle): N synthetic keyframes
at the EuRoC (752 x 480, cell 35) and KITTI (1241 x 376, cell 35) geometry, CLAHE'd like configs 2 / 3 and raw, half of them topping
up a tracked keypoint set (the keyframe case), under
  blur_half_even     GaussianBlur 3x3 rounding ties to even (float-kernel build) instead of half up (the fixed-point paths)
  subpix_generic     getRectSubPix's four-tap generic form for every patch instead of the optimised two-tap running form
  subpix_float_acc   cornerSubPix's five gradient sums in float instead of double
  sobel_dy_exact     the other evaluation order of cv::Sobel(dx = 0, dy = 1, scale)  (OV2_OPT_SOBEL_DY_ORDER exists on the device)
  all                the first three together
Per variant (sets matched by nearest neighbour): keypoints without a counterpart within 1 px (another arg-max of the min-eigenvalue
map won the cell, or the selection that follows changed), the largest distance of the others (sub-pixel refinement), bit-identical
fraction, frames with a different keypoint COUNT.  Output: profiles/archive/r4_detect_variants.json.
The variants are restated from the public sources: this bounds the deviation on restated code, it does not pin it (no OpenCV here)."""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from ov2slam_amd import synth           # noqa: E402

VARIANTS = {"blur_half_even": dict(blur=O.BLUR_HALF_EVEN), "subpix_generic": dict(subpix=O.SUBPIX_GENERIC),
            "subpix_float_acc": dict(subpix=O.SUBPIX_FLOAT_ACC), "sobel_dy_exact": dict(sobel=1),
            "all": dict(blur=O.BLUR_HALF_EVEN, subpix=O.SUBPIX_GENERIC)}


def detect(img, cell, cur, roi, blur=0, subpix=0, sobel=0):
    O.set_sobel_dy_order(sobel)
    try:
        with O.detect_variant(blur=blur, subpix=subpix):
            return O.detect_singlescale(img, cell, cur, roi, 0.001, True)[0]
    finally:
        O.set_sobel_dy_order(0)


def one(job):
    seed, (w, h), use_clahe = job
    rng = np.random.default_rng(seed)
    img = synth.frame_pair(w, h, seed=seed, shift=(rng.uniform(-4, 4), rng.uniform(-3, 3)), theta=rng.uniform(-0.01, 0.01))[0]
    if seed % 4 == 3:                                                   # a quarter of the frames: sensor-like noise on top
        img = np.clip(img.astype(np.int16) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)
    if use_clahe:
        img = O.clahe(img, 3.0, w // 50, h // 50)
    roi = (5, 5, w - 10, h - 10)
    cur = np.zeros((0, 2), np.float32)
    if seed % 2:                                                        # keyframe top-up: two thirds of the cells already occupied
        k = synth.grid_keypoints(w, h, 35, rng)
        cur = k[rng.uniform(size=len(k)) < 0.66]
    base = detect(img, 35, cur, roi)
    out = {"points": len(base)}
    for name, kw in VARIANTS.items():
        v = detect(img, 35, cur, roi, **kw)
        # sets are compared point by point through nearest neighbours (one moved winner changes the mask of the cells after it and
        # with it the ORDER of everything that follows: index-wise comparison would call all of it different)
        r = {"count_differs": int(len(v) != len(base)), "moved": 0, "identical": 0, "max_d_unmoved": 0.0, "max_d_moved": 0.0, "compared": len(base)}
        if len(v) and len(base):
            from scipy.spatial import cKDTree
            d, idx = cKDTree(v.astype(np.float64)).query(base.astype(np.float64))
            moved = d > 1.0
            r["moved"] = int(moved.sum()); r["identical"] = int((v[idx].view(np.uint32) == base.view(np.uint32)).all(1).sum())
            r["max_d_unmoved"] = float(d[~moved].max()) if (~moved).any() else 0.0
            r["max_d_moved"] = float(d[moved].max()) if moved.any() else 0.0
        out[name] = r
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    jobs = [(s, (752, 480) if s % 3 else (1241, 376), s % 5 != 0) for s in range(n)]
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        res = pool.map(one, jobs, chunksize=8)
    tot = {"keyframes": n, "keypoints": sum(r["points"] for r in res), "geometry": "2/3 EuRoC 752x480, 1/3 KITTI 1241x376, cell 35; 4/5 CLAHE'd (clip 3, 50-px tiles), "
           "1/5 raw; 1/2 topping up a tracked set; 1/4 with +-6 grey levels of noise", "variants": {}}
    for name in VARIANTS:
        rs = [r[name] for r in res]
        pts = sum(v["compared"] for v in rs)
        tot["variants"][name] = {
            "keyframes_with_a_different_keypoint_count": sum(v["count_differs"] for v in rs),
            "keypoints_compared": pts,
            "cell_winner_moved": sum(v["moved"] for v in rs),
            "cell_winner_moved_fraction": sum(v["moved"] for v in rs) / max(1, pts),
            "max_move_px": max(v["max_d_moved"] for v in rs),
            "bit_identical_fraction": sum(v["identical"] for v in rs) / max(1, pts),
            "max_abs_dpx_of_the_unmoved": max(v["max_d_unmoved"] for v in rs)}
    tot["note"] = ("variants restated from the public OpenCV sources (oracle/detect.c): a bound by measurement on restated code, not a pin. The HIP kernels implement "
                   "the canonical column (fixed-point blur, two-tap getRectSubPix, double sums, row-filter Sobel dy) bit for bit.")
    out = os.path.join(ROOT, "profiles", "r4_detect_variants.json")
    json.dump(tot, open(out, "w"), indent=1)
    print(json.dumps(tot["variants"], indent=1))


if __name__ == "__main__":
    main()
