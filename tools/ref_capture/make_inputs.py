#!/usr/bin/env python3
"""Writes the inputs of the reference-capture run (tests/golden/ref_inputs/*.npy): the same synthetic frames, keypoints
and priors the parity tests use (ov2slam_amd/synth.py, fixed seeds).  numpy only -- runs anywhere."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ov2slam_amd import synth  # noqa: E402

out = os.path.join(ROOT, "tests", "golden", "ref_inputs")
os.makedirs(out, exist_ok=True)
for tag, (w, h) in (("euroc", (752, 480)), ("kitti", (1241, 376))):
    prev, cur, flow = synth.frame_pair(w, h, seed=1234)
    rng = np.random.default_rng(7)
    kps = synth.grid_keypoints(w, h, 35, rng)
    pri = (flow(kps) + rng.normal(0, 1.5, kps.shape)).astype(np.float32)
    np.save(os.path.join(out, tag + "_prev.npy"), prev)
    np.save(os.path.join(out, tag + "_cur.npy"), cur)
    np.save(os.path.join(out, tag + "_kps.npy"), kps.astype(np.float32))
    np.save(os.path.join(out, tag + "_pri.npy"), pri)
    # a third of the cells already occupied: exercises the occupancy / mask prologue of the detectors
    np.save(os.path.join(out, tag + "_curkps.npy"), kps[::3].astype(np.float32))
# local-BA problems for the Ceres capture (capture_ba.cpp): the flat layout of ov2slam_amd/stream.py
from ov2slam_amd import stream  # noqa: E402
for tag, (n_kf, n_lm, obs, stereo, seed) in (("kf8_mono", (8, 200, 6, False, 5)), ("kf8_stereo", (8, 200, 6, True, 5)), ("kf12_stereo", (12, 400, 8, True, 3)),
                                             ("kf50_mono", (50, 2000, 30, False, 42)), ("kf50_stereo", (50, 2000, 30, True, 42))):
    with open(os.path.join(out, "ba_%s.bin" % tag), "wb") as f:
        stream.write_ba_problem(f, synth.make_ba_problem(n_kf, n_lm, obs, stereo=stereo, seed=seed))
print("wrote", out)
