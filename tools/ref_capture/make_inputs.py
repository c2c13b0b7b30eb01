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
print("wrote", out)
