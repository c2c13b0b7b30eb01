#!/usr/bin/env python3
"""configs[4] through the lock-step driver: where the SLAM thread's step goes (the driver's slam_step_breakdown_s).
    python tools/lockstep_breakdown.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ov2slam_amd import batch, stream, synth  # noqa: E402
d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ov2_lockstep_cases"); os.makedirs(d, exist_ok=True)
tex = synth.base_texture(1400, 1234)
names = sorted(batch.EUROC_FRAMES)
windows = [synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7 + i) for i in range(2)]
order = sorted(names, key=lambda s: -batch.EUROC_FRAMES[s])
cases = []
for i, s in enumerate(order):
    sq = batch.SyntheticSequence(s, batch.EUROC_FRAMES[s], seed=1000 + names.index(s), tex=tex, stereo=True)
    cases.append(os.path.join(d, "case%02d.bin" % i)); stream.write_case(cases[-1], sq, windows)
exe = stream.build_native_driver(d, "lockstep_driver")
stream.run_lockstep(exe, cases[:3])
for _ in range(2):
    st, sm = stream.run_lockstep(exe, cases)
    steps = sm["steps"]
    print(json.dumps({"fps": round(sm["frames"] / sm["seconds"]), "us_per_step": round(1e6 * sm["slam_thread_seconds"] / steps, 1),
                      "us_per_step_by_part": {k: round(1e6 * v / steps, 1) for k, v in sm["slam_step_breakdown_s"].items()},
                      "wait_for_loader_us": round(1e6 * sm["slam_wait_for_loader_s"] / steps, 1), "wait_for_mapper_us": round(1e6 * sm["slam_wait_for_mapper_s"] / steps, 1)}))
