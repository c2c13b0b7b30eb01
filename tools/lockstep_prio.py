#!/usr/bin/env python3
"""configs[4] through the lock-step driver with different stream priorities of the tracking / mapper / estimator contexts ("a,b,e").
    python tools/lockstep_prio.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ov2slam_amd import batch, stream, synth  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ov2_lockstep_cases")
    os.makedirs(d, exist_ok=True)
    tex = synth.base_texture(1400, 1234)
    names = sorted(batch.EUROC_FRAMES)
    windows = [synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7 + i) for i in range(2)]
    order = sorted(names, key=lambda s: -batch.EUROC_FRAMES[s])
    cases = []
    for i, s in enumerate(order):
        sq = batch.SyntheticSequence(s, batch.EUROC_FRAMES[s], seed=1000 + names.index(s), tex=tex, stereo=True)
        cases.append(os.path.join(d, "case%02d.bin" % i)); stream.write_case(cases[-1], sq, windows)
    exe = stream.build_native_driver(d, "lockstep_driver")
    stream.run_lockstep(exe, cases[:3], ba_policy="newest")
    res = []
    combos = [(p, 1) for p in ("0,0,0", "1,1,-1", "1,0,0", "1,1,0", "0,0,-1", "1,0,-1", "0,1,0", "0,0,0")]
    if len(sys.argv) > 2 and sys.argv[2] == "groups":
        combos = [("0,0,0", g) for g in (1, 2, 3, 4, 6, 1, 2)]
    workers = [3]
    if len(sys.argv) > 2 and sys.argv[2] == "workers":
        combos = [("0,0,0", 1)]; workers = [0, 1, 2, 3, 5, 8, 0, 3]
    for prio, groups, nw in [(p, g, w) for (p, g) in combos for w in workers]:
        for policy in ("newest", "all"):
            st, sm = stream.run_lockstep(exe, cases, ba_policy=policy, loader_threads=4, priorities=prio, batched_estimator=groups, host_workers=nw)
            r = {"priorities_tracker_mapper_estimator": prio, "estimator_groups": groups, "host_workers": nw, "policy": policy, "fps": round(sm["frames"] / sm["seconds"], 1), "seconds": sm["seconds"], "slam_library_s": sm["slam_library_s"],
                 "wait_mapper_s": sm["slam_wait_for_mapper_s"], "ba_solves": sum(s["ba_solves"] for s in st), "ba_batches": sm.get("ba_batches"), "ba_busy_s_sum": round(sum(s["ba_busy_s"] for s in st), 4)}
            res.append(r); print(json.dumps(r), flush=True)
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
