#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU through tools/lockstep_driver.cpp: where the wall clock goes (library calls of the SLAM thread, waits for
the loader threads / the mappers) for several loader-thread counts and both estimator policies, next to the per-sequence stream form.
    python tools/lockstep_sweep.py [scale=1] [out.json]        (scale divides the EuRoC frame counts)
Keeps the case files under $TMPDIR/ov2_lockstep_cases for a following rocprofv3 run (tools/lockstep_prof.sh)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ov2slam_amd import batch, stream, synth  # noqa: E402


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ov2_lockstep_cases")
    os.makedirs(d, exist_ok=True)
    tex = synth.base_texture(1400, 1234)
    names = sorted(batch.EUROC_FRAMES)
    windows = [synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7 + i) for i in range(2)]
    order = sorted(names, key=lambda s: -batch.EUROC_FRAMES[s])
    cases = []
    for i, s in enumerate(order):
        sq = batch.SyntheticSequence(s, max(12, batch.EUROC_FRAMES[s] // scale), seed=1000 + names.index(s), tex=tex, stereo=True)
        cases.append(os.path.join(d, "case%02d.bin" % i)); stream.write_case(cases[-1], sq, windows)
    exe_l = stream.build_native_driver(d, "lockstep_driver")
    exe_s = stream.build_native_driver(d, "stream_driver")
    res = {"scale": scale, "frames": sum(max(12, batch.EUROC_FRAMES[s] // scale) for s in order), "runs": []}
    stream.run_lockstep(exe_l, cases[:3], ba_policy="newest")                       # warm-up
    for policy, loaders, prio, eb in (("newest", 4, True, True), ("all", 4, True, True), ("newest", 4, False, True), ("newest", 1, True, True), ("all", 4, False, True),
                                      ("newest", 4, True, False), ("all", 4, True, False), ("newest", 4, True, True), ("all", 4, True, True)):
        if True:
            st, sm = stream.run_lockstep(exe_l, cases, ba_policy=policy, loader_threads=loaders, priorities=prio, batched_estimator=eb)
            r = {"mode": "lockstep", "policy": policy, "loader_threads": loaders, "stream_priorities": prio, "batched_estimator": eb, "ba_batches": sm.get("ba_batches"),
                 "fps": sm["frames"] / sm["seconds"], "seconds": sm["seconds"],
                 "slam_thread_seconds": sm["slam_thread_seconds"], "slam_library_s": sm["slam_library_s"],
                 "wait_loader_s": sm["slam_wait_for_loader_s"], "wait_mapper_s": sm["slam_wait_for_mapper_s"], "steps": sm["steps"],
                 "us_per_step_library": 1e6 * sm["slam_library_s"] / sm["steps"],
                 "ba_solves": sum(s["ba_solves"] for s in st), "ba_skipped": sum(s["ba_skipped_kfs"] for s in st), "keyframes": sum(s["keyframes"] for s in st),
                 "ba_busy_s_sum": sum(s["ba_busy_s"] for s in st), "ba_device_ms_sum": sum(s["ba_device_ms"] for s in st),
                 "mapper_busy_s_sum": sum(s["mapper_busy_s"] for s in st)}
            res["runs"].append(r)
            print(json.dumps(r), flush=True)
    # without the mapper / estimator load: sequences that carry no BA windows
    nob = []
    for i, s in enumerate(order):
        sq = batch.SyntheticSequence(s, max(12, batch.EUROC_FRAMES[s] // scale), seed=1000 + names.index(s), tex=tex, stereo=True)
        nob.append(os.path.join(d, "nob%02d.bin" % i)); stream.write_case(nob[-1], sq, [])
    st, sm = stream.run_lockstep(exe_l, nob, ba_policy="newest", loader_threads=4)
    r = {"mode": "lockstep, no localBA (front end + stereo matching only)", "fps": sm["frames"] / sm["seconds"], "seconds": sm["seconds"],
         "slam_library_s": sm["slam_library_s"], "wait_loader_s": sm["slam_wait_for_loader_s"], "wait_mapper_s": sm["slam_wait_for_mapper_s"]}
    res["runs"].append(r); print(json.dumps(r), flush=True)
    # round 4's form at 1/16 of the length
    small = []
    for i, s in enumerate(order):
        sq = batch.SyntheticSequence(s, max(12, batch.EUROC_FRAMES[s] // 16), seed=1000 + names.index(s), tex=tex, stereo=True)
        small.append(os.path.join(d, "small%02d.bin" % i)); stream.write_case(small[-1], sq, windows)
    for conc in (1, 2):
        st, sec = stream.run_native_concurrent(exe_s, small, concurrency=conc)
        r = {"mode": "per-sequence streams (tools/stream_driver.cpp), frame counts / 16", "concurrency": conc, "fps": sum(s["frames"] for s in st) / sec, "seconds": sec}
        res["runs"].append(r); print(json.dumps(r), flush=True)
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
