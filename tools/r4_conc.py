#!/usr/bin/env python3
"""Concurrency of independent sequences on one GPU: aggregate frames/s of N keyframe-cycle streams inside one driver process, for
several values of GPU_MAX_HW_QUEUES (the HIP runtime multiplexes streams onto that many hardware queues; default 4)."""
import json, os, subprocess, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ov2slam_amd import batch, stream, synth
td = tempfile.mkdtemp()
exe = stream.build_native_driver(td)
tex = synth.base_texture(1400, 1234)
windows = [synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7 + i) for i in range(2)]
cases = []
for i in range(8):
    sq = batch.SyntheticSequence("s%d" % i, 300, seed=1000 + i, tex=tex, stereo=True)
    cases.append(os.path.join(td, "c%d.bin" % i)); stream.write_case(cases[-1], sq, windows)
def run(n, env=None):
    e = dict(os.environ); e.update(env or {})
    r = subprocess.run(stream.native_argv(exe, cases[:n], "newest", 0), capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-400:]
    st = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    wall = max(s["t_end"] for s in st) - min(s["t_begin"] for s in st)
    return sum(s["frames"] for s in st) / wall, [round(s["frames"] / s["seconds"]) for s in st]
run(1)
for n, env in ((1, {}), (2, {}), (4, {}), (4, {"GPU_MAX_HW_QUEUES": "16"}), (8, {"GPU_MAX_HW_QUEUES": "32"}), (4, {"GPU_MAX_HW_QUEUES": "16", "HIP_FORCE_DEV_KERNARG": "1"}),
               (8, {"GPU_MAX_HW_QUEUES": "8"})):
    fps, per = run(n, env)
    print("streams %d %-50s aggregate %7.0f frames/s   per stream %s" % (n, json.dumps(env), fps, per), flush=True)
