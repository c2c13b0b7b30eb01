#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by tools/profile.sh into one small JSON (kept under profiles/)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir = sys.argv[1]
res = {"kernels": {}, "pmc": {}}
stats = glob.glob(os.path.join(out_dir, "trace_kernel_stats.csv"))
if stats:
    for r in csv.DictReader(open(stats[0])):
        res["kernels"][r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                   "total_us": float(r["TotalDurationNs"]) / 1e3, "pct": float(r["Percentage"])}
for tag in ("fetch", "write", "sq"):
    f = glob.glob(os.path.join(out_dir, tag + "_counter_collection.csv"))
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, vals in cs.items():
            res["pmc"].setdefault(k, {})[c] = {"mean_per_dispatch": sum(vals) / len(vals), "dispatches": len(vals)}
# which kernel sources the counters belong to: bench.py refuses counters whose source hash is not the tree's
import hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res["source_sha256"] = {f: hashlib.sha256(open(os.path.join(root, "ov2slam_amd", "csrc", f), "rb").read()).hexdigest()
                        for f in ("lk3.hip", "clahe.hip", "pyramid.hip") if os.path.exists(os.path.join(root, "ov2slam_amd", "csrc", f))}
path = os.path.join(out_dir, "summary.json")
json.dump(res, open(path, "w"), indent=1, sort_keys=True)
# refresh the "current" block of profiles/lk_traffic.json (what bench.py's roofline.traffic reads) when the LK kernel was profiled:
#   python tools/summarize_profile.py <dir> --update-lk-traffic <seqs_per_gpu> <algorithmic bytes per launch>
if "--update-lk-traffic" in sys.argv:
    i = sys.argv.index("--update-lk-traffic")
    seqs, alg = int(sys.argv[i + 1]), float(sys.argv[i + 2])
    kk = [k for k in res["pmc"] if "k_fb_klt3" in k]
    if kk and "FETCH_SIZE" in res["pmc"][kk[0]] and "WRITE_SIZE" in res["pmc"][kk[0]]:
        pm = res["pmc"][kk[0]]
        fetch_kb, write_kb = pm["FETCH_SIZE"]["mean_per_dispatch"], pm["WRITE_SIZE"]["mean_per_dispatch"]
        tj_path = os.path.join(root, "profiles", "lk_traffic.json")
        tj = json.load(open(tj_path))
        hbm = (2.0 * fetch_kb + write_kb) * 1024.0                      # gfx950: FETCH_SIZE tallies 128-byte lines at 64 B (calibration below)
        cur = {"seqs_per_gpu": seqs, "kernel": "k_fb_klt3", "kernel_source_sha256": res["source_sha256"].get("lk3.hip"),
               "fetch_size_kb_per_launch": fetch_kb, "write_size_kb_per_launch": write_kb, "fetch_correction": 2.0, "hbm_bytes_per_launch": hbm,
               "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
               "avg_launch_us_rocprof": res["kernels"].get(kk[0], {}).get("avg_us"), "source": os.path.relpath(path, root)}
        sq = {c: v["mean_per_dispatch"] for c, v in pm.items() if c.startswith("SQ_")}
        if sq.get("SQ_WAVE_CYCLES") and sq.get("SQ_ACTIVE_INST_ANY") and cur["avg_launch_us_rocprof"]:
            cur["sq_counters_per_dispatch"] = sq
            # quad-cycles in which a SIMD issued / (1024 SIMDs x kernel time in quad-cycles at 2.4 GHz)
            cur["valu_issue_frac"] = sq["SQ_ACTIVE_INST_ANY"] / (1024.0 * cur["avg_launch_us_rocprof"] * 1e-6 * 2.4e9 / 4.0)
            cur["limiter_kind"] = "valu_issue" if cur["valu_issue_frac"] > 0.5 else "latency"
        cur["limiter"] = ("%s: SQ_INSTS_VALU %.1f M wave-instructions per dispatch, traffic %.2f GB per launch = %.2fx the algorithmic bytes "
                          "(16-byte row gathers: 12.5 %% of every 128-byte line used)" % (cur.get("limiter_kind", "?"), sq.get("SQ_INSTS_VALU", 0) / 1e6, hbm / 1e9, hbm / alg))
        tj.setdefault("history", []).append(tj.get("current"))
        tj["current"] = cur
        json.dump(tj, open(tj_path, "w"), indent=1)
        print("updated", tj_path)
lk = [k for k in res["pmc"] if "k_fb_klt" in k]
for k in lk:
    print(k, json.dumps(res["pmc"][k]))
print("wrote", path)
