#!/usr/bin/env python3
"""Idle time between the kernels of the bench step: reads a rocprofv3 --kernel-trace CSV (Start_Timestamp / End_Timestamp per
dispatch) and prints, per kernel name, the mean duration and the mean gap to the PREVIOUS dispatch on the device."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0][:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append(e - s)
    if prev_end is not None:
        gap[n].append(s - prev_end)
    prev_end = max(prev_end or 0, e)
tot_d = tot_g = 0
for n in sorted(dur, key=lambda k: -sum(dur[k])):
    d = sum(dur[n]) / len(dur[n]); g = sum(gap[n]) / max(1, len(gap[n]))
    print("%-42s n=%5d  dur %9.1f us   gap before %7.1f us" % (n, len(dur[n]), d / 1e3, g / 1e3))
