#!/usr/bin/env python3
"""CPU (oracle) diagnostic behind DESIGN.md 4.1: distribution of Gauss-Newton trips per level visit on the bench's step, and what
a 20-keypoint lock-step wavefront pays for it -- as is, and with stragglers deferred to a second, densely packed pass after a cap."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import oracle as O

S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
views, kps, pri = bench.make_inputs(S, 1234)
lib = O.lib()
lib.orc_lk_trip_hist_enable(1)
tiles = (bench.W // 50, bench.H // 50)
pyr = [O.Pyramid(O.clahe(v, bench.CLAHE_CLIP, *tiles), bench.WIN, bench.LEVELS) for v in views]
NA = bench.N_PASS_A
for f in range(2 * bench.NF):
    a, b = pyr[bench.walk_view(f)], pyr[bench.walk_view(f + 1)]
    for s in range(S):
        O.fb_klt(a, b, bench.WIN, 1, 30.0, 0.5, kps[f, s, :NA], pri[f, s, :NA])                 # pass A: priors, 2 levels
        O.fb_klt(a, b, bench.WIN, bench.LEVELS, 30.0, 0.5, kps[f, s, NA:], pri[f, s, NA:])      # pass B: full pyramid
h = (C.c_int * 64)()
lib.orc_lk_trip_hist_get(h)
h = np.array(h[:], np.float64)
n = h.sum(); p = h / n; t = np.arange(64)
mean = (p * t).sum()
cdf = np.cumsum(p)
emax = lambda q, m: float(((1 - np.cumsum(q)[:-1] ** m)).sum() + 0)          # E[max of m iid] = sum_{k>=0} P(max > k)
print("level visits: %d   mean trips %.2f   P(trips >= 8) %.3f   P(>= 12) %.3f   P(>= 20) %.3f   P(= 30) %.4f"
      % (n, mean, 1 - cdf[7], 1 - cdf[11], 1 - cdf[19], p[30]))
print("histogram (trips: share):", ", ".join("%d: %.3f" % (k, p[k]) for k in range(31) if p[k] >= 0.0005))
E20 = emax(p, 20)
print("E[max of 20 independent visits] = %.2f  -> lock-step efficiency %.2f" % (E20, mean / E20))
BUILD = 4.4                                                                   # template build in trip units
base = BUILD + E20
print("cost per visit-wave in trip units: build %.1f + trips %.2f = %.2f" % (BUILD, E20, base))
for cap in (4, 5, 6, 8, 10):
    q = p.copy(); tail = q[cap + 1:].sum(); q[cap] += tail; q[cap + 1:] = 0          # main pass: visits stop at the cap
    main = BUILD + emax(q, 20)
    # deferred pass: the stragglers rebuild their template and run their remaining trips, 20 of them per wavefront
    r = np.zeros(64); r[1:64 - cap] = p[cap + 1:]; r /= max(r.sum(), 1e-30)
    deferred = tail * (BUILD + emax(r, 20))
    print("cap %2d: main %.2f + deferred %.3f (%.1f %% of the visits) = %.2f  -> %.3f x" % (cap, main, deferred, 100 * tail, main + deferred, (main + deferred) / base))
