#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity campaign (run on the GPU box: python tools/fuzz_parity.py [n_cases] [seed]).
Every case draws an image size, content, window / cell / tile parameters and point sets (including points on and
beyond the borders) and demands bit-exact agreement for the integer/float32 front-end paths; every few cases the
single-sequence tracker (ov2_tracker_*: preprocessImage + kltTracking), the pyramid-resident detectors and random bundle
adjustments (inverse depth, 3-D points, the large-problem path) are compared as well (BA: the tolerances of tests/test_gpu_ba.py)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ov2slam_amd
from ov2slam_amd import synth, _lib as L
from oracle import oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(SEED)
ctx = ov2slam_amd.Context(0)
trk = ov2slam_amd.FeatureTracker(ctx, 30, 0.01)
fails, loose = [], []
ba_pool = []


def rand_image(w, h, kind):
    if kind == 0:
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    a, _, _ = synth.frame_pair(w, h, seed=int(rng.integers(1 << 30)))
    if kind == 2:
        a = (a.astype(np.float32) * 0.3 + 100).astype(np.uint8)          # low contrast
    if kind == 3:
        a[: h // 3] = 128                                                  # flat band
    return a


def check(name, ok, info):
    if not ok:
        fails.append((name, info)); print("MISMATCH", name, info, flush=True)


t0 = time.time()
for case in range(N):
    w, h = int(rng.integers(64, 900)), int(rng.integers(48, 520))
    if rng.integers(0, 2): w = 4 * (w // 4)                            # dword-multiple widths: the strip kernel's geometry
    # batch mode's strip kernel (CLAHE apply + level 1 + borders in one walk) forced on / off per case (read at every launch)
    ctx.set_option(L.OV2_OPT_CLAHE_STRIPS, int(rng.integers(0, 3)))       # separate kernels / strip kernel / fused strip kernel
    kind = int(rng.integers(0, 4))
    prev = rand_image(w, h, kind)
    shift = rng.uniform(-6, 6, 2)
    cur = np.roll(np.roll(prev, int(round(shift[1])), 0), int(round(shift[0])), 1)
    cur = np.clip(cur.astype(np.int16) + rng.integers(-3, 4, cur.shape), 0, 255).astype(np.uint8)
    info = dict(case=case, w=w, h=h, kind=kind)
    # ---- pyramid (+ fused CLAHE) ----
    lvl = int(rng.integers(0, 4))
    Gp = ov2slam_amd.Pyramid(ctx, w, h, 9, lvl).build(prev); Rp = O.Pyramid(prev, 9, lvl)
    Gc = ov2slam_amd.Pyramid(ctx, w, h, 9, lvl).build(cur); Rc = O.Pyramid(cur, 9, lvl)
    check("pyr_levels", Gp.levels == Rp.levels, info)
    for l in range(min(Gp.levels, Rp.levels)):
        check("pyramid", np.array_equal(Gp.download(l, padded=True)[0], Rp.level(l, padded=True)[0]), dict(info, level=l))
    tx, ty = int(rng.integers(1, max(2, w // 40))), int(rng.integers(1, max(2, h // 40)))
    clip = float(rng.choice([0.0, 1.0, 2.0, 3.0, 8.0, 40.0]))
    if tx + 1 <= 48:
        g = ov2slam_amd.CLAHE(ctx, clip, (tx, ty)).apply(prev)
        check("clahe", np.array_equal(g, O.clahe(prev, clip, tx, ty)), dict(info, tiles=(tx, ty), clip=clip))
        Pf = ov2slam_amd.Pyramid(ctx, w, h, 9, lvl).build_clahe(prev, clip, tx, ty)
        Rf = O.Pyramid(O.clahe(prev, clip, tx, ty), 9, lvl)
        for l in range(Pf.levels):
            check("clahe_pyr", np.array_equal(Pf.download(l, padded=True)[0], Rf.level(l, padded=True)[0]), dict(info, level=l))
    # ---- LK, both kernels ----
    n = int(rng.integers(1, 400))
    kps = np.stack([rng.uniform(-12, w + 12, n), rng.uniform(-12, h + 12, n)], 1).astype(np.float32)
    pri = (kps - shift + rng.normal(0, rng.choice([0.3, 2.0, 8.0]), kps.shape)).astype(np.float32)
    nl = int(rng.integers(0, Gp.levels))
    for impl in ("row", "lane3"):
        ctx.set_option(L.OV2_OPT_LK_IMPL, L.OV2_LK_IMPL_ROW if impl == "row" else L.OV2_LK_IMPL_LANE3)
        go, gs, gst = trk.fbKltTracking(Gp, Gc, 9, nl, 30., 0.5, kps, pri, return_stats=True)
        ro, rs, rst = O.fb_klt(Rp, Rc, 9, nl, 30., 0.5, kps, pri)
        check("fbklt_" + impl, np.array_equal(gs, rs) and np.array_equal(go.view(np.uint32), ro.view(np.uint32)) and gst[0] == rst[0], dict(info, n=n, nl=nl))
    ctx.set_option(L.OV2_OPT_LK_IMPL, L.OV2_LK_IMPL_AUTO)
    # ---- LK with the float accumulators of an x86 OpenCV build (OV2_OPT_LK_ACC), against the oracle in the same mode ----
    with ctx.options(lk_acc=L.OV2_LK_ACC_FLOAT_UI4), O.lk_acc_mode(O.LK_ACC_FLOAT_UI4):
        go, gs, gst = trk.fbKltTracking(Gp, Gc, 9, nl, 30., 0.5, kps, pri, return_stats=True)
        ro, rs, rst = O.fb_klt(Rp, Rc, 9, nl, 30., 0.5, kps, pri)
        check("fbklt_float_acc", np.array_equal(gs, rs) and np.array_equal(go.view(np.uint32), ro.view(np.uint32)) and gst[0] == rst[0], dict(info, n=n, nl=nl))
    # ---- stereo SAD scan on a random level ----
    sl = int(rng.integers(0, Gp.levels))
    lw, lh = Gp.level_size(sl)
    pts = np.stack([rng.uniform(0, lw - 1, 60), rng.uniform(0, lh - 1, 60)], 1).astype(np.float32)
    for go_left in (True, False):
        a = trk.getLineMinSAD(Gp, Gc, sl, pts, int(rng.choice([3, 5, 7])), go_left)
    ws = int(rng.choice([3, 5, 7]))
    a = trk.getLineMinSAD(Gp, Gc, sl, pts, ws, True); b = O.line_min_sad(Rp.level(sl)[0], Rc.level(sl)[0], pts, ws, True)
    check("line_min_sad", np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), dict(info, level=sl, ws=ws))
    # ---- stereo matching: the fused entry (ov2_stereo_match) against the call sequence, random 3-D priors ----
    if case % 3 == 1:
        from ov2slam_amd import stereo as _st
        cal = ov2slam_amd.CameraCalibration(ctx, "pinhole", 458.654, 457.296, w / 2.0, h / 2.0, D=None)
        inside = (kps[:, 0] >= 0) & (kps[:, 0] < w) & (kps[:, 1] >= 0) & (kps[:, 1] < h)
        sk = kps[inside][:200]                                           # getLineMinSAD needs points inside the image
        if len(sk):
            p3 = {int(i): (float(sk[i, 0] - shift[0] + rng.normal(0, 4.0)), float(sk[i, 1] - shift[1] + rng.normal(0, 2.0)))
                  for i in np.nonzero(rng.uniform(size=len(sk)) < 0.5)[0]}
            rect = bool(rng.integers(0, 2))
            slv = min(lvl, Gp.levels - 1)                                  # small images have fewer levels than asked for
            F = None if rect else np.array([[0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])
            a_ok, a_r = _st.stereo_matching(trk, Gp, Gc, sk, sk, cal, rect=rect, Frl=F, nklt_pyr_lvl=slv, priors3d=p3)
            b_ok, b_r = _st.stereo_matching_fused(trk, Gp, Gc, sk, sk, cal, rect=rect, Frl=F, nklt_pyr_lvl=slv, priors3d=p3)
            check("stereo_fused", np.array_equal(a_ok, b_ok) and np.array_equal(a_r.view(np.uint32), b_r.view(np.uint32)), dict(info, n=len(sk), rect=rect, lvl=slv))
    # ---- detectors ----
    if w >= 120 and h >= 120:
        cell = int(rng.choice([20, 35, 45, 50, 53, 58]))
        curk = kps[(kps[:, 0] > 0) & (kps[:, 0] < w - 1) & (kps[:, 1] > 0) & (kps[:, 1] < h - 1)][: int(rng.integers(0, 80))]
        fx = ov2slam_amd.FeatureExtractor(ctx, nfast_th=int(rng.integers(5, 40)), dmaxquality=float(rng.choice([1e-4, 1e-3, 1e-2])))
        th0, q0 = fx.nfast_th_, fx.dmaxquality_
        g = fx.detectGridFAST(prev, cell, curk)
        r, rth = O.detect_grid_fast(prev, cell, curk, th0, O.MASK_AS_EXECUTED, True)
        check("grid_fast", np.array_equal(g.view(np.uint32), r.view(np.uint32)) and fx.nfast_th_ == rth, dict(info, cell=cell, th=th0))
        roi = (int(rng.integers(0, 10)), int(rng.integers(0, 10)), w - int(rng.integers(10, 30)), h - int(rng.integers(10, 30)))
        g = fx.detectSingleScale(prev, cell, curk, roi)
        r, rq = O.detect_singlescale(prev, cell, curk, roi, q0, True)
        check("singlescale", np.array_equal(g.view(np.uint32), r.view(np.uint32)) and fx.dmaxquality_ == rq, dict(info, cell=cell, q=q0))
    # ---- pyramid-resident detectors: the same answer as on the host image ----
    if w >= 120 and h >= 120 and case % 3 == 0:
        P0 = ov2slam_amd.Pyramid(ctx, w, h, 9, 0).build(prev)
        fa = ov2slam_amd.FeatureExtractor(ctx, nfast_th=th0, dmaxquality=q0); fb = ov2slam_amd.FeatureExtractor(ctx, nfast_th=th0, dmaxquality=q0)
        ga, gb = fa.detectGridFAST(prev, cell, curk), fb.detectGridFASTPyr(P0, cell, curk)
        check("grid_fast_d", np.array_equal(ga.view(np.uint32), gb.view(np.uint32)) and fa.nfast_th_ == fb.nfast_th_, dict(info, cell=cell))
        ga, gb = fa.detectSingleScale(prev, cell, curk, roi), fb.detectSingleScalePyr(P0, cell, curk, roi)
        check("singlescale_d", np.array_equal(ga.view(np.uint32), gb.view(np.uint32)) and fa.dmaxquality_ == fb.dmaxquality_, dict(info, cell=cell))
    # ---- single-sequence tracker: preprocessImage + kltTracking (visual_front_end.cpp:1143-1177, :132-275) ----
    if w >= 150 and h >= 150 and case % 2 == 0:
        import contextlib
        facc = int(rng.integers(0, 3)) == 0          # a third of the tracker cases with the float accumulators (both kernels: the wave kernel by default)
        with contextlib.ExitStack() as stack:
            if facc:
                stack.enter_context(ctx.options(lk_acc=L.OV2_LK_ACC_FLOAT_UI4)); stack.enter_context(O.lk_acc_mode(O.LK_ACC_FLOAT_UI4))
            use_clahe = bool(rng.integers(0, 2)); clipv = float(rng.choice([1.0, 3.0, 8.0]))
            cap = int(rng.choice([512, 512, 64, 150]))                         # small capacities: keypoints beyond it run as further chunks
            t = ov2slam_amd.VisualFrontEndTracker(ctx, w, h, use_clahe=use_clahe, fclahe_val=clipv, nbmaxkps=cap, use_graph=bool(rng.integers(0, 2)))
            with_cal = bool(rng.integers(0, 2))
            if with_cal:                                                       # Frame::computeKeypoint inside the frame's enqueue
                tcal = ov2slam_amd.CameraCalibration(ctx, "pinhole", 458.654, 457.296, w / 2.0, h / 2.0, D=(-0.2834, 0.0739, 0.00019, 1.76e-05))
                t.setCalibration(tcal)
            t.trackFrame(prev, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), None)
            nk = int(rng.integers(1, 500))
            tk = np.stack([rng.uniform(-4, w + 4, nk), rng.uniform(-4, h + 4, nk)], 1).astype(np.float32)
            hp = (rng.uniform(size=nk) < rng.uniform(0, 1)).astype(np.uint8)
            tp = np.where(hp[:, None] > 0, tk - shift + rng.normal(0, rng.choice([0.5, 3.0, 15.0]), tk.shape), tk).astype(np.float32)
            use_prior = bool(rng.integers(0, 4))
            if use_prior:
                go, gs, gp3p = t.trackFrame(cur, tk, tp, hp)
            else:
                t.preprocessImage(cur); go, gs, gp3p = t.kltTracking(tk, tp, hp, klt_use_prior=False)
            a, b = (O.clahe(prev, clipv, w // 50, h // 50), O.clahe(cur, clipv, w // 50, h // 50)) if use_clahe else (prev, cur)
            ro, rok, rretry, rp3p = O.klt_tracking(O.Pyramid(a, 9, 3), O.Pyramid(b, 9, 3), tk, tp, hp, klt_use_prior=use_prior)
            check("tracker", gp3p == rp3p and np.array_equal((gs & 1).astype(bool), rok) and np.array_equal((gs & 2).astype(bool), rretry)
                  and np.array_equal(np.ascontiguousarray(go, np.float32).view(np.uint32), np.ascontiguousarray(ro, np.float32).view(np.uint32)),
                  dict(info, n=nk, clahe=use_clahe, prior=use_prior, cap=cap, float_acc=facc))
            if with_cal:
                gu, gb = t.lastKeypoints(nk)
                ru, rb = O.compute_keypoints(O.CAM_PINHOLE, tcal.K, tcal.D, tcal.iK, go)
                check("tracker_keypoints", np.array_equal(gu.view(np.uint32), ru.view(np.uint32)) and np.array_equal(gb.view(np.uint64), rb.view(np.uint64)),
                      dict(info, n=nk, cap=cap))
            t.close()
    # ---- bundle adjustment: random small problems, both landmark forms, both problem-size paths ----
    if case % 4 == 0:
        from ov2slam_amd import optimizer
        n_kf, obs = int(rng.integers(3, 30)), int(rng.integers(2, 12))
        n_lm, stereo, bseed = int(rng.integers(20, 600)), bool(rng.integers(0, 2)), int(rng.integers(1 << 20))
        kw = [dict(), dict(max_iter=10, huber_delta=-1.0), dict(max_iter=12, function_tolerance=1e-9)][int(rng.integers(0, 3))]
        binfo = dict(case=case, n_kf=n_kf, n_lm=n_lm, obs=obs, stereo=stereo, seed=bseed, kw=kw)

        def ba_diff(g, r, key):
            return dict(it=(g["iterations"], r["iterations"]), term=(g["termination"], r["termination"]), ok_steps=(g["num_successful_steps"], r["num_successful_steps"]),
                        cost0=r["initial_cost"], cost=(g["final_cost"], r["final_cost"]), dpos=float(np.abs(g["poses"][:, :3] - r["poses"][:, :3]).max()),
                        dlm=float(np.abs(g[key] - r[key]).max()))

        def ba_close(g, r, key, k=1.0):
            scale = max(1e-9, np.abs(r["poses"][:, :3]).max())
            qg = g["poses"][:, 3:] * np.sign((g["poses"][:, 3:] * r["poses"][:, 3:]).sum(1))[:, None]
            m = np.isfinite(r["chi2"])
            return (g["iterations"] == r["iterations"] and g["termination"] == r["termination"] and g["num_successful_steps"] == r["num_successful_steps"]
                    and abs(g["final_cost"] - r["final_cost"]) <= k * 1e-8 * abs(r["final_cost"]) + 1e-12
                    and np.abs(g["poses"][:, :3] - r["poses"][:, :3]).max() <= k * 1e-7 * scale and np.abs(qg - r["poses"][:, 3:]).max() <= k * 1e-7
                    and np.allclose(g[key], r[key], rtol=k * 1e-6, atol=k * 1e-9) and np.array_equal(np.isfinite(g["chi2"]), m)
                    and np.allclose(g["chi2"][m], r["chi2"][m], rtol=k * 1e-6, atol=k * 1e-9))

        def ba_check(name, g, r, key):
            # tight = the tolerances of tests/test_gpu_ba.py (well-posed problems); random draws include barely constrained ones
            # (mono, 2 observations per point, not converged at max_iter) where summation order shows at 1e-7: those are
            # counted separately against 100x the tolerance and listed, anything beyond that is a mismatch
            if ba_close(g, r, key): return
            if ba_close(g, r, key, 100.0):
                loose.append((name, dict(binfo, **ba_diff(g, r, key)))); print("LOOSE", name, loose[-1][1], flush=True)
                return
            # beyond 100x: is it the problem or the solver?  The arithmetic noise floor of THIS problem = the oracle against
            # itself with the residual blocks in another (equally valid) order -- Ceres sums in block order, so a permuted problem
            # is the same Ceres problem with another rounding.  A device result inside 4x that spread, with identical iteration /
            # termination / step pattern, is listed as ill-conditioned with both numbers; anything else is a mismatch.
            fl = ba_noise_floor(key)
            d = ba_diff(g, r, key); m = np.isfinite(r["chi2"]) & np.isfinite(g["chi2"])
            dchi = float(np.abs(g["chi2"][m] - r["chi2"][m]).max()) if m.any() else 0.0
            same = (g["iterations"] == r["iterations"] and g["termination"] == r["termination"] and g["num_successful_steps"] == r["num_successful_steps"]
                    and np.array_equal(np.isfinite(g["chi2"]), np.isfinite(r["chi2"])))
            if (fl is not None and same and d["dpos"] <= 4 * fl["dpos"] and d["dlm"] <= 4 * fl["dlm"] and dchi <= 4 * fl["dchi2"]
                    and abs(g["final_cost"] - r["final_cost"]) <= 4 * fl["dcost"]):
                loose.append((name + " (ill-conditioned draw)", dict(binfo, **d, dchi2=dchi, oracle_self_spread=fl)))
                print("ILL-CONDITIONED", name, loose[-1][1], flush=True)
            else: check(name, False, dict(binfo, **d, dchi2=dchi, oracle_self_spread=fl))

        floor_cache = {}
        def ba_noise_floor(key, trials=3):
            if key in floor_cache: return floor_cache[key]
            q = pb; r0 = O.xyz_ba_solve(q, O.ba_default_options(**kw)) if key == "xyz" else O.ba_solve(q, O.ba_default_options(**kw))
            prng = np.random.default_rng(bseed + 77); out = dict(dpos=0.0, dlm=0.0, dchi2=0.0, dcost=0.0)
            for _ in range(trials):
                perm = prng.permutation(q["n_res"]); q2 = dict(q)
                for kk, v in q.items():
                    if kk.startswith("res_") or kk == "is_outlier": q2[kk] = np.ascontiguousarray(v[perm])
                r2 = O.xyz_ba_solve(q2, O.ba_default_options(**kw)) if key == "xyz" else O.ba_solve(q2, O.ba_default_options(**kw))
                if (r2["iterations"], r2["termination"], r2["num_successful_steps"]) != (r0["iterations"], r0["termination"], r0["num_successful_steps"]):
                    out = dict(dpos=np.inf, dlm=np.inf, dchi2=np.inf, dcost=np.inf, note="the oracle's own step pattern depends on the block order"); break
                c2 = np.empty_like(r2["chi2"]); c2[perm] = r2["chi2"]; m = np.isfinite(r0["chi2"]) & np.isfinite(c2)
                out["dpos"] = max(out["dpos"], float(np.abs(r2["poses"][:, :3] - r0["poses"][:, :3]).max()))
                out["dlm"] = max(out["dlm"], float(np.abs(r2[key] - r0[key]).max()))
                out["dchi2"] = max(out["dchi2"], float(np.abs(c2[m] - r0["chi2"][m]).max()) if m.any() else 0.0)
                out["dcost"] = max(out["dcost"], abs(r2["final_cost"] - r0["final_cost"]))
            floor_cache[key] = out
            return out
        pb = synth.make_ba_problem(n_kf, n_lm, min(obs, n_kf), stereo=stereo, seed=bseed)
        r = O.ba_solve(pb, O.ba_default_options(**kw))
        for big in (0, 1):
            ctx.set_option(L.OV2_OPT_BA_FORCE_LARGE, big)
            g = optimizer.solve(ctx, pb, optimizer.default_options(ctx.lib, **kw))
            ba_check("ba_invdepth_big%d" % big, g, r, "invdepth")
        # round 3: the lineariser without LDS pre-aggregation and the column-chunked sparse Schur complement, forced onto the small problem
        ctx.set_option(L.OV2_OPT_BA_LIN_DIRECT, 1); ctx.set_option(L.OV2_OPT_BA_SCHUR_CHUNK, int(rng.choice([36, 60, 96])))
        g = optimizer.solve(ctx, pb, optimizer.default_options(ctx.lib, **kw))
        ba_check("ba_invdepth_big_direct_chunked", g, r, "invdepth")
        for k_ in (L.OV2_OPT_BA_FORCE_LARGE, L.OV2_OPT_BA_LIN_DIRECT, L.OV2_OPT_BA_SCHUR_CHUNK):
            ctx.set_option(k_, 0)
        # round 3: both passes of localBA on the resident problem (ov2_local_ba) vs the two-call protocol on the oracle
        def osolver(prob, res_active, chi2_init, depthpos_init, **kk):
            return O.ba_solve(prob, O.ba_default_options(**kk), res_active, chi2_init, depthpos_init)
        gl = optimizer.Optimizer(ctx).localBA(pb); rl = optimizer.Optimizer(None, solver=osolver).localBA(pb)
        rits = (rl["pass1"]["iterations"], rl["pass2"]["iterations"] if rl["l2_done"] else 0)
        okl = (gl["l2_done"] == rl["l2_done"] and tuple(gl["iterations"]) == rits and np.array_equal(gl["bad_obs"], rl["bad_obs"])
               and np.abs(gl["poses"] - rl["poses"]).max() <= 1e-6 * max(1.0, np.abs(rl["poses"]).max()))
        if not okl:
            # an outlier verdict within rounding of the chi2 threshold may flip on barely constrained draws: listed, not failed, when the poses agree
            if np.abs(gl["poses"] - rl["poses"]).max() <= 1e-4 * max(1.0, np.abs(rl["poses"]).max()) and int((gl["bad_obs"] != rl["bad_obs"]).sum()) <= 2:
                loose.append(("local_ba", dict(binfo, its=(tuple(gl["iterations"]), rits)))); print("LOOSE local_ba", loose[-1][1], flush=True)
            else:
                check("local_ba", False, dict(binfo, its=(tuple(gl["iterations"]), rits), nbad=(int(gl["bad_obs"].sum()), int(rl["bad_obs"].sum()))))
        # round 5: the lock-step batch (ov2_local_ba_batch) -- the last few random windows in ONE call against their one-problem results
        ba_pool.append((pb, gl, dict(binfo)))
        if len(ba_pool) >= 5:
            stop = [bool(rng.integers(0, 4) == 0) for _ in ba_pool]
            gb, nb = optimizer.Optimizer(ctx).localBA_batch([q[0] for q in ba_pool], stop=stop)
            for (pq, g1, inf), b, st_ in zip(ba_pool, gb, stop):
                if st_:                                                     # a stopped problem keeps pass 1: compare with a stopped single call
                    o1 = optimizer.Optimizer(ctx); o1.signalStopLocalBA(); g1 = o1.localBA(pq)
                okb = (b["l2_done"] == g1["l2_done"] and tuple(b["iterations"]) == tuple(g1["iterations"]) and np.array_equal(b["bad_obs"], g1["bad_obs"])
                       and np.abs(b["poses"] - g1["poses"]).max() <= 1e-6 * max(1.0, np.abs(g1["poses"]).max()))
                if not okb:
                    if np.abs(b["poses"] - g1["poses"]).max() <= 1e-4 * max(1.0, np.abs(g1["poses"]).max()) and int((b["bad_obs"] != g1["bad_obs"]).sum()) <= 2:
                        loose.append(("local_ba_batch", dict(inf, its=(tuple(b["iterations"]), tuple(g1["iterations"]))))); print("LOOSE local_ba_batch", loose[-1][1], flush=True)
                    else:
                        check("local_ba_batch", False, dict(inf, stop=st_, shared=nb, its=(tuple(b["iterations"]), tuple(g1["iterations"])),
                                                            nbad=(int(b["bad_obs"].sum()), int(g1["bad_obs"].sum()))))
            ba_pool.clear()
        pb = synth.make_xyz_ba_problem(n_kf, n_lm, min(obs, n_kf), stereo=stereo, seed=bseed)
        g = optimizer.solve_xyz(ctx, pb, optimizer.default_options(ctx.lib, **kw)); r = O.xyz_ba_solve(pb, O.ba_default_options(**kw))
        ba_check("ba_xyz", g, r, "xyz")
    if (case + 1) % 10 == 0:
        print("case %d/%d  %.0f s  mismatches so far: %d" % (case + 1, N, time.time() - t0, len(fails)), flush=True)
print("FUZZ DONE: %d cases, %d mismatches, %d BA solves within 100x but not 1x the test tolerance" % (N, len(fails), len(loose)))
sys.exit(1 if fails else 0)
