#!/usr/bin/env python3
"""bench.py -- OV2SLAM front-end + local-BA hot path on MI355X (one JSON line on rank 0).

Metric (BASELINE.json): "frames/sec tracking + local-BA iters/sec, EuRoC MH_01 stereo 'accurate'".

One timed "step" = one camera frame of config[1] (EuRoC MH_01 stereo, 'accurate' parameters:
752x480, LK 9x9, 3+1 pyramid levels, 30 it / 0.01 px, 308 keypoints) pushed through the HIP hot
path for each of the `--seqs` sequences this GPU processes in lock-step (offline
batch-of-sequences mode of config[4]; default 4096 = 16.5 GB of HBM per GPU, `--seqs 1` is the single-sequence
drop-in case; the per-frame work does not depend on it, the fill of the GPU does -- DESIGN.md section 6):
    preprocessImage : CLAHE (clip 3.0, 15x9 tiles: use_clahe 1 in parameters_files/accurate) of the new left
                      image + device-resident pyramid build (/root/reference/src/visual_front_end.cpp:1143-1177)
    kltTracking     : fbKltTracking pass A (nbpyrlvl=1) on the keypoints that carry a 3-D prior,
                      pass B (nbpyrlvl=3) on the others (src/visual_front_end.cpp:186-268)
`value` = tracked frames/s over all sequences and ranks (max-over-ranks time).  Images, keypoints
and priors are synthetic (ov2slam_amd/synth.py; no dataset offline) and already resident in HBM.

Outside the K timed steps (rank 0, N=1 only) the same line also reports
    ba     : local-BA LM iterations/s on config[3] (50 KF x 10k landmarks x 30 obs, resident in HBM)
    detect : detectSingleScale (273 cells) + fbKltTracking through the host-buffer drop-in API
    cpu_baseline : the oracle (CPU port of the reference arithmetic) on the box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, WIN, LEVELS, CELL = 752, 480, 9, 3, 35
NKPS = 308                  # nbmaxkps_ for EuRoC accurate (slam_params.cpp:107-110)
N_PASS_A = 216              # keypoints with a 3-D prior -> pass A (2 levels)
N_PASS_B = NKPS - N_PASS_A  # pass B (4 levels)
NF = 6                      # distinct synthetic views per sequence (cycled)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
CLAHE_CLIP, CLAHE_TILES = 3.0, (W // 50, H // 50)   # ov2slam.cpp:85-89, accurate/euroc: fclahe_val 3


def make_inputs(seqs, seed):
    """NF+1 synthetic views (shared by the sequences of a rank), per-sequence keypoints / priors."""
    from ov2slam_amd import synth
    rng = np.random.default_rng(seed)
    tex = synth.base_texture(1400, seed)
    views, offs = [], []
    ox, oy, th = 150.0, 150.0, 0.0
    for _ in range(NF + 1):
        views.append(synth.warp(tex, W, H, ox, oy, th))
        offs.append((ox, oy, th))
        ox += rng.uniform(-5, 5); oy += rng.uniform(-4, 4); th += rng.uniform(-0.006, 0.006)
    views = np.stack(views)
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0

    def flow(pts, a, b):
        (ox0, oy0, t0), (ox1, oy1, t1) = offs[a], offs[b]
        dx, dy = pts[:, 0] - cx, pts[:, 1] - cy
        c, s = np.cos(t0), np.sin(t0)
        tx, ty = c * dx - s * dy + cx + ox0, s * dx + c * dy + cy + oy0
        dx, dy = tx - cx - ox1, ty - cy - oy1
        c, s = np.cos(-t1), np.sin(-t1)
        return np.stack([c * dx - s * dy + cx, s * dx + c * dy + cy], 1)

    kps = np.zeros((NF, seqs, NKPS, 2), np.float32)
    pri = np.zeros_like(kps)
    for f in range(NF):
        for s in range(seqs):
            k = synth.grid_keypoints(W, H, CELL, rng)[:NKPS]
            if len(k) < NKPS:
                extra = np.stack([rng.uniform(20, W - 20, NKPS - len(k)), rng.uniform(20, H - 20, NKPS - len(k))], 1)
                k = np.concatenate([k, extra.astype(np.float32)])
            rng.shuffle(k)
            kps[f, s] = k
            pri[f, s] = flow(k.astype(np.float64), f, f + 1) + rng.normal(0, 1.5, k.shape)
    return views, kps, pri


def lk_algorithmic_bytes(iters, visits, npts):
    # per (point, level) visit 500 B of template footprint (9x9 bilinear window of the u8 image and
    # the int16x2 derivative), 100 B of the other image per executed GN iteration, 29 B of point I/O per call
    return 500 * visits + 100 * iters + 29 * npts


def cpu_baseline(views, kps, pri, ba_problem, budget_s=10.0):
    """Oracle ('port') on the host cores: the same tracking step, and the same BA problem."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    # pick the thread count that is fastest for this step on this host (cv::parallel_for_ would use a pool)
    p0, p1 = O.Pyramid(views[0], WIN, LEVELS), O.Pyramid(views[1], WIN, LEVELS)
    best_nt, best_t = 1, 1e9
    for nt in (1, 2, 4, 8, 16, 32, 64):
        if nt > cores:
            break
        t0 = time.perf_counter()
        for _ in range(5):
            O.fb_klt(p0, p1, WIN, LEVELS, 30., 0.5, kps[0, 0], pri[0, 0], nthreads=nt)
        t = time.perf_counter() - t0
        if t < best_t:
            best_nt, best_t = nt, t
    t0 = time.perf_counter()
    frames = 0
    prevp = p0
    while True:
        f = frames % NF
        if f == 0:
            prevp = O.Pyramid(O.clahe(views[0], CLAHE_CLIP, *CLAHE_TILES), WIN, LEVELS)
        curp = O.Pyramid(O.clahe(views[f + 1], CLAHE_CLIP, *CLAHE_TILES), WIN, LEVELS)
        k, p = kps[f, 0], pri[f, 0]
        O.fb_klt(prevp, curp, WIN, 1, 30., 0.5, k[:N_PASS_A], p[:N_PASS_A], nthreads=best_nt)
        O.fb_klt(prevp, curp, WIN, LEVELS, 30., 0.5, k[N_PASS_A:], p[N_PASS_A:], nthreads=best_nt)
        prevp = curp
        frames += 1
        el = time.perf_counter() - t0
        if el > budget_s and frames >= 20:
            break
    out = {"value": frames / el, "unit": "frames/s", "cores": best_nt, "kind": "port", "host_cores": cores,
           "sample": "%d frames of the same synthetic 752x480 step (CLAHE + pyramid build + LK pass A/B) through "
                     "oracle/liboracle.so, LK over keypoints on %d pthreads (best of 1..64)" % (frames, best_nt)}
    if ba_problem is not None:
        t0 = time.perf_counter()
        r = O.ba_solve(ba_problem)
        el = time.perf_counter() - t0
        out["ba"] = {"iters_per_s": r["iterations"] / el, "iterations": r["iterations"], "seconds": el, "cores": 1,
                     "sample": "one robust pass (<=5 LM iterations) of the 50 KF x 10k landmark x 30 obs problem, "
                               "single thread like options.num_threads = 1 (optimizer.cpp:460)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--seqs", type=int, default=4096, help="sequences processed in lock-step per GPU (16.5 GB of HBM at 4096)")
    ap.add_argument("--workload", choices=["euroc", "kitti"], default="euroc",
                    help="euroc = the headline configuration (BASELINE.json configs[1]); kitti = configs[2] "
                         "(1241x376, wide-image stress) -- a side measurement, not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the BA / detect / single-sequence sections")
    args = ap.parse_args()
    if args.workload == "kitti":
        # KITTI 00 stereo, accurate params: 1241x376, nmaxdist 35 -> nbmaxkps 36*11 = 396
        global W, H, NKPS, N_PASS_A, N_PASS_B, CLAHE_TILES
        W, H, NKPS = 1241, 376, 396
        N_PASS_A = 277; N_PASS_B = NKPS - N_PASS_A
        CLAHE_TILES = (W // 50, H // 50)

    import torch
    import torch.distributed as dist
    import ov2slam_amd
    from ov2slam_amd import _lib as L, optimizer, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    S = args.seqs
    views, kps, pri = make_inputs(S, seed=1234 + rank)          # every rank owns different sequences

    stream = torch.cuda.current_stream()
    ctx = ov2slam_amd.Context(dev.index, stream=stream.cuda_stream)
    lib = ctx.lib
    frames_d = torch.from_numpy(views).to(dev)                   # (NF+1, H, W) shared by the S sequences of this rank
    frames_d = frames_d[:, None].expand(NF + 1, S, H, W).contiguous()
    kps_d = torch.from_numpy(kps).to(dev)
    pri_d = torch.from_numpy(pri).to(dev)
    pri_work = pri_d.clone()
    status_d = torch.zeros((S, NKPS), dtype=torch.uint8, device=dev)
    stats_d = torch.zeros(2, dtype=torch.int64, device=dev)
    nA_d = torch.full((S,), N_PASS_A, dtype=torch.int32, device=dev)
    nB_d = torch.full((S,), N_PASS_B, dtype=torch.int32, device=dev)
    pyrs = [ov2slam_amd.Pyramid(ctx, W, H, WIN, LEVELS, batch=S) for _ in range(2)]

    vp = lambda t: C.c_void_p(t.data_ptr())
    # pre-marshalled arguments: nothing but ctypes calls inside the timed loop
    a_img = [vp(frames_d[f]) for f in range(NF + 1)]
    a_k = [vp(kps_d[f]) for f in range(NF)]
    a_kB = [vp(kps_d[f][:, N_PASS_A:]) for f in range(NF)]
    a_p = [vp(pri_work[f]) for f in range(NF)]
    a_pB = [vp(pri_work[f][:, N_PASS_A:]) for f in range(NF)]
    a_st, a_stB, a_stats, a_nA, a_nB = vp(status_d), vp(status_d[:, N_PASS_A:]), vp(stats_d), vp(nA_d), vp(nB_d)
    hp = [p.h_pyr for p in pyrs]
    fb, build_clahe = lib.ov2_fb_klt_d, lib.ov2_pyr_build_clahe_d
    lk_events = []

    def preprocess(pyr, img):
        # clahe->apply + buildOpticalFlowPyramid (visual_front_end.cpp:1159, :1172) in one call
        L.check(build_clahe(ctx.h, pyr, img, W, W * H, CLAHE_CLIP, CLAHE_TILES[0], CLAHE_TILES[1]))

    def step(i, timed):
        f = i % NF
        prevp, curp = hp[i % 2], hp[(i + 1) % 2]
        if f == 0:
            pri_work.copy_(pri_d)                                   # priors are in/out: restore once per cycle
            preprocess(prevp, a_img[0])
        preprocess(curp, a_img[f + 1])                              # preprocessImage
        if timed:
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(stream)
        # pass A: first N_PASS_A points of every sequence, nbpyrlvl = 1   (visual_front_end.cpp:196)
        L.check(fb(ctx.h, prevp, curp, WIN, 1, 30, 0.01, 30.0, 0.5, a_k[f], a_p[f], NKPS, a_nA, a_st, a_stats))
        if timed:
            e1.record(stream)
        # pass B: the remaining points, nbpyrlvl = 3   (:242)
        L.check(fb(ctx.h, prevp, curp, WIN, LEVELS, 30, 0.01, 30.0, 0.5, a_kB[f], a_pB[f], NKPS, a_nB, a_stB, a_stats))
        if timed:
            e2.record(stream)
            lk_events.append((e0, e1, e2))

    for i in range(args.warmup):
        step(i, False)
    torch.cuda.synchronize()
    stats_d.zero_()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                  # RCCL: 8 bytes, timings only
    elapsed = float(t.item())

    # ---- roofline of the dominant kernel: k_fb_klt3 (lk3.hip) -------------------------------------------
    iters, visits = [int(v) for v in stats_d.tolist()]
    ms_A = sum(a.elapsed_time(b) for a, b, c in lk_events)
    ms_B = sum(b.elapsed_time(c) for a, b, c in lk_events)
    n_launch = 2 * len(lk_events)
    bytes_total = lk_algorithmic_bytes(iters, visits, args.steps * S * NKPS)
    avg_launch_ms = (ms_A + ms_B) / max(1, n_launch)
    achieved = bytes_total / max(1, n_launch) / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    tracked = float(status_d.float().mean().item())

    # HBM traffic of the dominant kernel: PMC counters cannot be collected from inside this process; they come
    # from the separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of tools/profile.sh whose summary
    # is committed under profiles/ (same workload and seqs_per_gpu, else null)
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "lk_traffic.json")))
        if tj.get("seqs_per_gpu") == S and args.workload == "euroc":
            traffic = tj["hbm_bytes_per_launch"]
    except Exception:
        pass

    if rank == 0:
        frames = args.steps * S * world
        out = {
            "metric": "frames/sec tracking + local-BA iters/sec, EuRoC MH_01 stereo 'accurate', 1 GPU vs CPU ref",
            "value": frames / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32 fixed point + f32 (LK), f64 (BA)", "data": "synthetic",
            "config": {"workload": "%s stereo 'accurate' tracking step on synthetic %dx%d frames: CLAHE + pyramid "
                                   "build (4 levels) + fbKltTracking pass A (%d kps, nbpyrlvl 1) + pass B "
                                   "(%d kps, nbpyrlvl 3), 9x9 window, 30 it / 0.01 px"
                                   % ("EuRoC MH_01" if args.workload == "euroc" else "KITTI 00", W, H, N_PASS_A, N_PASS_B),
                       "seqs_per_gpu": S, "keypoints_per_frame": NKPS, "parallelism": "replicas x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "k_fb_klt3", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": avg_launch_ms, "launches": n_launch,
                         "algorithmic_bytes_per_launch": bytes_total / max(1, n_launch),
                         "gn_iterations": iters, "patch_builds": visits},
            "lk_ms_per_step": (ms_A + ms_B) / args.steps, "tracked_fraction": tracked,
        }
        if world == 1 and not args.no_extras:
            # ---- local BA: config[3], problem resident in HBM, robust pass of optimizer.cpp:436-485 --
            pb = synth.make_ba_problem(50, 10000, 30, stereo=False, seed=42)
            rp = optimizer.ResidentProblem(ctx, pb)
            rp.solve()                                               # warm-up
            its, ms, wall0 = 0, 0.0, time.perf_counter()
            for _ in range(5):
                r = rp.solve()
                its += r["iterations"]; ms += r["solve_ms"]
            wall = time.perf_counter() - wall0
            out["ba"] = {"iters_per_s": its / (ms * 1e-3), "iters_per_s_wall_incl_d2h": its / wall,
                         "iterations_per_solve": its / 5, "solve_ms": ms / 5,
                         "workload": "50 KF x 10000 inverse-depth landmarks x 30 obs (290000 residual blocks), Huber "
                                     "sqrt(5.9915), max 5 LM iterations, function_tolerance 1e-3",
                         "termination": optimizer.TERMINATION.get(r["termination"])}
            rp.close()
            # ---- drop-in (host buffer) API on ONE sequence: per-call latency incl. PCIe -------------
            ctx1 = ov2slam_amd.Context(dev.index)
            fx = ov2slam_amd.FeatureExtractor(ctx1, dmaxquality=0.001)
            trk = ov2slam_amd.FeatureTracker(ctx1, 30, 0.01)
            P0 = ov2slam_amd.Pyramid(ctx1, W, H, WIN, LEVELS).build(views[0]); ctx1.sync()
            P1 = ov2slam_amd.Pyramid(ctx1, W, H, WIN, LEVELS)
            roi = (5, 5, W - 10, H - 10)
            fx.detectSingleScale(views[0], CELL, np.zeros((0, 2), np.float32), roi)
            t1 = time.perf_counter()
            for _ in range(20):
                det = fx.detectSingleScale(views[0], CELL, np.zeros((0, 2), np.float32), roi)
            det_ms = (time.perf_counter() - t1) / 20 * 1e3
            t1 = time.perf_counter()
            for _ in range(50):
                P1.build_clahe(views[1], CLAHE_CLIP, CLAHE_TILES[0], CLAHE_TILES[1])            # preprocessImage
                trk.fbKltTracking(P0, P1, WIN, 1, 30., 0.5, kps[0, 0][:N_PASS_A], pri[0, 0][:N_PASS_A])
                trk.fbKltTracking(P0, P1, WIN, LEVELS, 30., 0.5, kps[0, 0][N_PASS_A:], pri[0, 0][N_PASS_A:])
            trk_ms = (time.perf_counter() - t1) / 50 * 1e3
            out["drop_in_single_sequence"] = {"track_ms_per_frame_incl_pcie": trk_ms, "detect_singlescale_ms_incl_pcie": det_ms,
                                              "detected_points": int(len(det))}
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(views, kps, pri, pb)
                cb = out["cpu_baseline"]
                if "ba" in cb:
                    # combined LK-track + local-BA wall-clock per keyframe cycle (5 frames + 1 robust BA pass), CPU / GPU
                    cpu_s = 5.0 / cb["value"] + cb["ba"]["seconds"]
                    gpu_s = 5.0 / out["value"] + out["ba"]["solve_ms"] * 1e-3      # amortised over the S lock-step sequences
                    out["combined_speedup_vs_cpu"] = cpu_s / gpu_s
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
