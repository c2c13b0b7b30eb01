#!/usr/bin/env python3
"""bench.py -- OV2SLAM front-end + local-BA hot path on MI355X (one JSON line on rank 0).

Metric (BASELINE.json): "frames/sec tracking + local-BA iters/sec, EuRoC MH_01 stereo 'accurate'".

    python bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no RANK in the environment makes this process the LAUNCHER: it checks that N GPUs are
visible (fails loudly otherwise), picks a free port on 127.0.0.1 and spawns N ranks of itself (one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), exactly what `python -m torch.distributed.run --nproc-per-node N`
does; under torchrun (RANK already set) it is a rank.  Ranks talk RCCL (`--backend nccl`, default) or gloo
(`--backend gloo --dry`: launcher + collective path only, no GPU -- the CPU test of the N>1 plumbing).

One timed "step" = one camera frame of config[1] (EuRoC MH_01 stereo, 'accurate' parameters: 752x480, CLAHE, LK 9x9,
3+1 pyramid levels, 30 it / 0.01 px, 308 keypoints) pushed through the HIP hot path for each of the `--seqs` sequences
this GPU processes in lock-step (offline batch-of-sequences mode; default 4096 = 16.5 GB of HBM per GPU):
    preprocessImage : CLAHE + device-resident pyramid (/root/reference/src/visual_front_end.cpp:1143-1177)
    kltTracking     : fbKltTracking pass A (nbpyrlvl 1, keypoints with a 3-D prior) + pass B (nbpyrlvl 3)  (:132-275)
`value` = tracked frames/s over all sequences and ranks (max-over-ranks time), inputs resident in HBM.

The same line also carries:
    single_sequence : ONE camera stream through the drop-in entry (ov2_tracker_track_frame: one H2D of the frame, one
                      LK launch, one sync, hipGraph replay) -- PCIe-inclusive per-frame latency and frames/s; this is the
                      mode configs[1], [2] and [4] of BASELINE.json actually use
    config5         : the 11 synthetic EuRoC-length sequences (full length) sharded longest-first over the ranks; a rank's sequences
                      advance in lock-step (tools/lockstep_driver.cpp on ov2_btracker_*), keyframes to per-sequence mapper /
                      estimator contexts; the per-sequence-stream form beside it; aggregate over RCCL all-gather
    ba              : local-BA LM iterations/s on config[3] (50 KF x 10k landmarks x 30 obs) and variants (rank 0, N=1)
    parity          : the tracker's output on one frame vs the oracle (bit-exact) and the BA poses vs the oracle
    cpu_baseline    : the oracle (CPU port of the reference arithmetic) rebuilt -O3 -march=native on this host
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
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
METRIC = "frames/sec tracking + local-BA iters/sec, EuRoC MH_01 stereo 'accurate', 1 GPU vs CPU ref"


N_VIEW_SETS = 64      # distinct image contents a rank's sequences cycle through (VERDICT r5 "weak" 8)


def make_inputs(seqs, seed, uniform=False):
    """min(64, seqs) DIFFERENT view sets of NF+1 synthetic views each -- 16 textures with their own camera walks x 4 photometric variants
    (as drawn, low contrast, a gamma curve with a dark half, and for ONE set a low-entropy rendering: 80 % of the pixels on 8 grey
    levels) -- and per-sequence keypoints / priors from the ground-truth flow of the sequence's set (sequence s shows set s % 64).
    Round 5 showed the same seven views to every sequence, band-limited noise stretched to 0..255: the friendliest input there is.
    Returns (views (sets, NF+1, H, W) uint8, kps, pri (2 NF, seqs, NKPS, 2) float32); with uniform=True also the priors of the same
    keypoints when EVERY sequence shows set 0 (round 5's input, kept as a side measurement of the bench)."""
    from ov2slam_amd import synth
    rng = np.random.default_rng(seed)
    n_sets = max(1, min(N_VIEW_SETS, seqs))
    n_tex = (n_sets + 3) // 4
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    views = np.zeros((n_sets, NF + 1, H, W), np.uint8)
    offs_of = []
    for t in range(n_tex):
        tex = synth.base_texture(1400, seed + 101 * t, nblobs=int(rng.integers(200, 900)))
        offs = []
        ox, oy, th = float(rng.uniform(100, 250)), float(rng.uniform(100, 250)), float(rng.uniform(-0.05, 0.05))
        base = []
        for _ in range(NF + 1):
            base.append(synth.warp(tex, W, H, ox, oy, th)); offs.append((ox, oy, th))
            ox += rng.uniform(-5, 5); oy += rng.uniform(-4, 4); th += rng.uniform(-0.006, 0.006)
        base = np.stack(base).astype(np.float32)
        for v in range(4):
            k = 4 * t + v
            if k >= n_sets:
                break
            if v == 0:
                img = base
            elif v == 1:
                img = base * 0.35 + 90.0                                  # low contrast
            elif v == 2:
                img = 255.0 * (base / 255.0) ** 2.2                        # dark half
            else:
                img = 255.0 - base * 0.8                                  # inverted, compressed
            views[k] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
            offs_of.append(offs)
    if n_sets >= 8:                                                       # ONE low-entropy set: 80 % of the pixels on 8 grey levels near black
        k = n_sets - 1
        sel = rng.uniform(size=views[k].shape) < 0.8
        views[k] = np.where(sel, views[k] // 32, views[k] // 4).astype(np.uint8)

    def flow(offs, pts, a, b):
        (ox0, oy0, t0), (ox1, oy1, t1) = offs[a], offs[b]
        dx, dy = pts[:, 0] - cx, pts[:, 1] - cy
        c, s = np.cos(t0), np.sin(t0)
        tx, ty = c * dx - s * dy + cx + ox0, s * dx + c * dy + cy + oy0
        dx, dy = tx - cx - ox1, ty - cy - oy1
        c, s = np.cos(-t1), np.sin(-t1)
        return np.stack([c * dx - s * dy + cx, s * dx + c * dy + cy], 1)

    # transition t of the ping-pong walk over the views (0 -> 1 -> ... -> NF -> NF-1 -> ... -> 0 -> ...): 2*NF transitions per cycle,
    # every frame's `prev` pyramid is the previous step's `cur` -- no view is ever pre-processed twice in a row
    kps = np.zeros((2 * NF, seqs, NKPS, 2), np.float32)
    pri = np.zeros_like(kps)
    pri_u = np.zeros_like(kps) if uniform else None
    for f in range(2 * NF):
        va, vb = walk_view(f), walk_view(f + 1)
        for s in range(seqs):
            k = synth.grid_keypoints(W, H, CELL, rng)[:NKPS]
            if len(k) < NKPS:
                extra = np.stack([rng.uniform(20, W - 20, NKPS - len(k)), rng.uniform(20, H - 20, NKPS - len(k))], 1)
                k = np.concatenate([k, extra.astype(np.float32)])
            rng.shuffle(k)
            kps[f, s] = k
            noise = rng.normal(0, 1.5, k.shape)
            pri[f, s] = flow(offs_of[s % n_sets], k.astype(np.float64), va, vb) + noise
            if uniform:
                pri_u[f, s] = flow(offs_of[0], k.astype(np.float64), va, vb) + noise
    return (views, kps, pri, pri_u) if uniform else (views, kps, pri)


def walk_view(i):
    """view shown at step i of the ping-pong walk over the NF + 1 views"""
    k = i % (2 * NF)
    return k if k <= NF else 2 * NF - k


def lk_algorithmic_bytes(iters, visits, npts):
    # per (point, level) visit 500 B of template footprint (9x9 bilinear window of the u8 image and
    # the int16x2 derivative), 100 B of the other image per executed GN iteration, 29 B of point I/O per call
    return 500 * visits + 100 * iters + 29 * npts


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(views, kps, pri, ba_problem, budget_s=8.0):
    """Oracle ('port') on the host cores, rebuilt -O3 -march=native here; per slice of the reference's Profiler names
    (BASELINE.md section 2): CLAHE + pyramid over rows / tiles on the pool (cv::parallel_for_), LK over keypoints on the
    pool, BA single-threaded like options.num_threads = 1 (optimizer.cpp:460)."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    flags = O.use_native() or "-O3 -march=x86-64-v2 -ffp-contract=off (native rebuild failed: portable build)"
    p0 = O.Pyramid(O.clahe(views[0], CLAHE_CLIP, *CLAHE_TILES), WIN, LEVELS)
    hp = np.zeros(NKPS, bool); hp[:N_PASS_A] = True

    pbuf = [O.Pyramid(views[0], WIN, LEVELS), O.Pyramid(views[0], WIN, LEVELS)]     # cur_pyr_ / prev_pyr_: Mats re-used every frame

    def pre(slot=1, v=1):
        return pbuf[slot].rebuild(O.clahe(views[v], CLAHE_CLIP, *CLAHE_TILES))

    def lk(p1, nt):
        O.fb_klt(p0, p1, WIN, 1, 30., 0.5, kps[0, 0][:N_PASS_A], pri[0, 0][:N_PASS_A], nthreads=nt)
        O.fb_klt(p0, p1, WIN, LEVELS, 30., 0.5, kps[0, 0][N_PASS_A:], pri[0, 0][N_PASS_A:], nthreads=nt)

    def best(fn, reps=6):
        fn(); ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    # thread count per slice: whatever is fastest on this host (OpenCV would use its pool = all cores)
    cand = [n for n in (1, 2, 4, 8, 16, 32, 64, 128) if n <= cores]
    pre_t, lk_t = {}, {}
    p1 = pre()
    for nt in cand:
        O.set_num_threads(nt)
        pre_t[nt] = best(pre)
        lk_t[nt] = best(lambda: lk(p1, nt))
    nt_pre = min(pre_t, key=pre_t.get); nt_lk = min(lk_t, key=lk_t.get)
    # bounded sample of the whole step
    t0 = time.perf_counter(); frames = 0; t_pre = t_lk = 0.0
    prevp = p0
    while True:
        f = frames % NF
        a = time.perf_counter()
        O.set_num_threads(nt_pre)
        if f == 0:
            prevp = pre(frames % 2, 0)
            a = time.perf_counter()
        curp = pre((frames + 1) % 2, f + 1)
        b = time.perf_counter()
        O.set_num_threads(nt_lk)
        k, p = kps[f, 0], pri[f, 0]
        O.fb_klt(prevp, curp, WIN, 1, 30., 0.5, k[:N_PASS_A], p[:N_PASS_A], nthreads=nt_lk)
        O.fb_klt(prevp, curp, WIN, LEVELS, 30., 0.5, k[N_PASS_A:], p[N_PASS_A:], nthreads=nt_lk)
        c = time.perf_counter()
        t_pre += b - a; t_lk += c - b
        prevp = curp
        frames += 1
        if c - t0 > budget_s and frames >= 20:
            break
    el = t_pre + t_lk
    out = {"value": frames / el, "unit": "frames/s", "cores": max(nt_pre, nt_lk), "kind": "port", "host_cores": cores,
           "build": flags,
           "slices_ms": {"2.FE_TM_preprocessImage": t_pre / frames * 1e3, "2.FE_TM_KLT-Tracking": t_lk / frames * 1e3},
           "threads": {"2.FE_TM_preprocessImage": nt_pre, "2.FE_TM_KLT-Tracking": nt_lk, "2.BA_Optimize": 1},
           # every thread count that was timed, median ms per call: the slice runs on its fastest one
           "threads_tried": {"2.FE_TM_preprocessImage": {str(n): pre_t[n] * 1e3 for n in cand}, "2.FE_TM_KLT-Tracking": {str(n): lk_t[n] * 1e3 for n in cand}},
           "sample": "%d frames of the same synthetic 752x480 step through the oracle (a C restatement of the OpenCV / Ceres "
                     "arithmetic the reference calls -- NOT OpenCV / Ceres themselves, which are absent from this image), "
                     "persistent thread pool, CLAHE + pyramid over tiles / rows on %d threads, LK over keypoints on %d threads "
                     "(fastest of %s each)" % (frames, nt_pre, nt_lk, cand)}
    if ba_problem is not None:
        O.set_num_threads(1)
        t0 = time.perf_counter()
        r = O.ba_solve(ba_problem)
        el_ba = time.perf_counter() - t0
        out["ba"] = {"iters_per_s": r["iterations"] / el_ba, "iterations": r["iterations"], "seconds": el_ba, "cores": 1,
                     "slice": "2.BA_Optimize",
                     "sample": "one robust pass (<=5 LM iterations) of the 50 KF x 10k landmark x 30 obs problem, dense Schur "
                               "complement, single thread like options.num_threads = 1 (optimizer.cpp:460)"}
        out["_ba_result"] = r
    # ---- the same restatement under relaxed floating point (vectorisable, FMA, upper-triangle block Schur): a SPEED baseline,
    #      not the checker -- so that ratios are not only quoted against the strict-order build nobody would ship
    fflags = O.use_fast()
    if fflags:
        O.set_num_threads(nt_pre); p1f = pre(); tp = best(pre)
        O.set_num_threads(nt_lk); tl = best(lambda: lk(p1f, nt_lk))
        simd = {"value": 1.0 / (tp + tl), "unit": "frames/s", "cores": max(nt_pre, nt_lk), "kind": "port, relaxed-FP build", "build": fflags,
                "slices_ms": {"2.FE_TM_preprocessImage": tp * 1e3, "2.FE_TM_KLT-Tracking": tl * 1e3},
                "note": "NOT bit-exact with the checker (agrees to rounding); OpenCV's hand-written SIMD LK / pyrDown and Ceres' "
                        "Eigen kernels would still be faster than compiler-vectorised C: treat GPU / CPU ratios as upper bounds"}
        if ba_problem is not None:
            O.set_num_threads(1)
            t0 = time.perf_counter(); rf = O.ba_solve(ba_problem); el = time.perf_counter() - t0
            simd["ba"] = {"iters_per_s": rf["iterations"] / el, "iterations": rf["iterations"], "seconds": el, "cores": 1,
                          "pose_max_abs_diff_vs_strict_build": float(np.abs(rf["poses"] - out["_ba_result"]["poses"]).max())}
        out["_simd"] = simd
        O.use_native()
    return out


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(args):
    """Spawn args.gpus ranks of this script (one per GPU) and relay rank 0's JSON line."""
    n = args.gpus
    if not args.dry:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py --gpus %d: only %d GPU(s) visible -- refusing to run fewer ranks than requested\n" % (n, have))
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for r, p in enumerate(procs):
        c = p.wait()
        if c != 0:
            sys.stderr.write("bench.py: rank %d exited with code %d\n" % (r, c))
            rc = rc or c
    return rc


# ------------------------------------------------------------------------------------------------ who is here
def ranks_and_devices(world, device_index, dry):
    """Proof that the collective backend saw `world` ranks on `world` distinct GPUs: an all-reduce(SUM) of ones over the process group
    (RCCL when the backend is nccl) and every rank's PCI bus id, all-gathered.  Returns (ranks_seen, [bus id per rank])."""
    import torch
    import torch.distributed as dist
    bus = "n/a (dry run, no GPU)"
    if not dry:
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id"), getattr(pr, "pci_bus_id"), getattr(pr, "pci_device_id"))
        except Exception:
            try:
                hip = C.CDLL("libamdhip64.so")
                buf = C.create_string_buffer(64)
                bus = buf.value.decode() if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) == 0 else "unknown"
            except Exception:
                bus = "unknown"
    if world == 1 or not (dist.is_available() and dist.is_initialized()):
        return 1, [bus]
    one = torch.ones(1, dtype=torch.float64, device=("cuda:%d" % device_index) if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    ids = [None] * world
    dist.all_gather_object(ids, bus)
    return int(round(float(one.item()))), ids


# ------------------------------------------------------------------------------------------------ config 5
def config5_plan(world, scale):
    from ov2slam_amd import batch
    counts = {k: max(12, v // scale) for k, v in batch.EUROC_FRAMES.items()}
    return counts, batch.assign_sequences(counts, world)


def run_config5(ctx, rank, world, scale, dry=False, device=0, concurrency=2, stream_scale=16):
    """This rank's share of the 11 EuRoC-length synthetic sequences, ALL OF THEM IN LOCK-STEP through one native host process
    (tools/lockstep_driver.cpp on ov2_btracker_*: one enqueue per frame step covers every sequence of the rank; keyframes go to
    per-sequence mapper / estimator contexts), at the full frame counts (scale 1).  Beside it, for comparison, the same rank's
    sequences at 1/stream_scale length through the per-sequence stream driver (tools/stream_driver.cpp, `concurrency` at a time).
    All-gather of the counters over torch.distributed -- the only collective."""
    from ov2slam_amd import batch, stream
    counts, plan = config5_plan(world, scale)
    mine = plan[rank]
    keys = ("frames", "seconds", "tracked", "attempted", "ate_sq_sum", "ate_n", "sequences", "host_native", "keyframes", "stereo_ok", "stereo_kps",
            "ba_solves", "ba_iterations", "ba_seconds", "ba_skipped", "device", "steps", "slam_library_s", "slam_wait_loader_s", "slam_wait_mapper_s",
            "all_frames", "all_seconds", "all_ba_solves", "all_ba_iterations", "all_keyframes",
            "stream_frames", "stream_seconds", "stream_ba_solves", "stream_ba_skipped", "stream_keyframes",
            "ba_batches", "thr_frames", "thr_seconds", "thr_ba_solves", "run_seconds_min", "run_seconds_max")
    loc = {k: 0.0 for k in keys}
    loc["sequences"] = float(len(mine)); loc["device"] = float(device)
    argv_note = stream.lockstep_argv("lockstep_driver", ["<case:%s>" % s for s in mine], "newest", device)
    if dry:
        loc["frames"] = float(sum(counts[s] for s in mine)); loc["seconds"] = 1.0 + 0.25 * rank
        loc["tracked"] = loc["attempted"] = 300.0 * loc["frames"]; loc["steps"] = float(max([counts[s] for s in mine] + [0]))
        loc["keyframes"] = loc["frames"] / 5; loc["ba_solves"] = loc["keyframes"]; loc["ba_iterations"] = 5 * loc["ba_solves"]; loc["ba_seconds"] = 0.5
    elif mine:
        from ov2slam_amd import synth
        import tempfile
        tex = synth.base_texture(1400, 1234)
        names = sorted(batch.EUROC_FRAMES)
        windows = [synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7 + i) for i in range(2)]    # local-BA windows (optimizer.cpp:150-188)
        td = tempfile.TemporaryDirectory()
        exe_l = stream.build_native_driver(td.name, "lockstep_driver")            # no Python fallback: the lock-step host IS the measurement
        exe_s = stream.build_native_driver(td.name, "stream_driver")
        loc["host_native"] = 1.0

        seqs = [batch.SyntheticSequence(sname, counts[sname], seed=1000 + names.index(sname), tex=tex, stereo=True) for sname in mine]   # generation untimed

        def cases_for(cnts, tag):
            out = []
            for i, (sname, sq) in enumerate(zip(mine, seqs)):
                sq.n_frames = int(cnts[sname])                                    # the views cycle: the length is a header field
                out.append(os.path.join(td.name, "%s%d.bin" % (tag, i))); stream.write_case(out[-1], sq, windows)
            return out
        cases = cases_for(counts, "full")
        warm = cases_for({k: 40 for k in counts}, "warm")
        stream.run_lockstep(exe_l, warm, device=device)                               # warm-up (page cache, clocks, code objects)
        # three full-length runs; the MEDIAN one is reported (every run's fps is listed: the SLAM thread shares the command processor
        # with the estimator's batches, run-to-run spread ~ +-8 %)
        runs = sorted((stream.run_lockstep(exe_l, cases, device=device, ba_policy="newest") for _ in range(3)), key=lambda r: r[1]["seconds"])
        stats, summ = runs[1]
        loc["run_seconds_min"], loc["run_seconds_max"] = runs[0][1]["seconds"], runs[2][1]["seconds"]
        loc["frames"] = summ["frames"]; loc["seconds"] = summ["seconds"]; loc["steps"] = summ["steps"]
        loc["slam_library_s"] = summ["slam_library_s"]; loc["slam_wait_loader_s"] = summ["slam_wait_for_loader_s"]; loc["slam_wait_mapper_s"] = summ["slam_wait_for_mapper_s"]
        for st in stats:
            loc["tracked"] += st["tracked"]; loc["attempted"] += st["attempted"]; loc["ate_sq_sum"] += st["err_sq_sum"]; loc["ate_n"] += st["err_n"]
            loc["keyframes"] += st["keyframes"]; loc["stereo_ok"] += st["stereo_ok"]; loc["stereo_kps"] += st["stereo_kps"]
            loc["ba_solves"] += st["ba_solves"]; loc["ba_iterations"] += st["ba_iterations"]
            loc["ba_skipped"] += st["ba_skipped_kfs"]
        loc["ba_seconds"] = summ["seconds"]                                          # the estimator runs beside the SLAM thread: wall clock
        loc["ba_batches"] = summ.get("ba_batches", 0)
        # equal work for CPU / GPU comparisons: every keyframe gets its localBA (the estimator contexts bound this one)
        # (four estimator groups: while one group's batch is on the GPU the next group's problems are staged -- the best form when the
        # estimator bounds the run, profiles/r5_lockstep_estimator_groups.json)
        stats_a, summ_a = stream.run_lockstep(exe_l, cases, device=device, ba_policy="all", batched_estimator=4)
        loc["all_frames"] = summ_a["frames"]; loc["all_seconds"] = summ_a["seconds"]
        loc["all_ba_solves"] = sum(st["ba_solves"] for st in stats_a); loc["all_ba_iterations"] = sum(st["ba_iterations"] for st in stats_a)
        loc["all_keyframes"] = sum(st["keyframes"] for st in stats_a)
        # the first lock-step form of this round (an estimator thread + context per sequence calling ov2_local_ba, stream priorities), full length
        stats_t, summ_t = stream.run_lockstep(exe_l, cases, device=device, ba_policy="newest", priorities=True, batched_estimator=False)
        loc["thr_frames"] = summ_t["frames"]; loc["thr_seconds"] = summ_t["seconds"]; loc["thr_ba_solves"] = sum(st["ba_solves"] for st in stats_t)
        # the per-sequence stream form (round 4's config 5) at reduced length, for the comparison
        small = cases_for({k: max(12, v // max(1, stream_scale)) for k, v in batch.EUROC_FRAMES.items()}, "small")
        st_s, sec_s = stream.run_native_concurrent(exe_s, small, device=device, concurrency=max(1, concurrency))
        loc["stream_frames"] = sum(st["frames"] for st in st_s); loc["stream_seconds"] = sec_s
        loc["stream_ba_solves"] = sum(st["ba_solves"] for st in st_s); loc["stream_ba_skipped"] = sum(st["ba_skipped_kfs"] for st in st_s)
        loc["stream_keyframes"] = sum(st["keyframes"] for st in st_s)
        td.cleanup()
    stats = batch.gather_stats(loc)
    agg = batch.aggregate(stats)
    sec_all, sec_str, sec_thr = max(stats["all_seconds"]), max(stats["stream_seconds"]), max(stats["thr_seconds"])
    return {"workload": "11 synthetic stereo sequences with EuRoC frame counts / %d (%d frames), longest-first over %d rank(s); the sequences of a rank "
                        "advance IN LOCK-STEP through tools/lockstep_driver.cpp: per frame step ONE ov2_btracker_track_frame (frame upload, CLAHE + "
                        "pyramid, fused kltTracking, computeKeypoint for every sequence of the rank), every 5th step one batched detectSingleScale; "
                        "keyframes go to the rank's mapper thread (one batched right-image CLAHE + pyramid, one ov2_stereo_match_batch) and to the "
                        "rank's estimator thread (ONE ov2_local_ba_batch per round over the sequences with a keyframe waiting: two-pass 25-KF "
                        "localBA, per sequence the newest keyframe only like estimator.cpp:195-205); sequences that end drop out"
                        % (scale, int(sum(counts.values())), world),
            "fps": agg["fps"], "frames": agg["frames"], "seconds_slowest_rank": agg["seconds"],
            "runs": {"n": 3, "reported": "median", "seconds_timed_total_slowest_rank": (max(stats["run_seconds_min"]) + agg["seconds"] + max(stats["run_seconds_max"])) if not dry else None,
                     "fps_fastest_run": agg["frames"] / max(stats["run_seconds_min"]) if max(stats["run_seconds_min"]) > 0 else None,
                     "fps_slowest_run": agg["frames"] / max(stats["run_seconds_max"]) if max(stats["run_seconds_max"]) > 0 else None},
            "frames_per_rank": stats["frames"], "seconds_per_rank": stats["seconds"], "sequences_per_rank": stats["sequences"],
            "steps_per_rank": stats["steps"],
            "device_per_rank": [int(d) for d in stats["device"]],
            "slam_thread_per_rank": {"library_s": stats["slam_library_s"], "wait_for_loader_s": stats["slam_wait_loader_s"], "wait_for_mapper_s": stats["slam_wait_mapper_s"]},
            "tracked_fraction": sum(stats["tracked"]) / max(1.0, sum(stats["attempted"])),
            "keyframes": sum(stats["keyframes"]), "stereo_ok_fraction": sum(stats["stereo_ok"]) / max(1.0, sum(stats["stereo_kps"])),
            "ba_solves": sum(stats["ba_solves"]), "ba_keyframes_skipped_while_busy": sum(stats["ba_skipped"]), "ba_batches": sum(stats["ba_batches"]),
            "ba_iters_per_s": agg.get("ba_iters_per_s", 0.0),
            "every_keyframe_optimised": {"estimator_groups": 4, "fps": sum(stats["all_frames"]) / sec_all if sec_all > 0 else None, "seconds_slowest_rank": sec_all,
                                         "ba_solves": sum(stats["all_ba_solves"]), "keyframes": sum(stats["all_keyframes"]),
                                         "ba_iters_per_s": sum(stats["all_ba_iterations"]) / sec_all if sec_all > 0 else None},
            "estimator_thread_per_sequence": {"what": "the same lock-step front end with an estimator thread + context per sequence calling ov2_local_ba "
                                                      "(stream priorities on), full length: this round's first form",
                                              "fps": sum(stats["thr_frames"]) / sec_thr if sec_thr > 0 else None, "seconds_slowest_rank": sec_thr,
                                              "ba_solves": sum(stats["thr_ba_solves"])},
            "per_sequence_streams": {"what": "round 4's form of this config: each sequence through its own SLAM thread + ov2_tracker (tools/stream_driver.cpp), "
                                             "%d at a time per GPU, frame counts / %d" % (int(concurrency), int(stream_scale)),
                                     "fps": sum(stats["stream_frames"]) / sec_str if sec_str > 0 else None, "frames": sum(stats["stream_frames"]),
                                     "seconds_slowest_rank": sec_str, "ba_solves": sum(stats["stream_ba_solves"]),
                                     "ba_keyframes_skipped_while_busy": sum(stats["stream_ba_skipped"]), "keyframes": sum(stats["stream_keyframes"])},
            "lockstep_speedup_vs_streams": (agg["fps"] / (sum(stats["stream_frames"]) / sec_str)) if sec_str > 0 and sum(stats["stream_frames"]) > 0 else None,
            # how to read a multi-GPU run of this config: with the 11 sequences spread over N GPUs a rank holds 11 / N of them; at 1-2
            # sequences per rank the lock-step form has nothing to batch and a rank delivers about the per-stream rate -- expect ~N x that,
            # NOT N x the 11-sequence figure above, and nothing like the 4096-sequence headline (that is another workload)
            "expected_fps_per_gpu_at_1_to_2_sequences": ((sum(stats["stream_frames"]) / sec_str) / max(1, int(concurrency)) * 1.0 if sec_str > 0 and sum(stats["stream_frames"]) > 0 else None),
            "expected_note": "per-stream rate of one sequence on one GPU (per_sequence_streams.fps / concurrency): what a rank of an 8-GPU run of this config, "
                             "holding 1-2 sequences, can be expected to deliver; launch-bound (profiles/r5_final_lockstep_kernel_gaps.txt)",
            "host": "tools/lockstep_driver.cpp (native, one process per rank)" if min(stats["host_native"]) > 0 or dry else "none",
            "lockstep_argv_rank0": argv_note,
            "parity": "per-sequence results of the lock-step form are bit-identical to the per-stream form (digests of every tracked position, status, "
                      "undistorted pixel, bearing, detection and stereo match): tests/test_gpu_stream.py",
            "track_rmse_px": agg.get("ate_rmse", 0.0),
            "ate": "not computable on THIS run: pose estimation (P3P, motion model) and triangulation stay on the CPU in the reference and are "
                   "outside SURVEY.md section 8, and no dataset exists offline; track_rmse_px is the tracking error against the synthetic flow.  "
                   "A pose trajectory through the section-8 functions (preprocessImage + kltTracking + ceresPnP per frame; detectSingleScale + "
                   "stereoMatching + a two-pass localBA over the last eight keyframes per keyframe; exact synthetic ground truth) is "
                   "tools/ate_synthetic.py / tests/test_gpu_ate.py: profiles/r6_ate_synthetic.json -- 401 frames, 7.6 m path, 80 local BAs of "
                   "~5000 blocks: ATE 0.196 mm on the GPU = 0.196 mm on the oracle (difference 8e-20 m, every pose within 1e-15 m, identical "
                   "tracked / matched / gated point counts, BA iteration counts and outlier blocks); without the BA 0.152 mm on both",
            "assignment": plan}


# ------------------------------------------------------------------------------------------------ extras (rank 0, N = 1)
def single_sequence(dev_index, views, kps, pri, n_frames=600):
    """ONE camera stream through the drop-in entry point, host buffers in, host buffers out, PCIe and the sync included.
    `pageable`: the frame comes from an ordinary numpy array (the library stages it through its pinned buffer);
    `pinned`: the frame was written into ov2_tracker_image_buffer() by its producer (no host-side copy)."""
    import ov2slam_amd
    from ov2slam_amd import _lib as L
    ctx = ov2slam_amd.Context(dev_index)
    lib = ctx.lib
    hp = np.zeros(NKPS, np.uint8); hp[:N_PASS_A] = 1
    res = {}
    for mode in ("pageable", "pinned"):
        trk = ov2slam_amd.VisualFrontEndTracker(ctx, W, H, fclahe_val=CLAHE_CLIP, nbmaxkps=512, use_graph=True)
        pris = [np.ascontiguousarray(np.where(hp[:, None] > 0, pri[f, 0], kps[f, 0]), np.float32) for f in range(NF)]
        out = np.zeros((NKPS, 2), np.float32); st = np.zeros(NKPS, np.uint8); p3p = C.c_int(0)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        a_k = [vp(kps[f, 0]) for f in range(NF)]; a_p = [vp(p) for p in pris]
        a_hp, a_out, a_st, a_p3p = vp(hp), vp(out), vp(st), C.byref(p3p)
        imgs = [np.ascontiguousarray(v) for v in views]
        a_img = [vp(v) for v in imgs]
        buf = trk.image_buffer
        a_buf = vp(buf)
        fn, h = lib.ov2_tracker_track_frame, trk.h_trk
        ts, tracked = [], 0
        for i in range(n_frames + 30):
            f = i % NF
            if f == 0:                                                     # restart the cycle: (re)build the pyramid of view 0
                L.check(fn(h, a_img[0], W, None, None, None, 0, 1, None, None, None))
            if mode == "pinned":
                buf[:, :W] = imgs[f + 1]                                   # the producer's write, not part of the call
                t0 = time.perf_counter()
                rc = fn(h, a_buf, trk.stride, a_k[f], a_p[f], a_hp, NKPS, 1, a_out, a_st, a_p3p)
            else:
                t0 = time.perf_counter()
                rc = fn(h, a_img[f + 1], W, a_k[f], a_p[f], a_hp, NKPS, 1, a_out, a_st, a_p3p)
            t1 = time.perf_counter()
            L.check(rc)
            if i >= 30:
                ts.append(t1 - t0); tracked += int((st & 1).sum())
        ts = np.array(ts)
        res[mode] = {"fps_incl_pcie": float(1.0 / ts.mean()), "ms_per_frame_mean": float(ts.mean() * 1e3),
                     "ms_per_frame_median": float(np.median(ts) * 1e3), "ms_per_frame_p99": float(np.percentile(ts, 99) * 1e3),
                     "tracked_fraction": tracked / (len(ts) * NKPS), "hip_graph": trk.uses_graph}
        trk.close()
    ctx.close()
    res["entry"] = "ov2_tracker_track_frame: preprocessImage + kltTracking, 1 H2D (361 KB) + 5 kernels + 1 fused LK launch, 1 sync"
    res["frames"] = n_frames
    return res


def config2_stream(dev_index, n_frames=400, kf_every=5, with_cpu=True, cpu_frames=40):
    """BASELINE.json configs[1] as the reference runs it -- ONE camera, three threads on three contexts (ov2slam_amd/stream.py):
    front-end per frame, detection + stereo matching per keyframe, two-pass localBA per keyframe, concurrently -- measured as
    one stream, and the CPU oracle executing the same schedule stage by stage (the reference runs the stages on three threads:
    its pipeline rate is bounded by the slowest stage, its single-thread cost is their sum)."""
    import ov2slam_amd
    from ov2slam_amd import batch, stream, synth
    tex = synth.base_texture(1400, 1234)
    seq = batch.SyntheticSequence("MH_01", n_frames, seed=1000, tex=tex, stereo=True)
    windows = [synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7 + i) for i in range(2)]
    ctx = ov2slam_amd.Context(dev_index)
    stream.run_stream(ctx, batch.SyntheticSequence("warm", 12, seed=5, tex=tex, stereo=True), kf_every=kf_every, ba_problems=windows)
    st = stream.run_stream(ctx, seq, kf_every=kf_every, ba_problems=windows)
    st_all = stream.run_stream(ctx, seq, kf_every=kf_every, ba_problems=windows, ba_policy="all")
    trk_only = stream.run_stream(ctx, batch.SyntheticSequence("MH_01", n_frames, seed=1000, tex=tex), kf_every=kf_every, do_stereo=False)
    ctx.close()
    # the same cycle with a native host: tools/stream_driver.cpp (three std::threads, nothing but the C ABI)
    native = None
    try:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            exe = stream.build_native_driver(td)
            case = os.path.join(td, "case.bin")
            stream.write_case(case, seq, windows, kf_every=kf_every)
            stream.run_native(exe, case, "newest", dev_index)
            n1, n2 = stream.run_native(exe, case, "newest", dev_index), stream.run_native(exe, case, "all", dev_index)
        native = {"frames_per_s": n1["frames"] / n1["seconds"], "frames_per_s_slam_thread": n1["frames"] / n1["slam_thread_seconds"],
                  "frames_per_s_every_keyframe_optimised": n2["frames"] / n2["seconds"],
                  "slam_thread_ms_per_frame": {"total": n1["slam_thread_seconds"] / n1["frames"] * 1e3, "inside_library_calls": n1["slam_library_s"] / n1["frames"] * 1e3},
                  "mapper_ms_per_keyframe": n1["mapper_busy_s"] / max(1, n1["stereo_kfs"]) * 1e3,
                  "ba_solves": n1["ba_solves"], "ba_keyframes_skipped_while_busy": n1["ba_skipped_kfs"],
                  "ba_wall_ms_per_solve": n1["ba_busy_s"] / max(1, n1["ba_solves"]) * 1e3,
                  "ba_iters_per_s_wall": n1["ba_iterations"] / max(1e-9, n1["ba_busy_s"]),
                  "tracked_fraction": n1["tracked"] / max(1, n1["attempted"]), "track_rmse_px": (n1["err_sq_sum"] / max(1, n1["err_n"])) ** 0.5,
                  "stereo_ok_fraction": n1["stereo_ok"] / max(1, n1["stereo_kps"]),
                  "host": "tools/stream_driver.cpp: g++ -O2, three std::threads on three contexts, the C ABI only (own random stream: same statistics, "
                          "not the same noise as the Python driver)"}
    except Exception as e:
        native = {"error": repr(e)[-400:]}
    out = {"workload": "one synthetic EuRoC-sized stereo stream, %d frames, keyframe every %d: SLAM thread (track_frame + computeKeypoint "
                       "per frame, detectSingleScale + computeKeypoint per keyframe) / mapper thread (right CLAHE + pyramid + "
                       "ov2_stereo_match per keyframe) / estimator thread (ov2_local_ba on a 25 KF x 3000 landmark x 12 obs stereo "
                       "window, newest keyframe only) on three contexts of one GPU, host buffers in and out" % (n_frames, kf_every),
           "frames_per_s": st["frames"] / st["seconds"], "frames_per_s_slam_thread": st["frames"] / st["slam_thread_seconds"],
           "frames_per_s_every_keyframe_optimised": st_all["frames"] / st_all["seconds"],
           "ba_solves_every_keyframe_optimised": st_all["ba_solves"],
           "frames_per_s_front_end_alone": trk_only["frames"] / trk_only["seconds"],
           "slam_thread_ms_per_frame": {"total": st["slam_thread_seconds"] / st["frames"] * 1e3,
                                        "inside_library_calls": st["slam_library_s"] / st["frames"] * 1e3,
                                        "note": "the rest is this Python driver (synthetic flow / priors / bookkeeping standing in for "
                                                "the reference's CPU-side pose estimation and map updates)"},
           "keyframes": st["keyframes"], "mapper_ms_per_keyframe": st["mapper_busy_s"] / max(1, st["stereo_kfs"]) * 1e3,
           "slam_thread_waited_for_mapper_ms_total": st["slam_wait_for_mapper_s"] * 1e3,
           "stereo_ok_fraction": st["stereo_ok"] / max(1, st["stereo_kps"]),
           "ba_solves": st["ba_solves"], "ba_keyframes_skipped_while_busy": st["ba_skipped_kfs"],
           "ba_wall_ms_per_solve": st["ba_busy_s"] / max(1, st["ba_solves"]) * 1e3,
           "ba_device_ms_per_solve": st["ba_device_ms"] / max(1, st["ba_solves"]),
           "ba_iters_per_s_wall": st["ba_iterations"] / max(1e-9, st["ba_busy_s"]),
           "tracked_fraction": st["tracked"] / max(1, st["attempted"]),
           "track_rmse_px": (st["err_sq_sum"] / max(1, st["err_n"])) ** 0.5}
    out["python_driver_frames_per_s"] = out["frames_per_s"]
    out["native_host"] = native
    if native and "frames_per_s" in native:
        # the figures of record are the native host's: the reference's host side is C++ too
        out["frames_per_s"] = native["frames_per_s"]
        out["frames_per_s_every_keyframe_optimised"] = native["frames_per_s_every_keyframe_optimised"]
    if with_cpu:
        out["cpu_same_schedule"] = cpu_stream(batch.SyntheticSequence("MH_01", cpu_frames, seed=1000, tex=tex, stereo=True), windows, kf_every)
        c = out["cpu_same_schedule"]
        # equal work on both sides: every keyframe gets its stereo matching and its two-pass localBA
        out["speedup_vs_cpu_pipeline_bound"] = out["frames_per_s_every_keyframe_optimised"] / c["frames_per_s_pipeline_bound"]
        out["speedup_vs_cpu_serial"] = out["frames_per_s_every_keyframe_optimised"] / c["frames_per_s_serial"]
        out["front_end_and_mapper_speedup_vs_cpu"] = out["frames_per_s"] / c["frames_per_s_front_end_and_mapper_bound"]
    return out


def cpu_stream(seq, windows, kf_every):
    """The schedule of ov2slam_amd.stream.run_stream on the oracle (CPU restatement of the OpenCV / Ceres arithmetic), stage by
    stage: front-end per frame on the fastest thread count, detection + stereo matching per keyframe, two-pass localBA per
    keyframe single-threaded like options.num_threads = 1 (optimizer.cpp:460)."""
    from oracle import oracle as O
    import ov2slam_amd
    O.build()
    O.use_native()
    cores = os.cpu_count() or 1
    nt = min(16, cores)
    O.set_num_threads(nt)
    w, h = seq.w, seq.h
    rng = np.random.default_rng(seq.seed + 17)
    roi = (5, 5, w - 10, h - 10)
    K = (458.654, 457.296, 367.215, 248.375); D = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)
    iK = np.linalg.inv(np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1.0]]))
    t_fe = t_det = t_st = t_ba = 0.0

    def pre(img):
        return O.Pyramid(O.clahe(img, CLAHE_CLIP, w // 50, h // 50), WIN, LEVELS)

    def oracle_solver(prob, res_active, chi2_init, depthpos_init, **kw):
        return O.ba_solve(prob, O.ba_default_options(**kw), res_active, chi2_init, depthpos_init)
    opt = ov2slam_amd.Optimizer(None, solver=oracle_solver)
    a = time.perf_counter()
    prevp = pre(seq.frame(0))
    t_fe += time.perf_counter() - a
    a = time.perf_counter()
    kps, q = O.detect_singlescale(O.clahe(seq.frame(0), CLAHE_CLIP, w // 50, h // 50), CELL, np.zeros((0, 2), np.float32), roi, 0.001, True)
    t_det += time.perf_counter() - a
    age = np.zeros(len(kps), np.int32)
    nkf, nba, its = 1, 0, 0
    for f in range(1, seq.n_frames):
        gt = seq.flow(kps, f - 1, f)
        hp = age > 0
        pri = np.where(hp[:, None], gt + rng.normal(0, 1.5, gt.shape), kps).astype(np.float32)
        a = time.perf_counter()
        img = O.clahe(seq.frame(f), CLAHE_CLIP, w // 50, h // 50)
        curp = O.Pyramid(img, WIN, LEVELS)
        out, ok, _, _ = O.klt_tracking(prevp, curp, kps, pri, hp)
        O.compute_keypoints(O.CAM_PINHOLE, K, D, iK, out[ok])
        t_fe += time.perf_counter() - a
        kps, age = out[ok], age[ok] + 1
        inside = (kps[:, 0] > 8) & (kps[:, 0] < w - 9) & (kps[:, 1] > 8) & (kps[:, 1] < h - 9)
        kps, age = kps[inside], age[inside]
        prevp = curp
        if f % kf_every == 0:
            nkf += 1
            a = time.perf_counter()
            new, q = O.detect_singlescale(img, CELL, kps, roi, q, True)
            new = new[:max(0, NKPS - len(kps))]
            t_det += time.perf_counter() - a
            kps = np.concatenate([kps, new]); age = np.concatenate([age, np.zeros(len(new), np.int32)])
            a = time.perf_counter()
            unpx, _ = O.compute_keypoints(O.CAM_PINHOLE, K, D, iK, kps)
            pr = pre(seq.right_frame(f))
            p3 = {int(i): (kps[i, 0] - seq.disparity + rng.normal(0, 1.0), kps[i, 1] + rng.normal(0, 1.0)) for i in np.nonzero(age > 0)[0]}
            O.stereo_matching(curp, pr, kps, unpx, O.CAM_PINHOLE, K, D, True, priors3d=p3)
            t_st += time.perf_counter() - a
            a = time.perf_counter()
            O.set_num_threads(1)
            r = opt.localBA(windows[nba % len(windows)])
            O.set_num_threads(nt)
            t_ba += time.perf_counter() - a
            nba += 1; its += r["pass1"]["iterations"] + (r["pass2"]["iterations"] if r["l2_done"] else 0)
    n = seq.n_frames
    per_frame = {"front_end": t_fe / n, "detect": t_det / n, "stereo": t_st / n, "local_ba": t_ba / n}
    slam = per_frame["front_end"] + per_frame["detect"]                 # the SLAM thread runs both
    return {"frames": n, "keyframes": nkf, "threads_front_end": nt, "threads_ba": 1, "kind": "port",
            "ms_per_frame": {k: v * 1e3 for k, v in per_frame.items()},
            "ms_per_keyframe": {"detect": t_det / nkf * 1e3, "stereo": t_st / max(1, nkf - 1) * 1e3, "local_ba": t_ba / max(1, nba) * 1e3},
            "ba_iters_per_s": its / max(1e-9, t_ba),
            "frames_per_s_serial": n / (t_fe + t_det + t_st + t_ba),
            "frames_per_s_pipeline_bound": 1.0 / max(slam, per_frame["stereo"], per_frame["local_ba"]),
            "frames_per_s_front_end_and_mapper_bound": 1.0 / max(slam, per_frame["stereo"]),
            "note": "oracle = C restatement of the OpenCV / Ceres arithmetic (not OpenCV / Ceres); stages timed one after the other; "
                    "pipeline_bound = rate of the slowest of the reference's three threads if they overlapped perfectly"}


def parity_check(dev_index, views, kps, pri):
    """In-bench parity: the tracker's output for two frames vs the oracle's preprocessImage + kltTracking (bit-exact)."""
    import ov2slam_amd
    from oracle import oracle as O
    O.build()
    ctx = ov2slam_amd.Context(dev_index)
    trk = ov2slam_amd.VisualFrontEndTracker(ctx, W, H, fclahe_val=CLAHE_CLIP, nbmaxkps=512, use_graph=True)
    hp = np.zeros(NKPS, np.uint8); hp[:N_PASS_A] = 1
    empty = np.zeros((0, 2), np.float32)
    trk.trackFrame(views[0], empty, empty, None)
    ok_all, n = True, 0
    prevp = O.Pyramid(O.clahe(views[0], CLAHE_CLIP, *CLAHE_TILES), WIN, LEVELS)
    for f in range(2):
        k = kps[f, 0]; p = np.where(hp[:, None] > 0, pri[f, 0], k).astype(np.float32)
        gout, gst, _ = trk.trackFrame(views[f + 1], k, p, hp)
        curp = O.Pyramid(O.clahe(views[f + 1], CLAHE_CLIP, *CLAHE_TILES), WIN, LEVELS)
        rout, rok, rret, _ = O.klt_tracking(prevp, curp, k, p, hp)
        ok_all &= bool(np.array_equal((gst & 1).astype(bool), rok) and np.array_equal((gst & 2).astype(bool), rret)
                       and np.array_equal(gout.view(np.uint32), rout.view(np.uint32)))
        n += len(k); prevp = curp
    trk.close(); ctx.close()
    # the oracle (and therefore the HIP path) accumulates the LK sums in int64; stock OpenCV builds accumulate in float: the measured
    # distance on this frame pair (the 44 856-track campaign is profiles/archive/r3_lk_acc_modes.json, tools/lk_acc_campaign.py)
    acc = O.lk_acc_mode_report(O.Pyramid(O.clahe(views[0], CLAHE_CLIP, *CLAHE_TILES), WIN, LEVELS),
                               O.Pyramid(O.clahe(views[1], CLAHE_CLIP, *CLAHE_TILES), WIN, LEVELS), kps[0, 0],
                               np.where(hp[:, None] > 0, pri[0, 0], kps[0, 0]).astype(np.float32))
    worst = {"status_flips": max(m["status_flips"] for m in acc["modes"].values()),
             "max_abs_dpx": max(m["max_abs_dpx"] for m in acc["modes"].values()), "points": acc["points"]}
    # the detector: the GPU's keyframe detection on this frame vs the oracle (bit-exact), and the oracle's distance to the arithmetic
    # variants another OpenCV build would run (blur rounding, getRectSubPix path, accumulator type): this frame here, 1000 keyframes /
    # 200 k keypoints in profiles/archive/r4_detect_variants.json (tools/detect_variant_campaign.py)
    det = {}
    try:
        ctx2 = ov2slam_amd.Context(dev_index)
        fx = ov2slam_amd.FeatureExtractor(ctx2, dmaxquality=0.001)
        img0 = O.clahe(views[0], CLAHE_CLIP, *CLAHE_TILES)
        roi = (5, 5, W - 10, H - 10)
        gk = fx.detectSingleScale(img0, CELL, np.zeros((0, 2), np.float32), roi)
        rk, _ = O.detect_singlescale(img0, CELL, np.zeros((0, 2), np.float32), roi, 0.001, True)
        det["detector_bit_exact_vs_oracle"] = bool(np.array_equal(gk.view(np.uint32), rk.view(np.uint32)))
        det["detector_points_compared"] = int(len(rk))
        from scipy.spatial import cKDTree
        var = {}
        for name, kw in (("blur_ties_to_even", dict(blur=O.BLUR_HALF_EVEN)), ("getRectSubPix_generic", dict(subpix=O.SUBPIX_GENERIC)),
                         ("cornerSubPix_float_sums", dict(subpix=O.SUBPIX_FLOAT_ACC))):
            with O.detect_variant(**kw):
                vk, _ = O.detect_singlescale(img0, CELL, np.zeros((0, 2), np.float32), roi, 0.001, True)
            d, _ = cKDTree(vk.astype(np.float64)).query(rk.astype(np.float64))
            var[name] = {"keypoints_without_counterpart_within_1px": int((d > 1.0).sum()), "max_abs_dpx_of_the_others": float(d[d <= 1.0].max())}
        det["detector_canonical_vs_other_opencv_arithmetic"] = var
        det["detector_variant_campaign"] = "profiles/archive/r4_detect_variants.json: 200465 keypoints of 1000 keyframes -- blur ties-to-even: 1.0 % of the keypoints " \
                                           "move to another arg-max, the others identical; getRectSubPix generic form / float sums: <= 0.062 px, 1 / 0 moved"
        ctx2.close()
    except Exception:
        import traceback
        det = {"detector_parity_error": traceback.format_exc()[-600:]}
    return {"lk_bit_exact_vs_oracle": ok_all, "lk_points_compared": n, **det,
            "front_end_oracle_pin": "none: no OpenCV in this image or on the GPU box (gpurun_out/r3probe) -- parity is GPU = oracle, and the "
                                    "oracle's distance to float-accumulator OpenCV builds is bounded by measurement",
            "lk_int64_vs_float_accumulator_opencv_orders": worst}


def ba_section(ctx):
    """local-BA timings: config[3] (mono, robust pass, resident) is the figure comparable across rounds; the others are the
    stereo variant, the full two-pass localBA protocol (robust + L2, host buffers) and a realistic window."""
    from ov2slam_amd import optimizer, synth
    out = {}

    def resident(label, pb, desc, reps=5):
        rp = optimizer.ResidentProblem(ctx, pb)
        rp.solve()
        its, ms, wall0 = 0, 0.0, time.perf_counter()
        for _ in range(reps):
            r = rp.solve()
            its += r["iterations"]; ms += r["solve_ms"]
        wall = time.perf_counter() - wall0
        res = {"iters_per_s": its / (ms * 1e-3), "iters_per_s_wall_incl_d2h": its / wall, "iterations_per_solve": its / reps,
               "solve_ms": ms / reps, "us_per_iteration": ms / max(1, its) * 1e3, "workload": desc,
               "termination": optimizer.TERMINATION.get(r["termination"])}
        poses = r["poses"].copy()
        rp.close()
        return res, poses

    pb = synth.make_ba_problem(50, 10000, 30, stereo=False, seed=42)
    out, gpu_poses = resident("config3_mono", pb, "50 KF x 10000 inverse-depth landmarks x 30 obs (290000 residual blocks), Huber "
                              "sqrt(5.9915), max 5 LM iterations, function_tolerance 1e-3, problem resident in HBM")
    pbs = synth.make_ba_problem(50, 10000, 30, stereo=True, seed=42)
    out["config3_stereo"], _ = resident("config3_stereo", pbs, "same, stereo: + 290000 right + 10000 right-anchor residual blocks", reps=3)
    pbw = synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7)
    out["window_25kf_3k_stereo"], _ = resident("window", pbw, "realistic local-BA window: 25 KF x 3000 landmarks x 12 obs, stereo "
                                               "(optimizer.cpp:150-188)", reps=5)
    # the whole localBA protocol of optimizer.cpp:436-735 (robust pass, outlier removal, L2 pass, second test), host buffers in /
    # out: ONE ov2_local_ba call (problem resident between the passes, outlier handling on the device; the per-block chi2 arrays,
    # which the reference's write-back does not use, are not downloaded) -- and the two-call form it replaces, for comparison
    opt = optimizer.Optimizer(ctx)

    def wall_of(fn, reps=3):
        fn()
        best, res = 1e9, None
        for _ in range(reps):
            t0 = time.perf_counter()
            res = fn()
            best = min(best, time.perf_counter() - t0)
        return best, res
    for key, pbx, what in (("localba_two_pass_stereo", pbs, "the stereo config[3] problem (590000 residual blocks)"),
                           ("localba_two_pass_window", pbw, "the 25 KF x 3000 landmark x 12 obs stereo window")):
        wall, r = wall_of(lambda: opt.localBA(pbx, want_chi2=False))
        wall2, r2 = wall_of(lambda: opt.localBA_two_calls(pbx), reps=2)
        it1, it2 = r["iterations"]
        ms = r["solve_ms"][0] + r["solve_ms"][1]
        out[key] = {"wall_ms_incl_h2d_d2h": wall * 1e3, "device_ms": ms, "iterations_robust": it1, "iterations_l2": it2,
                    "iters_per_s_device": (it1 + it2) / (ms * 1e-3), "iters_per_s_wall": (it1 + it2) / wall,
                    "outliers_removed": int(r["bad_obs"].sum()),
                    "two_ov2_ba_solve_calls_wall_ms": wall2 * 1e3, "same_outlier_set_as_two_calls": bool(np.array_equal(r["bad_obs"], r2["bad_obs"])),
                    "workload": "Optimizer.localBA on " + what + ": robust pass (<=5 it) + outlier removal + L2 pass (<=10 it) + second "
                                "test in ONE ov2_local_ba call, host arrays in, poses / inverse depths / outlier flags out"}
    # the estimator side of configs[4]: the windows of a rank's eleven sequences in ONE ov2_local_ba_batch call (grid.z = problem)
    wins = [synth.make_ba_problem(25, 3000, 12, stereo=True, seed=7 + i) for i in range(11)]
    wall_b, rb = wall_of(lambda: opt.localBA_batch(wins, want_chi2=False), reps=5)
    wall_s, rs = wall_of(lambda: [opt.localBA(w, want_chi2=False) for w in wins], reps=2)
    res_b, n_shared = rb
    its_b = sum(sum(r["iterations"]) for r in res_b)
    out["localba_batch_11_windows"] = {
        "wall_ms_incl_h2d_d2h": wall_b * 1e3, "device_ms": sum(res_b[0]["solve_ms"]), "problems_sharing_the_launches": n_shared,
        "iterations_total": its_b, "iters_per_s_wall": its_b / wall_b,
        "eleven_ov2_local_ba_calls_wall_ms": wall_s * 1e3, "eleven_calls_device_ms": sum(sum(r["solve_ms"]) for r in rs),
        "same_outlier_sets_as_single_calls": bool(all(np.array_equal(a["bad_obs"], b["bad_obs"]) for a, b in zip(res_b, rs))),
        "same_iteration_counts_as_single_calls": bool(all(a["iterations"] == b["iterations"] for a, b in zip(res_b, rs))),
        "workload": "eleven 25 KF x 3000 landmark x 12 obs stereo windows (different seeds), two-pass localBA each, host arrays in / results out; "
                    "wall clock includes the Python packing of the eleven problems"}
    return out, pb, gpu_poses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--seqs", type=int, default=4096, help="sequences processed in lock-step per GPU (16.5 GB of HBM at 4096)")
    ap.add_argument("--workload", choices=["euroc", "kitti"], default="euroc",
                    help="euroc = the headline configuration (BASELINE.json configs[1]); kitti = configs[2] "
                         "(1241x376, wide-image stress) -- a side measurement, not the headline")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--dry", action="store_true", help="no GPU work: launcher + process group + config-5 plan / all-gather only (CPU test)")
    ap.add_argument("--config5-scale", type=int, default=1, help="EuRoC frame counts are divided by this for the lock-step config-5 run (1 = the full 27 049 frames)")
    ap.add_argument("--config5-stream-scale", type=int, default=16, help="divisor for the per-sequence-stream comparison run of config 5")
    ap.add_argument("--config5-concurrency", type=int, default=2,
                    help="sequences of a rank that stream concurrently on its GPU inside one driver process (1 = one after another).  Measured "
                         "(tools/r4_conc.py, profiles/archive/r4_stream_concurrency.txt): 1 stream 4.4 k frames/s, 2 streams 7.3 k, 4 streams 6.4 k, 8 streams "
                         "5.9 k -- the streams are chains of small launches and the aggregate is bound by the rate the GPU's command processor "
                         "retires them, which is what the lock-step batch entry points (the headline mode) exist to avoid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-sequence / config-5 / BA / detect / parity sections")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch(args))

    if args.workload == "kitti":
        # KITTI 00 stereo, accurate params: 1241x376, nmaxdist 35 -> nbmaxkps 36*11 = 396
        global W, H, NKPS, N_PASS_A, N_PASS_B, CLAHE_TILES
        W, H, NKPS = 1241, 376, 396
        N_PASS_A = 277; N_PASS_B = NKPS - N_PASS_A
        CLAHE_TILES = (W // 50, H // 50)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if not args.dry:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    if args.dry:
        # the N>1 plumbing without a GPU: barrier, max-reduce of the elapsed time, config-5 plan + all-gather
        if world > 1:
            dist.barrier()
        t = torch.tensor([1.0 + 0.1 * rank], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c5 = run_config5(None, rank, world, args.config5_scale, dry=True, device=local_rank if world > 1 else 0)
        seen, bus_ids = ranks_and_devices(world, local_rank if world > 1 else 0, True)
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": None, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "dry": True, "backend": args.backend, "elapsed_max_over_ranks": float(t.item()),
                              "ranks_seen": seen, "pci_bus_id_per_rank": bus_ids, "config5": c5}))
        if world > 1:
            dist.destroy_process_group()
        return

    import ov2slam_amd
    from ov2slam_amd import _lib as L

    if world == 1:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    S = args.seqs
    view_sets, kps, pri, pri_uniform = make_inputs(S, seed=1234 + rank, uniform=True)      # every rank owns different sequences; sequence s shows view set s % 64
    views = view_sets[0]                                        # the single-sequence sections below run sequence 0
    n_sets = int(view_sets.shape[0])

    stream = torch.cuda.current_stream()
    ctx = ov2slam_amd.Context(dev.index, stream=stream.cuda_stream)
    lib = ctx.lib
    # raw frames in HBM with a 16-byte-aligned row pitch (what ov2_pyr_build_h / the tracker stage host images into as well):
    # 1241-byte KITTI rows would otherwise send every CLAHE kernel down its unaligned-row instance (2.3x slower, r2_kitti_*)
    PITCH = (W + 15) & ~15
    vpad_sets = np.zeros((n_sets, NF + 1, H, PITCH), np.uint8); vpad_sets[:, :, :, :W] = view_sets
    vpad = vpad_sets[0]
    sets_d = torch.from_numpy(vpad_sets).to(dev)                 # (sets, NF+1, H, PITCH)
    set_of = torch.arange(S, device=dev) % n_sets
    frames_d = torch.stack([sets_d[:, f][set_of] for f in range(NF + 1)])      # (NF+1, S, H, PITCH): every sequence's own copy of its frames
    del sets_d
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
    NT = 2 * NF                                                  # transitions of the ping-pong walk over the views
    a_k = [vp(kps_d[f]) for f in range(NT)]
    a_kB = [vp(kps_d[f][:, N_PASS_A:]) for f in range(NT)]
    a_p = [vp(pri_work[f]) for f in range(NT)]
    a_pB = [vp(pri_work[f][:, N_PASS_A:]) for f in range(NT)]
    a_st, a_stB, a_stats, a_nA, a_nB = vp(status_d), vp(status_d[:, N_PASS_A:]), vp(stats_d), vp(nA_d), vp(nB_d)
    hp = [p.h_pyr for p in pyrs]
    fb, build_clahe = lib.ov2_fb_klt_d, lib.ov2_pyr_build_clahe_d
    lk_events = []

    def preprocess(pyr, img):
        # clahe->apply + buildOpticalFlowPyramid (visual_front_end.cpp:1159, :1172) in one call
        L.check(build_clahe(ctx.h, pyr, img, PITCH, PITCH * H, CLAHE_CLIP, CLAHE_TILES[0], CLAHE_TILES[1]))

    def step(i, timed):
        # One camera frame per sequence: EXACTLY one preprocessImage and the two fbKltTracking calls.  The views are walked
        # back and forth (0, 1, .., NF, NF-1, .., 0, ..), so `prev` is always the previous step's `cur` pyramid (rounds 1-2
        # restarted the cycle every NF steps with an extra preprocessImage of view 0 and a 60 MB restore of all priors:
        # ~0.45 ms per step of bench bookkeeping inside the timed region).
        f = i % NT
        prevp, curp = hp[i % 2], hp[(i + 1) % 2]
        pri_work[f].copy_(pri_d[f])                                 # priors are in/out: restore this transition's (10 MB)
        preprocess(curp, a_img[walk_view(i + 1)])                   # preprocessImage
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

    preprocess(hp[0], a_img[walk_view(0)])                          # the frame before the first step
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
    ranks_seen, bus_ids = ranks_and_devices(world, dev.index, False)     # (after the timed region: an all-reduce of ones + the PCI bus ids)

    # ---- the pre-processing of a step on LOW-ENTROPY frames (side measurement, rank 0): the CLAHE histogram is built with LDS atomics,
    # and the bench frames (band-limited noise stretched to 0..255) are the friendliest input there is -- a constant, an over- or an
    # under-exposed frame puts a wavefront's 64 adds on a handful of bins (round 4's kernel: 3.0x slower on a constant frame)
    pre_entropy = None
    if rank == 0 and not args.no_extras:
        try:
            ev = lambda: torch.cuda.Event(enable_timing=True)
            rngp = np.random.default_rng(5)
            base = vpad[0].astype(np.int32)
            sel = rngp.uniform(size=base.shape) < 0.8
            kinds = {"bench_frames": None, "constant": np.full_like(base, 128), "saturated": np.where(sel, 255, 255 - base // 8),
                     "dark": np.where(sel, base // 32, base // 4), "levels16": np.where(sel, (base // 16) * 16 + 8, base)}
            pre_entropy = {}
            for name, img in kinds.items():
                if img is None:
                    src = a_img[walk_view(0)]
                else:
                    one = torch.from_numpy(np.clip(img, 0, 255).astype(np.uint8)).to(dev)
                    buf = one[None].expand(S, H, PITCH).contiguous()
                    src = C.c_void_p(buf.data_ptr())
                ts = []
                for r in range(5):
                    e0, e1 = ev(), ev()
                    e0.record(stream); preprocess(hp[r % 2], src); e1.record(stream)
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                pre_entropy[name] = float(np.median(ts[1:]))
                if img is not None:
                    del buf, one
            pre_entropy = {"ms_per_step": pre_entropy, "worst_over_bench_frames": max(pre_entropy.values()) / pre_entropy["bench_frames"],
                           "note": "preprocessImage of %d frames per call, HIP events; constant = every pixel 128, saturated = 80 %% of the pixels 255, dark = 80 %% in 8 "
                                   "levels near 0, levels16 = 80 %% on 16 grey levels; the histogram wavefronts keep four staggered copies since round 5 "
                                   "(profiles/r5_clahe_histogram_variants.txt)" % S}
        except Exception:
            import traceback
            pre_entropy = {"error": traceback.format_exc()[-600:]}

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
    traffic, traffic_note, limiter_kind, valu_frac = None, None, None, None
    try:
        import hashlib
        tj = json.load(open(os.path.join(ROOT, "profiles", "lk_traffic.json")))["current"]
        sha = hashlib.sha256(open(os.path.join(ROOT, "ov2slam_amd", "csrc", "lk3.hip"), "rb").read()).hexdigest()
        if tj.get("kernel_source_sha256") != sha:
            # counters of another kernel: refuse them (null + reason) instead of carrying stale numbers in the line
            traffic_note = ("counters refused: profiles/lk_traffic.json was measured on lk3.hip %s..., the tree has %s... -- re-run tools/profile.sh and "
                            "tools/summarize_profile.py --update-lk-traffic" % (str(tj.get("kernel_source_sha256"))[:12], sha[:12]))
        elif tj.get("seqs_per_gpu") != S or args.workload != "euroc":
            traffic_note = "no committed PMC pass for this workload / batch (profiles/lk_traffic.json holds euroc at %s sequences per GPU)" % tj.get("seqs_per_gpu")
        else:
            traffic = tj["hbm_bytes_per_launch"]; traffic_note = tj.get("limiter") + "; source: " + str(tj.get("source"))
            limiter_kind, valu_frac = tj.get("limiter_kind"), tj.get("valu_issue_frac")
    except Exception as e:
        traffic_note = "profiles/lk_traffic.json unreadable: %r" % (e,)

    # ---- the same K steps with pre-processing and tracking on TWO streams (side measurement, never `value`) ------------------
    # Offline batch mode knows frame t+1 while frame t is tracked: preprocessImage of the next frame (LDS-atomic bound histogram,
    # memory-bound strip kernel) runs on its own context / stream beside the VALU-bound LK kernels of the current one.  Three
    # pyramids rotate; a pyramid's `ready` event orders producer -> consumer (ov2_pyr_wait_ready inside ov2_fb_klt_d), a torch
    # event per tracking step orders consumer -> the producer that overwrites the buffer two steps later.  The headline `value`
    # and `roofline` stay the serial ones above: a kernel's launch time measured while it shares the CUs says nothing about it.
    pipelined = None
    if not args.no_extras and world == 1:
      try:
        s_pre = torch.cuda.Stream(device=dev)
        ctx_pre = ov2slam_amd.Context(dev.index, stream=s_pre.cuda_stream)
        pyr3 = pyrs + [ov2slam_amd.Pyramid(ctx, W, H, WIN, LEVELS, batch=S)]
        h3 = [p.h_pyr for p in pyr3]
        ev_lk = {}
        def pre_on(idx, frame_no):                                   # frame `frame_no` -> buffer idx, on the pre-processing stream
            L.check(build_clahe(ctx_pre.h, h3[idx], a_img[walk_view(frame_no)], PITCH, PITCH * H, CLAHE_CLIP, CLAHE_TILES[0], CLAHE_TILES[1]))
        def pstep(i):
            f = i % NT
            if i - 2 in ev_lk: s_pre.wait_event(ev_lk.pop(i - 2))     # buffer (i+1) % 3 held frame i-2: its last reader was step i-2
            pre_on((i + 1) % 3, i + 1)
            pri_work[f].copy_(pri_d[f])
            prevp, curp = h3[i % 3], h3[(i + 1) % 3]
            L.check(fb(ctx.h, prevp, curp, WIN, 1, 30, 0.01, 30.0, 0.5, a_k[f], a_p[f], NKPS, a_nA, a_st, a_stats))
            L.check(fb(ctx.h, prevp, curp, WIN, LEVELS, 30, 0.01, 30.0, 0.5, a_kB[f], a_pB[f], NKPS, a_nB, a_stB, a_stats))
            e = torch.cuda.Event(); e.record(stream); ev_lk[i] = e
        torch.cuda.synchronize()
        pre_on(0, 0)
        nw = min(args.warmup, 12)
        for i in range(nw):
            pstep(i)
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for i in range(args.steps):
            pstep(nw + i)
        torch.cuda.synchronize()
        tp = time.perf_counter() - tp0
        pipelined = {"ms_per_step": tp / args.steps * 1e3, "frames_per_s": args.steps * S / tp, "streams": 2, "pyramid_buffers": 3,
                     "tracked_fraction": float(status_d.float().mean().item()),
                     "note": "same work per step as the headline (one preprocessImage + two fbKltTracking calls over %d sequences), "
                             "preprocessImage of frame t+1 on a second context beside the tracking of frame t; side measurement" % S}
        del pyr3, h3
        ctx_pre.close() if hasattr(ctx_pre, "close") else None
      except Exception as e:                                          # a side measurement must not cost the headline line
        pipelined = {"error": repr(e)}

    # free the batch buffers before the single-sequence / BA sections
    # ---- keyframe detection for the whole batch: every sequence's detector on level 0 of its pyramid, one call -------------
    det_batch = None
    if not args.no_extras and world == 1:
      try:
        ncells = (W // CELL) * (H // CELL)
        cap = 2 * ncells
        det_out = torch.zeros((S, cap, 2), dtype=torch.float32, device=dev)
        qual = np.full(S, 1e-3)
        roi = (5, 5, W - 10, H - 10)
        def timed_detect(cur_ptr, cur_cap, n_ptr):
            torch.cuda.synchronize()
            nd = ov2slam_amd.FeatureExtractor.detectSingleScaleBatch(ctx, pyrs[0], CELL, cur_ptr, cur_cap, n_ptr, roi, qual, det_out.data_ptr(), cap)
            t1 = time.perf_counter()
            for _ in range(3):
                qual[:] = 1e-3
                nd = ov2slam_amd.FeatureExtractor.detectSingleScaleBatch(ctx, pyrs[0], CELL, cur_ptr, cur_cap, n_ptr, roi, qual, det_out.data_ptr(), cap)
            return (time.perf_counter() - t1) / 3 * 1e3, nd
        det_ms, n_det = timed_detect(0, 0, 0)
        # the keyframe case: the tracked keypoints of every sequence occupy their cells, the detector tops the set up
        # (every second keypoint of the frame: about half of the cells stay occupied, the usual state at a keyframe request)
        cur_half = kps_d[0][:, ::2].contiguous()
        n_half = int(cur_half.shape[1])
        ncur_d = torch.full((S,), n_half, dtype=torch.int32, device=dev)
        top_ms, n_top = timed_detect(cur_half.data_ptr(), n_half, ncur_d.data_ptr())
        step_ms = elapsed / args.steps * 1e3
        # ... and MEASURED: twenty tracking steps with the top-up detector on the current frame after every fifth (one call, its counts come back:
        # one synchronisation per keyframe step, as a lock-step host would see it)
        base = args.warmup + args.steps
        base += base % 2                                            # (even: the walk's pyramid parity)
        preprocess(hp[base % 2], a_img[walk_view(base)])            # (the sections above left other frames in the pyramids: re-prime the walk)
        for i in range(5): step(base + i, False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(5, 25):
            step(base + i, False)
            if i % 5 == 4:
                qual[:] = 1e-3
                ov2slam_amd.FeatureExtractor.detectSingleScaleBatch(ctx, pyrs[(base + i + 1) % 2], CELL, cur_half.data_ptr(), n_half, ncur_d.data_ptr(), roi, qual,
                                                                    det_out.data_ptr(), cap)
        torch.cuda.synchronize()
        kf_ms = (time.perf_counter() - t1) / 20 * 1e3
        det_batch = {"detect_singlescale_batch_ms": det_ms, "images": S, "us_per_image": det_ms * 1e3 / S,
                     "frames_per_s_with_keyframe_every_5th_measured": S / (kf_ms * 1e-3), "ms_per_step_with_keyframe_every_5th_measured": kf_ms,
                     "points_per_image": float(n_det.mean()),
                     "topup_ms": top_ms, "topup_us_per_image": top_ms * 1e3 / S, "topup_points_per_image": float(n_top.mean()),
                     "topup_current_keypoints": n_half,
                     "frames_per_s_with_keyframe_every_5th": S / ((step_ms + top_ms / 5.0) * 1e-3),
                     "entry": "ov2_detect_singlescale_batch_d on level 0 of the batch pyramid (device-resident lists, one sync): "
                              "no current keypoints / topping up %d tracked keypoints per sequence; the last figure adds a fifth of the "
                              "top-up call to the tracking step (keyframe every 5th frame)" % n_half}
        # a keyframe's detection reads the CLAHE'd level 0 once (W x H bytes); what this design also moves is the fp32 response map
        # of the free cells (4 B per pixel, written by the response kernel, re-read by the selection's slow path only)
        b_alg = W * H
        det_batch["roofline_detect"] = {
            "bound": "hbm", "algorithmic_bytes_per_image": b_alg, "achieved": b_alg * S / (top_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": b_alg * S / (top_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_counting_the_response_map": (b_alg + 4 * (W // CELL) * (H // CELL) * CELL * CELL // 2) * S / (top_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "limiter": "not bytes: k_mineig_strip is VALU-issue bound (~80 instructions per pixel of exact OpenCV arithmetic: fp64 row / column sums, "
                       "correctly rounded sqrt), k_corner_subpix iterates ~20 Gauss-Newton trips per point on these textures, k_grid_select is a dependency "
                       "chain over the cells at two work-groups per CU; counters and knock-outs: profiles/r6_detect_batch_strip_counters.json, profiles/r6_detect_notes.txt"}
        del det_out
      except Exception:
        import traceback
        det_batch = {"error": traceback.format_exc()[-1000:]}
    # ---- round 5's input as a side measurement (rank 0): EVERY sequence shows view set 0 (own copies in HBM, own keypoints).  Same kernels,
    # same work shape; what differs is how much the Gauss-Newton trip counts of a wavefront's twenty keypoints spread (the kernel idles
    # in lock step behind its slowest keypoint) -- the only link between the 0.28 of rounds 3-5 and this round's figure on 64 contents
    uniform = None
    if rank == 0 and not args.no_extras:
        try:
            for f in range(NF + 1):
                frames_d[f].copy_(frames_d[f, 0:1].expand(S, H, PITCH))
            pri_d.copy_(torch.from_numpy(pri_uniform).to(dev))
            lk_events.clear(); stats_d.zero_()
            preprocess(hp[0], a_img[walk_view(0)])
            nu = min(args.steps, 20)
            for i in range(4):
                step(i, False)
            torch.cuda.synchronize(); stats_d.zero_()
            tu0 = time.perf_counter()
            for i in range(nu):
                step(4 + i, True)
            torch.cuda.synchronize()
            tu = time.perf_counter() - tu0
            itu, viu = [int(v) for v in stats_d.tolist()]
            msu = sum(a.elapsed_time(c) for a, b_, c in lk_events) / max(1, 2 * len(lk_events))
            bu = lk_algorithmic_bytes(itu, viu, nu * S * NKPS) / max(1, 2 * len(lk_events))
            uniform = {"frames_per_s": nu * S / tu, "ms_per_step": tu / nu * 1e3, "roofline_frac": bu / (msu * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "avg_launch_ms": msu, "algorithmic_bytes_per_launch": bu, "gn_iterations": itu, "patch_builds": viu, "steps": nu,
                       "note": "all %d sequences show view set 0 (round 5's input: one texture, identical content in every sequence); side measurement, never `value`" % S}
        except Exception:
            import traceback
            uniform = {"error": traceback.format_exc()[-600:]}
    del frames_d, kps_d, pri_d, pri_work, status_d
    for p in pyrs:
        p.close()
    torch.cuda.empty_cache()

    c5 = None
    if not args.no_extras and args.workload == "euroc":
        try:
            ctx5 = ov2slam_amd.Context(dev.index)
            c5 = run_config5(ctx5, rank, world, args.config5_scale, device=dev.index, concurrency=args.config5_concurrency, stream_scale=args.config5_stream_scale)
            ctx5.close()
        except Exception:
            import traceback
            c5 = {"error": traceback.format_exc()[-1000:]} if world == 1 else None      # (with several ranks a failure must stay loud: collectives)
            if world > 1:
                raise

    if rank == 0:
        frames = args.steps * S * world
        out = {
            "metric": METRIC,
            "value": frames / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # what the process group actually spanned: all_reduce(SUM) of ones over the backend (RCCL), and the PCI bus id of every rank's GPU
            "ranks_seen": ranks_seen, "pci_bus_id_per_rank": bus_ids, "distinct_gpus": len(set(bus_ids)) == world,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32 fixed point + f32 (LK), f64 (BA)", "data": "synthetic",
            "config": {"workload": "%s stereo 'accurate' tracking step on synthetic %dx%d frames: CLAHE + pyramid "
                                   "build (4 levels) + fbKltTracking pass A (%d kps, nbpyrlvl 1) + pass B "
                                   "(%d kps, nbpyrlvl 3), 9x9 window, 30 it / 0.01 px; offline batch-of-sequences mode "
                                   "(the single-camera figure is `single_sequence`)"
                                   % ("EuRoC MH_01" if args.workload == "euroc" else "KITTI 00", W, H, N_PASS_A, N_PASS_B),
                       "seqs_per_gpu": S, "keypoints_per_frame": NKPS, "parallelism": "replicas x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "k_fb_klt3", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         # what actually limits the kernel (SQ counters of the committed PMC pass): the fraction of SIMD issue slots in use
                         "limiter": limiter_kind, "valu_issue_frac": valu_frac,
                         # what THIS data layout allows: the kernel gathers 16-byte row segments (12.5 % of every 128-byte line), the
                         # same pattern without arithmetic tops out at 5.2 TB/s of line traffic (profiles/lk_traffic.json) and the
                         # kernel's lines are 1.66x its algorithmic bytes: 5.2 / 1.66 / 8 -- 0.60 needs a tiled layout
                         "layout_ceiling_frac": 0.39 if traffic is None else min(1.0, 5.2e12 / (traffic / (bytes_total / max(1, n_launch))) / (HBM_PEAK_GBS * 1e9)),
                         "avg_launch_ms": avg_launch_ms, "launches": n_launch,
                         "algorithmic_bytes_per_launch": bytes_total / max(1, n_launch),
                         "gn_iterations": iters, "patch_builds": visits},
            "lk_ms_per_step": (ms_A + ms_B) / args.steps, "tracked_fraction": tracked,
        }
        out["config"]["image_content"] = ("%d distinct view sets per rank (16 textures x 4 photometric variants, one set low-entropy: 80 %% of the pixels on 8 grey "
                                          "levels), sequence s shows set s %% %d; own camera walk per texture" % (n_sets, n_sets))
        # SURVEY 8(d)(i) defines the tracking rate INCLUDING detection on keyframes: the step plus a fifth of the batched top-up detection
        # (a keyframe every 5th frame).  `value` stays the tracking step alone, as in every earlier round, so the two can be compared.
        if uniform is not None:
            out["same_content_in_every_sequence"] = uniform
        out["value_incl_detect"] = det_batch.get("frames_per_s_with_keyframe_every_5th") if isinstance(det_batch, dict) else None
        out["value_incl_detect_measured"] = det_batch.get("frames_per_s_with_keyframe_every_5th_measured") if isinstance(det_batch, dict) else None
        # ---- pre-processing (everything of the step that is not k_fb_klt3): SURVEY.md 8(d) bytes against the time the step spends
        # there; per kernel, duration and HBM bytes by the counters of the committed PMC passes (tools/profile.sh, FETCH_SIZE x 2:
        # the gfx950 correction of the guide, WRITE_SIZE as read), same workload and batch only
        try:
            pre_ms = elapsed / args.steps * 1e3 - (ms_A + ms_B) / args.steps
            px = [(max(1, (W + (1 << l) - 1) >> l)) * (max(1, (H + (1 << l) - 1) >> l)) for l in range(LEVELS + 1)]
            b_alg = W * H + sum(px[1:]) + 4 * sum(px) + 2 * W * H        # B_pyr (read L0, write L>=1, int16x2 derivatives) + CLAHE 2 W0 H0
            b_must = 2 * W * H + sum(px)                                # what this design has to move: raw frame twice (histogram, blend), every level once
            pre = {"ms_per_step": pre_ms, "algorithmic_bytes_per_image": b_alg, "achieved": b_alg * S / (pre_ms * 1e-3) / 1e9, "unit": "GB/s",
                   "frac": b_alg * S / (pre_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "bytes_the_design_must_move_per_image": b_must, "frac_of_those": b_must * S / (pre_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "note": "8(d) counts the int16x2 derivative pyramid (4 of its 5.33 B/px), which this design never materialises (LK evaluates it in registers)"}
            # per kernel: duration and HBM bytes by the counters of the newest committed PMC summary WHOSE KERNEL SOURCES ARE THE TREE'S
            # (tools/summarize_profile.py records the sha256 of clahe.hip / pyramid.hip): anything else is refused, not quoted
            import glob
            import hashlib
            tree = {f: hashlib.sha256(open(os.path.join(ROOT, "ov2slam_amd", "csrc", f), "rb").read()).hexdigest() for f in ("clahe.hip", "pyramid.hip")}
            pre["kernels"] = None
            pre["kernels_source"] = "no committed rocprofv3 summary matches the kernel sources in the tree (run tools/profile.sh + tools/summarize_profile.py)"
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rocprof_summary_seqs%d.json" % S)), key=os.path.getmtime, reverse=True)
            for pth in cands if args.workload == "euroc" else []:
                sj = json.load(open(pth))
                if any(sj.get("source_sha256", {}).get(f) != h for f, h in tree.items()):
                    continue
                ks = {}
                n_steps = max([kv["calls"] for kn, kv in sj.get("kernels", {}).items() if "k_clahe_apply_pyr" in kn] + [1])
                for kn, kv in sj.get("kernels", {}).items():
                    short = kn.replace("void ", "").split("<")[0]
                    if short in ("k_clahe_lut", "k_clahe_apply_pyr", "k_pyr_level"):
                        pm = sj.get("pmc", {}).get(kn, {})
                        rd = 2.0 * pm.get("FETCH_SIZE", {}).get("mean_per_dispatch", 0.0) * 1e3
                        wr = pm.get("WRITE_SIZE", {}).get("mean_per_dispatch", 0.0) * 1e3
                        ks[short] = {"avg_us": kv["avg_us"], "launches_per_step": round(kv["calls"] / max(1, n_steps)),
                                     "hbm_read_bytes_per_launch": rd, "hbm_written_bytes_per_launch": wr,
                                     "frac_by_counters": (rd + wr) / (kv["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS}
                pre["kernels"] = ks
                pre["kernels_source"] = "profiles/" + os.path.basename(pth)
                break
            if pre_entropy is not None:
                pre["low_entropy_ms"] = pre_entropy
            out["roofline_pre"] = pre
        except Exception:
            pass
        if c5 is not None:
            out["config5"] = c5
        if det_batch is not None:
            out["detect_batch"] = det_batch
        if pipelined is not None:
            out["pipelined_two_streams"] = pipelined
        # The sections below are side measurements: a failure in one of them must not cost the headline line
        def extras():
            ss = single_sequence(dev.index, views, kps, pri)
            out["single_sequence"] = ss
            out["single_sequence_fps"] = ss["pageable"]["fps_incl_pcie"]
            ctx1 = ov2slam_amd.Context(dev.index)
            out["ba"], pb, gpu_poses = ba_section(ctx1)
            # B_BA = 2 N_res (16 + 12 + 8) + 2 N_lm (8 + 16 + 4) + 2 * 56 N_kf + 8 (6 N_kf_opt)^2 per LM iteration
            b_ba = 2 * 290000 * 36 + 2 * 10000 * 28 + 2 * 56 * 50 + 8 * (6 * 50) ** 2
            us_it = out["ba"]["us_per_iteration"]
            out["ba"]["roofline"] = {"bound": "latency", "algorithmic_bytes_per_iteration": b_ba, "achieved": b_ba / (us_it * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                                     "unit": "GB/s", "frac": b_ba / (us_it * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                     "limiter": "a chain of nine launches per LM iteration, half of it the one-work-group fp64 Cholesky of the 300 x 300 reduced system "
                                                "(208 us: 32-column pivot chains, trailing tiles through L2, backward substitution), then the lineariser (74 us: LDS fp64 "
                                                "atomics and dependent fp64 chains on one wavefront per SIMD); profiles/archive/r4_v3_ba_timeline_config4_mono.txt, "
                                                "profiles/archive/r4_v3_ba_kernel_stats_*.csv; HBM bytes are under 1 % of the roofline by design (SURVEY 8(d): 'explain, don't hide')"}
            # ---- detection through the host-buffer drop-in API on ONE image: per-call latency incl. PCIe -----
            fx = ov2slam_amd.FeatureExtractor(ctx1, dmaxquality=0.001)
            roi = (5, 5, W - 10, H - 10)
            fx.detectSingleScale(views[0], CELL, np.zeros((0, 2), np.float32), roi)
            t1 = time.perf_counter()
            for _ in range(20):
                det = fx.detectSingleScale(views[0], CELL, np.zeros((0, 2), np.float32), roi)
            det_ms = (time.perf_counter() - t1) / 20 * 1e3
            # the keyframe case of the drop-in: the detector reads the CLAHE'd frame from level 0 of the tracker's pyramid
            trk = ov2slam_amd.VisualFrontEndTracker(ctx1, W, H, fclahe_val=CLAHE_CLIP, nbmaxkps=512)
            trk.trackFrame(views[0], np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), None)
            fx.detectSingleScalePyr(trk.cur_pyr, CELL, np.zeros((0, 2), np.float32), roi)
            t1 = time.perf_counter()
            for _ in range(20):
                det_d = fx.detectSingleScalePyr(trk.cur_pyr, CELL, np.zeros((0, 2), np.float32), roi)
            det_d_ms = (time.perf_counter() - t1) / 20 * 1e3
            trk.close()
            out["detect"] = {"detect_singlescale_ms_incl_pcie": det_ms, "detected_points": int(len(det)),
                             "detect_singlescale_pyramid_resident_ms": det_d_ms, "detected_points_pyramid_resident": int(len(det_d))}
            # ---- a stereo keyframe on the mapper's context: right image CLAHE + pyramid, then MapManager::stereoMatching's data path
            #      (SAD priors, 1-level prior tracks + retry, full-pyramid tracks, epipolar gate) -- one call vs the call sequence
            from ov2slam_amd import stereo as _st
            left_img = views[0]; right_img = np.roll(views[0], -20, axis=1)
            lp = ov2slam_amd.Pyramid(ctx1, W, H, WIN, LEVELS).build_clahe(left_img, CLAHE_CLIP, CLAHE_TILES[0], CLAHE_TILES[1])
            rp = ov2slam_amd.Pyramid(ctx1, W, H, WIN, LEVELS)
            ftrk = ov2slam_amd.FeatureTracker(ctx1, 30, 0.01)
            rcal = ov2slam_amd.CameraCalibration(ctx1, "pinhole", 458.654, 457.296, 367.215, 248.375, D=None)
            skps = kps[0, 0][:NKPS].astype(np.float32)
            p3d = {i: (float(skps[i, 0] - 20.0), float(skps[i, 1])) for i in range(0, NKPS, 2)}
            def stereo_kf(fn):
                rp.build_clahe(right_img, CLAHE_CLIP, CLAHE_TILES[0], CLAHE_TILES[1])
                return fn(ftrk, lp, rp, skps, skps, rcal, rect=True, priors3d=p3d)
            res_ms = {}
            for name, fn in (("fused", _st.stereo_matching_fused), ("call_sequence", _st.stereo_matching)):
                ok_s, _ = stereo_kf(fn)
                t1 = time.perf_counter()
                for _ in range(20):
                    ok_s, _ = stereo_kf(fn)
                res_ms[name] = (time.perf_counter() - t1) / 20 * 1e3
            out["stereo_keyframe"] = {"fused_ms": res_ms["fused"], "call_sequence_ms": res_ms["call_sequence"], "keypoints": int(len(skps)),
                                      "with_3d_prior": len(p3d), "stereo_ok_fraction": float(ok_s.mean()),
                                      "entry": "right-image ov2_pyr_build_clahe_h (asynchronous) + ov2_stereo_match, host buffers, one sync; "
                                               "call_sequence = ov2_line_min_sad + 2 x ov2_fb_klt + ov2_stereo_epipolar_check (4 syncs)"}
            ctx1.close()
            if not args.no_cpu_baseline:
                out["parity"] = parity_check(dev.index, views, kps, pri)
                cb = cpu_baseline(views, kps, pri, pb)
                rb = cb.pop("_ba_result", None)
                simd = cb.pop("_simd", None)
                out["cpu_baseline"] = cb
                if simd is not None:
                    out["cpu_baseline_simd"] = simd
                    out["tracking_speedup_vs_cpu_simd_single_sequence"] = ss["pageable"]["fps_incl_pcie"] / simd["value"]
                    if "ba" in simd:
                        out["ba_iters_per_s_ratio_vs_cpu_simd"] = out["ba"]["iters_per_s"] / simd["ba"]["iters_per_s"]
                if rb is not None:
                    rel = float(np.abs(gpu_poses - rb["poses"]).max() / max(1e-30, np.abs(rb["poses"]).max()))
                    out["parity"]["ba_pose_max_rel_err_vs_oracle"] = rel
                    out["parity"]["ba_within_1e-4"] = bool(rel <= 1e-4)
                    out["tracking_speedup_vs_cpu_single_sequence"] = ss["pageable"]["fps_incl_pcie"] / cb["value"]
            # configs[1] as ONE measured stream (three threads, three contexts) beside the CPU oracle on the same schedule; the
            # combined figure is the MEASURED stream rate over the CPU's pipeline bound (its three threads overlapping perfectly)
            # -- it replaces the arithmetic "5 frames + 1 BA pass" composite of rounds 1-2
            c2 = config2_stream(dev.index, with_cpu=not args.no_cpu_baseline)
            out["config2_stream"] = c2
            if "speedup_vs_cpu_pipeline_bound" in c2:
                out["combined_speedup_vs_cpu"] = c2["speedup_vs_cpu_pipeline_bound"]
                out["combined_speedup_vs_cpu_serial_cpu"] = c2["speedup_vs_cpu_serial"]
        if world == 1 and not args.no_extras:
            try:
                extras()
            except Exception:
                import traceback
                out["extras_error"] = traceback.format_exc()[-1500:]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
