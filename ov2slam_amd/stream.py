"""One camera, three contexts: the keyframe cycle of BASELINE.json configs[1] (EuRoC stereo, 'accurate') the way the
reference schedules it -- the front-end on the SLAM thread, stereo matching on the mapper thread, localBA on the estimator
thread, concurrently (/root/reference/src/ov2slam.cpp:116-237, src/mapper.cpp:62-95, src/estimator.cpp:33-98):

    SLAM thread      (context A)  per frame   preprocessImage + kltTracking       ov2_tracker_track_frame
                                              + Frame::computeKeypoint            (same enqueue: ov2_tracker_set_calibration)
                                  keyframe    MapManager::extractKeypoints        ov2_detect_singlescale_d on cur_pyr_
                                              computeKeypoint of the new points   ov2_compute_keypoints
                                              Mapper::addNewKf                    (queue, FIFO: mapper.cpp:784-809)
    mapper thread    (context B)  keyframe    clahe->apply + buildOpticalFlowPyramid of the RIGHT image, stereoMatching
                                              (mapper.cpp:74-83)                  ov2_pyr_build_clahe_h + ov2_stereo_match on the
                                                                                  front-end's left pyramid (cross-context event)
                                              Estimator::addNewKf
    estimator thread (context C)  keyframe    Optimizer::localBA                  ov2_local_ba (two passes, problem resident);
                                              only the LAST queued keyframe is processed (estimator.cpp:185-205)

What stays on the CPU in the reference and is NOT part of this path (pose estimation, triangulation, the map walk that builds the
BA problem) is stood in for by the synthetic ground truth: priors come from the true flow, the BA problems are pre-generated
windows.  Python threads: every library call releases the GIL (ctypes), so the three contexts really overlap on the GPU."""
import queue
import threading
import time

import numpy as np

from . import frontend, stereo, optimizer

K_EUROC = (458.654, 457.296, 367.215, 248.375)
D_EUROC = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)


def run_stream(ctx, seq, kf_every=5, cell=35, nbmaxkps=308, prior_sigma=1.5, use_graph=True, do_stereo=True, ba_problems=None,
               distortion=None, ba_policy="newest"):
    """seq: batch.SyntheticSequence (stereo=True for do_stereo).  ba_problems: list of flat local-BA problems (cycled, one
    localBA per keyframe) or None.  ba_policy: "newest" = the reference's estimator (only the last queued keyframe is optimised
    when several wait, estimator.cpp:195-205), "all" = every keyframe gets its localBA (equal work for CPU / GPU comparisons).
    distortion: coefficients of both cameras for computeKeypoint / the epipolar gate (None: the synthetic views are ideal pinhole
    images; D_EUROC costs the same kernels a few more instructions).  ctx = context A; contexts B and C are created on the same device.
    Returns a dict of counters and times (seconds)."""
    from . import Context
    rng = np.random.default_rng(seq.seed + 17)
    w, h = seq.w, seq.h
    dev = ctx.device
    ctxB = Context(dev) if do_stereo else None
    ctxC = Context(dev) if ba_problems else None
    trk = frontend.VisualFrontEndTracker(ctx, w, h, nbmaxkps=2 * nbmaxkps, use_graph=use_graph)
    fx = frontend.FeatureExtractor(ctx, dmaxquality=0.001)
    calL = frontend.CameraCalibration(ctx, "pinhole", *K_EUROC, D=distortion)
    trk.setCalibration(calL)                         # Frame::computeKeypoint of the tracked positions rides in the per-frame enqueue
    roi = (5, 5, w - 10, h - 10)
    empty = np.zeros((0, 2), np.float32)
    st = dict(frames=0, tracked=0, attempted=0, err_sq_sum=0.0, err_n=0, detect_calls=0, keyframes=0, stereo_kfs=0, stereo_ok=0,
              stereo_kps=0, mapper_busy_s=0.0, ba_solves=0, ba_skipped_kfs=0, ba_iterations=0, ba_busy_s=0.0, ba_device_ms=0.0,
              slam_wait_for_mapper_s=0.0, slam_library_s=0.0)
    errors = []
    # ---- mapper thread -----------------------------------------------------------------------------------------------
    map_q = queue.Queue()
    ba_q = queue.Queue()
    kf_consumed = {}                                   # keyframe frame index -> Event: the mapper is done with the left pyramid

    def mapper():
        try:
            ftrk = frontend.FeatureTracker(ctxB, 30, 0.01)
            calR = frontend.CameraCalibration(ctxB, "pinhole", *K_EUROC, D=distortion)
            pyrR = frontend.Pyramid(ctxB, w, h, 9, 3)
            while True:
                item = map_q.get()
                if item is None:
                    break
                f, left_pyr, right_img, kps, unpx, p3, hp = item
                t0 = time.perf_counter()
                pyrR.build_clahe(right_img, 3.0, w // 50, h // 50)                                   # asynchronous
                ok, right = stereo.stereo_match_arrays(ftrk, left_pyr, pyrR, kps, unpx, p3, hp, calR, rect=True)
                st["mapper_busy_s"] += time.perf_counter() - t0
                kf_consumed[f].set()
                st["stereo_kfs"] += 1; st["stereo_ok"] += int(ok.sum()); st["stereo_kps"] += len(kps)
                if ba_problems:
                    ba_q.put(f)
            pyrR.close()
        except Exception as e:                          # surfaced by the caller
            errors.append(e)
            for ev in list(kf_consumed.values()):     # snapshot: the SLAM thread may insert while this runs
                ev.set()
        finally:
            ba_q.put(None)

    def estimator():
        try:
            opt = optimizer.Optimizer(ctxC)
            n = 0
            while True:
                item = ba_q.get()
                if item is None:
                    break
                # In SLAM mode only the last received keyframe is processed (estimator.cpp:195-205)
                while ba_policy == "newest":
                    try:
                        nxt = ba_q.get_nowait()
                    except queue.Empty:
                        break
                    if nxt is None:
                        ba_q.put(None)
                        break
                    st["ba_skipped_kfs"] += 1
                    item = nxt
                pb = ba_problems[n % len(ba_problems)]; n += 1
                t0 = time.perf_counter()
                r = opt.localBA(pb, want_chi2=False)
                st["ba_busy_s"] += time.perf_counter() - t0
                st["ba_solves"] += 1; st["ba_iterations"] += r["iterations"][0] + r["iterations"][1]
                st["ba_device_ms"] += r["solve_ms"][0] + r["solve_ms"][1]
        except Exception as e:
            errors.append(e)

    th_map = threading.Thread(target=mapper, daemon=True) if do_stereo else None
    th_ba = threading.Thread(target=estimator, daemon=True) if ba_problems else None
    if th_map:
        th_map.start()
    if th_ba:
        th_ba.start()
        if not th_map:
            pass

    def keyframe(f, kps, age):
        """createKeyframe: detection tops the keypoint set up; the keyframe goes to the mapper."""
        tl = time.perf_counter()
        new = fx.detectSingleScalePyr(trk.cur_pyr, cell, kps, roi)[:max(0, nbmaxkps - len(kps))]
        st["slam_library_s"] += time.perf_counter() - tl
        st["detect_calls"] += 1; st["keyframes"] += 1
        if len(new):
            kps = np.concatenate([kps, new]); age = np.concatenate([age, np.zeros(len(new), np.int32)])
        noise = rng.normal(0, 1.0, kps.shape).astype(np.float32)            # (drawn in every mode: same random stream with and without stereo)
        if do_stereo:
            unpx, _ = calL.computeKeypoints(kps, want_bv=True)
            hp = (age > 0).astype(np.uint8)                                  # keypoints with a map point: right-image prior (:402-413)
            p3 = kps.copy()
            p3[:, 0] -= np.float32(seq.disparity)
            p3 += noise
            kf_consumed[f] = threading.Event()
            map_q.put((f, trk.cur_pyr, seq.right_frame(f), kps.copy(), unpx, p3, hp))
        elif ba_problems:
            ba_q.put(f)
        return kps, age

    t0 = time.perf_counter()
    trk.trackFrame(seq.frame(0), empty, empty, None)
    st["frames"] = 1
    kps, age = keyframe(0, empty, np.zeros(0, np.int32))                     # frame 0 is a keyframe (visual_front_end.cpp:87-95)
    last_kf = 0
    for f in range(1, seq.n_frames):
        if errors:
            break
        gt = seq.flow(kps, f - 1, f)
        has_prior = (age > 0).astype(np.uint8)
        pri = np.where(has_prior[:, None] > 0, gt + rng.normal(0, prior_sigma, gt.shape), kps).astype(np.float32)
        if do_stereo and f >= 2:
            # this frame's preprocessImage overwrites the pyramid of frame f - 2 (two pyramids alternate); if that frame was a
            # keyframe the mapper may still be reading it (the reference's cv::Mat buffers are re-used the same way): wait until
            # stereo matching has consumed it -- whatever kf_every is (with kf_every = 1 every frame's pyramid is shared)
            ev = kf_consumed.get(f - 2)
            if ev is not None and not ev.is_set():
                tw = time.perf_counter(); ev.wait(); st["slam_wait_for_mapper_s"] += time.perf_counter() - tw
        tl = time.perf_counter()
        out, sb, _ = trk.trackFrame(seq.frame(f), kps, pri, has_prior)
        st["slam_library_s"] += time.perf_counter() - tl
        ok = (sb & 1).astype(bool)
        st["frames"] += 1; st["attempted"] += len(kps); st["tracked"] += int(ok.sum())
        if ok.any():
            d = out[ok].astype(np.float64) - gt[ok]
            st["err_sq_sum"] += float((d ** 2).sum()); st["err_n"] += int(ok.sum())
        tl = time.perf_counter()
        unpx_all, bv_all = trk.lastKeypoints(len(out))                       # Frame::updateKeypoint -> computeKeypoint (frame.cpp:246-254):
        st["slam_library_s"] += time.perf_counter() - tl                     # computed by the same enqueue, this is a host copy
        kps, age = out[ok], age[ok] + 1
        inside = (kps[:, 0] > 8) & (kps[:, 0] < w - 9) & (kps[:, 1] > 8) & (kps[:, 1] < h - 9)
        kps, age = kps[inside], age[inside]
        if f % kf_every == 0:
            kps, age = keyframe(f, kps, age)
            last_kf = f
    ctx.sync()
    st["slam_thread_seconds"] = time.perf_counter() - t0
    if th_map:
        map_q.put(None); th_map.join()
    elif th_ba:
        ba_q.put(None)
    if th_ba:
        th_ba.join()
    st["seconds"] = time.perf_counter() - t0                                  # until the mapper and the estimator have drained
    trk.close()
    if ctxB:
        ctxB.close()
    if ctxC:
        ctxC.close()
    if errors:
        raise errors[0]
    return st


# ---------------------------------------------------------------------------------------------------------------------
# the same keyframe cycle driven by a native host program (tools/stream_driver.cpp): what a C++ front-end achieves when the
# per-frame bookkeeping is not numpy
# ---------------------------------------------------------------------------------------------------------------------
def write_case(path, seq, ba_problems, kf_every=5, cell=35, nbmaxkps=308, prior_sigma=1.5):
    """Case file of tools/stream_driver.cpp (little-endian, see read_case there)."""
    import struct
    ba_problems = ba_problems or []
    with open(path, "wb") as f:
        f.write(struct.pack("<8i", seq.w, seq.h, len(seq.views), seq.n_frames, kf_every, cell, nbmaxkps, len(ba_problems)))
        f.write(struct.pack("<2d", seq.disparity, prior_sigma))
        f.write(np.asarray(seq.offs, np.float64).tobytes())
        for v in seq.views:
            f.write(np.ascontiguousarray(v, np.uint8).tobytes())
        for v in seq.right_views:
            f.write(np.ascontiguousarray(v, np.uint8).tobytes())
        for pb in ba_problems:
            write_ba_problem(f, pb)


def write_ba_problem(f, pb):
    """One flat inverse-depth BA problem (synth.make_ba_problem layout) as little-endian binary: the reader is read_case of
    tools/stream_driver.cpp and load() of tools/ref_capture/capture_ba.cpp."""
    import struct
    f.write(struct.pack("<3i", int(pb["n_kf"]), int(pb["n_lm"]), int(pb["n_res"])))
    for name, dt in (("poses", np.float64), ("kf_const", np.uint8), ("invdepth", np.float64), ("lm_anchor_kf", np.int32),
                     ("lm_anchor_uv", np.float64), ("res_type", np.uint8), ("res_kf", np.int32), ("res_lm", np.int32),
                     ("res_uv", np.float64), ("res_sigma", np.float64), ("calib_l", np.float64), ("calib_r", np.float64),
                     ("T_rl", np.float64)):
        f.write(np.ascontiguousarray(pb[name], dt).tobytes())


def read_ba_problem(f):
    """inverse of write_ba_problem: the dict layout of synth.make_ba_problem (without its ground-truth extras)"""
    import struct
    n_kf, n_lm, n_res = struct.unpack("<3i", f.read(12))
    pb = dict(n_kf=n_kf, n_lm=n_lm, n_res=n_res)
    for name, dt, cnt, shape in (("poses", np.float64, 7 * n_kf, (n_kf, 7)), ("kf_const", np.uint8, n_kf, None), ("invdepth", np.float64, n_lm, None),
                                 ("lm_anchor_kf", np.int32, n_lm, None), ("lm_anchor_uv", np.float64, 2 * n_lm, (n_lm, 2)), ("res_type", np.uint8, n_res, None),
                                 ("res_kf", np.int32, n_res, None), ("res_lm", np.int32, n_res, None), ("res_uv", np.float64, 2 * n_res, (n_res, 2)),
                                 ("res_sigma", np.float64, n_res, None), ("calib_l", np.float64, 4, None), ("calib_r", np.float64, 4, None),
                                 ("T_rl", np.float64, 7, None)):
        a = np.frombuffer(f.read(cnt * np.dtype(dt).itemsize), dt).copy()
        pb[name] = a.reshape(shape) if shape else a
    return pb


def build_native_driver(out_dir, name="stream_driver"):
    """g++ tools/<name>.cpp (stream_driver: one camera stream per SLAM thread; lockstep_driver: a rank's sequences in lock-step)
    against the in-tree library; returns the executable's path (raises on failure)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(out_dir, name)
    libdir = os.path.join(root, "ov2slam_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(root, "tools", name + ".cpp"), "-I", root,
                           "-L", libdir, "-lov2slam_hip", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def native_argv(exe, case_path, ba_policy="newest", device=0):
    """Command line of tools/stream_driver.cpp: the rank's GPU travels as argv[3] (one process per GPU, SURVEY 8(e);
    the reference's own protocol starts one process per sequence, benchmark_scripts/euroc_bench.sh:3-27).  `case_path` may be a
    list: the sequences then run CONCURRENTLY inside the one process, three threads and three contexts each."""
    cases = case_path if isinstance(case_path, str) else ",".join(case_path)
    return [exe, cases, ba_policy, str(int(device))]


def run_native_concurrent(exe, case_paths, device=0, ba_policy="newest", concurrency=4):
    """The sequences of `case_paths` on ONE GPU, `concurrency` at a time inside one driver process (a single stream keeps the GPU
    ~5 % busy: the offline batch mode runs several).  The sequences of a wave start streaming together once all of them have
    initialised; the wave lasts from the earliest t_begin to the latest t_end they report.  Returns (per-sequence stats,
    seconds = sum of the waves)."""
    import json
    import subprocess
    stats, seconds = [None] * len(case_paths), 0.0
    for w0 in range(0, len(case_paths), max(1, concurrency)):
        idx = list(range(w0, min(len(case_paths), w0 + max(1, concurrency))))
        r = subprocess.run(native_argv(exe, [case_paths[i] for i in idx], ba_policy, device), capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise RuntimeError("stream_driver failed (%d): %s" % (r.returncode, r.stderr[-500:]))
        lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
        if len(lines) != len(idx):
            raise RuntimeError("stream_driver returned %d result lines for %d sequences" % (len(lines), len(idx)))
        for i, l in zip(idx, lines):
            stats[i] = json.loads(l)
            if int(stats[i].get("device", -1)) != int(device):
                raise RuntimeError("stream_driver ran on device %s, asked for %d" % (stats[i].get("device"), device))
        seconds += max(stats[i]["t_end"] for i in idx) - min(stats[i]["t_begin"] for i in idx)
    return stats, seconds


def run_native(exe, case_path, ba_policy="newest", device=0):
    import json
    import subprocess
    r = subprocess.run(native_argv(exe, case_path, ba_policy, device), capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("stream_driver failed (%d): %s" % (r.returncode, r.stderr[-500:]))
    st = json.loads(r.stdout.strip().splitlines()[-1])
    if int(st.get("device", -1)) != int(device):
        raise RuntimeError("stream_driver ran on device %s, asked for %d" % (st.get("device"), device))
    return st


def lockstep_argv(exe, case_paths, ba_policy="newest", device=0, loader_threads=4, priorities=False, batched_estimator=True, host_workers=3):
    """Command line of tools/lockstep_driver.cpp: ALL of the rank's sequences in one process, advanced one frame per step through
    the lock-step tracker (ov2_btracker_*); the rank's GPU travels as argv[3] like stream_driver's."""
    return [exe, ",".join(case_paths), ba_policy, str(int(device)), str(int(loader_threads)), priorities if isinstance(priorities, str) else ("1" if priorities else "0"), str(int(batched_estimator)), str(int(host_workers))]


def run_lockstep(exe, case_paths, device=0, ba_policy="newest", loader_threads=4, timeout=1800, priorities=False, batched_estimator=True, host_workers=3):
    """-> (per-sequence stats in the order of case_paths, summary dict with frames / seconds / steps of the whole rank)"""
    import json
    import subprocess
    r = subprocess.run(lockstep_argv(exe, case_paths, ba_policy, device, loader_threads, priorities, batched_estimator, host_workers), capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError("lockstep_driver failed (%d): %s" % (r.returncode, r.stderr[-500:]))
    lines = [json.loads(l) for l in r.stdout.strip().splitlines() if l.startswith("{")]
    summary = [l for l in lines if l.get("lockstep_summary")]
    stats = [l for l in lines if not l.get("lockstep_summary")]
    if len(summary) != 1 or len(stats) != len(case_paths):
        raise RuntimeError("lockstep_driver returned %d result lines for %d sequences" % (len(stats), len(case_paths)))
    for st in stats + summary:
        if int(st.get("device", -1)) != int(device):
            raise RuntimeError("lockstep_driver ran on device %s, asked for %d" % (st.get("device"), device))
    return stats, summary[0]
