"""Offline batch-of-sequences launcher logic (BASELINE.json config[4]).

The hot path does not shard inside a sequence, so multi-GPU = replicas: every rank (one process per
GPU) processes whole sequences; the only cross-rank traffic is a barrier and an all-gather of a few
floats per rank (frames, seconds, BA iterations, ATE) over torch.distributed (backend "nccl" = RCCL
on the GPU box, "gloo" in the CPU tests).  Nothing here touches the data path."""
import numpy as np

# stereo frame counts of the 11 EuRoC MAV sequences (public dataset figures; the reference's
# benchmark script loops MH_01..MH_05 only, benchmark_scripts/euroc_bench.sh:7)
EUROC_FRAMES = {"MH_01": 3682, "MH_02": 3040, "MH_03": 2700, "MH_04": 2033, "MH_05": 2273,
                "V1_01": 2912, "V1_02": 1710, "V1_03": 2149, "V2_01": 2280, "V2_02": 2348, "V2_03": 1922}


def assign_sequences(frame_counts, world):
    """Longest-processing-time-first assignment of whole sequences to ranks.
    frame_counts: dict name -> frames.  Returns a list (per rank) of lists of names."""
    loads = [0] * world
    out = [[] for _ in range(world)]
    for name, n in sorted(frame_counts.items(), key=lambda kv: (-kv[1], kv[0])):
        r = int(np.argmin(loads))
        out[r].append(name)
        loads[r] += n
    return out


def gather_stats(local, group=None):
    """All-gather a dict of floats from every rank; returns {key: [v_rank0, v_rank1, ...]}.
    Works without an initialised process group (single process)."""
    import torch
    import torch.distributed as dist
    keys = sorted(local)
    t = torch.tensor([float(local[k]) for k in keys], dtype=torch.float64)
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        outs = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
        dist.all_gather(outs, t, group=group)
        outs = [o.cpu() for o in outs]
    else:
        outs = [t]
    return {k: [float(o[i]) for o in outs] for i, k in enumerate(keys)}


def aggregate(stats):
    """Whole-job numbers from gather_stats output: throughput uses the slowest rank's time."""
    frames = sum(stats["frames"])
    seconds = max(stats["seconds"])
    out = {"frames": frames, "seconds": seconds, "fps": frames / seconds if seconds > 0 else 0.0}
    if "ba_iterations" in stats and "ba_seconds" in stats:
        s = max(stats["ba_seconds"])
        out["ba_iters_per_s"] = sum(stats["ba_iterations"]) / s if s > 0 else 0.0
    if "ate_sq_sum" in stats and "ate_n" in stats:
        n = sum(stats["ate_n"])
        out["ate_rmse"] = (sum(stats["ate_sq_sum"]) / n) ** 0.5 if n > 0 else 0.0
    return out


def ate_rmse(est_xyz, gt_xyz):
    """Absolute trajectory error after a rigid (Umeyama, no scale) alignment -- what is evaluated off-repo
    from the TUM files written by include/logger.hpp:135-160.  Returns (rmse, sum of squared errors, n)."""
    est = np.asarray(est_xyz, np.float64); gt = np.asarray(gt_xyz, np.float64)
    mu_e, mu_g = est.mean(0), gt.mean(0)
    Hm = (est - mu_e).T @ (gt - mu_g)
    U, _, Vt = np.linalg.svd(Hm)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    err = (R @ (est - mu_e).T).T + mu_g - gt
    sq = float((err ** 2).sum())
    return (sq / len(est)) ** 0.5, sq, len(est)


# ---------------------------------------------------------------------------------------------------------
# config 5 runner: whole sequences through the single-sequence hot path, one process per GPU
# ---------------------------------------------------------------------------------------------------------
class SyntheticSequence:
    """Synthetic stand-in of one EuRoC sequence (no dataset offline): `n_frames` 752x480 views of a
    seeded texture under a smooth similarity motion (a short cycle of distinct views, replayed back and forth, so that a
    sequence costs 0.2 s to generate instead of 30 ms per frame), with the ground-truth flow between consecutive frames."""

    def __init__(self, name, n_frames, seed, w=752, h=480, n_views=7, tex=None, stereo=False, disparity=20.0):
        from . import synth
        self.name, self.n_frames, self.w, self.h = name, int(n_frames), w, h
        rng = np.random.default_rng(seed)
        tex = synth.base_texture(1400, 1234) if tex is None else tex
        ox, oy, th = float(rng.uniform(100, 400)), float(rng.uniform(100, 400)), 0.0
        self.views, self.offs, self.right_views = [], [], []
        self.disparity = float(disparity)
        for _ in range(n_views):
            self.views.append(synth.warp(tex, w, h, ox, oy, th)); self.offs.append((ox, oy, th))
            if stereo:      # rectified right camera: right(x, y) = left(x + d, y) -- a fronto-parallel scene at constant disparity d
                self.right_views.append(synth.warp(tex, w, h, ox + disparity * np.cos(th), oy + disparity * np.sin(th), th))
            ox += rng.uniform(-5, 5); oy += rng.uniform(-4, 4); th += rng.uniform(-0.006, 0.006)
        self.seed = seed

    def view_index(self, f):
        n = len(self.views)
        k = f % (2 * n - 2)
        return k if k < n else 2 * n - 2 - k

    def frame(self, f):
        return self.views[self.view_index(f)]

    def right_frame(self, f):
        return self.right_views[self.view_index(f)]

    def flow(self, pts, fa, fb):
        """ground-truth position in frame fb of pixels `pts` of frame fa"""
        w, h = self.w, self.h
        cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
        (ox0, oy0, t0), (ox1, oy1, t1) = self.offs[self.view_index(fa)], self.offs[self.view_index(fb)]
        pts = np.asarray(pts, np.float64)
        dx, dy = pts[:, 0] - cx, pts[:, 1] - cy
        c, s = np.cos(t0), np.sin(t0)
        tx, ty = c * dx - s * dy + cx + ox0, s * dx + c * dy + cy + oy0
        dx, dy = tx - cx - ox1, ty - cy - oy1
        c, s = np.cos(-t1), np.sin(-t1)
        return np.stack([c * dx - s * dy + cx, s * dx + c * dy + cy], 1)


def run_sequence(ctx, seq, kf_every=5, cell=35, nbmaxkps=308, prior_sigma=1.5, use_graph=True, ba_problems=None):
    """One sequence through the GPU hot path the way the reference schedules it (ov2slam_amd.stream.run_stream): the SLAM
    thread on `ctx` (per frame preprocessImage + kltTracking + computeKeypoint, at every `kf_every`-th frame the min-eigenvalue
    grid detector tops the keypoint set up, MapManager::extractKeypoints, map_manager.cpp:286-341), and -- when the sequence
    carries right images / `ba_problems` is given -- the mapper thread's right-image pyramid + stereo matching and the
    estimator thread's two-pass localBA per keyframe on contexts of their own, concurrently.  Keypoints that were tracked
    before carry a prior (true flow + noise, standing in for the motion model's projection of their map point).
    Returns the counters of run_stream (frames, seconds, tracked, attempted, err_sq_sum, err_n, detect_calls, stereo_*, ba_*)."""
    from . import stream
    return stream.run_stream(ctx, seq, kf_every=kf_every, cell=cell, nbmaxkps=nbmaxkps, prior_sigma=prior_sigma, use_graph=use_graph,
                             do_stereo=bool(seq.right_views), ba_problems=ba_problems)
