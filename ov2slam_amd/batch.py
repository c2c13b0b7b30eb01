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
