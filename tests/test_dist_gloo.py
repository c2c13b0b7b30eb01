"""N>1 path on CPU: world_size-2 gloo process group exercising the batch launcher logic
(sequence assignment + all-gather of timings / ATE) that bench.py --gpus N and the offline
batch-of-sequences mode rely on."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ov2slam_amd import batch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = batch.assign_sequences(batch.EUROC_FRAMES, world)
    mine = plan[rank]
    frames = sum(batch.EUROC_FRAMES[s] for s in mine)
    # a synthetic trajectory per rank: rotated + translated + 1 cm noise -> ATE ~ noise level
    rng = np.random.default_rng(rank)
    gt = np.cumsum(rng.normal(0, 0.1, (200, 3)), 0)
    Rz = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    est = (Rz @ gt.T).T + np.array([1.0, 2.0, 3.0]) + rng.normal(0, 0.01, gt.shape)
    rmse, sq, n = batch.ate_rmse(est, gt)
    dist.barrier()
    stats = batch.gather_stats({"frames": frames, "seconds": 10.0 + rank, "ba_iterations": 100 * (rank + 1),
                                "ba_seconds": 2.0, "ate_sq_sum": sq, "ate_n": n})
    agg = batch.aggregate(stats)
    t = torch.tensor([10.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)             # what bench.py does with the elapsed time
    q.put((rank, mine, agg, float(t.item()), rmse))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, seqs0, agg0, tmax0, rmse0), (r1, seqs1, agg1, tmax1, rmse1) = res
    assert sorted(seqs0 + seqs1) == sorted(batch.EUROC_FRAMES)           # every sequence exactly once
    assert agg0 == agg1                                                  # all ranks see the same aggregate
    assert agg0["frames"] == sum(batch.EUROC_FRAMES.values())
    assert agg0["seconds"] == 11.0 and tmax0 == tmax1 == 11.0            # max over ranks
    assert abs(agg0["fps"] - agg0["frames"] / 11.0) < 1e-9
    assert agg0["ba_iters_per_s"] == 150.0
    assert 0.005 < agg0["ate_rmse"] < 0.03 and rmse0 < 0.03 and rmse1 < 0.03


def test_assign_sequences_is_balanced():
    for world in (1, 2, 4, 8):
        plan = batch.assign_sequences(batch.EUROC_FRAMES, world)
        assert sorted(sum(plan, [])) == sorted(batch.EUROC_FRAMES)
        loads = [sum(batch.EUROC_FRAMES[s] for s in p) for p in plan]
        assert max(loads) <= 1.35 * (sum(loads) / world) + max(batch.EUROC_FRAMES.values()) * (world >= 8)


def test_gather_stats_single_process():
    s = batch.gather_stats({"frames": 5, "seconds": 2.0})
    assert s == {"frames": [5.0], "seconds": [2.0]} and batch.aggregate(s)["fps"] == 2.5


def test_bench_launcher_dry_gloo():
    """bench.py's own launcher: `--gpus 2 --backend gloo --dry` spawns two ranks from ONE command and rank 0 reports
    n_gpus: 2; the config-5 plan covers every sequence once and the aggregate uses the slowest rank's time."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["dry"] is True
    assert out["ranks_seen"] == 2 and len(out["pci_bus_id_per_rank"]) == 2      # all_reduce(SUM) of ones over the process group
    c5 = out["config5"]
    assert sorted(sum(c5["assignment"], [])) == sorted(batch.EUROC_FRAMES)
    assert len(c5["frames_per_rank"]) == 2 and c5["frames"] == sum(c5["frames_per_rank"])
    assert abs(c5["seconds_slowest_rank"] - max(c5["seconds_per_rank"])) < 1e-12
    assert abs(out["elapsed_max_over_ranks"] - 1.1) < 1e-12                # max over ranks of (1.0, 1.1)
    assert c5["device_per_rank"] == [0, 1]                                  # every rank's config-5 worker on ITS GPU
    # the lock-step driver: ONE process per rank takes all of the rank's sequences (full EuRoC lengths by default) and its device
    argv = c5["lockstep_argv_rank0"]
    assert argv[1].split(",") == ["<case:%s>" % s for s in c5["assignment"][0]] and argv[2] == "newest" and argv[3] == "0"
    assert c5["frames"] == sum(batch.EUROC_FRAMES.values())
    assert c5["steps_per_rank"] == [max(batch.EUROC_FRAMES[s] for s in plan) for plan in c5["assignment"]]    # a rank steps as long as its longest sequence


def test_native_driver_receives_the_ranks_device():
    """The config-5 worker is a separate process (tools/stream_driver.cpp): its three contexts must sit on the rank's GPU.
    The device index travels as argv[3]; the driver source creates every context on it and echoes it back."""
    from ov2slam_amd import stream
    assert stream.native_argv("drv", "case.bin", "newest", 5) == ["drv", "case.bin", "newest", "5"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "tools", "stream_driver.cpp")).read()
    assert "ov2_ctx_create(0" not in src and src.count("ov2_ctx_create(device") == 3
    # (defaults: no stream priorities, the rank's estimator batched over its sequences; the round's first form stays selectable)
    assert stream.lockstep_argv("drv", ["a.bin", "b.bin"], "all", 3, 2) == ["drv", "a.bin,b.bin", "all", "3", "2", "0", "1", "3"]
    assert stream.lockstep_argv("drv", ["a.bin"], "newest", 0, 4, True, False)[-3:] == ["1", "0", "3"]
    src = open(os.path.join(root, "tools", "lockstep_driver.cpp")).read()          # the rank's lock-step host: SLAM, mapper and estimator contexts
    assert "ov2_ctx_create(0" not in src and "ov2_ctx_create_with_priority(0" not in src and src.count("ov2_ctx_create_with_priority(device") == 4     # (+ the batched estimator's)
    import inspect
    import bench
    assert "device=dev.index" in inspect.getsource(bench.main) and "device=device" in inspect.getsource(bench.run_config5)


def test_bench_launcher_refuses_missing_gpus():
    """`--gpus 2` without two visible GPUs must fail loudly instead of silently running one rank."""
    import subprocess
    import sys
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode == 2 and "GPU(s) visible" in p.stderr
