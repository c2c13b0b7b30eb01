"""The keyframe cycle on three contexts (ov2slam_amd/stream.py): the front-end results do not depend on whether the mapper and
estimator threads run beside it, every keyframe is stereo-matched in order, the estimator follows the reference's queue policy."""
import numpy as np
import pytest

import ov2slam_amd
from ov2slam_amd import batch, stream, synth

pytestmark = pytest.mark.gpu


def test_stream_runs_the_three_stages_and_front_end_is_unaffected(gpu_ctx):
    tex = synth.base_texture(1400, 1234)
    seq = batch.SyntheticSequence("t", 31, seed=1000, tex=tex, stereo=True)
    windows = [synth.make_ba_problem(10, 300, 6, stereo=True, seed=3)]
    full = stream.run_stream(gpu_ctx, seq, kf_every=5, ba_problems=windows, ba_policy="all")
    alone = stream.run_stream(gpu_ctx, batch.SyntheticSequence("t", 31, seed=1000, tex=tex), kf_every=5, do_stereo=False)
    # same frames, same keypoints tracked, same tracking error: the other two contexts only read what the front-end produced
    for k in ("frames", "tracked", "attempted", "err_n", "keyframes"):
        assert full[k] == alone[k], k
    assert abs(full["err_sq_sum"] - alone["err_sq_sum"]) < 1e-9
    assert full["frames"] == 31 and full["keyframes"] == 7
    assert full["stereo_kfs"] == full["keyframes"] and full["stereo_ok"] > 0.9 * full["stereo_kps"]
    assert full["ba_solves"] == full["keyframes"] and full["ba_skipped_kfs"] == 0 and full["ba_iterations"] > 0
    assert full["tracked"] > 0.95 * full["attempted"]
    # the reference's estimator policy: solves + skipped keyframes = keyframes
    newest = stream.run_stream(gpu_ctx, seq, kf_every=5, ba_problems=windows, ba_policy="newest")
    assert newest["ba_solves"] + newest["ba_skipped_kfs"] == newest["keyframes"] and newest["ba_solves"] >= 1
    # run_sequence (config 5) is this loop
    st = batch.run_sequence(gpu_ctx, seq, ba_problems=windows)
    assert st["frames"] == 31 and st["stereo_kfs"] == st["keyframes"]


def test_lockstep_driver_is_bit_identical_to_the_per_stream_driver(tmp_path):
    """BASELINE configs[4] on one rank, two ways: every sequence through its own SLAM thread and tracker (tools/stream_driver.cpp) and
    all of them in lock-step through ov2_btracker_* (tools/lockstep_driver.cpp).  Same frames, same keypoints, same random streams:
    the digests of everything the library returned per sequence -- tracked positions, status bits, undistorted pixels and bearing
    vectors of every frame; the detector's points and adaptive threshold at every keyframe; right-image positions and verdicts of
    stereo matching -- must be equal, and so must the counters.  Sequences of different length drop out of the batch on the way."""
    tex = synth.base_texture(1400, 1234)
    windows = [synth.make_ba_problem(10, 300, 6, stereo=True, seed=3)]
    lengths = [23, 41, 31, 41, 12]                              # unsorted on purpose: the driver orders its batch longest first
    cases = []
    for i, n in enumerate(lengths):
        seq = batch.SyntheticSequence("s%d" % i, n, seed=2000 + i, tex=tex, stereo=True)
        cases.append(str(tmp_path / ("case%d.bin" % i)))
        stream.write_case(cases[-1], seq, windows)
    exe_s = stream.build_native_driver(str(tmp_path), "stream_driver")
    exe_l = stream.build_native_driver(str(tmp_path), "lockstep_driver")
    ref = [stream.run_native(exe_s, c, ba_policy="all") for c in cases]
    # (estimator forms: one batch over the rank's sequences, two groups, the round's first form -- a thread and context per sequence)
    # and the per-sequence host work of a step on 0 .. 4 worker threads beside the SLAM thread
    for loaders, est, prio, workers in ((1, 1, False, 3), (3, 1, False, 0), (2, 2, False, 4), (2, 0, True, 1)):
        got, summary = stream.run_lockstep(exe_l, cases, ba_policy="all", loader_threads=loaders, batched_estimator=est, priorities=prio, host_workers=workers)
        assert summary["batched_estimator"] == est and (summary["ba_problems"] == sum(b["ba_solves"] for b in got)) == (est > 0)
        assert summary["frames"] == sum(lengths) and summary["steps"] == max(lengths) and summary["sequences"] == len(lengths)
        for i, (a, b) in enumerate(zip(ref, got)):
            for k in ("frames", "tracked", "attempted", "err_n", "keyframes", "stereo_kfs", "stereo_ok", "stereo_kps", "ba_solves",
                      "track_digest", "detect_digest", "stereo_digest"):
                assert a[k] == b[k], "sequence %d: %s differs (%s vs %s)" % (i, k, a[k], b[k])
            assert abs(a["err_sq_sum"] - b["err_sq_sum"]) < 1e-6 * max(1.0, a["err_sq_sum"])
            assert b["mode"] == "lockstep" and a["mode"] == "stream" and b["frames"] == lengths[i]
            assert b["ba_solves"] == b["keyframes"] and b["ba_skipped_kfs"] == 0
    # the reference's estimator policy in lock-step: solves + skipped keyframes = keyframes, per sequence
    got, _ = stream.run_lockstep(exe_l, cases, ba_policy="newest")
    for b in got:
        assert b["ba_solves"] + b["ba_skipped_kfs"] == b["keyframes"] and b["ba_solves"] >= 1
