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
