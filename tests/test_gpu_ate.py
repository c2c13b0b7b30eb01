"""A pose trajectory through the hot path: the synthetic stereo odometry of tools/ate_synthetic.py (per frame preprocessImage + kltTracking +
ceresPnP, every fifth frame detectSingleScale + stereoMatching + new map points anchored with the estimated pose -- and, in its second mode,
Optimizer::localBA over the last eight keyframes at every keyframe; exact ground truth: a fronto-parallel plane under a rolling, translating
pinhole camera) run once on the product and once on the oracle, each with its own state.
The absolute trajectory error of the two must agree (BASELINE.json's "at matched ATE"), pose by pose."""
import os, sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_trajectory_and_ate_match_the_oracle(gpu_ctx, oracle):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ate_synthetic as A
    res, traj = A.main(121, 30)
    for mode, tol in (("pnp", 1e-7), ("pnp_and_local_ba", 1e-6)):
        g, o, d = res[mode]["gpu"], res[mode]["oracle"], res[mode]["gpu_vs_oracle"]
        # the front end is bit-exact, so both runs track, match and gate the same points and take the same solver decisions (iteration counts,
        # outlier blocks); the poses differ by the solvers' rounding only (the BA sums with fp64 atomics in arrival order)
        assert d["same_counters"], (mode, g, o)
        assert d["max_position_difference_m"] <= tol and d["max_quaternion_difference"] <= tol, (mode, d)
        assert d["ate_difference_m"] <= 0.1 * tol, (mode, d)
        # and the odometry works: sub-millimetre on a 2 m path, every frame's pose from PnP, keyframes every fifth frame
        assert g["ate_rmse_m"] < 1e-3 and o["ate_rmse_m"] < 1e-3 and g["pnp_failed"] == 0 and g["keyframes"] == 25, (mode, g)
        assert g["tracked"] > 0.95 * g["attempted"] > 30000 and g["pnp_points"] > 30000
        assert np.isfinite(traj[mode]["gpu"]).all()
    assert res["pnp_and_local_ba"]["gpu"]["ba_solves"] == 24 and res["pnp_and_local_ba"]["gpu"]["ba_blocks"] > 50000
