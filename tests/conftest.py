import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def _gpu_available():
    try:
        import ov2slam_amd
        ctx = ov2slam_amd.Context(0)
        ctx.close()
        return True
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    import ov2slam_amd
    ctx = ov2slam_amd.Context(0)     # raises loudly if the .so or the GPU is missing
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def euroc_pair():
    """Synthetic 752x480 frame pair + grid keypoints + noisy priors."""
    from ov2slam_amd import synth
    prev, cur, flow = synth.frame_pair(752, 480, seed=1234)
    rng = np.random.default_rng(7)
    kps = synth.grid_keypoints(752, 480, 35, rng)
    gt = flow(kps)
    pri = (gt + rng.normal(0, 1.5, gt.shape)).astype(np.float32)
    return dict(prev=prev, cur=cur, flow=flow, kps=kps, gt=gt, pri=pri)
