"""ov2slam_amd -- MI355X-native (gfx950 HIP) implementation of OV2SLAM's front-end +
local-BA hot path behind the reference's FeatureExtractor / FeatureTracker /
Optimizer::localBA interfaces.  See DESIGN.md and INTEGRATION.md."""
from ._lib import load, Ov2Error, LIB_PATH  # noqa: F401
from .frontend import Context, Pyramid, FeatureTracker, FeatureExtractor, CLAHE, CameraCalibration, VisualFrontEndTracker, LockstepTracker  # noqa: F401
from .optimizer import Optimizer, MultiViewGeometry  # noqa: F401
from . import optimizer  # noqa: F401
