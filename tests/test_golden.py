"""Oracle regression digests (tests/golden/oracle_digests.json, produced by tests/golden/make_golden.py).
Self-pinned: the reference provides no golden vectors for this path (see the script's docstring)."""
import importlib.util
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_matches_committed_digests(oracle):
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(HERE, "golden", "oracle_digests.json")))
    got = mod.compute()
    assert got == want, {k: (got[k], want[k]) for k in want if got.get(k) != want[k]}


def test_cpp_adapters_compile():
    """The C++ mirrors of FeatureTracker / FeatureExtractor / Optimizer (ov2slam_amd/host) stay in sync with the C ABI."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "ov2slam_amd", "host", "compile_check.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_cpp_adapters_compile_with_cv_types():
    """The same adapters with the reference's own value types (-DOV2_WITH_OPENCV: cv::Point2f, cv::Rect, cv::Mat), checked
    against a minimal stand-in for <opencv2/core.hpp> (tests/fake_opencv) because this image has no OpenCV."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "ov2slam_amd", "host", "compile_check.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-DOV2_WITH_OPENCV",
                        "-I" + os.path.join(HERE, "fake_opencv"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_verbatim_signatures_compile_and_link():
    """ov2slam_amd/host/verbatim.hpp (the reference's exact cv:: signatures over a thread-local context and a hash-validated pyramid
    cache) and its GPU test driver compile and LINK against the library with -DOV2_WITH_OPENCV and the stand-in <opencv2/core.hpp>."""
    import tempfile
    root = os.path.dirname(HERE)
    libdir = os.path.join(root, "ov2slam_amd")
    if not os.path.exists(os.path.join(libdir, "libov2slam_hip.so")):
        import pytest
        pytest.skip("libov2slam_hip.so not built")
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run(["g++", "-std=c++17", "-O0", "-Wall", "-Werror", "-DOV2_WITH_OPENCV", "-I" + os.path.join(HERE, "fake_opencv"),
                            os.path.join(HERE, "cpp", "verbatim_run.cpp"), "-o", os.path.join(td, "v"), "-L", libdir, "-lov2slam_hip",
                            "-Wl,-rpath," + libdir], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
