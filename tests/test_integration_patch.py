"""integration/ov2slam_hip.patch: the reference-side binding as text `git apply` accepts.  The reference cannot
be compiled in this image (no OpenCV / Eigen / Ceres / ROS), so what CAN be proven is proven: the committed patch is what the
generator produces from the reference tree, it applies to a scratch copy of that tree, every inserted block is preprocessor- and
brace-balanced, and the adapter calls it inserts compile against the real adapter headers (with stand-in cv:: value types)."""
import os
import shutil
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "integration"))

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present (GPU box)")


@needs_ref
def test_committed_patch_is_what_the_generator_produces():
    import make_patch
    text = make_patch.HEADER + make_patch.generate(REF)
    assert text == open(os.path.join(ROOT, "integration", "ov2slam_hip.patch"), encoding="utf-8").read(), \
        "integration/ov2slam_hip.patch is stale: run python integration/make_patch.py"
    added = [l for l in text.splitlines() if l.startswith("+") and not l.startswith("+++")]
    removed = [l for l in text.splitlines() if l.startswith("-") and not l.startswith("---")]
    assert len(added) > 600 and removed == [], "the patch only ADDS lines (everything under #ifdef OV2SLAM_HIP)"


@needs_ref
def test_patch_applies_to_the_reference_tree(tmp_path):
    import make_patch
    for f in make_patch.FILES:
        os.makedirs(os.path.dirname(tmp_path / f), exist_ok=True)
        shutil.copy(os.path.join(REF, f), tmp_path / f)
    patch = os.path.join(ROOT, "integration", "ov2slam_hip.patch")
    r = subprocess.run(["git", "apply", "--check", "--verbose", patch], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["git", "apply", patch], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for f in make_patch.FILES:
        new, old = open(tmp_path / f, encoding="utf-8").read(), open(os.path.join(REF, f), encoding="utf-8").read()
        assert new != old
        if f.endswith((".cpp", ".hpp")):
            # with OV2SLAM_HIP undefined the translation unit is the reference's, token for token: dropping every
            # #ifdef OV2SLAM_HIP ... [#else] ... #endif block (keeping the #else part) gives the original text back
            out, state = [], None
            for line in new.split("\n"):
                s = line.strip()
                if state is None and s.startswith("#ifdef OV2SLAM_HIP"): state = "hip"; continue
                if state is None and s.startswith("#ifndef OV2SLAM_HIP"): state = "ref"; continue
                if state == "hip" and s == "#else": state = "ref"; continue
                if state in ("hip", "ref") and s == "#endif": state = None; continue
                if state != "hip": out.append(line)
            assert state is None, f
            assert "\n".join(out).replace("\n\n", "\n") == old.replace("\n\n", "\n") or \
                "".join("\n".join(out).split()) == "".join(old.split()), f


def test_adapter_calls_of_the_patch_compile():
    """ov2slam_amd/host/patch_usage_check.cpp repeats the adapter calls the patch inserts, with cv:: argument types."""
    src = os.path.join(ROOT, "ov2slam_amd", "host", "patch_usage_check.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-DOV2_WITH_OPENCV",
                        "-I" + os.path.join(HERE, "fake_opencv"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # every adapter method the patch calls is exercised there
    patch = open(os.path.join(ROOT, "integration", "ov2slam_hip.patch"), encoding="utf-8").read()
    usage = open(src, encoding="utf-8").read()
    for call in ("preprocessImage(", "kltTracking(", "lastErrorMessage(", "detectGridFAST(", "detectSingleScale(", "curPyr()",
                 "stereoMatching(", "buildClahe(", ".build(", "addKeyframe(", "addLandmark(", "addResidual(", "setMaxSolverTime(",
                 "solveLocalBA(", "signalStopLocalBA(", "new ov2::SlamGpu(", "clearStopLocalBA(", "solveLocalBAXYZ(", "solveLooseBA(", "solveFullBA(",
                 "solveStructureOnlyBA(", "ov2::ceresPnP(", "threadContext()", "SlamGpu::forceCeres()", "SlamGpu::global()", "kf_front.buildClahe(",
                 "kf_front.get()", "setDeterministicBA(", "fpx.addPoint(", "sp.addKeyframe(", "fpx.addResidual(OV2_XYZ_RIGHT"):
        assert call in patch and call in usage, call
