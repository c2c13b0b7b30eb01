"""The C++ adapters of ov2slam_amd/host/*.hpp -- the classes a maintainer of the reference would actually call (FrameTracker,
FeatureExtractor, FeatureTracker + Pyramid, Optimizer) -- EXECUTED, not only syntax-checked: tests/cpp/
adapter_run.cpp is compiled with g++ against libov2slam_hip.so, run on the GPU on a case file, and what the adapters return
must equal what the ctypes mirrors return for the same inputs (same library: bit-exact for the front end) and, for the tracker,
what the oracle computes."""
import os
import struct
import subprocess

import numpy as np
import pytest

import ov2slam_amd
from ov2slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _wr(f, a):
    b = np.ascontiguousarray(a).tobytes()
    f.write(struct.pack("<q", len(b))); f.write(b)


def _rd(f, dt):
    (n,) = struct.unpack("<q", f.read(8))
    return np.frombuffer(f.read(n), dt).copy()


def _bits(a):
    return np.ascontiguousarray(a, np.float32).reshape(-1).view(np.uint32)


def test_cpp_adapters_run_and_match(gpu_ctx, oracle, tmp_path):
    exe = tmp_path / "adapter_run"
    libdir = os.path.join(ROOT, "ov2slam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "adapter_run.cpp"),
                           "-o", str(exe), "-L", libdir, "-lov2slam_hip", "-Wl,-rpath," + libdir])
    w, h, cell = 752, 480, 35
    prev, cur, flow = synth.frame_pair(w, h, seed=11, shift=(3.1, -2.2), theta=0.004)
    rng = np.random.default_rng(4)
    kps = synth.grid_keypoints(w, h, cell, rng)
    hp = (rng.uniform(size=len(kps)) < 0.7).astype(np.uint8)
    pri = np.where(hp[:, None] > 0, flow(kps) + rng.normal(0, 1.0, kps.shape), kps).astype(np.float32)
    bad = (hp > 0) & (rng.uniform(size=len(kps)) < 0.2)
    pri[bad] += rng.normal(0, 12.0, (int(bad.sum()), 2)).astype(np.float32)      # some priors far off: the retry path
    pb = synth.make_ba_problem(12, 400, 8, stereo=True, seed=3)
    case, res = tmp_path / "case.bin", tmp_path / "res.bin"
    with open(case, "wb") as f:
        _wr(f, np.array([w, h, cell], np.int32)); _wr(f, prev); _wr(f, cur)
        _wr(f, kps.astype(np.float32)); _wr(f, pri); _wr(f, hp)
        _wr(f, np.asarray(pb["poses"], np.float64)); _wr(f, np.asarray(pb["kf_const"], np.uint8)); _wr(f, np.asarray(pb["invdepth"], np.float64))
        _wr(f, np.asarray(pb["lm_anchor_kf"], np.int32)); _wr(f, np.asarray(pb["lm_anchor_uv"], np.float64))
        _wr(f, np.asarray(pb["res_type"], np.uint8)); _wr(f, np.asarray(pb["res_kf"], np.int32)); _wr(f, np.asarray(pb["res_lm"], np.int32))
        _wr(f, np.asarray(pb["res_uv"], np.float64)); _wr(f, np.asarray(pb["res_sigma"], np.float64))
        _wr(f, np.concatenate([pb["calib_l"], pb["calib_r"], pb["T_rl"]]).astype(np.float64))
    r = subprocess.run([str(exe), str(case), str(res)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    with open(res, "rb") as f:
        c_out, c_st, c_p3p = _rd(f, np.float32).reshape(-1, 2), _rd(f, np.uint8), int(_rd(f, np.int32)[0])
        c_det, c_q = _rd(f, np.float32).reshape(-1, 2), float(_rd(f, np.float64)[0])
        c_back, c_bst = _rd(f, np.float32).reshape(-1, 2), _rd(f, np.uint8)
        c_fast, c_th = _rd(f, np.float32).reshape(-1, 2), int(_rd(f, np.int32)[0])
        c_ss = _rd(f, np.float32).reshape(-1, 2)
        c_fb, c_fbst = _rd(f, np.float32).reshape(-1, 2), _rd(f, np.uint8)
        c_sr, c_sok = _rd(f, np.float32).reshape(-1, 2), _rd(f, np.uint8)
        c_flags, c_poses, c_lam, c_badobs = _rd(f, np.int32), _rd(f, np.float64), _rd(f, np.float64), _rd(f, np.uint8)
        c_stop = _rd(f, np.int32)

    # ---- FrameTracker vs the ctypes mirror and vs the oracle ----
    roi = (5, 5, w - 10, h - 10)
    empty = np.zeros((0, 2), np.float32)
    trk = ov2slam_amd.VisualFrontEndTracker(gpu_ctx, w, h, use_clahe=True, fclahe_val=3.0, nbmaxkps=512)
    trk.trackFrame(prev, empty, empty, None)
    g_out, g_st, g_p3p = trk.trackFrame(cur, kps, pri, hp)
    assert np.array_equal(_bits(c_out), _bits(g_out)) and np.array_equal(c_st.astype(bool), (g_st & 1).astype(bool)) and c_p3p == int(g_p3p)
    a, b = oracle.clahe(prev, 3.0, w // 50, h // 50), oracle.clahe(cur, 3.0, w // 50, h // 50)
    o_out, o_ok, o_retry, o_p3p = oracle.klt_tracking(oracle.Pyramid(a, 9, 3), oracle.Pyramid(b, 9, 3), kps, pri, hp)
    assert np.array_equal(_bits(c_out), _bits(o_out)) and np.array_equal(c_st.astype(bool), o_ok) and o_retry.any()
    fx = ov2slam_amd.FeatureExtractor(gpu_ctx, nfast_th=10, dmaxquality=0.001)
    cur_kps = g_out[(g_st & 1) > 0][:40]
    g_det = fx.detectSingleScalePyr(trk.cur_pyr, cell, cur_kps, roi)
    assert len(c_det) > 50 and np.array_equal(_bits(c_det), _bits(g_det)) and c_q == fx.dmaxquality_
    trk.preprocessImage(prev)
    g_back, g_bst, _ = trk.kltTracking(kps, kps, hp, klt_use_prior=False)
    assert np.array_equal(_bits(c_back), _bits(g_back)) and np.array_equal(c_bst.astype(bool), (g_bst & 1).astype(bool))
    trk.close()
    # ---- FeatureExtractor on the host image ----
    fx = ov2slam_amd.FeatureExtractor(gpu_ctx, nfast_th=10, dmaxquality=0.001)
    g_fast = fx.detectGridFAST(cur, cell, empty)
    assert len(c_fast) > 50 and np.array_equal(_bits(c_fast), _bits(g_fast)) and c_th == fx.nfast_th_
    g_ss = fx.detectSingleScale(cur, cell, empty, roi)
    assert np.array_equal(_bits(c_ss), _bits(g_ss))
    # ---- FeatureTracker::fbKltTracking on two Pyramids ----
    Gp = ov2slam_amd.Pyramid(gpu_ctx, w, h, 9, 3).build(prev); Gc = ov2slam_amd.Pyramid(gpu_ctx, w, h, 9, 3).build(cur)
    g_fb, g_fbst = ov2slam_amd.FeatureTracker(gpu_ctx, 30, 0.01).fbKltTracking(Gp, Gc, 9, 3, 30., 0.5, kps, pri)
    assert np.array_equal(_bits(c_fb), _bits(g_fb)) and np.array_equal(c_fbst.astype(bool), g_fbst.astype(bool)) and c_fbst.mean() > 0.5
    # ---- FeatureTracker::stereoMatching (ov2_stereo_match) == the ctypes mirror's fused and multi-call flows ----
    from ov2slam_amd import stereo
    cal = ov2slam_amd.CameraCalibration(gpu_ctx, "pinhole", 458.654, 457.296, 367.215, 248.375, D=None)
    p3d = {int(i): (float(pri[i, 0]), float(pri[i, 1])) for i in np.nonzero(hp)[0]}
    ftrk = ov2slam_amd.FeatureTracker(gpu_ctx, 30, 0.01)
    g_sok, g_sr = stereo.stereo_matching_fused(ftrk, Gp, Gc, kps, kps, cal, rect=True, priors3d=p3d)
    m_sok, m_sr = stereo.stereo_matching(ftrk, Gp, Gc, kps, kps, cal, rect=True, priors3d=p3d)
    assert np.array_equal(c_sok.astype(bool), g_sok) and np.array_equal(_bits(c_sr), _bits(g_sr))
    assert np.array_equal(g_sok, m_sok) and np.array_equal(_bits(g_sr), _bits(m_sr))
    # ---- Optimizer::localBA: same protocol, same library ----
    g = ov2slam_amd.Optimizer(gpu_ctx).localBA(pb)
    assert c_flags[0] == 1 and bool(c_flags[1]) == bool(g["l2_done"])
    # the adapter's stop flag lives as long as the reference's (cleared where src/optimizer.cpp:896 clears its own, by the caller)
    assert g["l2_done"] and list(c_stop) == [1, 1, 1, 1], c_stop
    assert c_flags[2] == g["iterations"][0] and (not g["l2_done"] or c_flags[3] == g["iterations"][1])
    assert np.array_equal(c_badobs.astype(bool), np.asarray(g["bad_obs"]).astype(bool))
    assert np.allclose(c_poses.reshape(-1, 7), g["poses"], rtol=0, atol=1e-9) and np.allclose(c_lam, g["invdepth"], rtol=1e-9, atol=1e-12)


def test_verbatim_signatures_route(gpu_ctx, oracle, tmp_path):
    """ov2slam_amd/host/verbatim.hpp: the reference's EXACT signatures -- fbKltTracking(const std::vector<cv::Mat> &, ...),
    detectSingleScale / detectGridFAST(const cv::Mat &, ...) (include/feature_tracker.hpp:45, include/feature_extractor.hpp:40-46) -- on a
    thread-local context with the device pyramid looked up by the level-0 Mat (hash-validated: the reference re-uses its buffers).
    tests/cpp/verbatim_run.cpp (compiled with -DOV2_WITH_OPENCV against tests/fake_opencv: the image has no OpenCV) checks inside that
    this route returns the bits of the Context / Pyramid route and that the cache hits / refreshes as it must; here its outputs are
    compared with the ctypes mirror and the oracle."""
    exe = tmp_path / "verbatim_run"
    libdir = os.path.join(ROOT, "ov2slam_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-DOV2_WITH_OPENCV", "-I", os.path.join(ROOT, "tests", "fake_opencv"),
                           os.path.join(ROOT, "tests", "cpp", "verbatim_run.cpp"), "-o", str(exe), "-L", libdir, "-lov2slam_hip", "-Wl,-rpath," + libdir])
    w, h, cell = 752, 480, 35
    prev, cur, flow = synth.frame_pair(w, h, seed=12, shift=(2.6, -1.7), theta=0.003)
    rng = np.random.default_rng(9)
    kps = synth.grid_keypoints(w, h, cell, rng)
    pri = (flow(kps) + rng.normal(0, 1.0, kps.shape)).astype(np.float32)
    case, res = tmp_path / "case.bin", tmp_path / "res.bin"
    with open(case, "wb") as f:
        _wr(f, np.array([w, h, cell], np.int32)); _wr(f, prev); _wr(f, cur); _wr(f, kps.astype(np.float32)); _wr(f, pri)
    r = subprocess.run([str(exe), str(case), str(res)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "verbatim ok" in r.stdout, r.stdout + r.stderr
    with open(res, "rb") as f:
        c_fb, c_st, c_det = _rd(f, np.float32).reshape(-1, 2), _rd(f, np.uint8), _rd(f, np.float32).reshape(-1, 2)
    Gp = ov2slam_amd.Pyramid(gpu_ctx, w, h, 9, 3).build(prev); Gc = ov2slam_amd.Pyramid(gpu_ctx, w, h, 9, 3).build(cur)
    g_fb, g_st = ov2slam_amd.FeatureTracker(gpu_ctx, 30, 0.01).fbKltTracking(Gp, Gc, 9, 3, 30., 0.5, kps, pri)
    assert np.array_equal(_bits(c_fb), _bits(g_fb)) and np.array_equal(c_st.astype(bool), g_st.astype(bool)) and c_st.mean() > 0.8
    o_fb, o_st, _ = oracle.fb_klt(oracle.Pyramid(prev, 9, 3), oracle.Pyramid(cur, 9, 3), 9, 3, 30., 0.5, kps, pri)
    assert np.array_equal(_bits(c_fb), _bits(o_fb)) and np.array_equal(c_st.astype(bool), o_st)
    o_det, _ = oracle.detect_singlescale(cur, cell, kps[:len(kps) // 3], (5, 5, w - 10, h - 10), 0.001, True)
    assert np.array_equal(_bits(c_det), _bits(o_det))
