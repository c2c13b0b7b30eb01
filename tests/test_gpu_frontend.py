"""GPU parity tests (through the C ABI) of the HIP pyramid + LK path against the oracle.
Bar: bit-exact (integer pyramid; LK positions/status compared as raw float32 bits)."""
import numpy as np
import pytest

import ov2slam_amd
from ov2slam_amd import synth

pytestmark = pytest.mark.gpu


def _pyr_pair(ctx, oracle, img, win=9, lvl=3):
    h, w = img.shape
    G = ov2slam_amd.Pyramid(ctx, w, h, win, lvl).build(img)
    R = oracle.Pyramid(img, win, lvl)
    return G, R


@pytest.mark.parametrize("wh", [(752, 480), (1241, 376), (95, 61), (40, 40)])
def test_pyramid_bit_exact(gpu_ctx, oracle, wh):
    w, h = wh
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    G, R = _pyr_pair(gpu_ctx, oracle, img)
    assert G.levels == R.levels
    for l in range(R.levels):
        assert G.level_size(l) == R.level_size(l)
        gi, gd = G.download(l, padded=True)
        ri, rd = R.level(l, padded=True)
        assert np.array_equal(gi, ri), "image level %d" % l
        assert np.array_equal(gd, rd), "derivative level %d" % l


def test_pyramid_batch_matches_single(gpu_ctx, oracle):
    rng = np.random.default_rng(12)
    imgs = rng.integers(0, 256, (3, 120, 168), dtype=np.uint8)
    G = ov2slam_amd.Pyramid(gpu_ctx, 168, 120, 9, 3, batch=3).build(imgs)
    for b in range(3):
        R = oracle.Pyramid(imgs[b], 9, 3)
        for l in range(R.levels):
            gi, gd = G.download(l, b=b)
            ri, rd = R.level(l)
            assert np.array_equal(gi, ri) and np.array_equal(gd, rd)


def _assert_same_float_bits(a, b, what):
    a = np.ascontiguousarray(a, np.float32).view(np.uint32)
    b = np.ascontiguousarray(b, np.float32).view(np.uint32)
    bad = np.nonzero(a != b)[0]
    assert bad.size == 0, "%s: %d mismatching floats, first at %s" % (what, bad.size, bad[:5])


@pytest.mark.parametrize("nbpyrlvl", [3, 1, 0])
def test_fbklt_bit_exact_euroc(gpu_ctx, oracle, euroc_pair, nbpyrlvl):
    d = euroc_pair
    Gp, Rp = _pyr_pair(gpu_ctx, oracle, d["prev"])
    Gc, Rc = _pyr_pair(gpu_ctx, oracle, d["cur"])
    trk = ov2slam_amd.FeatureTracker(gpu_ctx, 30, 0.01)
    gout, gst, gstats = trk.fbKltTracking(Gp, Gc, 9, nbpyrlvl, 30., 0.5, d["kps"], d["pri"], return_stats=True)
    rout, rst, rstats = oracle.fb_klt(Rp, Rc, 9, nbpyrlvl, 30., 0.5, d["kps"], d["pri"])
    assert np.array_equal(gst, rst)
    _assert_same_float_bits(gout, rout, "tracked positions")
    assert gstats[0] == rstats[0]                       # identical GN iteration counts
    if nbpyrlvl == 3:
        assert gst.mean() > 0.9


def test_lk_single_call_bit_exact(gpu_ctx, oracle, euroc_pair):
    d = euroc_pair
    Gp, Rp = _pyr_pair(gpu_ctx, oracle, d["prev"])
    Gc, Rc = _pyr_pair(gpu_ctx, oracle, d["cur"])
    trk = ov2slam_amd.FeatureTracker(gpu_ctx, 30, 0.01)
    gp, gs, ge, gi = trk.calcOpticalFlowPyrLK(Gp, Gc, d["kps"], d["pri"], 9, 3)
    rp, rs, re_, ri = oracle.lk_track(Rp, Rc, d["kps"], d["pri"], 9, 3)
    assert np.array_equal(gs, rs) and np.array_equal(gi, ri)
    _assert_same_float_bits(gp, rp, "next points")
    _assert_same_float_bits(ge, re_, "min-eigenvalue err")


def test_fbklt_edge_cases(gpu_ctx, oracle):
    """Points near / outside the border, flat regions, large motion, KITTI-sized image."""
    rng = np.random.default_rng(21)
    prev, cur, flow = synth.frame_pair(1241, 376, seed=99, shift=(7.5, 3.25), theta=-0.006)
    prev[:60, :200] = 90; cur[:60, :200] = 90                      # textureless corner
    Gp, Rp = _pyr_pair(gpu_ctx, oracle, prev)
    Gc, Rc = _pyr_pair(gpu_ctx, oracle, cur)
    kps = np.concatenate([
        synth.grid_keypoints(1241, 376, 35, rng),
        np.array([[0.2, 0.3], [1240.6, 375.4], [-4.0, 100.0], [600.0, -3.5], [1249.5, 200.0], [3.0, 372.9],
                  [20.0, 20.0], [100.0, 30.0], [620.5, 188.5]], np.float32)])
    pri = (flow(kps) + rng.normal(0, 4.0, kps.shape)).astype(np.float32)
    pri[5] = [5000.0, -3000.0]                                     # absurd prior
    trk = ov2slam_amd.FeatureTracker(gpu_ctx, 30, 0.01)
    for lvl in (3, 1):
        gout, gst = trk.fbKltTracking(Gp, Gc, 9, lvl, 30., 0.5, kps, pri)
        rout, rst, _ = oracle.fb_klt(Rp, Rc, 9, lvl, 30., 0.5, kps, pri)
        assert np.array_equal(gst, rst)
        _assert_same_float_bits(gout, rout, "edge-case positions (lvl %d)" % lvl)


def test_fbklt_empty_is_noop(gpu_ctx, oracle):
    img = np.zeros((64, 64), np.uint8)
    G = ov2slam_amd.Pyramid(gpu_ctx, 64, 64, 9, 1).build(img)
    trk = ov2slam_amd.FeatureTracker(gpu_ctx)
    out, st = trk.fbKltTracking(G, G, 9, 1, 30., 0.5, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32))
    assert out.shape == (0, 2) and st.shape == (0,)


def test_unsupported_window_fails_loudly(gpu_ctx):
    img = np.zeros((64, 64), np.uint8)
    G = ov2slam_amd.Pyramid(gpu_ctx, 64, 64, 8, 1).build(img)
    trk = ov2slam_amd.FeatureTracker(gpu_ctx)
    pts = np.array([[30.0, 30.0]], np.float32)
    with pytest.raises(ov2slam_amd.Ov2Error):
        trk.fbKltTracking(G, G, 8, 1, 30., 0.5, pts, pts)


def test_fbklt_size_independent_properties_full_size(gpu_ctx):
    """At full EuRoC size with many points: identical images -> every textured point is tracked to
    itself (fb distance 0); swapping prev/cur on a pure translation negates the flow."""
    rng = np.random.default_rng(5)
    tex = synth.base_texture(1100, 77)
    a = synth.warp(tex, 752, 480, 120, 120)
    b = synth.warp(tex, 752, 480, 123, 118)          # integer shift (+3,-2): exact resampling
    Ga = ov2slam_amd.Pyramid(gpu_ctx, 752, 480, 9, 3).build(a)
    Gb = ov2slam_amd.Pyramid(gpu_ctx, 752, 480, 9, 3).build(b)
    trk = ov2slam_amd.FeatureTracker(gpu_ctx)
    kps = np.stack([rng.uniform(30, 720, 4000), rng.uniform(30, 450, 4000)], 1).astype(np.float32)
    out, st = trk.fbKltTracking(Ga, Ga, 9, 3, 30., 0.5, kps, kps)
    assert st.mean() > 0.95
    assert np.abs(out[st] - kps[st]).max() < 1e-3
    fwd, s1 = trk.fbKltTracking(Ga, Gb, 9, 3, 30., 0.5, kps, kps)
    good = s1
    assert good.mean() > 0.9
    assert np.abs((fwd[good] - kps[good]) - np.array([-3.0, 2.0])).max() < 0.05
    bwd, s2 = trk.fbKltTracking(Gb, Ga, 9, 3, 30., 0.5, fwd, fwd)
    both = s1 & s2
    assert np.abs(bwd[both] - kps[both]).max() < 0.1


@pytest.mark.parametrize("win", [5, 7, 11, 13])
def test_fbklt_other_window_sizes_bit_exact(gpu_ctx, oracle, euroc_pair, win):
    """Every kernel instance (row-byte / neighbourhood register layouts differ per WIN) against the oracle."""
    d = euroc_pair
    Gp, Rp = _pyr_pair(gpu_ctx, oracle, d["prev"], win, 3)
    Gc, Rc = _pyr_pair(gpu_ctx, oracle, d["cur"], win, 3)
    trk = ov2slam_amd.FeatureTracker(gpu_ctx, 30, 0.01)
    rng = np.random.default_rng(win)
    kps = np.concatenate([d["kps"], np.array([[0.4, 0.2], [751.5, 479.5], [-3.0, 100.0], [300.0, -2.5]], np.float32)])
    pri = np.concatenate([d["pri"], kps[-4:] + rng.normal(0, 2, (4, 2)).astype(np.float32)])
    for lvl in (3, 0):
        gout, gst, gstats = trk.fbKltTracking(Gp, Gc, win, lvl, 30., 0.5, kps, pri, return_stats=True)
        rout, rst, rstats = oracle.fb_klt(Rp, Rc, win, lvl, 30., 0.5, kps, pri)
        assert np.array_equal(gst, rst)
        _assert_same_float_bits(gout, rout, "win %d lvl %d" % (win, lvl))
        assert gstats[0] == rstats[0]
    # large drifts force neighbourhood re-fetches: priors 12 px off
    far = (d["gt"] + 12.0).astype(np.float32)
    gout, gst = trk.fbKltTracking(Gp, Gc, win, 3, 30., 0.5, d["kps"], far)
    rout, rst, _ = oracle.fb_klt(Rp, Rc, win, 3, 30., 0.5, d["kps"], far)
    assert np.array_equal(gst, rst)
    _assert_same_float_bits(gout, rout, "far priors win %d" % win)
