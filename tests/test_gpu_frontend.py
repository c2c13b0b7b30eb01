"""GPU parity tests (through the C ABI) of the HIP pyramid + LK path against the oracle.
Bar: bit-exact (integer pyramid; LK positions/status compared as raw float32 bits)."""
import numpy as np
import pytest

import ov2slam_amd
from ov2slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["row", "lane3"], autouse=True)
def lk_impl(request, gpu_ctx):
    """Every test of this file runs with both LK kernels (lk.hip: row per lane, lk3.hip: 3 lanes per keypoint), pinned through
    ov2_ctx_set_option(OV2_OPT_LK_IMPL); without it the library picks by launch size and these small cases would only see the
    first one."""
    from ov2slam_amd import _lib as L
    with gpu_ctx.options(lk_impl=L.OV2_LK_IMPL_ROW if request.param == "row" else L.OV2_LK_IMPL_LANE3):
        yield request.param


def _pyr_pair(ctx, oracle, img, win=9, lvl=3):
    h, w = img.shape
    G = ov2slam_amd.Pyramid(ctx, w, h, win, lvl).build(img)
    R = oracle.Pyramid(img, win, lvl)
    return G, R


# (264, 100) and (530, 70): the last work-group column of level 1 owns fewer than win + 1 columns, so the level kernel
# cannot mirror the right border itself and the stand-alone border kernel runs (pyramid.hip: fuse condition)
@pytest.mark.parametrize("wh", [(752, 480), (1241, 376), (95, 61), (40, 40), (264, 100), (530, 70)])
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


@pytest.mark.parametrize("nbpyrlvl", [0, 1])
def test_fbklt_backward_reuse_and_refetch_paths(gpu_ctx, oracle, nbpyrlvl):
    """lk3.hip runs the backward level 0 from the neighbourhoods the forward level left in LDS.  Cover the common
    path (small forward motion) together with the ones that leave it: forward Gauss-Newton walks of several pixels
    at level 0 (search block re-centred, template neighbourhood no longer inside it -> refetch), and a current
    image whose content near some keypoints is shifted back so that the backward track drifts off the reused block."""
    rng = np.random.default_rng(5)
    prev, cur, flow = synth.frame_pair(752, 480, seed=31, shift=(1.3, -0.8), theta=0.002)
    cur = cur.copy()
    cur[200:260, 300:420] = np.roll(cur[200:260, 300:420], 3, axis=1)      # locally inconsistent motion
    Gp, Rp = _pyr_pair(gpu_ctx, oracle, prev)
    Gc, Rc = _pyr_pair(gpu_ctx, oracle, cur)
    kps = synth.grid_keypoints(752, 480, 25, rng)
    trk = ov2slam_amd.FeatureTracker(gpu_ctx, 30, 0.01)
    travelled = 0
    for sigma in (0.2, 2.5, 6.0):
        pri = (flow(kps) + rng.normal(0, sigma, kps.shape)).astype(np.float32)
        gout, gst, gstats = trk.fbKltTracking(Gp, Gc, 9, nbpyrlvl, 30., 0.5, kps, pri, return_stats=True)
        rout, rst, rstats = oracle.fb_klt(Rp, Rc, 9, nbpyrlvl, 30., 0.5, kps, pri)
        assert np.array_equal(gst, rst)
        _assert_same_float_bits(gout, rout, "positions (prior sigma %.1f)" % sigma)
        assert gstats[0] == rstats[0] and gstats[1] == rstats[1]
        travelled += int((np.abs(gout - pri).max(axis=1)[gst > 0] > 4.0).sum())
    assert travelled > 20            # some tracked points did walk more than the 3-pixel margin of the search block


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


_BATCH_SCRIPT = r"""
import ctypes as C, os, sys, numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, sys.argv[1])
import ov2slam_amd
from ov2slam_amd import synth, _lib as L
from oracle import oracle as O
ctx = ov2slam_amd.Context(0)
B, NMAX, W, H = 11, 45, 376, 240            # 11 = 8 + 3: exercises the XCD-aware block map and its remainder path
rng = np.random.default_rng(3)
prevs, curs, kps, pri = [], [], np.zeros((B, NMAX, 2), np.float32), np.zeros((B, NMAX, 2), np.float32)
n_item = np.array([45, 0, 1, 19, 20, 21, 40, 45, 7, 33, 45], np.int32)
for b in range(B):
    p, c, flow = synth.frame_pair(W, H, seed=50 + b, shift=(1.5 + b * 0.3, -1.0), theta=0.002 * b)
    prevs.append(p); curs.append(c)
    k = synth.grid_keypoints(W, H, 35, rng)[:NMAX]
    kps[b, :len(k)] = k; pri[b, :len(k)] = (flow(k) + rng.normal(0, 1.0, k.shape)).astype(np.float32)
Pp = ov2slam_amd.Pyramid(ctx, W, H, 9, 3, batch=B).build(np.stack(prevs))
Pc = ov2slam_amd.Pyramid(ctx, W, H, 9, 3, batch=B).build(np.stack(curs))
ctx.sync()
vp = lambda t: C.c_void_p(t.data_ptr())
for impl in ("row", "lane3"):
    ctx.set_option(L.OV2_OPT_LK_IMPL, L.OV2_LK_IMPL_ROW if impl == "row" else L.OV2_LK_IMPL_LANE3)
    for lvl in (3, 1):
        dk = torch.from_numpy(kps).cuda(); dp = torch.from_numpy(pri).cuda(); dn = torch.from_numpy(n_item).cuda()
        st = torch.full((B, NMAX), 7, dtype=torch.uint8, device="cuda"); stats = torch.zeros(2, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        L.check(ctx.lib.ov2_fb_klt_d(ctx.h, Pp.h_pyr, Pc.h_pyr, 9, lvl, 30, 0.01, 30.0, 0.5, vp(dk), vp(dp), NMAX, vp(dn), vp(st), vp(stats)))
        ctx.sync()
        gp, gs = dp.cpu().numpy(), st.cpu().numpy()
        tot = 0
        for b in range(B):
            n = int(n_item[b])
            rp, rs, rstats = O.fb_klt(O.Pyramid(prevs[b], 9, 3), O.Pyramid(curs[b], 9, 3), 9, lvl, 30.0, 0.5, kps[b, :n], pri[b, :n])
            assert np.array_equal(gs[b, :n].astype(bool), rs), (impl, lvl, b)
            assert np.array_equal(gp[b, :n].view(np.uint32), rp.view(np.uint32)), (impl, lvl, b)
            assert np.all(gs[b, n:] == 7) and np.array_equal(gp[b, n:], pri[b, n:]), "slots beyond n_per_item must stay untouched"
            tot += rstats[0]
        assert int(stats[0].item()) == tot, (impl, lvl)
print("BATCH_LK_OK")
"""


def test_fbklt_batched_device_path_both_kernels():
    """ov2_fb_klt_d on a batch of 11 image pairs with ragged per-item keypoint counts, both kernels, in its own
    process (torch owns the device buffers and has to initialise HIP first)."""
    import os, subprocess, sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _BATCH_SCRIPT, root], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BATCH_LK_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_two_contexts_concurrently(oracle, euroc_pair):
    """The library keeps no global mutable state: the SLAM thread and the mapper thread call fbKltTracking concurrently
    (src/visual_front_end.cpp:196 / src/map_manager.cpp:510), each on its own context / HIP stream."""
    import threading
    d = euroc_pair
    Rp, Rc = oracle.Pyramid(d["prev"], 9, 3), oracle.Pyramid(d["cur"], 9, 3)
    ref = {lvl: oracle.fb_klt(Rp, Rc, 9, lvl, 30., 0.5, d["kps"], d["pri"])[:2] for lvl in (1, 3)}
    errors = []

    def worker(lvl):
        try:
            ctx = ov2slam_amd.Context(0)
            trk = ov2slam_amd.FeatureTracker(ctx, 30, 0.01)
            for _ in range(15):
                Gp = ov2slam_amd.Pyramid(ctx, 752, 480, 9, 3).build(d["prev"])
                Gc = ov2slam_amd.Pyramid(ctx, 752, 480, 9, 3).build(d["cur"])
                out, st = trk.fbKltTracking(Gp, Gc, 9, lvl, 30., 0.5, d["kps"], d["pri"])
                if not (np.array_equal(st, ref[lvl][1]) and np.array_equal(out.view(np.uint32), ref[lvl][0].view(np.uint32))):
                    errors.append("mismatch in thread lvl=%d" % lvl)
            ctx.close()
        except Exception as e:                       # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(lvl,)) for lvl in (1, 3, 1, 3)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errors, errors[:3]


@pytest.mark.parametrize("win", [9, 5, 7, 11, 13])
def test_fbklt_float_accumulators_as_executed_on_x86(gpu_ctx, oracle, euroc_pair, win):
    """OV2_OPT_LK_ACC = OV2_LK_ACC_FLOAT_UI4: the sums of calcOpticalFlowPyrLK in FLOAT accumulators, in the order of an x86 OpenCV 4.x
    build (acctype float, 128-bit universal intrinsics: 4 lane accumulators + a scalar for the normal matrix, v_dotprod pairs in 8 lanes
    + a scalar for the mismatch vector) -- what src/feature_tracker.cpp:66-69 executes on a desktop.  Bit for bit (positions, status,
    Gauss-Newton trips, min-eigenvalue err) against the oracle in ORC_LK_ACC_FLOAT_UI4, every window instance of the row kernel
    (SIMD widths 0 / 4 / 8 / 12 columns), forward-backward and single calls, edge points, drifting tracks; and it is a different
    computation from the default: some positions must differ from the INT64 mode."""
    from ov2slam_amd import _lib as L
    d = euroc_pair
    # contrast x 5: with the soft texture of the synthetic pair every sum stays below 2^24 and float accumulation is exact -- the two
    # modes would return the same bits and the test would prove nothing
    gain = lambda im: np.clip((im.astype(np.float32) - 128.0) * 5.0 + 128.0, 0, 255).astype(np.uint8)
    Gp, Rp = _pyr_pair(gpu_ctx, oracle, gain(d["prev"]), win, 3)
    Gc, Rc = _pyr_pair(gpu_ctx, oracle, gain(d["cur"]), win, 3)
    trk = ov2slam_amd.FeatureTracker(gpu_ctx, 30, 0.01)
    rng = np.random.default_rng(100 + win)
    kps = np.concatenate([d["kps"], np.array([[0.4, 0.2], [751.5, 479.5], [-3.0, 100.0], [300.0, -2.5]], np.float32)])
    pri = np.concatenate([d["pri"], kps[-4:] + rng.normal(0, 2, (4, 2)).astype(np.float32)])
    int_out, int_st = trk.fbKltTracking(Gp, Gc, win, 3, 30., 0.5, kps, pri)
    n_diff = 0
    with gpu_ctx.options(lk_acc=L.OV2_LK_ACC_FLOAT_UI4), oracle.lk_acc_mode(oracle.LK_ACC_FLOAT_UI4):
        for lvl in (3, 0):
            gout, gst, gstats = trk.fbKltTracking(Gp, Gc, win, lvl, 30., 0.5, kps, pri, return_stats=True)
            rout, rst, rstats = oracle.fb_klt(Rp, Rc, win, lvl, 30., 0.5, kps, pri)
            assert np.array_equal(gst, rst)
            _assert_same_float_bits(gout, rout, "float accumulators, win %d lvl %d" % (win, lvl))
            assert gstats[0] == rstats[0]
            if lvl == 3:
                both = gst.astype(bool) & int_st.astype(bool)
                n_diff = int((np.ascontiguousarray(gout[both]).view(np.uint32) != np.ascontiguousarray(int_out[both]).view(np.uint32)).any(axis=1).sum())
                assert np.abs(gout[both] - int_out[both]).max() < 5e-3          # the two modes stay within a few 1e-4 px of each other
        gp, gs, ge, gi = trk.calcOpticalFlowPyrLK(Gp, Gc, d["kps"], d["pri"], win, 3)
        rp, rs, re_, ri = oracle.lk_track(Rp, Rc, d["kps"], d["pri"], win, 3)
        assert np.array_equal(gs, rs) and np.array_equal(gi, ri)
        _assert_same_float_bits(gp, rp, "next points")
        _assert_same_float_bits(ge, re_, "min-eigenvalue err")
        far = (d["gt"] + 12.0).astype(np.float32)                              # re-fetched search blocks
        gout, gst = trk.fbKltTracking(Gp, Gc, win, 3, 30., 0.5, d["kps"], far)
        rout, rst, _ = oracle.fb_klt(Rp, Rc, win, 3, 30., 0.5, d["kps"], far)
        assert np.array_equal(gst, rst)
        _assert_same_float_bits(gout, rout, "far priors win %d" % win)
    if win >= 9:       # (small windows: fewer terms, the sums stay exact more often)
        assert n_diff > 0, "the float mode returned the INT64 mode's bits everywhere: it did not run"
    assert gpu_ctx.get_option(L.OV2_OPT_LK_ACC) == L.OV2_LK_ACC_INT64
