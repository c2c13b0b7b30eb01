"""GPU parity (through the C ABI) of the HIP CLAHE against the oracle: bit-exact u8 output."""
import ctypes as C
import numpy as np
import pytest

import ov2slam_amd
from ov2slam_amd import synth
from ov2slam_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wh,tiles,clip", [((752, 480), (15, 9), 3.0), ((1241, 376), (24, 7), 2.0), ((640, 480), (12, 9), 3.0),
                                            ((96, 64), (4, 4), 40.0), ((100, 50), (2, 1), 0.0), ((103, 57), (3, 2), 1.0),
                                            # 41 padded columns over 3-pixel tiles: tiles whose mirror sources lie in OTHER tiles (byte-wise path)
                                            ((100, 50), (47, 3), 2.0), ((333, 90), (10, 4), 4.0)])
def test_clahe_bit_exact(gpu_ctx, oracle, wh, tiles, clip):
    w, h = wh
    img, _, _ = synth.frame_pair(w, h, seed=w * 3 + h)
    img = (img.astype(np.float32) * 0.5 + 40).astype(np.uint8)
    g = ov2slam_amd.CLAHE(gpu_ctx, clip, tiles).apply(img)
    r = oracle.clahe(img, clip, tiles[0], tiles[1])
    assert np.array_equal(g, r)
    rng = np.random.default_rng(w)
    noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
    assert np.array_equal(ov2slam_amd.CLAHE(gpu_ctx, clip, tiles).apply(noise), oracle.clahe(noise, clip, tiles[0], tiles[1]))


_DEVICE_PATH_SCRIPT = r"""
import ctypes as C, sys, numpy as np
import torch                      # torch's HIP runtime must be initialised before libov2slam_hip.so in one process
torch.cuda.init()
sys.path.insert(0, sys.argv[1])
import ov2slam_amd
from ov2slam_amd import synth, _lib as L
from oracle import oracle as O
ctx = ov2slam_amd.Context(0)
rng = np.random.default_rng(4)
imgs = np.stack([synth.frame_pair(752, 480, seed=s)[0] for s in (1, 2)] + [rng.integers(0, 256, (480, 752), dtype=np.uint8)])
src = torch.from_numpy(imgs).cuda(); dst = torch.empty_like(src); torch.cuda.synchronize()
L.check(ctx.lib.ov2_clahe_d(ctx.h, C.c_void_p(src.data_ptr()), 752, 480, 752, 752 * 480, 3, 3.0, 15, 9,
                            C.c_void_p(dst.data_ptr()), 752, 752 * 480))
ctx.sync()
out = dst.cpu().numpy()
for b in range(3):
    assert np.array_equal(out[b], O.clahe(imgs[b], 3.0, 15, 9)), b
# fused preprocessImage: CLAHE straight into the pyramid == ov2_clahe_d + ov2_pyr_build_d == oracle
pf = ov2slam_amd.Pyramid(ctx, 752, 480, 9, 3, batch=3).build_clahe_from_device(src.data_ptr(), 3.0, 15, 9)
ps = ov2slam_amd.Pyramid(ctx, 752, 480, 9, 3, batch=3).build_from_device(dst.data_ptr())
ctx.sync()
for b in range(3):
    ref = O.Pyramid(O.clahe(imgs[b], 3.0, 15, 9), 9, 3)
    for lvl in range(pf.levels):
        a = pf.download(lvl, b, padded=True)[0]; c = ps.download(lvl, b, padded=True)[0]
        assert np.array_equal(a, c), (b, lvl)
        assert np.array_equal(pf.download(lvl, b, padded=True)[0], ref.level(lvl, padded=True)[0]), (b, lvl)
# the same through the strip kernel (what a large batch selects by itself: apply + level 1 + borders in one walk)
ctx.set_option(L.OV2_OPT_CLAHE_STRIPS, 1)
pq = ov2slam_amd.Pyramid(ctx, 752, 480, 9, 3, batch=3).build_clahe_from_device(src.data_ptr(), 3.0, 15, 9)
ctx.sync()
for b in range(3):
    for lvl in range(pq.levels):
        assert np.array_equal(pq.download(lvl, b, padded=True)[0], pf.download(lvl, b, padded=True)[0]), ("strips", b, lvl)
ctx.set_option(L.OV2_OPT_CLAHE_STRIPS, -1)
# images whose rows are NOT 4-byte aligned (KITTI: 1241-byte rows; 103x57) through the device entry points: this is the
# only way to reach the unaligned-source kernel instances (the host entry points stage with an aligned pitch)
for (w, h), tiles in (((1241, 376), (24, 7)), ((103, 57), (3, 2)), ((751, 97), (15, 2))):
    im = np.stack([synth.frame_pair(w, h, seed=s)[0] for s in (5, 6)] + [rng.integers(0, 256, (h, w), dtype=np.uint8)])
    for off in (0, 1, 3):                                       # also a misaligned base address
        buf = torch.zeros(im.size + 8, dtype=torch.uint8, device="cuda")
        s_ = buf[off:off + im.size].view(3, h, w); s_.copy_(torch.from_numpy(im))
        d_ = torch.empty((3, h, w), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
        L.check(ctx.lib.ov2_clahe_d(ctx.h, C.c_void_p(s_.data_ptr()), w, h, w, w * h, 3, 3.0, tiles[0], tiles[1], C.c_void_p(d_.data_ptr()), w, w * h))
        pf = ov2slam_amd.Pyramid(ctx, w, h, 9, 3, batch=3).build_clahe_from_device(s_.data_ptr(), 3.0, tiles[0], tiles[1])
        ctx.sync()
        o = d_.cpu().numpy()
        for b in range(3):
            ref = O.clahe(im[b], 3.0, tiles[0], tiles[1])
            assert np.array_equal(o[b], ref), (w, h, off, b)
            rp = O.Pyramid(ref, 9, 3)
            for lvl in range(pf.levels):
                assert np.array_equal(pf.download(lvl, b, padded=True)[0], rp.level(lvl, padded=True)[0]), (w, h, off, b, lvl)
print("DEVICE_PATH_OK")
"""


def test_clahe_batched_device_path():
    """ov2_clahe_d on torch-owned HBM buffers (the bench path); runs in its own process because torch has to
    initialise its HIP runtime before the library does."""
    import os, subprocess, sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _DEVICE_PATH_SCRIPT, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DEVICE_PATH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_fused_preprocess_from_host_image(gpu_ctx, oracle):
    """ov2_pyr_build_clahe_h == CLAHE (oracle) followed by the pyramid (oracle), borders included."""
    for (w, h), tiles in (((752, 480), (15, 9)), ((1241, 376), (24, 7)), ((103, 57), (3, 2))):
        img, _, _ = synth.frame_pair(w, h, seed=w + h)
        P = ov2slam_amd.Pyramid(gpu_ctx, w, h, 9, 3).build_clahe(img, 3.0, tiles[0], tiles[1])
        ref = oracle.Pyramid(oracle.clahe(img, 3.0, tiles[0], tiles[1]), 9, 3)
        assert P.levels == ref.levels
        for lvl in range(P.levels):
            assert np.array_equal(P.download(lvl, padded=True)[0], ref.level(lvl, padded=True)[0]), (w, h, lvl)


@pytest.mark.parametrize("wh,tiles", [((752, 480), (15, 9)), ((640, 480), (12, 9)), ((376, 240), (7, 4)), ((264, 100), (5, 2)),
                                       ((512, 97), (10, 1)), ((1000, 64), (20, 1)), ((64, 48), (1, 1)), ((260, 50), (3, 1)), ((508, 33), (47, 2)),
                                       # round 4: widths that are no multiple of 4 (KITTI 00-02: 1241 x 376, 03: 1242 x 375, 04-: 1226 x 370;
                                       # w % 4 = 1, 2, 3; odd and even level-1 widths; one strip and several)
                                       ((1241, 376), (24, 7)), ((1242, 375), (24, 7)), ((1226, 370), (24, 7)), ((103, 57), (3, 2)),
                                       ((751, 97), (15, 2)), ((253, 64), (5, 1)), ((66, 40), (2, 1)), ((67, 41), (1, 1))])
def test_strip_kernel_equals_oracle_pyramid(gpu_ctx, oracle, wh, tiles):
    """k_clahe_apply_pyr (batch mode's CLAHE apply + level 1 + both borders in one walk, forced here with OV2_OPT_CLAHE_STRIPS = 1, and = 2: fused with the LUT computation):
    every level, borders included, equals CLAHE (oracle) followed by the pyramid (oracle) -- one to four column strips,
    odd heights, cell columns narrower than a strip."""
    w, h = wh
    rng = np.random.default_rng(w + h)
    for img in (synth.frame_pair(w, h, seed=w + h)[0], rng.integers(0, 256, (h, w), dtype=np.uint8)):
        ref = oracle.Pyramid(oracle.clahe(img, 3.0, tiles[0], tiles[1]), 9, 3)
        for force in (2, 1, 0):
            with gpu_ctx.options(clahe_strips=force):
                P = ov2slam_amd.Pyramid(gpu_ctx, w, h, 9, 3).build_clahe(img, 3.0, tiles[0], tiles[1])
            assert P.levels == ref.levels
            for lvl in range(P.levels):
                assert np.array_equal(P.download(lvl, padded=True)[0], ref.level(lvl, padded=True)[0]), (w, h, force, lvl)
            P.close()


def test_strip_kernel_equals_separate_kernels_random_geometry(gpu_ctx):
    """Randomised geometry sweep: the strip kernel and the separate kernels (both bit-exact against the oracle above on the
    named sizes) must produce identical padded pyramids for any width (round 4: also width % 4 != 0), any height, any tile grid."""
    rng = np.random.default_rng(77)
    for case in range(60):
        win = 9 if case < 30 else int(rng.choice([5, 7, 13, 21]))         # LK window = border width of every level
        w = 4 * int(rng.integers(16, 280)) + (int(rng.integers(0, 4)) if case % 3 else 0); h = int(rng.integers(2 * win + 6, 260))
        tx = int(rng.integers(1, max(2, min(30, w // 8)))); ty = int(rng.integers(1, max(2, min(12, h // 8))))
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        out = {}
        for force in (2, 1, 0):
            with gpu_ctx.options(clahe_strips=force):
                P = ov2slam_amd.Pyramid(gpu_ctx, w, h, win, 3).build_clahe(img, 2.5, tx, ty)
            out[force] = [P.download(l, padded=True)[0] for l in range(P.levels)]
            P.close()
        for force in (1, 2):
            for l, (a, b) in enumerate(zip(out[force], out[0])):
                assert np.array_equal(a, b), (case, force, win, w, h, tx, ty, l, np.argwhere(a != b)[:4].tolist())


@pytest.mark.parametrize("wh,tiles", [((752, 480), (15, 9)), ((1241, 376), (24, 7)), ((376, 240), (7, 4)), ((103, 57), (3, 2))])
def test_low_entropy_frames_bit_exact(gpu_ctx, oracle, wh, tiles):
    """Frames whose histograms have a handful of bins (constant, over- / under-exposed, posterised, flat blocks with a texture island):
    the inputs on which a wavefront's ds_adds collide.  Round 5 gave the histogram wavefronts four staggered copies and adds a dword that
    many lanes share once per byte with the lane count as weight (clahe_lut_tiles) -- every CLAHE path must still equal the oracle."""
    w, h = wh
    rng = np.random.default_rng(w * 7 + h)
    base, _, _ = synth.frame_pair(w, h, seed=w + 11 * h)
    base = base.astype(np.int32)
    sel = rng.uniform(size=base.shape) < 0.8
    blocks = np.repeat(np.repeat(rng.integers(0, 256, ((h + 31) // 32, (w + 47) // 48)), 32, 0), 48, 1)[:h, :w]       # flat 48 x 32 patches
    island = blocks.copy(); island[h // 3:h // 3 + 40, w // 4:w // 4 + 90] = base[h // 3:h // 3 + 40, w // 4:w // 4 + 90]
    frames = {"constant": np.full_like(base, 128), "black": np.zeros_like(base), "white": np.full_like(base, 255),
              "saturated": np.where(sel, 255, 255 - base // 8), "dark": np.where(sel, base // 32, base // 4),
              "levels16": np.where(sel, (base // 16) * 16 + 8, base), "two_levels": np.where(base > 128, 200, 17),
              "flat_blocks": blocks, "flat_blocks_with_island": island}
    for name, f in frames.items():
        img = np.clip(f, 0, 255).astype(np.uint8)
        ref = oracle.Pyramid(oracle.clahe(img, 3.0, tiles[0], tiles[1]), 9, 3)
        for force in (2, 1, 0):
            with gpu_ctx.options(clahe_strips=force):
                P = ov2slam_amd.Pyramid(gpu_ctx, w, h, 9, 3).build_clahe(img, 3.0, tiles[0], tiles[1])
            for lvl in range(P.levels):
                assert np.array_equal(P.download(lvl, padded=True)[0], ref.level(lvl, padded=True)[0]), (name, w, h, force, lvl)
            P.close()
        assert np.array_equal(ov2slam_amd.CLAHE(gpu_ctx, 3.0, tiles).apply(img), oracle.clahe(img, 3.0, tiles[0], tiles[1])), name
