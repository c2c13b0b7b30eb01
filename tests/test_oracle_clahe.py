"""CPU tests of the oracle's CLAHE restatement against an independent numpy implementation."""
import numpy as np
import pytest

from ov2slam_amd import synth


def np_clahe(img, clip_limit, tx, ty):
    """Independent CLAHE (vectorised numpy, float32 arithmetic in OpenCV's order)."""
    h, w = img.shape
    if w % tx == 0 and h % ty == 0:
        ext = img
    else:
        ext = np.pad(img, ((0, ty - h % ty), (0, tx - w % tx)), mode="reflect")
    eh, ew = ext.shape
    tw, th = ew // tx, eh // ty
    total = tw * th
    clip = max(int(clip_limit * total / 256), 1) if clip_limit > 0 else 0
    scale = np.float32(255) / np.float32(total)
    luts = np.zeros((ty, tx, 256), np.uint8)
    for j in range(ty):
        for i in range(tx):
            hist = np.bincount(ext[j * th:(j + 1) * th, i * tw:(i + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if clip > 0:
                clipped = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip)
                batch, res = divmod(clipped, 256)
                hist += batch
                if res:
                    step = max(256 // res, 1)
                    idx = np.arange(0, 256, step)[:res]
                    hist[idx] += 1
            cs = np.cumsum(hist).astype(np.float32) * scale
            luts[j, i] = np.clip(np.rint(cs), 0, 255).astype(np.uint8)
    ys, xs = np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32)
    tyf = ys * (np.float32(1) / np.float32(th)) - np.float32(0.5)
    txf = xs * (np.float32(1) / np.float32(tw)) - np.float32(0.5)
    ty1, tx1 = np.floor(tyf).astype(int), np.floor(txf).astype(int)
    ya, xa = (tyf - ty1).astype(np.float32), (txf - tx1).astype(np.float32)
    ty2, tx2 = np.minimum(ty1 + 1, ty - 1), np.minimum(tx1 + 1, tx - 1)
    ty1, tx1 = np.maximum(ty1, 0), np.maximum(tx1, 0)
    v = img.astype(int)
    L = lambda a, b: luts[a[:, None], b[None, :], v].astype(np.float32)
    xa_, xa1 = xa[None, :], (np.float32(1) - xa)[None, :]
    ya_, ya1 = ya[:, None], (np.float32(1) - ya)[:, None]
    res = (L(ty1, tx1) * xa1 + L(ty1, tx2) * xa_) * ya1 + (L(ty2, tx1) * xa1 + L(ty2, tx2) * xa_) * ya_
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("wh,tiles,clip", [((752, 480), (15, 9), 3.0), ((1241, 376), (24, 7), 2.0), ((640, 480), (12, 9), 3.0),
                                            ((96, 64), (4, 4), 40.0), ((100, 50), (2, 1), 0.0)])
def test_clahe_matches_numpy(oracle, wh, tiles, clip):
    w, h = wh
    img, _, _ = synth.frame_pair(w, h, seed=w + h)
    img = (img.astype(np.float32) * 0.5 + 40).astype(np.uint8)          # low-contrast input
    out = oracle.clahe(img, clip, tiles[0], tiles[1])
    ref = np_clahe(img, clip, tiles[0], tiles[1])
    assert np.array_equal(out, ref)
    if clip > 0:
        assert out.std() > img.std()                                        # contrast is enhanced


def test_clahe_flat_and_extremes(oracle):
    flat = np.full((480, 752), 100, np.uint8)
    out = oracle.clahe(flat, 3.0, 15, 9)
    assert out.min() == out.max()                                            # a flat image stays flat
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 256, (480, 752), dtype=np.uint8)
    out = oracle.clahe(noise, 3.0, 15, 9)
    assert np.array_equal(out, np_clahe(noise, 3.0, 15, 9))
