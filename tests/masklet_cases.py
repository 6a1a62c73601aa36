"""Seeded mask inputs shared by tests/golden/gen_masklet_golden.py (which feeds them to the imported reference) and the
CPU / GPU parity tests (which regenerate them and compare against the committed outputs)."""
import hashlib

import numpy as np

# (source h, w) -> target of seg_utils.reshape_masklet's default rule, production sizes of MeViS / Ref-YTVOS frames
PRODUCTION_SHAPES = [(720, 1280), (480, 854), (1080, 1920), (360, 640), (1280, 720), (540, 960)]
# (n, h, w, H, W) stored in full in the golden file
SMALL_SHAPES = [(3, 24, 40, 27, 48), (2, 48, 27, 33, 19), (1, 17, 9, 40, 33), (4, 36, 64, 54, 96), (2, 20, 30, 20, 30),
                (2, 40, 60, 20, 30), (2, 1, 1, 5, 7), (3, 7, 5, 1, 1), (2, 30, 1, 3, 50)]


def parity_images(h, w):
    """16 images v[y,x] = a[y%2][x%2] over all 16 binary 2x2 tables a: every output pixel of a bilinear resample sees
    all 16 combinations of its four taps (the taps are at (y0, y0+1) x (x0, x0+1) except at clamped borders)."""
    yy, xx = np.arange(h)[:, None] % 2, np.arange(w)[None, :] % 2
    out = np.zeros((16, h, w), np.uint8)
    for k in range(16):
        a = [[(k >> 0) & 1, (k >> 1) & 1], [(k >> 2) & 1, (k >> 3) & 1]]
        out[k] = np.where(yy == 0, np.where(xx == 0, a[0][0], a[0][1]), np.where(xx == 0, a[1][0], a[1][1]))
    return out


def blob_masklet(T, h, w, seed):
    """T frames of drifting rectangles + discs, one empty frame, one full frame, one white-noise frame."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.zeros((T, h, w), np.uint8)
    for t in range(T):
        m = np.zeros((h, w), bool)
        for _ in range(3):
            y0, x0 = rng.integers(0, h), rng.integers(0, w)
            hh, ww = rng.integers(1, max(2, h // 2)), rng.integers(1, max(2, w // 2))
            m[y0:y0 + hh, x0:x0 + ww] = True
        cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(1, max(2, min(h, w) // 3))
        m |= (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
        out[t] = m
    if T >= 3:
        out[T - 3] = 0
        out[T - 2] = 1
        out[T - 1] = rng.random((h, w)) < 0.5
    return out


def production_masklet(h, w, seed=0):
    return np.concatenate([parity_images(h, w), blob_masklet(6, h, w, seed)], 0)


def digest(bits_u8):
    """sha256 of a {0,1} uint8 array packed to bits (row-major, numpy packbits big-endian)."""
    return hashlib.sha256(np.packbits(np.asarray(bits_u8, np.uint8).reshape(-1)).tobytes()).hexdigest()
