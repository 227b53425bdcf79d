"""shared generators for the loop-filter tests (deblock / CDEF / restoration)."""
import numpy as np

TXS = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (32, 64),
       (64, 32), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]


def test_image(rng, h, w, bd):
    """piecewise-smooth picture: smooth gradients (+-1 LSB texture, so the flat filters fire), blocky steps of every
    size, and a noisy band"""
    yy, xx = np.mgrid[0:h, 0:w]
    img = 60 + 40 * np.sin(xx / 37.0) + 30 * np.cos(yy / 23.0)
    steps = rng.integers(-6, 7, (h // 8 + 1, w // 8 + 1))
    img = img + np.kron(steps, np.ones((8, 8)))[:h, :w]
    big = rng.integers(-25, 26, (h // 32 + 1, w // 32 + 1))
    img = img + np.kron(big, np.ones((32, 32)))[:h, :w]
    img = img + rng.integers(-1, 2, (h, w))
    nb = min(24, h - h // 2)
    img[h // 2:h // 2 + nb] += rng.integers(-40, 41, (nb, w))
    img = np.clip(img, 0, 255)
    if bd == 10:
        img = img * 4 + rng.integers(0, 4, (h, w))
    return np.clip(np.rint(img), 0, (1 << bd) - 1).astype(np.uint8 if bd == 8 else np.uint16)


def random_mi(rng, O, h, w, is_chroma, zero_level_frac=0.15):
    """one random transform size, level pair and skip pattern per 64x64 (32x32 chroma) region"""
    mi = np.zeros((h // 4, w // 4), np.uint32)
    sb = 32 if is_chroma else 64
    for y0 in range(0, h, sb):
        for x0 in range(0, w, sb):
            while True:
                tw, th = TXS[int(rng.integers(0, 19))]
                if max(tw, th) <= sb:
                    break
            lv = int(rng.integers(1, 64)) if rng.random() > zero_level_frac else 0
            lh = int(rng.integers(1, 64)) if rng.random() > zero_level_frac else 0
            skip = int(rng.random() < 0.3)
            pb = int(rng.choice([8, 16, 32]))   # prediction-block size for the block-edge flags
            for uy in range(y0 // 4, min(h, y0 + sb) // 4):
                for ux in range(x0 // 4, min(w, x0 + sb) // 4):
                    bl = int((ux * 4) % pb == 0)
                    bt = int((uy * 4) % pb == 0)
                    mi[uy, ux] = O.lf_mi(int(np.log2(tw)), int(np.log2(th)), lv, lh, skip, bl, bt)
    return mi
