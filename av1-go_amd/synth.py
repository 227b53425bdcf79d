"""Deterministic synthetic YUV 4:2:0 frames (BASELINE.md §2 / SURVEY.md §8d recipe).

Texture: xorshift32-hashed noise (seed 0xA51C0DE), 3x3 box blur twice, plus a diagonal ramp, with a
64 px border.  Frame t samples the texture bilinearly at offset (1.25 t, 0.75 t) px and adds +-2 LSB
(8-bit scale) noise from seed 0xA51C0DE + t.  Chroma: same at half resolution, seeds +1, +2.
Ranges: 8-bit [16,235]; 10-bit [64,940] in uint16.  Nothing is read from disk.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

SEED = 0xA51C0DE
BORDER = 64


def _xorshift32(x):
    x = x.astype(np.uint32)
    for _ in range(2):
        x ^= (x << np.uint32(13))
        x ^= (x >> np.uint32(17))
        x ^= (x << np.uint32(5))
    return x


def _noise(shape, seed):
    idx = np.arange(shape[0] * shape[1], dtype=np.uint32).reshape(shape)
    return _xorshift32(idx * np.uint32(2654435761) + np.uint32(seed & 0xFFFFFFFF))


def _blur3(a):
    p = np.pad(a, 1, mode="edge")
    return sum(p[dy:dy + a.shape[0], dx:dx + a.shape[1]] for dy in range(3) for dx in range(3)) / 9.0


def texture(w, h, seed, max_t):
    tw, th = w + 2 * BORDER + int(1.25 * max_t) + 2, h + 2 * BORDER + int(0.75 * max_t) + 2
    n = (_noise((th, tw), seed) >> np.uint32(24)).astype(np.float64)  # 0..255
    n = _blur3(_blur3(n))
    yy, xx = np.mgrid[0:th, 0:tw]
    ramp = ((xx + yy) % 512) / 511.0 * 96.0 - 48.0
    t = (n - n.mean()) * 3.0 + 128.0 + ramp
    return np.clip(t, 0, 255)


def plane(tex, w, h, t, seed, bd, scale=1.0):
    ox, oy = BORDER + 1.25 * t * scale, BORDER + 0.75 * t * scale
    x0, y0 = int(np.floor(ox)), int(np.floor(oy))
    fx, fy = ox - x0, oy - y0
    a = tex[y0:y0 + h, x0:x0 + w]
    b = tex[y0:y0 + h, x0 + 1:x0 + w + 1]
    c = tex[y0 + 1:y0 + h + 1, x0:x0 + w]
    d = tex[y0 + 1:y0 + h + 1, x0 + 1:x0 + w + 1]
    v = (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy
    nz = (_noise((h, w), seed + t) % np.uint32(5)).astype(np.float64) - 2.0
    v = v + nz
    lo, hi = (16, 235) if bd == 8 else (64, 940)
    v = lo + (v / 255.0) * (hi - lo)
    return np.clip(np.rint(v), lo, hi).astype(np.uint8 if bd == 8 else np.uint16)


def frames(w, h, n, bd=8, first=0):
    """returns (Y [n,h,w], U [n,h/2,w/2], V [n,h/2,w/2])"""
    ty = texture(w, h, SEED, first + n)
    tu = texture(w // 2, h // 2, SEED + 1, first + n)
    tv = texture(w // 2, h // 2, SEED + 2, first + n)
    dt = np.uint8 if bd == 8 else np.uint16
    Y, U, V = np.empty((n, h, w), dt), np.empty((n, h // 2, w // 2), dt), np.empty((n, h // 2, w // 2), dt)

    def one(t):
        Y[t] = plane(ty, w, h, first + t, SEED, bd)
        U[t] = plane(tu, w // 2, h // 2, first + t, SEED + 1, bd, 0.5)
        V[t] = plane(tv, w // 2, h // 2, first + t, SEED + 2, bd, 0.5)

    workers = min(n, os.cpu_count() or 1, 16)
    if workers > 1 and n * w * h >= (1 << 24):     # frames are independent; numpy releases the GIL inside the array ops
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(one, range(n)))
    else:
        for t in range(n):
            one(t)
    return Y, U, V
