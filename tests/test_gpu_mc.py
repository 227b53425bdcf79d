"""GPU parity for K4 (sub-pel motion compensation) vs the CPU oracle.  north_star allows +-1 LSB; the kernel
restates the same two-stage rounding, so the test demands bit-exactness and records the tolerance it could use."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MC_TOLERANCE_LSB = 0   # north_star: "sub-pel MC within +-1 LSB"; achieved: exact

SIZES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (32, 64),
         (64, 32), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]


@pytest.mark.parametrize("bd", [8, 10])
def test_mc_all_sizes_phases_filters(ctx, O, av1mi, bd):
    rng = np.random.default_rng(70 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    H, W = 200, 264
    ref = rng.integers(0, 1 << bd, (H, W)).astype(dt)
    d_ref = ctx.to_device(ref)
    for sid, (w, h) in enumerate(SIZES):
        nbx, nby = W // w, H // h
        n = min(nbx * nby, 48)
        lst = np.zeros(n, av1mi.MC_BLK_DTYPE)
        sel = rng.choice(nbx * nby, n, replace=False)
        for i, s in enumerate(sel):
            by, bx = divmod(int(s), nbx)
            far = i % 7 == 0   # some vectors point far outside the plane: edge clamping
            mv = rng.integers(-4000, 4000, 2) if far else rng.integers(-300, 300, 2)
            lst[i] = (bx * w, by * h, mv[0], mv[1], rng.integers(0, 4), rng.integers(0, 4), 0)
        lst[0]["mvx"], lst[0]["mvy"] = 0, 0          # integer copy
        lst[1]["mvx"], lst[1]["mvy"] = 8, 8          # half-pel both ways
        d_lst = ctx.to_device(lst)
        d_dst = ctx.to_device(np.zeros((H, W), dt))
        ctx.mc_list(sid, d_ref, W, W, H, d_dst, W, bd, d_lst, n)
        got = d_dst.download((H, W), dt)
        d_lst.free(); d_dst.free()
        exp = np.zeros((H, W), dt)
        for b in lst:
            x, y = int(b["x"]), int(b["y"])
            exp[y:y + h, x:x + w] = O.mc_block(ref, bd, x, y, w, h, int(b["mvx"]), int(b["mvy"]), int(b["filt_x"]), int(b["filt_y"]))
        assert np.abs(got.astype(int) - exp.astype(int)).max() <= MC_TOLERANCE_LSB, ((w, h), bd)
    d_ref.free()


def test_mc_every_phase_pair(ctx, O, av1mi):
    rng = np.random.default_rng(77)
    ref = rng.integers(0, 256, (64, 64)).astype(np.uint8)
    d_ref = ctx.to_device(ref)
    for filt in range(4):
        lst = np.zeros(256, av1mi.MC_BLK_DTYPE)
        for p in range(256):
            lst[p] = (24, 24, 16 + (p & 15), -32 + (p >> 4), filt, filt, 0)
        for p in range(256):   # same destination: one launch per phase pair
            d_l, d_dst = ctx.to_device(lst[p:p + 1]), ctx.to_device(np.zeros((64, 64), np.uint8))
            ctx.mc_list(1, d_ref, 64, 64, 64, d_dst, 64, 8, d_l, 1)
            got = d_dst.download((64, 64), np.uint8)[24:32, 24:32]
            d_l.free(); d_dst.free()
            assert (got == O.mc_block(ref, 8, 24, 24, 8, 8, 16 + (p & 15), -32 + (p >> 4), filt, filt)).all(), (filt, p)
    d_ref.free()
