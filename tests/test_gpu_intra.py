"""GPU parity for K3 (intra prediction) — HIP kernel through the C ABI vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TXS = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (32, 64),
       (64, 32), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]


def _cases(rng, bw, bh, n):
    """random block descriptors incl. every availability pattern the oracle distinguishes"""
    out = []
    for i in range(n):
        mode = int(rng.integers(0, 13))
        delta = int(rng.integers(-3, 4)) if 1 <= mode <= 8 else 0
        pat = i % 6
        nt, ntr, nl, nbl = bw, bw, bh, bh
        if pat == 1: nt = ntr = 0
        elif pat == 2: nl = nbl = 0
        elif pat == 3: nt = ntr = nl = nbl = 0
        elif pat == 4: ntr = int(rng.integers(0, bw + 1)) // 4 * 4; nbl = 0
        elif pat == 5:
            nt = max(4, int(rng.integers(1, bw + 1)) // 4 * 4); ntr = 0 if nt < bw else bw
            nl = max(4, int(rng.integers(1, bh + 1)) // 4 * 4); nbl = 0 if nl < bh else bh
        out.append((mode, delta, int(rng.integers(0, 2)), int(rng.integers(0, 2)), nt, ntr, nl, nbl))
    return out


@pytest.mark.parametrize("bd", [8, 10])
def test_intra_pred_all_sizes(ctx, O, av1mi, bd):
    rng = np.random.default_rng(40 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    for ts, (bw, bh) in enumerate(TXS):
        # blocks on a sparse lattice so that every block has its own full neighbourhood in `ref`
        px, py = 2 * bw + bh + 8, 2 * bh + bw + 8
        px = (px + 3) // 4 * 4
        nbx, nby = 6, 5
        W, H = nbx * px + 8, nby * py + 8
        W = (W + 3) // 4 * 4
        ref = rng.integers(0, 1 << bd, (H, W)).astype(dt)
        cases = _cases(rng, bw, bh, nbx * nby)
        lst = np.zeros(nbx * nby, av1mi.INTRA_BLK_DTYPE)
        exp = np.zeros((H, W), dt)
        for i, (mode, delta, dis, ft, nt, ntr, nl, nbl) in enumerate(cases):
            x, y = 4 + (i % nbx) * px + 4, 4 + (i // nbx) * py + 4
            x = x // 4 * 4
            lst[i] = (x, y, mode, delta, dis | (ft << 1), nt, ntr, nl, nbl, 0)
            exp[y:y + bh, x:x + bw] = O.intra_predict(ref, x, y, bw, bh, mode, delta, bd, nt, ntr, nl, nbl, dis, ft)
        d_ref, d_lst = ctx.to_device(ref), ctx.to_device(lst)
        d_dst = ctx.to_device(np.zeros((H, W), dt))
        ctx.intra_pred_list(ts, d_ref, W, d_dst, W, bd, d_lst, len(lst))
        got = d_dst.download((H, W), dt)
        for b in (d_ref, d_lst, d_dst):
            b.free()
        for i, c in enumerate(cases):
            x, y = int(lst[i]["x"]), int(lst[i]["y"])
            assert (got[y:y + bh, x:x + bw] == exp[y:y + bh, x:x + bw]).all(), ((bw, bh), bd, O.INTRA_MODE_NAMES[c[0]], c)
        assert (got == exp).all()   # nothing outside the listed blocks is written


def test_intra_pred_every_angle_8x8_16x16(ctx, O, av1mi):
    rng = np.random.default_rng(44)
    for ts, (bw, bh) in ((1, (8, 8)), (2, (16, 16)), (0, (4, 4))):
        W = H = 96
        ref = rng.integers(0, 256, (H, W)).astype(np.uint8)
        combos = [(m, d, ft, dis) for m in range(1, 9) for d in range(-3, 4) for ft in (0, 1) for dis in (0, 1)]
        lst = np.zeros(len(combos), av1mi.INTRA_BLK_DTYPE)
        for i, (m, d, ft, dis) in enumerate(combos):
            lst[i] = (40, 40, m, d, dis | (ft << 1), bw, bw, bh, bh, 0)
        # all blocks at the same place: launch one at a time into separate outputs
        d_ref = ctx.to_device(ref)
        for i, (m, d, ft, dis) in enumerate(combos):
            d_l = ctx.to_device(lst[i:i + 1])
            d_dst = ctx.to_device(np.zeros((H, W), np.uint8))
            ctx.intra_pred_list(ts, d_ref, W, d_dst, W, 8, d_l, 1)
            got = d_dst.download((H, W), np.uint8)[40:40 + bh, 40:40 + bw]
            exp = O.intra_predict(ref, 40, 40, bw, bh, m, d, 8, bw, bw, bh, bh, dis, ft)
            d_l.free(); d_dst.free()
            assert (got == exp).all(), ((bw, bh), O.INTRA_MODE_NAMES[m], d, ft, dis)
        d_ref.free()


@pytest.mark.parametrize("bd", [8, 10])
def test_cfl_pred_matches_oracle(ctx, O, av1mi, bd):
    """chroma-from-luma (spec 7.11.5) for every chroma transform size up to 32x32, all alphas, with and without the
    MaxLumaW / MaxLumaH limits, many blocks per launch"""
    rng = np.random.default_rng(90 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    LH, LW = 256, 384
    luma = rng.integers(0, 1 << bd, (LH, LW)).astype(dt)
    sizes = {0: (4, 4), 1: (8, 8), 2: (16, 16), 3: (32, 32), 5: (4, 8), 6: (8, 4), 7: (8, 16), 8: (16, 8), 9: (16, 32), 10: (32, 16),
             13: (4, 16), 14: (16, 4), 15: (8, 32), 16: (32, 8)}
    d_luma = ctx.to_device(luma)
    for ts, (bw, bh) in sizes.items():
        dc = rng.integers(0, 1 << bd, (LH // 2, LW // 2)).astype(dt)
        nbx, nby = (LW // 2) // bw, (LH // 2) // bh
        n = min(nbx * nby, 40)
        sel = rng.choice(nbx * nby, n, replace=False)
        lst = np.zeros(n, av1mi.CFL_BLK_DTYPE)
        exp = dc.copy()
        for i, s in enumerate(sel):
            by, bx = divmod(int(s), nbx)
            x, y = bx * bw, by * bh
            alpha = int(rng.integers(-16, 17))
            mw, mh = LW, LH
            if i % 3 == 0:        # only part of the luma block is available
                mw = max(2, 2 * x + int(rng.integers(1, bw + 1)) * 2 - 2 * (i % 2))
                mh = max(2, 2 * y + int(rng.integers(1, bh + 1)) * 2)
            lst[i] = (x, y, mw, mh, alpha, 0)
            exp = O.cfl_predict(luma, exp, bd, x, y, bw, bh, alpha, mw, mh)
        d_dst, d_l = ctx.to_device(dc), ctx.to_device(lst)
        ctx.cfl_pred_list(ts, d_luma, LW, d_dst, LW // 2, bd, d_l, n)
        got = d_dst.download(dc.shape, dt)
        d_dst.free(); d_l.free()
        assert (got == exp).all(), ((bw, bh), bd)
    d = ctx.alloc(64)
    with pytest.raises(av1mi.Av1miError, match="up to 32x32"):
        ctx.cfl_pred_list(4, d_luma, LW, d, 64, bd, d, 1)
    d.free(); d_luma.free()
