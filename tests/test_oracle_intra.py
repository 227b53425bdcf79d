"""CPU tests of the intra-prediction oracle (K3).  No reference vectors exist, so the oracle
is pinned by structural identities that do not depend on recalled constants."""
import numpy as np
import pytest

SIZES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (32, 64),
         (64, 32), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]
BASE_ANGLE = {1: 90, 2: 180, 3: 45, 4: 135, 5: 113, 6: 157, 7: 203, 8: 67}
ANGLE_TO_MODE = {}
for m, a in BASE_ANGLE.items():
    for d in range(-3, 4):
        ANGLE_TO_MODE.setdefault(a + 3 * d, (m, d))


def _plane(rng, bd, n=224):
    return rng.integers(0, 1 << bd, (n, n)).astype(np.uint8 if bd == 8 else np.uint16)


def test_constant_neighbourhood_predicts_constant(O):
    for bd in (8, 10):
        p = np.full((224, 224), 77 << (bd - 8), np.uint8 if bd == 8 else np.uint16)
        for bw, bh in SIZES:
            for mode in range(13):
                for delta in ((-3, 0, 3) if 1 <= mode <= 8 else (0,)):
                    out = O.intra_predict(p, 72, 72, bw, bh, mode, delta, bd, bw, bw, bh, bh)
                    assert (out == 77 << (bd - 8)).all(), (bw, bh, mode, delta)


def test_v_and_h_copy_edges(O):
    rng = np.random.default_rng(1)
    p = _plane(rng, 8)
    for bw, bh in SIZES:
        v = O.intra_predict(p, 72, 72, bw, bh, 1, 0, 8, bw, bw, bh, bh)
        h = O.intra_predict(p, 72, 72, bw, bh, 2, 0, 8, bw, bw, bh, bh)
        assert (v == p[71, 72:72 + bw][None, :]).all() and (h == p[72:72 + bh, 71][:, None]).all()


def test_unavailable_edges(O):
    rng = np.random.default_rng(2)
    p = _plane(rng, 8)
    # nothing available: DC = 128, V = 127 (above base-1), H = 129 (left base+1)
    assert (O.intra_predict(p, 72, 72, 8, 8, 0, 0, 8, 0, 0, 0, 0) == 128).all()
    assert (O.intra_predict(p, 72, 72, 8, 8, 1, 0, 8, 0, 0, 0, 0) == 127).all()
    assert (O.intra_predict(p, 72, 72, 8, 8, 2, 0, 8, 0, 0, 0, 0) == 129).all()
    assert (O.intra_predict(p, 72, 72, 8, 8, 1, 0, 10, 0, 0, 0, 0) == 511).all()
    # only left available: V copies the left neighbour of row 0; DC averages the left column
    assert (O.intra_predict(p, 72, 72, 8, 8, 1, 0, 8, 0, 0, 8, 0) == p[72, 71]).all()
    assert (O.intra_predict(p, 72, 72, 8, 8, 0, 0, 8, 0, 0, 8, 0) == (int(p[72:80, 71].sum()) + 4) // 8).all()
    # only top available: H copies the pixel above column 0
    assert (O.intra_predict(p, 72, 72, 8, 8, 2, 0, 8, 8, 0, 0, 0) == p[71, 72]).all()
    # no top-right: D45 replicates the last above sample beyond the block width
    a = O.intra_predict(p, 72, 72, 8, 8, 3, 0, 8, 8, 0, 8, 0, disable_edge_filter=1)
    assert a[7, 7] == p[71, 79] and a[0, 0] == (int(p[71, 73]) * 32 + 16) >> 5


def test_d45_is_a_diagonal_shift_without_filter(O):
    rng = np.random.default_rng(3)
    p = _plane(rng, 8)
    out = O.intra_predict(p, 72, 72, 16, 16, 3, 0, 8, 16, 16, 16, 16, disable_edge_filter=1)
    for r in range(16):
        for c in range(16):
            i = r + c + 1
            exp = p[71, 72 + i] if i < 31 else p[71, 72 + 31]
            assert out[r, c] == exp


@pytest.mark.parametrize("bd", [8, 10])
def test_transpose_symmetry(O, bd):
    """pred(plane^T, w<->h, angle 270-a) == pred(plane, angle a)^T: pins zone 1 against zone 3 and the
    above/left edge filter + upsampling paths against each other (zone 2 only approximately)."""
    rng = np.random.default_rng(4 + bd)
    p = _plane(rng, bd)
    pt = np.ascontiguousarray(p.T)
    n = 0
    for bw, bh in SIZES:
        for a, (m, d) in ANGLE_TO_MODE.items():
            at = 270 - a
            if at not in ANGLE_TO_MODE:
                continue
            mt, dtl = ANGLE_TO_MODE[at]
            for ft in (0, 1):
                for dis in (0, 1):
                    x = O.intra_predict(p, 72, 80, bw, bh, m, d, bd, bw, bw, bh, bh, dis, ft)
                    y = O.intra_predict(pt, 80, 72, bh, bw, mt, dtl, bd, bh, bh, bw, bw, dis, ft)
                    if 90 < a < 180:
                        # zone 2 gives the above edge priority where both edges are hit: only nearly symmetric
                        assert (x == y.T).mean() > 0.6, (bw, bh, a, ft, dis)
                    else:
                        assert (x == y.T).all(), (bw, bh, a, ft, dis)
                    n += 1
        for m, mt in ((0, 0), (12, 12), (9, 9), (10, 11)):
            x = O.intra_predict(p, 72, 80, bw, bh, m, 0, bd, bw, bw, bh, bh)
            y = O.intra_predict(pt, 80, 72, bh, bw, mt, 0, bd, bh, bh, bw, bw)
            assert (x == y.T).all(), (bw, bh, m)
    assert n > 2000


def test_linear_ramp_is_reproduced_by_directional_modes(O):
    """a plane that is linear along the prediction direction's normal is predicted (nearly) exactly"""
    yy, xx = np.mgrid[0:224, 0:224]
    p = np.clip(100 + (xx - yy) * 2, 0, 1023).astype(np.uint16)   # constant along the 135-degree diagonal
    out = O.intra_predict(p, 72, 72, 16, 16, 4, 0, 10, 16, 16, 16, 16)
    assert np.abs(out.astype(int) - p[72:88, 72:88]).max() <= 1


def test_paeth_and_smooth_against_numpy(O):
    rng = np.random.default_rng(5)
    p = _plane(rng, 10)
    for bw, bh in ((4, 4), (8, 16), (32, 8), (64, 64)):
        top = p[71, 72:72 + bw].astype(int); left = p[72:72 + bh, 71].astype(int); tl = int(p[71, 71])
        base = top[None, :] + left[:, None] - tl
        pl, pt, ptl = np.abs(base - left[:, None]), np.abs(base - top[None, :]), np.abs(base - tl)
        exp = np.where((pl <= pt) & (pl <= ptl), left[:, None], np.where(pt <= ptl, top[None, :], tl))
        assert (O.intra_predict(p, 72, 72, bw, bh, 12, 0, 10, bw, bw, bh, bh) == exp).all()
        sm = O.intra_predict(p, 72, 72, bw, bh, 9, 0, 10, bw, bw, bh, bh).astype(int)
        assert sm.min() >= min(top.min(), left.min()) and sm.max() <= max(top.max(), left.max())
        assert sm[0, 0] == (255 * top[0] + 1 * left[-1] + 255 * left[0] + 1 * top[-1] + 256) >> 9


def test_cfl_prediction_properties(O):
    """chroma-from-luma (spec 7.11.5) against an independent numpy restatement, plus what the definition implies:
    alpha = 0 leaves the DC prediction alone, flat luma does too, the sign of alpha mirrors the correction"""
    rng = np.random.default_rng(21)
    for bd in (8, 10):
        dt = np.uint8 if bd == 8 else np.uint16
        luma = rng.integers(0, 1 << bd, (96, 128)).astype(dt)
        dc = np.full((48, 64), 100 << (bd - 8), dt)
        for (bw, bh) in ((4, 4), (8, 8), (16, 8), (4, 16), (32, 32), (32, 8)):
            for alpha in (-16, -5, 0, 3, 16):
                x, y = 8, 4
                got = O.cfl_predict(luma, dc, bd, x, y, bw, bh, alpha)
                blk = luma[2 * y:2 * (y + bh), 2 * x:2 * (x + bw)].astype(np.int64)
                L = (blk[0::2, 0::2] + blk[0::2, 1::2] + blk[1::2, 0::2] + blk[1::2, 1::2]) << 1
                avg = (L.sum() + (bw * bh) // 2) // (bw * bh)
                v = alpha * (L - avg)
                sl = np.where(v >= 0, (v + 32) >> 6, -((-v + 32) >> 6))
                exp = dc.copy()
                exp[y:y + bh, x:x + bw] = np.clip(dc[y:y + bh, x:x + bw].astype(np.int64) + sl, 0, (1 << bd) - 1)
                assert (got == exp).all(), (bd, bw, bh, alpha)
                if alpha == 0:
                    assert (got == dc).all()
        flat = np.full_like(luma, 77)
        assert (O.cfl_predict(flat, dc, bd, 0, 0, 8, 8, 9) == dc).all()
        # availability: beyond max_luma_w / max_luma_h the last 2x2 group repeats
        got = O.cfl_predict(luma, dc, bd, 0, 0, 16, 16, 7, max_luma_w=20, max_luma_h=12)
        ext = luma.copy().astype(np.int64)
        ext[:, 20:] = np.tile(ext[:, 18:20], (1, 54))
        ext[12:, :] = np.tile(ext[10:12, :], (42, 1))
        assert (got == O.cfl_predict(ext.astype(dt), dc, bd, 0, 0, 16, 16, 7)).all()
