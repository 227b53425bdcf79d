"""GPU parity tests for K1/K2/K8 — HIP kernels through the C ABI vs the CPU oracle,
bit-exact (integer work).  Run with -m gpu on the MI355X box."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "txfm_kat.npz")


def _valid(O):
    return [(ts, tt) for ts in range(19) for tt in range(16) if O.txfm_valid(ts, tt)]


def test_inv_golden_single_block(ctx, O):
    """committed goldens through the host-pointer entry point (av1mi_inv_txfm2d_add)."""
    g = np.load(GOLD)
    for ts, tt in _valid(O):
        for bd in (8, 10):
            key = "inv_%d_%d_%d" % (ts, tt, bd)
            pred = g[key + "_pred"]
            for c, r in zip(g[key + "_coef"], g[key + "_rec"]):
                got = ctx.inv_txfm2d_add(c, pred, ts, tt, bd)
                assert (got == r).all(), (O.TX_NAMES[ts], O.TX_TYPE_NAMES[tt], bd)


def test_fwd_golden_single_block(ctx, O):
    g = np.load(GOLD)
    for ts, tt in _valid(O):
        got = ctx.fwd_txfm2d(g["fwd_%d_%d_res" % (ts, tt)], ts, tt)
        assert (got == g["fwd_%d_%d_coef" % (ts, tt)]).all(), (O.TX_NAMES[ts], O.TX_TYPE_NAMES[tt])


def _grid_case(O, rng, ts, bd, nbx, nby, amp):
    h, w = O.TX_H[ts], O.TX_W[ts]
    ch, cw = O.coef_shape(ts)
    nb = nbx * nby
    types = np.array([t for t in range(16) if O.txfm_valid(ts, t)], np.uint8)
    tts = types[rng.integers(0, len(types), nb)]
    coef = rng.integers(-amp, amp + 1, (nb, ch, cw)).astype(np.int32)
    dt = np.uint8 if bd == 8 else np.uint16
    pred = rng.integers(0, 1 << bd, (nby * h, nbx * w)).astype(dt)
    exp = pred.copy()
    for b in range(nb):
        by, bx = divmod(b, nbx)
        sl = (slice(by * h, by * h + h), slice(bx * w, bx * w + w))
        exp[sl] = O.inv_txfm2d_add(coef[b], pred[sl], ts, int(tts[b]), bd)
    return coef, pred, exp, tts


@pytest.mark.parametrize("bd", [8, 10])
def test_inv_grid_mixed_types_all_sizes(ctx, O, bd):
    rng = np.random.default_rng(100 + bd)
    for ts in range(19):
        nbx, nby = 5, 3   # ragged: 15 blocks never fill a workgroup evenly
        for amp in (300, 1 << (bd + 7), 1 << 20):
            coef, pred, exp, tts = _grid_case(O, rng, ts, bd, nbx, nby, amp)
            d_coef, d_plane, d_types = ctx.to_device(coef), ctx.to_device(pred), ctx.to_device(tts)
            ctx.inv_txfm_add_grid(ts, d_coef, d_plane, pred.shape[1], bd, nbx, nbx * nby, d_types)
            got = d_plane.download(pred.shape, pred.dtype)
            for b in (d_coef, d_plane, d_types):
                b.free()
            assert (got == exp).all(), (O.TX_NAMES[ts], bd, amp)


def test_inv_list_scattered_blocks(ctx, O, av1mi):
    """list form: blocks at arbitrary (x,y) with arbitrary coefficient offsets, untouched pixels stay."""
    rng = np.random.default_rng(9)
    ts, bd = 1, 8
    H, W = 64, 96
    pred = rng.integers(0, 256, (H, W)).astype(np.uint8)
    pos = [(8, 0), (40, 8), (88, 56), (0, 56), (16, 24)]
    coef = rng.integers(-2000, 2000, (len(pos) + 2, 8, 8)).astype(np.int32)
    lst = np.zeros(len(pos), av1mi.TXB_DTYPE)
    exp = pred.copy()
    for i, (x, y) in enumerate(pos):
        slot = len(pos) + 1 - i   # reversed, with a gap
        tt = [0, 3, 9, 5, 12][i]
        lst[i] = (slot * 64, x, y, tt, 0)
        exp[y:y + 8, x:x + 8] = O.inv_txfm2d_add(coef[slot], pred[y:y + 8, x:x + 8], ts, tt, bd)
    d_coef, d_plane, d_list = ctx.to_device(coef), ctx.to_device(pred), ctx.to_device(lst)
    ctx.inv_txfm_add_list(ts, d_coef, d_plane, W, bd, d_list, len(pos))
    got = d_plane.download(pred.shape, pred.dtype)
    assert (got == exp).all()


def test_inv_empty_and_invalid(ctx, av1mi, O):
    d = ctx.alloc(1024)
    ctx.inv_txfm_add_grid(1, d, d, 8, 8, 1, 0)           # zero blocks: no-op
    with pytest.raises(av1mi.Av1miError):
        ctx.inv_txfm_add_grid(4, d, d, 64, 8, 1, 1, None, 1)  # ADST on 64x64 does not exist
    with pytest.raises(av1mi.Av1miError):
        ctx.inv_txfm_add_grid(1, d, d, 6, 8, 1, 1)        # stride not a multiple of 4
    with pytest.raises(av1mi.Av1miError):
        ctx.inv_txfm_add_grid(1, d, d, 8, 12, 1, 1)       # 12-bit unsupported
    d.free()


def test_fwd_grid_all_sizes(ctx, O):
    rng = np.random.default_rng(11)
    for ts in range(19):
        h, w = O.TX_H[ts], O.TX_W[ts]
        ch, cw = O.coef_shape(ts)
        nbx, nby = 3, 3
        types = np.array([t for t in range(16) if O.txfm_valid(ts, t)], np.uint8)
        tts = types[rng.integers(0, len(types), nbx * nby)]
        for amp in (255, 1023):
            res = rng.integers(-amp, amp + 1, (nby * h, nbx * w)).astype(np.int16)
            d_res, d_types = ctx.to_device(res), ctx.to_device(tts)
            d_coef = ctx.alloc(nbx * nby * ch * cw * 4)
            ctx.fwd_txfm_grid(ts, d_res, res.shape[1], d_coef, nbx, nbx * nby, d_types)
            got = d_coef.download((nbx * nby, ch, cw), np.int32)
            for b in (d_res, d_types, d_coef):
                b.free()
            for b in range(nbx * nby):
                by, bx = divmod(b, nbx)
                exp = O.fwd_txfm2d(res[by * h:by * h + h, bx * w:bx * w + w], ts, int(tts[b]))
                assert (got[b] == exp).all(), (O.TX_NAMES[ts], O.TX_TYPE_NAMES[tts[b]], amp)


@pytest.mark.parametrize("bd", [8, 10])
def test_wht_lossless_blocks(ctx, O, bd):
    """tx_type AV1MI_WHT_WHT (16) on 4x4 grids, mixed with the 16 regular types per block: GPU == oracle both ways, and the
    GPU pair alone reconstructs every residual exactly"""
    rng = np.random.default_rng(40 + bd)
    nbx, nby = 9, 7
    amp = (1 << bd) - 1
    tts = rng.choice(np.array(list(range(17)), np.uint8), nbx * nby)
    tts[::2] = 16
    res = rng.integers(-amp, amp + 1, (nby * 4, nbx * 4)).astype(np.int16)
    d_res, d_types = ctx.to_device(res), ctx.to_device(tts)
    d_coef = ctx.alloc(nbx * nby * 16 * 4)
    ctx.fwd_txfm_grid(0, d_res, res.shape[1], d_coef, nbx, nbx * nby, d_types)
    coef = d_coef.download((nbx * nby, 4, 4), np.int32)
    dt = np.uint8 if bd == 8 else np.uint16
    pred = rng.integers(0, 1 << bd, res.shape).astype(dt)
    d_plane = ctx.to_device(pred)
    ctx.inv_txfm_add_grid(0, d_coef, d_plane, pred.shape[1], bd, nbx, nbx * nby, d_types)
    got = d_plane.download(pred.shape, dt)
    for b in range(nbx * nby):
        by, bx = divmod(b, nbx)
        sl = (slice(by * 4, by * 4 + 4), slice(bx * 4, bx * 4 + 4))
        assert (coef[b] == O.fwd_txfm2d(res[sl], 0, int(tts[b]))).all(), (b, tts[b])
        assert (got[sl] == O.inv_txfm2d_add(coef[b], pred[sl], 0, int(tts[b]), bd)).all(), (b, tts[b])
        if tts[b] == 16:
            assert (got[sl].astype(np.int64) == np.clip(pred[sl].astype(np.int64) + res[sl], 0, amp)).all()
    for b in (d_res, d_types, d_coef, d_plane):
        b.free()
    # uniform-type form and the validity rule
    d = ctx.alloc(64 * 64 * 4)
    with pytest.raises(Exception):
        ctx.inv_txfm_add_grid(1, d, d, 8, 8, 1, 1, None, 16)      # WHT is defined for 4x4 only
    d.free()


def test_full_frame_round_trip_1080p(ctx, O):
    """BASELINE size, size-independent property: inv(fwd(residual)) reconstructs within +-2 on a whole
    1920x1088 8-bit plane of 8x8 DCT blocks, and the GPU result equals the oracle on sampled blocks."""
    rng = np.random.default_rng(12)
    W, H, ts = 1920, 1088, 1
    res = rng.integers(-200, 201, (H, W)).astype(np.int16)
    nbx, nby = W // 8, H // 8
    d_res = ctx.to_device(res)
    d_coef = ctx.alloc(W * H * 4)
    pred = np.full((H, W), 128, np.uint8)
    pred[:, ::3] = 100
    d_plane = ctx.to_device(pred)
    ctx.fwd_txfm_grid(ts, d_res, W, d_coef, nbx, nbx * nby)
    ctx.inv_txfm_add_grid(ts, d_coef, d_plane, W, 8, nbx, nbx * nby)
    rec = d_plane.download((H, W), np.uint8).astype(np.int32)
    coef = d_coef.download((nbx * nby, 8, 8), np.int32)
    want = np.clip(pred.astype(np.int32) + res, 0, 255)
    assert np.abs(rec - want).max() <= 2
    for b in rng.integers(0, nbx * nby, 64):
        by, bx = divmod(int(b), nbx)
        sl = (slice(by * 8, by * 8 + 8), slice(bx * 8, bx * 8 + 8))
        assert (coef[b] == O.fwd_txfm2d(res[sl], ts, 0)).all()
        assert (rec[sl] == O.inv_txfm2d_add(coef[b], pred[sl], ts, 0, 8)).all()
    for b in (d_res, d_coef, d_plane):
        b.free()


def test_quantize_dequantize_parity(ctx, O):
    rng = np.random.default_rng(13)
    for ts, bd, q in ((0, 8, 0), (1, 8, 128), (3, 10, 128), (4, 10, 255), (9, 8, 60)):
        ch, cw = O.coef_shape(ts)
        nb = 37
        ls = O.tx_scale(ts)
        dcq, acq = O.dc_q(q, bd), O.ac_q(q, bd)
        coef = rng.integers(-60000, 60000, (nb, ch, cw)).astype(np.int32)
        coef[0].flat[:6] = [0, 1, -1, 2 ** 31 - 1, -2 ** 31, 32767]
        d_coef = ctx.to_device(coef)
        d_lv, d_dq, d_dq2 = ctx.alloc(coef.size * 2), ctx.alloc(coef.size * 4), ctx.alloc(coef.size * 4)
        ctx.quantize(d_coef, d_lv, d_dq, coef.size, ch * cw, dcq, acq, ls)
        ctx.dequantize(d_lv, d_dq2, coef.size, ch * cw, dcq, acq, ls, bd)
        lv, dq, dq2 = d_lv.download(coef.shape, np.int16), d_dq.download(coef.shape, np.int32), d_dq2.download(coef.shape, np.int32)
        for b in range(nb):
            elv, edq, _ = O.quantize(coef[b], dcq, acq, ls)
            assert (lv[b] == elv).all() and (dq[b] == edq).all(), (ts, bd, q, b)
            assert (dq2[b] == O.dequantize(elv, dcq, acq, ls, bd)).all()
        for b in (d_coef, d_lv, d_dq, d_dq2):
            b.free()
