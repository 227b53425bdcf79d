"""CPU tests of the motion-compensation oracle (K4): the filter bank's structural identities and the block process."""
import numpy as np


def test_filter_bank_identities(O):
    f = O.subpel_filters().astype(int)
    assert (f.sum(axis=2) == 128).all()                       # unit DC gain, every filter and phase
    assert (f[:, 0] == [0, 0, 0, 128, 0, 0, 0, 0]).all()      # phase 0 is the identity
    for k in range(6):
        for p in range(1, 16):
            assert (f[k, p] == f[k, 16 - p][::-1]).all(), (k, p)  # phase p mirrors phase 16-p
    assert (f[4, :, :2] == 0).all() and (f[4, :, 6:] == 0).all() and (f[5, :, :2] == 0).all() and (f[5, :, 6:] == 0).all()
    assert (f[3, :, :3] == 0).all() and (f[3, :, 5:] == 0).all()
    assert (f[2, 8] == [-4, 12, -24, 80, 80, -24, 12, -4]).all() and (f[0, 8] == [0, 2, -14, 76, 76, -14, 2, 0]).all()


def test_integer_mv_is_a_copy_with_edge_clamp(O):
    """both ways: through the two filter stages (phase 0 is exact) and through the oracle's whole-sample shortcut"""
    for fast in (0, 1):
        O.lib().av1o_mc_set_fast_path(fast)
        try:
            _integer_mv_cases(O)
        finally:
            O.lib().av1o_mc_set_fast_path(1)


def _integer_mv_cases(O):
    rng = np.random.default_rng(1)
    for bd in (8, 10):
        ref = rng.integers(0, 1 << bd, (72, 96)).astype(np.uint8 if bd == 8 else np.uint16)
        got = O.mc_block(ref, bd, 16, 24, 16, 8, 5 * 16, -3 * 16)
        assert (got == ref[21:29, 21:37]).all()
        got = O.mc_block(ref, bd, 80, 60, 16, 16, 40 * 16, 30 * 16)     # far outside: clamps to the corner sample
        assert (got[8:, 8:] == ref[-1, -1]).all()
        got = O.mc_block(ref, bd, 0, 0, 8, 8, -64 * 16, -64 * 16, 2, 2)
        assert (got == ref[0, 0]).all()


def test_constant_and_range(O):
    ref = np.full((64, 64), 1023, np.uint16)
    for fx in range(4):
        for ph in range(16):
            assert (O.mc_block(ref, 10, 16, 16, 8, 8, ph, (ph * 7) & 15, fx, (fx + 1) & 3) == 1023).all()
    rng = np.random.default_rng(2)
    ref = (rng.integers(0, 2, (64, 64)) * 255).astype(np.uint8)   # worst-case ringing stays inside the pixel range
    out = O.mc_block(ref, 8, 16, 16, 16, 16, 7, 9, 2, 2)
    assert out.min() >= 0 and out.max() <= 255


def test_half_pel_bilinear_matches_closed_form(O):
    rng = np.random.default_rng(3)
    ref = rng.integers(0, 256, (40, 40)).astype(np.uint8)
    got = O.mc_block(ref, 8, 8, 8, 8, 8, 8, 0, 3, 3).astype(int)      # x half-pel, bilinear: (a+b)/2 through two roundings
    a, b = ref[8:16, 8:16].astype(int), ref[8:16, 9:17].astype(int)
    inter = (64 * a + 64 * b + 4) >> 3
    assert (got == ((128 * inter + 1024) >> 11)).all()


def test_transpose_symmetry(O):
    rng = np.random.default_rng(4)
    ref = rng.integers(0, 1024, (80, 80)).astype(np.uint16)
    reft = np.ascontiguousarray(ref.T)
    for (w, h, mvx, mvy, fx, fy) in ((8, 8, 5, 11, 0, 2), (16, 4, 37, -21, 1, 0), (4, 16, -7, 3, 2, 2), (32, 8, 100, 50, 0, 1)):
        a = O.mc_block(ref, 10, 24, 32, w, h, mvx, mvy, fx, fy)
        # horizontal-then-vertical is not exactly vertical-then-horizontal (different intermediate rounding): +-1
        b = O.mc_block(reft, 10, 32, 24, h, w, mvy, mvx, fy, fx)
        assert np.abs(a.astype(int) - b.T.astype(int)).max() <= 1


def test_small_blocks_use_four_tap_filters(O):
    rng = np.random.default_rng(5)
    ref = rng.integers(0, 256, (40, 40)).astype(np.uint8)
    f = O.subpel_filters().astype(int)
    got = O.mc_block(ref, 8, 16, 16, 4, 8, 5, 0, 0, 0).astype(int)     # w = 4: horizontal filter index 4
    taps = f[4, 5]
    exp = np.zeros((8, 4), int)
    for r in range(8):
        for c in range(4):
            s = sum(taps[t] * int(ref[16 + r, 16 + c + t - 3]) for t in range(8))
            exp[r, c] = min(max((128 * ((s + 4) >> 3) + 1024) >> 11, 0), 255)
    assert (got == exp).all()
