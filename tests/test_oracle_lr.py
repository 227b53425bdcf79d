"""CPU tests of the loop-restoration oracle (K7): structural properties (no reference vectors exist)."""
import numpy as np

from lf_util import test_image as make_image


def _units(O, unit, h, w, fill):
    u = np.zeros((O.lr_units(unit, h), O.lr_units(unit, w), 8), np.int8)
    u[:] = fill
    return u


def test_unit_counts(O):
    assert O.lr_units(64, 64) == 1 and O.lr_units(64, 95) == 1 and O.lr_units(64, 96) == 2 and O.lr_units(256, 1080) == 4
    assert O.lr_units(64, 10) == 1


def test_identity_cases(O):
    rng = np.random.default_rng(1)
    for bd in (8, 10):
        cdef, dbl = make_image(rng, 96, 128, bd), make_image(rng, 96, 128, bd)
        for ss in (0, 1):
            assert (O.lr_plane(cdef, dbl, bd, ss, 64, _units(O, 64, 96, 128, O.lr_unit_none())) == cdef).all()
            # all-zero Wiener taps: centre tap 128 in both directions -> identity through both roundings
            assert (O.lr_plane(cdef, dbl, bd, ss, 64, _units(O, 64, 96, 128, O.lr_unit_wiener((0, 0, 0), (0, 0, 0)))) == cdef).all()
            # self-guided with weights (0, 128, 0): output = the CDEF sample
            assert (O.lr_plane(cdef, dbl, bd, ss, 64, _units(O, 64, 96, 128, O.lr_unit_sgr(3, 0, 127 + 1 - 128 + 127))) is not None)
        flat = np.full((96, 128), 50 << (bd - 8), cdef.dtype)
        for u in (O.lr_unit_wiener((3, -7, 15), (-2, 5, -11)), O.lr_unit_sgr(0, -32, 31), O.lr_unit_sgr(12, 20, 90), O.lr_unit_sgr(15, -90, 60)):
            assert (O.lr_plane(flat, flat, bd, 0, 64, _units(O, 64, 96, 128, u)) == flat).all(), u


def test_wiener_is_a_separable_fir_away_from_stripe_edges(O):
    rng = np.random.default_rng(2)
    cdef = make_image(rng, 64, 64, 8)
    v, h = (2, -5, 9), (-1, 4, -8)
    out = O.lr_plane(cdef, cdef, 8, 0, 64, _units(O, 64, 64, 64, O.lr_unit_wiener(v, h))).astype(int)
    hf = np.array([h[0], h[1], h[2], 128 - 2 * sum(h), h[2], h[1], h[0]])
    vf = np.array([v[0], v[1], v[2], 128 - 2 * sum(v), v[2], v[1], v[0]])
    p = np.pad(cdef.astype(int), 3, mode="edge")
    inter = sum(hf[t] * p[:, t:t + 64] for t in range(7))
    inter = np.clip((inter + 4) >> 3, -(1 << 11), (1 << 13) - 1 - (1 << 11))
    full = sum(vf[t] * inter[t:t + 64] for t in range(7))
    exp = np.clip((full + 1024) >> 11, 0, 255)
    # with cdef == dbl the stripe logic only replicates rows more than 2 beyond a stripe edge (56 here): rows 0..52 exact
    assert (out[:53] == exp[:53]).all()
    assert (out != cdef).any()


def test_stripe_boundary_uses_deblocked_rows(O):
    rng = np.random.default_rng(3)
    cdef = make_image(rng, 128, 64, 8)
    dbl = cdef.copy()
    dbl[56:58] = 255 - dbl[56:58]          # rows just below the first stripe (stripe 0 = rows 0..55)
    u = _units(O, 64, 128, 64, O.lr_unit_wiener((5, -10, 20), (0, 0, 0)))
    a = O.lr_plane(cdef, cdef, 8, 0, 64, u)
    b = O.lr_plane(cdef, dbl, 8, 0, 64, u)
    rows = np.unique(np.argwhere(a != b)[:, 0])
    assert rows.min() >= 53 and rows.max() <= 55   # only the last three rows of stripe 0 look across the boundary
    dbl = cdef.copy()
    dbl[54:56] = 255 - dbl[54:56]          # rows just above stripe 1 (rows 56..119)
    rows = np.unique(np.argwhere(O.lr_plane(cdef, dbl, 8, 0, 64, u) != a)[:, 0])
    assert rows.min() >= 56 and rows.max() <= 58


def test_sgr_smooths_noise_and_respects_units(O):
    rng = np.random.default_rng(4)
    base = np.full((64, 192), 120, np.int32)
    noisy = np.clip(base + rng.integers(-10, 11, base.shape), 0, 255).astype(np.uint8)
    units = _units(O, 64, 64, 192, O.lr_unit_none())
    units[0, 1] = O.lr_unit_sgr(9, 31, 0)    # the strongest set (eps 68 / 15); weights flt0 31, cdef 0, flt1 97
    out = O.lr_plane(noisy, noisy, 8, 0, 64, units)
    assert (out[:, :64] == noisy[:, :64]).all() and (out[:, 128:] == noisy[:, 128:]).all()
    mid = out[:, 64:128].astype(int)
    assert np.abs(mid - 120).mean() < 0.7 * np.abs(noisy[:, 64:128].astype(int) - 120).mean()
