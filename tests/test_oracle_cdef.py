"""CPU tests of the CDEF oracle (K6): structural properties (no reference vectors exist)."""
import numpy as np

from lf_util import test_image as make_image


def _frame(rng, h, w, bd):
    return make_image(rng, h, w, bd), make_image(rng, h // 2, w // 2, bd), make_image(rng, h // 2, w // 2, bd)


def test_direction_search_finds_oriented_edges(O):
    yy, xx = np.mgrid[0:8, 0:8]
    # cdef direction 2 = horizontal structure (constant along rows), 6 = vertical, 0 = 45 deg up-right, 4 = 135 deg
    for expr, d in ((yy * 30, 2), (xx * 30, 6), ((xx + yy) * 15, 0), ((xx - yy) * 15 + 120, 4)):
        got, var = O.cdef_find_dir(np.clip(expr, 0, 255).astype(np.uint8), 8)
        assert got == d and var > 0, (d, got, var)
    assert O.cdef_find_dir(np.full((8, 8), 90, np.uint8), 8) == (0, 0)
    blk = np.clip(xx * 30, 0, 255)
    assert O.cdef_find_dir((blk * 4 + 3).astype(np.uint16), 10) == O.cdef_find_dir(blk.astype(np.uint8), 8)


def test_flat_off_and_skip_are_identity(O):
    rng = np.random.default_rng(1)
    for bd in (8, 10):
        Y, U, V = _frame(rng, 64, 128, bd)
        skip = np.zeros((8, 16), np.uint8)
        off = np.array([[255, 0, 0, 0], [255, 3, 15, 3]], np.uint8)
        out = O.cdef_frame(Y, U, V, bd, 3, off, skip)
        assert all((a == b).all() for a, b in zip(out, (Y, U, V)))
        on = np.array([[9, 2, 5, 1], [4, 3, 15, 3]], np.uint8)
        out = O.cdef_frame(Y, U, V, bd, 5, on, np.ones((8, 16), np.uint8))
        assert all((a == b).all() for a, b in zip(out, (Y, U, V)))
        zero = np.zeros((2, 4), np.uint8)
        out = O.cdef_frame(Y, U, V, bd, 4, zero, skip)
        assert all((a == b).all() for a, b in zip(out, (Y, U, V)))
        flat = [np.full_like(a, 300 if bd == 10 else 77) for a in (Y, U, V)]
        out = O.cdef_frame(*flat, bd, 6, on, skip)
        assert all((a == b).all() for a, b in zip(out, flat))


def test_output_stays_within_local_range_and_changes_something(O):
    rng = np.random.default_rng(2)
    Y, U, V = _frame(rng, 128, 128, 8)
    st = np.array([[15, 3, 15, 3]] * 4, np.uint8)
    oy, ou, ov = O.cdef_frame(Y, U, V, 8, 3, st, np.zeros((16, 16), np.uint8))
    assert (oy != Y).any() and (ou != U).any()
    pad = np.pad(Y.astype(int), 2, mode="edge")
    win = np.lib.stride_tricks.sliding_window_view(pad, (5, 5))
    assert (oy >= win.min(axis=(2, 3))).all() and (oy <= win.max(axis=(2, 3))).all()


def test_only_unskipped_blocks_change(O):
    rng = np.random.default_rng(3)
    Y, U, V = _frame(rng, 64, 64, 8)
    skip = np.ones((8, 8), np.uint8); skip[2, 5] = 0
    oy, ou, ov = O.cdef_frame(Y, U, V, 8, 4, np.array([[12, 2, 12, 2]], np.uint8), skip)
    ch = np.argwhere(oy != Y)
    assert len(ch) and ch[:, 0].min() >= 16 and ch[:, 0].max() < 24 and ch[:, 1].min() >= 40 and ch[:, 1].max() < 48
    chc = np.argwhere(ou != U)
    assert len(chc) and chc[:, 0].min() >= 8 and chc[:, 0].max() < 12 and chc[:, 1].min() >= 20 and chc[:, 1].max() < 24
