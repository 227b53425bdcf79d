"""CPU tests of the deblocking oracle (K5): structural properties (no reference vectors exist)."""
import numpy as np

from lf_util import random_mi, test_image as make_image


def test_constant_plane_is_a_fixed_point(O):
    for bd in (8, 10):
        p = np.full((64, 96), 200 << (bd - 8), np.uint8 if bd == 8 else np.uint16)
        mi = np.full((16, 24), O.lf_mi(2, 2, 63, 63), np.uint32)
        assert (O.deblock_plane(p, bd, 0, mi) == p).all()


def test_level_zero_and_non_edges_untouched(O):
    rng = np.random.default_rng(1)
    p = make_image(rng, 64, 128, 8)
    assert (O.deblock_plane(p, 8, 0, np.full((16, 32), O.lf_mi(3, 3, 0, 0), np.uint32)) == p).all()
    # 64x64 transforms, level 40: only samples within 6 of x=64 may change (y edges: none inside 64 rows)
    out = O.deblock_plane(p, 8, 0, np.full((16, 32), O.lf_mi(6, 6, 40, 40), np.uint32))
    ch = np.argwhere(out != p)
    assert len(ch) > 0 and ch[:, 1].min() >= 58 and ch[:, 1].max() <= 69


def test_filter_reach_per_size(O):
    """a step across an edge is smoothed over exactly the reach of the selected filter"""
    p = np.full((16, 64), 100, np.uint8)
    p[:, 32:] = 104
    for log2, reach in ((2, 2), (3, 3), (4, 6), (5, 6)):
        out = O.deblock_plane(p, 8, 0, np.full((4, 16), O.lf_mi(log2, log2, 32, 32), np.uint32), pass_mask=1)
        cols = np.unique(np.argwhere(out != p)[:, 1])
        assert cols.min() >= 32 - reach and cols.max() <= 31 + reach, (log2, cols)
        assert (np.diff(out[0, 24:40].astype(int)) >= 0).all()   # monotone ramp across the edge
    out = O.deblock_plane(p, 8, 1, np.full((4, 16), O.lf_mi(4, 4, 32, 32), np.uint32), pass_mask=1)  # chroma: 6-tap
    cols = np.unique(np.argwhere(out != p)[:, 1])
    assert cols.min() >= 30 and cols.max() <= 33


def test_passes_compose(O):
    rng = np.random.default_rng(2)
    p = make_image(rng, 128, 128, 10)
    mi = random_mi(rng, O, 128, 128, 0)
    both = O.deblock_plane(p, 10, 0, mi)
    seq = O.deblock_plane(O.deblock_plane(p, 10, 0, mi, pass_mask=1), 10, 0, mi, pass_mask=2)
    assert (both == seq).all() and (both != p).any()


def test_transpose_symmetry(O):
    """filtering the transposed picture with transposed mode info, pass order swapped, is NOT the same (pass order
    matters) but single passes are: pass0(P) == pass1(P^T)^T"""
    rng = np.random.default_rng(3)
    p = make_image(rng, 128, 192, 8)
    mi = random_mi(rng, O, 128, 192, 0)
    tw, th, lv, lh, fl = mi & 15, (mi >> 4) & 15, (mi >> 8) & 255, (mi >> 16) & 255, mi >> 24
    flt = (fl & 1) | (((fl >> 2) & 1) << 1) | (((fl >> 1) & 1) << 2)
    mit = (th | (tw << 4) | (lh << 8) | (lv << 16) | (flt << 24)).T.astype(np.uint32)
    a = O.deblock_plane(p, 8, 0, mi, pass_mask=1)
    b = O.deblock_plane(np.ascontiguousarray(p.T), 8, 0, np.ascontiguousarray(mit), pass_mask=2)
    assert (a == b.T).all()


def test_skip_inter_inner_edges(O):
    p = np.full((16, 32), 100, np.uint8)
    p[:, 8:] = 110
    inner = np.full((4, 8), O.lf_mi(3, 3, 30, 30, skip_inter=1, blk_left=0, blk_top=0), np.uint32)
    assert (O.deblock_plane(p, 8, 0, inner) == p).all()
    edge = inner.copy()
    edge[:, 2] = O.lf_mi(3, 3, 30, 30, skip_inter=1, blk_left=1, blk_top=0)
    assert (O.deblock_plane(p, 8, 0, edge) != p).any()
