"""GPU parity for K5 (deblocking) — one fused two-pass HIP kernel vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from lf_util import random_mi, test_image as make_image

pytestmark = pytest.mark.gpu


def _run(ctx, p, bd, is_chroma, mi, sharp):
    h, w = p.shape
    d_src, d_mi = ctx.to_device(p), ctx.to_device(mi)
    d_dst = ctx.to_device(np.zeros_like(p))
    ctx.deblock_plane(d_src, w, d_dst, w, w, h, bd, is_chroma, d_mi, mi.shape[1], sharp)
    out = d_dst.download(p.shape, p.dtype)
    for b in (d_src, d_mi, d_dst):
        b.free()
    return out


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("is_chroma", [0, 1])
def test_deblock_random_mode_info(ctx, O, bd, is_chroma):
    rng = np.random.default_rng(50 + bd + is_chroma)
    for (h, w), sharp in (((128, 192), 0), ((200, 328), 3), ((64, 64), 7), ((68, 132), 5), ((4, 8), 0)):
        p = make_image(rng, h, w, bd)
        mi = random_mi(rng, O, h, w, is_chroma)
        exp = O.deblock_plane(p, bd, is_chroma, mi, sharp)
        got = _run(ctx, p, bd, is_chroma, mi, sharp)
        assert (got == exp).all(), ((h, w), sharp, np.argwhere(got != exp)[:5])
        assert (exp != p).any() or h <= 68


def test_deblock_uniform_sizes_strong_levels(ctx, O):
    rng = np.random.default_rng(60)
    for log2 in (2, 3, 4, 5, 6):
        p = make_image(rng, 192, 256, 8)
        mi = np.full((48, 64), O.lf_mi(log2, log2, 63, 50), np.uint32)
        assert (_run(ctx, p, 8, 0, mi, 0) == O.deblock_plane(p, 8, 0, mi, 0)).all(), log2


def test_deblock_1080p_luma_properties(ctx, O):
    """full BASELINE size: constant picture is a fixed point; a textured one equals the oracle on sampled tiles"""
    rng = np.random.default_rng(61)
    h, w = 1088, 1920
    mi = np.full((h // 4, w // 4), O.lf_mi(3, 3, 32, 32), np.uint32)
    flat = np.full((h, w), 77, np.uint8)
    assert (_run(ctx, flat, 8, 0, mi, 0) == flat).all()
    p = make_image(rng, h, w, 8)
    got = _run(ctx, p, 8, 0, mi, 0)
    exp = O.deblock_plane(p, 8, 0, mi, 0)
    assert (got == exp).all()
