"""GPU parity for K6 (CDEF) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from lf_util import test_image as make_image

pytestmark = pytest.mark.gpu


def _frames(rng, nf, h, w, bd):
    mk = lambda hh, ww: np.stack([make_image(rng, hh, ww, bd) for _ in range(nf)])
    return mk(h, w), mk(h // 2, w // 2), mk(h // 2, w // 2)


@pytest.mark.parametrize("bd", [8, 10])
def test_cdef_random_strengths(ctx, O, bd):
    rng = np.random.default_rng(80 + bd)
    for (h, w), damping in (((128, 192), 3), ((136, 200), 4), ((64, 64), 5), ((8, 8), 6), ((200, 72), 6)):
        nf = 2
        Y, U, V = _frames(rng, nf, h, w, bd)
        nsb = ((h + 63) // 64) * ((w + 63) // 64)
        st = np.stack([rng.integers(0, 16, (nf, nsb)), rng.integers(0, 4, (nf, nsb)), rng.integers(0, 16, (nf, nsb)),
                       rng.integers(0, 4, (nf, nsb))], axis=2).astype(np.uint8)
        st[0, 0, 0] = 255 if nsb > 1 else st[0, 0, 0]      # one superblock with CDEF off
        st[1, -1] = (0, 0, 9, 2)                           # luma strengths zero, chroma on
        skip = (rng.random((nf, h // 8, w // 8)) < 0.25).astype(np.uint8)
        got = ctx.cdef_arrays(Y, U, V, bd, damping, st, skip)
        for f in range(nf):
            exp = O.cdef_frame(Y[f], U[f], V[f], bd, damping, st[f], skip[f])
            for g, e, name in zip(got, exp, "YUV"):
                assert (g[f] == e).all(), (name, (h, w), bd, damping, f, np.argwhere(g[f] != e)[:4])
        assert any((g != s).any() for g, s in zip(got, (Y, U, V))) or h <= 8


def test_cdef_shared_maps_and_1080p(ctx, O):
    rng = np.random.default_rng(83)
    h, w = 1080, 1920
    Y, U, V = _frames(rng, 1, h, w, 8)
    nsb = 17 * 30
    st = np.tile(np.array([[[6, 1, 4, 1]]], np.uint8), (1, nsb, 1))
    skip = np.zeros((1, h // 8, w // 8), np.uint8)
    got = ctx.cdef_arrays(Y, U, V, 8, 5, st, skip)
    exp = O.cdef_frame(Y[0], U[0], V[0], 8, 5, st[0], skip[0])
    for g, e in zip(got, exp):
        assert (g[0] == e).all()
    flat = [np.full_like(a, 99) for a in (Y, U, V)]
    got = ctx.cdef_arrays(*flat, 8, 5, st, skip)
    assert all((g == 99).all() for g in got)     # fixed point at full size
