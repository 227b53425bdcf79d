"""GPU parity for the inter (P-frame) pipeline (BASELINE config 3): vectors, skip flags, levels and reconstruction of
whole frames equal the CPU oracle's inter encoder loop bit for bit (so MC is inside the +-1 LSB north_star allows: 0)."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
KEYS = ("mvs", "skip", "lev_y", "lev_u", "lev_v", "rec_y", "rec_u", "rec_v")


def _ref_from_intra(O, Y, U, V, bd, q):
    k = O.intra_encode_frame(Y, U, V, bd, 8, q)
    return k["rec_y"], k["rec_u"], k["rec_v"]


@pytest.mark.parametrize("bd", [8, 10])
def test_inter_pipe_matches_oracle(ctx, O, bd):
    for (w, h, q, rng_) in ((192, 128, 100, 8), (200, 104, 40, 4), (64, 64, 200, 15), (72, 40, 128, 0)):
        nf = 3
        Y, U, V = synth.frames(w, h, nf + 1, bd, first=3)
        refs = [_ref_from_intra(O, Y[f], U[f], V[f], bd, q) for f in range(nf)]       # reference = coded previous frame
        src = (Y[1:], U[1:], V[1:])
        ref = tuple(np.stack([r[i] for r in refs]) for i in range(3))
        got = ctx.inter_encode_arrays(src, ref, bd, q, rng_)
        for f in range(nf):
            exp = O.inter_encode_frame((src[0][f], src[1][f], src[2][f]), refs[f], bd, q, rng_)
            for k in KEYS:
                assert (got[k][f] == exp[k]).all(), (k, (w, h), bd, q, rng_, f, np.argwhere(got[k][f] != exp[k])[:4])
        if rng_ >= 4 and w >= 128:
            # the synthetic sequence moves by (1.25, 0.75) px per frame: most vectors are (10, 6) in 1/8 px
            mv = got["mvs"][0]
            assert np.mean((mv[:, 0] == 10) & (mv[:, 1] == 6)) > 0.5


def test_inter_static_scene_is_all_skip(ctx):
    """size-independent property: identical source and reference -> zero vectors, no levels, skip everywhere"""
    rng = np.random.default_rng(5)
    Y = rng.integers(16, 236, (2, 128, 192)).astype(np.uint8)
    U = rng.integers(16, 236, (2, 64, 96)).astype(np.uint8)
    V = rng.integers(16, 236, (2, 64, 96)).astype(np.uint8)
    got = ctx.inter_encode_arrays((Y, U, V), (Y, U, V), 8, 128, 8)
    assert (got["mvs"] == 0).all() and (got["skip"] == 1).all() and not got["lev_y"].any()
    assert (got["rec_y"] == Y).all() and (got["rec_u"] == U).all()


def test_inter_1080p_frame(ctx, O):
    w, h, q = 1920, 1080, 128
    Y, U, V = synth.frames(w, h, 2, 8, first=7)
    ref = _ref_from_intra(O, Y[0], U[0], V[0], 8, q)
    got = ctx.inter_encode_arrays((Y[1:], U[1:], V[1:]), tuple(r[None] for r in ref), 8, q, 8)
    exp = O.inter_encode_frame((Y[1], U[1], V[1]), ref, 8, q, 8)
    for k in KEYS:
        assert (got[k][0] == exp[k]).all(), k


def test_inter_4k_10bit_frame(ctx, O):
    """BASELINE configs[3] size, the bench's default workload: one 3840x2160 10-bit P frame — integer search, sub-pel
    refinement, vectors, skip flags, levels and reconstruction — equals the oracle's inter encoder loop bit for bit"""
    w, h, q = 3840, 2160, 128
    Y, U, V = synth.frames(w, h, 2, 10, first=11)
    ref = _ref_from_intra(O, Y[0], U[0], V[0], 10, q)
    got = ctx.inter_encode_arrays((Y[1:], U[1:], V[1:]), tuple(r[None] for r in ref), 10, q, 8)
    exp = O.inter_encode_frame((Y[1], U[1], V[1]), ref, 10, q, 8)
    for k in KEYS:
        assert (got[k][0] == exp[k]).all(), (k, np.argwhere(got[k][0] != exp[k])[:4])
    mv = got["mvs"][0]
    assert np.mean((mv[:, 0] == 10) & (mv[:, 1] == 6)) > 0.5      # the clip moves by (1.25, 0.75) px per frame
