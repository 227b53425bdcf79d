"""GPU end-to-end: the whole segment pipeline (intra coding + deblock + CDEF + loop restoration, 8 launches) against
the oracle chain, frame by frame, bit-exact; plus the quality sanity the metric needs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_chain(O, pipe, f):
    Y, U, V = pipe.src
    r = O.intra_encode_frame(Y[f], U[f], V[f], pipe.bd, pipe.bs, pipe.qindex)
    dbl = [O.deblock_plane(r["rec_y"], pipe.bd, 0, pipe.mi_y), O.deblock_plane(r["rec_u"], pipe.bd, 1, pipe.mi_c),
           O.deblock_plane(r["rec_v"], pipe.bd, 1, pipe.mi_c)]
    cdef = O.cdef_frame(dbl[0], dbl[1], dbl[2], pipe.bd, pipe.cdef_damping, pipe.cdef_sb, pipe.cdef_skip)
    out = [O.lr_plane(cdef[0], dbl[0], pipe.bd, 0, pipe.lr_unit, pipe.lr_units_y),
           O.lr_plane(cdef[1], dbl[1], pipe.bd, 1, pipe.lr_unit, pipe.lr_units_c),
           O.lr_plane(cdef[2], dbl[2], pipe.bd, 1, pipe.lr_unit, pipe.lr_units_c)]
    return r, dbl, cdef, out


@pytest.mark.parametrize("w,h,bd,bs", [(320, 192, 8, 8), (256, 144, 10, 8), (192, 128, 10, 16)])
def test_segment_pipeline_matches_oracle_chain(ctx, O, w, h, bd, bs):
    import pipeline
    pipe = pipeline.IntraPipeline(ctx, w, h, bd, 3, 100, first_frame=2, block_size=bs)
    pipe.step()
    for f in (0, 2):
        got = pipe.download(f)
        r, dbl, cdef, out = _oracle_chain(O, pipe, f)
        for i, p in enumerate("yuv"):
            assert (got["rec_" + p] == r["rec_" + p]).all(), ("rec", p, f)
            assert (got["dbl_" + p] == dbl[i]).all(), ("dbl", p, f)
            assert (got["cdef_" + p] == cdef[i]).all(), ("cdef", p, f)
            assert (got["out_" + p] == out[i]).all(), ("out", p, f)
    Y = pipe.src[0][0].astype(np.float64)
    psnr = lambda a: 10 * np.log10(((1 << bd) - 1) ** 2 / np.mean((a.astype(np.float64) - Y) ** 2))
    g = pipe.download(0)
    assert psnr(g["out_y"]) > 28 and psnr(g["out_y"]) > psnr(g["rec_y"]) - 1.0   # the filter chain must not wreck the picture
    pipe.close()


def test_4k_10bit_full_chain_one_frame(ctx, O):
    """BASELINE config 4 size (3840x2160 10-bit): the whole chain on one frame equals the oracle chain bit for bit."""
    import pipeline
    pipe = pipeline.IntraPipeline(ctx, 3840, 2160, 10, 1, 128, first_frame=1, block_size=8)
    pipe.step()
    got = pipe.download(0)
    r, dbl, cdef, out = _oracle_chain(O, pipe, 0)
    for i, p in enumerate("yuv"):
        assert (got["rec_" + p] == r["rec_" + p]).all() and (got["dbl_" + p] == dbl[i]).all()
        assert (got["cdef_" + p] == cdef[i]).all() and (got["out_" + p] == out[i]).all()
    pipe.close()
