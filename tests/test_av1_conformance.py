"""The AV1 streams the host bitstream writer (av1-go_amd/host/av1_bitstream.cpp) emits, decoded by dav1d — a conformant
third-party decoder that ships inside this image (tests/dav1d_ref.py) — must reproduce the ORACLE's reconstruction bit for
bit, stage by stage: Dav1dSettings.inloop_filters switches deblocking / CDEF / loop restoration off at decode time, so the
pre-filter reconstruction (dequantiser + inverse transforms + intra prediction, rows K2 K3 K8 of SURVEY.md §8a), the
deblocked (K5), the CDEF (K6) and the restored (K7) planes are each compared on their own.

This is the pin of the oracle (SURVEY.md §8c asked for golden vectors; the reference holds none): the oracle's normative
stages == dav1d 1.5.3 on the same symbols, and the GPU tests carry that to the kernels (GPU == oracle, tests/test_gpu_*).
CPU only; skipped when the bundled libavif with dav1d is absent."""
import numpy as np
import pytest

import dav1d_ref as D

pytestmark = pytest.mark.skipif(not D.available(), reason="no dav1d in this image (pillow.libs/libavif)")


def _chain(O, P, Y, U, V, bd, q):
    """the oracle's key-frame chain with the pipeline's parameter policy; returns symbols + the four stages"""
    h, w = Y.shape
    r = O.intra_encode_frame(Y, U, V, bd, 8, q)
    acq = O.ac_q(q, bd)
    lvl = P.lf_level_from_q(acq, bd)
    mi_y = np.full((h // 4, w // 4), P.lf_mi_word(3, 3, lvl, lvl), np.uint32)
    mi_c = np.full((h // 8, w // 8), P.lf_mi_word(2, 2, lvl, lvl), np.uint32)
    dbl = [O.deblock_plane(r["rec_y"], bd, 0, mi_y), O.deblock_plane(r["rec_u"], bd, 1, mi_c), O.deblock_plane(r["rec_v"], bd, 1, mi_c)]
    nsb = ((h + 63) // 64) * ((w + 63) // 64)
    st = P.cdef_strength_from_q(acq, bd)
    damping = 3 + (acq >> (bd - 8) > 100) + (acq >> (bd - 8) > 300)
    cdef = O.cdef_frame(dbl[0], dbl[1], dbl[2], bd, damping, np.tile(st, (nsb, 1)), np.zeros((h // 8, w // 8), np.uint8))
    ur = lambda n: max(1, (n + 32) // 64)
    uy = np.tile(np.array(P.WIENER_DEFAULT_LUMA, np.int8), (ur(h), ur(w), 1))
    uc = np.tile(np.array(P.WIENER_DEFAULT_CHROMA, np.int8), (ur(h // 2), ur(w // 2), 1))
    out = [O.lr_plane(cdef[0], dbl[0], bd, 0, 64, uy), O.lr_plane(cdef[1], dbl[1], bd, 1, 64, uc), O.lr_plane(cdef[2], dbl[2], bd, 1, 64, uc)]
    hdr = dict(lf_level=(lvl,) * 4, cdef_damping=damping, cdef_y=(int(st[0]) << 2 | int(st[1]),), cdef_uv=(int(st[2]) << 2 | int(st[3]),),
               lr_type=(1, 1, 1), lr_units=(uy, uc, uc))
    return r, hdr, [[r["rec_y"], r["rec_u"], r["rec_v"]], dbl, list(cdef), out]


@pytest.mark.parametrize("w,h,bd,q", [(64, 64, 8, 128), (192, 128, 8, 128), (200, 136, 10, 60), (328, 184, 8, 200), (72, 72, 10, 230),
                                      (128, 320, 8, 15)])
def test_key_frame_stream_decodes_to_the_oracle_chain(O, w, h, bd, q):
    import av1stream
    import pipeline as P
    import synth
    Y, U, V = synth.frames(w, h, 1, bd, 3)
    r, hdr, stages = _chain(O, P, Y[0], U[0], V[0], bd, q)
    tu = av1stream.temporal_unit(w, h, bd, q, y_mode=r["modes_y"], uv_mode=r["modes_uv"], lev_y=r["lev_y"], lev_u=r["lev_u"], lev_v=r["lev_v"],
                                 **hdr)
    for name, flt, ref in (("reconstruction", 0, stages[0]), ("deblocked", D.INLOOP_DEBLOCK, stages[1]),
                           ("cdef", D.INLOOP_DEBLOCK | D.INLOOP_CDEF, stages[2]), ("restored", D.INLOOP_ALL, stages[3])):
        got = D.decode(tu, inloop_filters=flt)
        assert len(got) == 1
        for i in range(3):
            assert got[0][i].shape == ref[i].shape and (got[0][i] == ref[i]).all(), "%s plane %d differs from dav1d" % (name, i)


def test_threads_and_cdf_update_flag_do_not_change_the_pixels(O):
    import av1stream
    import pipeline as P
    import synth
    w, h, bd, q = 192, 128, 8, 100
    Y, U, V = synth.frames(w, h, 1, bd, 5)
    r, hdr, stages = _chain(O, P, Y[0], U[0], V[0], bd, q)
    sym = dict(y_mode=r["modes_y"], uv_mode=r["modes_uv"], lev_y=r["lev_y"], lev_u=r["lev_u"], lev_v=r["lev_v"])
    a = av1stream.temporal_unit(w, h, bd, q, threads=1, **sym, **hdr)
    assert a == av1stream.temporal_unit(w, h, bd, q, threads=4, **sym, **hdr)
    b = av1stream.temporal_unit(w, h, bd, q, disable_cdf_update=1, **sym, **hdr)
    c = av1stream.temporal_unit(w, h, bd, q, reduced_tx_set=1, **sym, **hdr)
    assert len(b) > len(a)          # adaptation pays
    for tu in (b, c):
        got = D.decode(tu)[0]
        assert all((got[i] == stages[3][i]).all() for i in range(3))


def _gop(O, P, w, h, bd, q, nframes, first=3):
    """oracle closed-GOP chain (key + P frames, every frame through deblock + CDEF + Wiener LR) and its AV1 stream"""
    import av1stream
    import synth
    Y, U, V = synth.frames(w, h, nframes, bd, first)
    stream, refs, ref, hdr = b"", [], None, None
    for t in range(nframes):
        if t == 0:
            r, hdr, stages = _chain(O, P, Y[0], U[0], V[0], bd, q)
            stream += av1stream.temporal_unit(w, h, bd, q, y_mode=r["modes_y"], uv_mode=r["modes_uv"], lev_y=r["lev_y"], lev_u=r["lev_u"],
                                              lev_v=r["lev_v"], **hdr)
            ref = stages[3]
        else:
            r = O.inter_encode_frame((Y[t], U[t], V[t]), ref, bd, q, 8)
            stream += av1stream.temporal_unit(w, h, bd, q, frame_type=1, with_sequence_header=False, mv=r["mvs"], skip=r["skip"],
                                              lev_y=r["lev_y"], lev_u=r["lev_u"], lev_v=r["lev_v"], **hdr)
            acq = O.ac_q(q, bd)
            lvl = hdr["lf_level"][0]
            mi_y = np.full((h // 4, w // 4), P.lf_mi_word(3, 3, lvl, lvl), np.uint32)
            mi_c = np.full((h // 8, w // 8), P.lf_mi_word(2, 2, lvl, lvl), np.uint32)
            dbl = [O.deblock_plane(r["rec_y"], bd, 0, mi_y), O.deblock_plane(r["rec_u"], bd, 1, mi_c), O.deblock_plane(r["rec_v"], bd, 1, mi_c)]
            nsb = ((h + 63) // 64) * ((w + 63) // 64)
            cdef = O.cdef_frame(dbl[0], dbl[1], dbl[2], bd, hdr["cdef_damping"], np.tile(P.cdef_strength_from_q(acq, bd), (nsb, 1)),
                                r["skip"].reshape(h // 8, w // 8))
            uy, uc = hdr["lr_units"][0], hdr["lr_units"][1]
            ref = [O.lr_plane(cdef[0], dbl[0], bd, 0, 64, uy), O.lr_plane(cdef[1], dbl[1], bd, 1, 64, uc), O.lr_plane(cdef[2], dbl[2], bd, 1, 64, uc)]
        refs.append(ref)
    return stream, refs


@pytest.mark.parametrize("w,h,bd,q,n", [(64, 64, 8, 128, 2), (192, 128, 8, 128, 4), (200, 136, 10, 60, 3), (328, 184, 8, 200, 3)])
def test_closed_gop_stream_decodes_to_the_oracle_chain(O, w, h, bd, q, n):
    """P frames: sub-pel motion compensation (K4, all of a frame's vectors), inter residual, the filters on inter frames, a P
    frame that references a P frame — the whole decoded sequence equals the oracle's reference frames."""
    import pipeline as P
    stream, refs = _gop(O, P, w, h, bd, q, n)
    got = D.decode(stream)
    assert len(got) == n
    for t in range(n):
        for i in range(3):
            assert (got[t][i] == refs[t][i]).all(), "frame %d plane %d differs from dav1d" % (t, i)
