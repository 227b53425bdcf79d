"""K9 (csrc/entropy_kernels.hip): the GPU tile entropy coder is byte-exact against the C oracle (oracle/av1o_entropy.c) and its
output decodes back to the input symbols with the host decoder (av1-go_amd/host/entropy.cpp)."""
import ctypes as C
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "av1-go_amd", "host", "libav1mi_host.so")


def _oracle_records(O, w, h, key, tile, ly, lu, lv, my=None, muv=None, mvs=None, skip=None):
    pick = lambda a, i: None if a is None else a[i]
    return [O.entropy_encode_frame(w, h, key, tile, ly[i], lu[i], lv[i], pick(my, i), pick(muv, i), pick(mvs, i), pick(skip, i))
            for i in range(ly.shape[0])]


def _assert_same(got, ref):
    assert len(got) == len(ref)
    for i, (g, r) in enumerate(zip(got, ref)):
        assert len(g) == len(r), "frame %d: %d bytes on the GPU, %d from the oracle" % (i, len(g), len(r))
        if g != r:
            k = next(j for j in range(len(g)) if g[j] != r[j])
            raise AssertionError("frame %d differs first at byte %d of %d" % (i, k, len(g)))


@pytest.mark.parametrize("bd,q,tile", [(8, 60, 64), (8, 140, 32), (10, 200, 128), (8, 255, 64)])
def test_key_frames_match_oracle(ctx, O, bd, q, tile):
    w, h, nf = 200, 136, 3                     # ragged in both directions for every tile size
    Y, U, V = synth.frames(w, h, nf, bd)
    o = [O.intra_encode_frame(Y[i], U[i], V[i], bd, 8, q) for i in range(nf)]
    st = lambda k: np.stack([x[k] for x in o])
    ly, lu, lv, my, muv = st("lev_y"), st("lev_u"), st("lev_v"), st("modes_y"), st("modes_uv")
    got = ctx.entropy_encode_arrays(w, h, 1, tile, ly, lu, lv, my, muv)
    _assert_same(got, _oracle_records(O, w, h, 1, tile, ly, lu, lv, my, muv))


@pytest.mark.parametrize("tile", [32, 64])
def test_p_frames_match_oracle(ctx, O, tile):
    w, h = 192, 136
    Y, U, V = synth.frames(w, h, 3, 8)
    k = O.intra_encode_frame(Y[0], U[0], V[0], 8, 8, 100)
    p = [O.inter_encode_frame((Y[i], U[i], V[i]), (k["rec_y"], k["rec_u"], k["rec_v"]), 8, 100 + 60 * i) for i in (1, 2)]
    st = lambda n: np.stack([x[n] for x in p])
    ly, lu, lv, mvs, skip = st("lev_y"), st("lev_u"), st("lev_v"), st("mvs"), st("skip")
    assert mvs.any()
    got = ctx.entropy_encode_arrays(w, h, 0, tile, ly, lu, lv, mvs=mvs, skip=skip)
    _assert_same(got, _oracle_records(O, w, h, 0, tile, ly, lu, lv, mvs=mvs, skip=skip))


def test_adversarial_symbols_match_oracle_and_decode(ctx, O):
    rng = np.random.default_rng(5)
    w, h, nf = 136, 72, 2
    nb = (w // 8) * (h // 8)
    ly = rng.integers(-32768, 32768, (nf, nb, 8, 8)).astype(np.int16)     # full range incl. -32768: 14-bit escapes, carries
    ly[:, ::3] = 0
    ly[:, 1::3] = 0
    ly[:, 1::3, 7, 7] = -1                                                # eob 64 with a single coefficient
    lu = rng.integers(-3, 4, (nf, nb, 4, 4)).astype(np.int16)
    lv = np.zeros((nf, nb, 4, 4), np.int16)
    lv[:, :, 0, 0] = 32767
    my = rng.integers(0, 13, (nf, nb)).astype(np.uint8)
    muv = np.full((nf, nb), 12, np.uint8)
    got = ctx.entropy_encode_arrays(w, h, 1, 64, ly, lu, lv, my, muv)
    _assert_same(got, _oracle_records(O, w, h, 1, 64, ly, lu, lv, my, muv))
    mvs = rng.integers(-32768, 32768, (nf, nb, 2)).astype(np.int16)       # wrapping differences, class 15
    mvs[:, :5] = 0
    skip = (rng.random((nf, nb)) < 0.5).astype(np.uint8)
    for a in (ly, lu, lv):
        a[skip == 1] = 0
    got = ctx.entropy_encode_arrays(w, h, 0, 32, ly, lu, lv, mvs=mvs, skip=skip)
    _assert_same(got, _oracle_records(O, w, h, 0, 32, ly, lu, lv, mvs=mvs, skip=skip))
    # and the stream is complete: the host decoder returns the symbols
    lib = C.CDLL(HOST)
    P = C.c_void_p
    lib.av1mi_host_entropy_decode.argtypes = [P, C.c_longlong, C.c_int, C.c_int, C.c_int, P, P, P, P, P, P, P]
    vp = lambda a: a.ctypes.data_as(P)
    for i in range(nf):
        rec = np.frombuffer(got[i], np.uint8).copy()
        d = [np.zeros((nb, 8, 8), np.int16), np.zeros((nb, 4, 4), np.int16), np.zeros((nb, 4, 4), np.int16), np.zeros(nb, np.uint8),
             np.zeros(nb, np.uint8), np.zeros((nb, 2), np.int16), np.zeros(nb, np.uint8)]
        assert lib.av1mi_host_entropy_decode(vp(rec), rec.size, w, h, 0, *[vp(a) for a in d]) == 0
        assert np.array_equal(d[0], ly[i]) and np.array_equal(d[1], lu[i]) and np.array_equal(d[2], lv[i])
        assert np.array_equal(d[5], mvs[i]) and np.array_equal(d[6], skip[i])


def test_skewed_stream_runs_of_ff_bytes(ctx, O):
    # long 0xFF runs + late carries exercise the byte hold / run counter of the GPU coder
    w, h = 256, 256
    nb = (w // 8) * (h // 8)
    ly = np.zeros((1, nb, 8, 8), np.int16)
    ly[:, :, 0, 0] = 1
    ly[:, 5::7, 0, 1] = -2
    z = np.zeros((1, nb, 4, 4), np.int16)
    my = np.zeros((1, nb), np.uint8)
    for tile in (64, 128):
        got = ctx.entropy_encode_arrays(w, h, 1, tile, ly, z, z, my, my)
        _assert_same(got, _oracle_records(O, w, h, 1, tile, ly, z, z, my, my))


def test_full_hd_frame_matches_oracle(ctx, O):
    w, h = 1920, 1080
    Y, U, V = synth.frames(w, h, 1, 8)
    o = O.intra_encode_frame(Y[0], U[0], V[0], 8, 8, 120)
    a = [o[k][None] for k in ("lev_y", "lev_u", "lev_v", "modes_y", "modes_uv")]
    got = ctx.entropy_encode_arrays(w, h, 1, 64, *a)
    _assert_same(got, _oracle_records(O, w, h, 1, 64, *a))


def test_capacity_and_argument_errors(ctx, av1mi):
    w, h = 64, 64
    ly = np.ones((1, 64, 8, 8), np.int16)
    z = np.zeros((1, 64, 4, 4), np.int16)
    m = np.zeros((1, 64), np.uint8)
    with pytest.raises(av1mi.Av1miError, match="out_cap 16 <"):
        ctx.entropy_encode_arrays(w, h, 1, 64, ly, z, z, m, m, out_cap=16)
    with pytest.raises(av1mi.Av1miError, match="entropy tile 48 not supported"):
        ctx.entropy_encode_arrays(w, h, 1, 48, ly, z, z, m, m)
    with pytest.raises(av1mi.Av1miError, match="null device pointer"):
        ctx.entropy_encode_arrays(w, h, 0, 64, ly, z, z, m, m)       # P frame without vectors / skip flags
