"""The op-stream formulation of the tile syntax (av1-go_amd/csrc/av1_ops.hpp: tokenize every block in parallel with all contexts
taken from neighbour data, then one serial range-coder pass per tile) — the code the GPU tile entropy coder runs — compiled for
the host (host/av1_opstream.cpp) must produce EXACTLY the bytes of the block-sequential writer (host/av1_bitstream.cpp, which
dav1d verifies in test_av1_conformance.py).  CPU only."""
import numpy as np
import pytest


def _key(O, P, w, h, bd, q, first=3):
    import synth
    from test_av1_conformance import _chain
    Y, U, V = synth.frames(w, h, 1, bd, first)
    r, hdr, _ = _chain(O, P, Y[0], U[0], V[0], bd, q)
    return dict(y_mode=r["modes_y"], uv_mode=r["modes_uv"], lev_y=r["lev_y"], lev_u=r["lev_u"], lev_v=r["lev_v"]), hdr


@pytest.mark.parametrize("w,h,bd,q", [(64, 64, 8, 128), (192, 128, 8, 128), (200, 136, 10, 60), (328, 184, 8, 200), (72, 72, 10, 230),
                                      (128, 320, 8, 15), (1920, 1080, 8, 128)])
def test_key_frame_bytes_equal_the_sequential_writer(O, w, h, bd, q):
    import av1stream
    import pipeline as P
    sym, hdr = _key(O, P, w, h, bd, q)
    a = av1stream.temporal_unit(w, h, bd, q, **sym, **hdr)
    b = av1stream.temporal_unit(w, h, bd, q, opstream=True, **sym, **hdr)
    assert a == b


@pytest.mark.parametrize("w,h,bd,q,n", [(64, 64, 8, 128, 2), (192, 128, 8, 128, 4), (200, 136, 10, 60, 3), (328, 184, 8, 200, 3), (640, 360, 8, 128, 3)])
def test_inter_frame_bytes_equal_the_sequential_writer(O, w, h, bd, q, n):
    """P frames of the oracle's closed-GOP chain: the MV prediction list, the mode contexts and the NEWMV flags come out of two
    parallel passes over neighbour data instead of a sequential walk"""
    import av1stream
    import pipeline as P
    import synth
    from test_av1_conformance import _chain, _filters
    Y, U, V = synth.frames(w, h, n, bd, 3)
    r, hdr, st = _chain(O, P, Y[0], U[0], V[0], bd, q)
    ref = st[3]
    for t in range(1, n):
        r = O.inter_encode_frame((Y[t], U[t], V[t]), ref, bd, q, 8)
        hp, st = _filters(O, P, r, bd, q, 1, w, h, r["skip"].reshape(h // 8, w // 8), (Y[t], U[t], V[t]))
        ref = st[2]
        sym = dict(frame_type=1, with_sequence_header=False, mv=r["mvs"], skip=r["skip"], lev_y=r["lev_y"], lev_u=r["lev_u"], lev_v=r["lev_v"])
        assert av1stream.temporal_unit(w, h, bd, q, **sym, **hp) == av1stream.temporal_unit(w, h, bd, q, opstream=True, **sym, **hp), "frame %d" % t


def test_random_vectors_and_escape_levels(O):
    """arbitrary vectors (clustered + far outliers), skip flags and levels large enough for the Golomb escape"""
    import av1stream
    rng = np.random.default_rng(5)
    w, h, bd, q = 192, 136, 10, 90
    nb = (w // 8) * (h // 8)
    base = rng.integers(-6, 7, (4, 2)) * 2
    mv = base[rng.integers(0, 4, nb)].astype(np.int16)
    far = rng.random(nb) < 0.2
    mv[far] = (rng.integers(-300, 301, (int(far.sum()), 2)) * 2).astype(np.int16)
    mv[rng.random(nb) < 0.1] = 0
    skip = (rng.random(nb) < 0.25).astype(np.uint8)

    def levels(k):
        a = np.zeros((nb, k), np.int16)
        m = rng.random((nb, k)) < 0.3 * np.linspace(1, 0.05, k)[None, :]
        a[m] = rng.integers(-4, 5, int(m.sum()))
        big = rng.random((nb, k)) < 0.01
        a[big] = rng.integers(-900, 901, int(big.sum()))
        return a
    sym = dict(frame_type=1, with_sequence_header=False, mv=mv, skip=skip, lev_y=levels(64), lev_u=levels(16), lev_v=levels(16), lf_level=(5, 5, 5, 5))
    assert av1stream.temporal_unit(w, h, bd, q, **sym) == av1stream.temporal_unit(w, h, bd, q, opstream=True, **sym)


def test_outside_the_tool_set_is_refused(O):
    import av1stream
    w, h = 64, 64
    nb = 64
    z = dict(y_mode=np.zeros(nb, np.uint8), uv_mode=np.zeros(nb, np.uint8), lev_y=np.zeros((nb, 64), np.int16), lev_u=np.zeros((nb, 16), np.int16),
             lev_v=np.zeros((nb, 16), np.int16))
    with pytest.raises(ValueError):
        av1stream.temporal_unit(w, h, 8, 100, opstream=True, angle_y=np.ones(nb, np.int8), **z)
    with pytest.raises(ValueError):
        av1stream.temporal_unit(w, h, 8, 100, opstream=True, reduced_tx_set=1, **z)


def test_lazy_byte_output_equals_the_eager_coder_including_late_carries():
    """The op-stream coder keeps sixteen finished bits back and stores two bytes at a time; a carry into bytes already stored
    then needs sixteen 1 bits in a row.  Millions of random interval updates make that happen a few times: the bytes must
    equal the host writer's coder (carry checked after every symbol) and the rare path must have run."""
    import ctypes
    import av1stream
    L = av1stream.lib()
    f = L.av1mi_host_coder_selftest
    f.restype = ctypes.c_longlong
    f.argtypes = [ctypes.c_uint64, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]
    total = 0
    for seed in range(4):
        c = ctypes.c_longlong(0)
        assert f(seed, 2000000, ctypes.byref(c)) == 0, "seed %d" % seed
        total += c.value
    assert total >= 1


@pytest.mark.parametrize("w,h,bd,q,lr", [(64, 64, 8, 128, False), (192, 128, 8, 60, True), (256, 168, 10, 23, True), (128, 104, 8, 200, False),
                                         (320, 192, 10, 128, True), (96, 136, 8, 90, True), (224, 64, 10, 150, False), (1440, 1080, 8, 128, True),
                                         (1920, 1080, 8, 128, True)])
def test_key_frames_in_32x32_blocks_twin_equals_the_block_writer(O, w, h, bd, q, lr):
    """key frames of a key_block_size 32 session: 32x32 blocks over the complete superblock rows, 8x8 blocks in a last
    partial row.  The serial tile tokenizer of csrc/av1_ops32.hpp + the unchanged chains and range coder (host/av1_opstream.cpp: what
    the GPU runs) must give the bytes of the general block writer for the same symbols, and dav1d must decode them to the oracle's
    reconstruction."""
    import av1stream
    import synth
    import av1_blocks as B
    import dav1d_ref as D
    Y, U, V = (a[0] for a in synth.frames(w, h, 1, bd, 3))
    hA = h // 64 * 64
    a = O.intra_encode_frame(Y[:hA], U[:hA // 2], V[:hA // 2], bd, 32, q)
    b = O.intra_encode_frame(Y[hA:], U[hA // 2:], V[hA // 2:], bd, 8, q) if hA < h else None
    # the general block writer's view of the frame
    lay = B.Layout(w, h)
    w32, w8 = w // 32, w // 8

    def chooser(r, c, bsize, allowed):
        return B.uniform_chooser(9 if r * 4 < hA else 3)(r, c, bsize, allowed)
    parts, tree = B.build_tree(lay, chooser)
    blocks = []
    for r, c, bsize, tb in tree:
        if r * 4 < hA:
            o, i = a, (r // 8) * w32 + c // 8
        else:
            o, i = b, ((r * 4 - hA) // 8) * w8 + c // 2
        blocks.append(dict(r=r, c=c, bsize=bsize, tile=tb, skip=0, is_inter=0, y_mode=int(o["modes_y"][i]), uv_mode=int(o["modes_uv"][i]), angle_y=0, angle_uv=0,
                           cfl=(0, 0), tx_depth=0, filt=0, mv=(0, 0), tx=B.max_tx_rect(bsize), tx_types=[0], levels=[[o["lev_y"][i]], [o["lev_u"][i]], [o["lev_v"][i]]]))
    hdr = {}
    if lr:
        ur = lambda n: max(1, (n + 32) // 64)
        uy = np.tile(np.array([1, 3, -7, 15, 3, -7, 15, 0], np.int8), (ur(h), ur(w), 1))
        uc = np.tile(np.array([1, 0, -7, 15, 0, -7, 15, 0], np.int8), (ur(h // 2), ur(w // 2), 1))
        hdr = dict(lr_type=(1, 0, 1), lr_units=(uy, uc, uc), lf_level=(9, 7, 5, 5), cdef_y=(5,), cdef_uv=(4,), cdef_damping=4)
    ref = B.encode(lay, bd, q, parts, blocks, **hdr)
    # the session's layout of the same symbols
    nb = (h // 8) * w8
    ym, uvm = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8)
    ly, lu, lv = np.zeros(h * w, np.int16), np.zeros(h * w // 4, np.int16), np.zeros(h * w // 4, np.int16)
    nA = (hA // 32) * w32
    ym[:nA], uvm[:nA] = a["modes_y"], a["modes_uv"]
    ly[:hA * w], lu[:hA * w // 4], lv[:hA * w // 4] = a["lev_y"].reshape(-1), a["lev_u"].reshape(-1), a["lev_v"].reshape(-1)
    if b is not None:
        o8 = (hA // 8) * w8
        ym[o8:], uvm[o8:] = b["modes_y"], b["modes_uv"]
        ly[hA * w:], lu[hA * w // 4:], lv[hA * w // 4:] = b["lev_y"].reshape(-1), b["lev_u"].reshape(-1), b["lev_v"].reshape(-1)
    twin = av1stream.temporal_unit(w, h, bd, q, opstream=True, key_rows32=hA, y_mode=ym, uv_mode=uvm, lev_y=ly, lev_u=lu, lev_v=lv, **hdr)
    assert twin == ref
    if D.available():
        pic = D.decode(twin, inloop_filters=0)[0]
        rec = np.concatenate([a["rec_y"]] + ([b["rec_y"]] if b is not None else []))
        assert np.array_equal(np.asarray(pic[0]), rec)
