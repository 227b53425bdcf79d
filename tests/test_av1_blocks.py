"""The dav1d pin of the oracle for EVERY size north_star names: the general block-structured writer
(av1-go_amd/host/av1_blockstream.cpp, include/av1mi_host.h av1mi_obu_write_blocks_temporal_unit) codes arbitrary symbols over
random partition trees — block sizes 4x4..64x64 incl. the 1:2 / 2:1 / 1:4 / 4:1 shapes, all ten partition types, TX_MODE_LARGEST and
TX_MODE_SELECT (tx_depth, txfm_split), every transform size with every transform type its set allows (FLIPADST, IDTX, the 1-D
classes, reduced_tx_set), all 13 intra modes x angle deltas on every transform size (edge filter / upsampling / top-right and
bottom-left availability of each partition shape), inter blocks of every size with the regular / smooth / sharp / bilinear filters
at eighth-sample vectors — dav1d 1.5.3 decodes the stream with its in-loop filters off, and the result must equal the ORACLE's
primitives (av1o_inv_txfm2d_add, av1o_dequantize, av1o_intra_predict, av1o_cfl_predict, av1o_mc_block) applied to the same
symbols in decoding order (tests/av1_blocks.py).  Rows K2, K3, K4, K8 of SURVEY.md §8a; the GPU kernels are compared with the
same oracle functions at all of these sizes in tests/test_gpu_txfm.py / test_gpu_intra.py / test_gpu_mc.py.  CPU only."""
import numpy as np
import pytest

import av1_blocks as B
import dav1d_ref as D

pytestmark = pytest.mark.skipif(not D.available(), reason="no dav1d in this image (pillow.libs/libavif)")

SIZES = [b for b in range(22) if b not in (13, 14, 15)]      # every BLOCK_* up to 64x64
SIZE_IDS = [B.BLOCK_NAMES[b] for b in SIZES]


def _same(got, rec, what):
    for i in range(3):
        assert got[i].shape == rec[i].shape
        assert (got[i] == rec[i]).all(), "%s: plane %d differs from dav1d (first at %s)" % (what, i, np.argwhere(got[i] != rec[i])[0])


def _key(O, rng, lay, bd, q, **kw):
    sel, red = kw.get("tx_mode_select", 0), kw.get("reduced_tx_set", 0)
    parts, blocks = B.random_frame(O, rng, lay, bd, q, **kw)
    tu = B.encode(lay, bd, q, parts, blocks, tx_mode_select=sel, reduced_tx_set=red)
    return tu, blocks, B.decode(O, lay, bd, q, blocks, red)


@pytest.mark.parametrize("w,h,bd,q,tiles,sel,red,seed", [
    (64, 64, 8, 100, -1, 0, 0, 1), (72, 88, 10, 128, -1, 0, 0, 2), (192, 136, 8, 200, 0, 0, 0, 3), (136, 200, 10, 60, -1, 0, 0, 4),
    (128, 64, 10, 30, 0, 1, 0, 5), (72, 88, 8, 128, 0, 1, 0, 6), (192, 136, 10, 200, -1, 1, 1, 7), (136, 200, 8, 60, 0, 1, 0, 8),
    (64, 128, 10, 230, -1, 0, 1, 9), (200, 72, 8, 15, -1, 1, 0, 10)])
def test_random_partition_trees_of_intra_blocks(O, w, h, bd, q, tiles, sel, red, seed):
    """key frames: random trees (every partition type, blocks straddling the frame edge, one-superblock tiles and one tile for the
    whole frame: prediction edges across superblocks), random modes / angles / CfL / transform types / tx_depth"""
    rng = np.random.default_rng(seed)
    lay = B.Layout(w, h, tiles, tiles)
    tu, blocks, rec = _key(O, rng, lay, bd, q, tx_mode_select=sel, reduced_tx_set=red)
    assert len({b["bsize"] for b in blocks}) >= (8 if lay.sb_cols * lay.sb_rows >= 4 else 1)
    _same(D.decode(tu, inloop_filters=0)[0], rec, "key frame")


MODE_ANGLES = [(m, a) for m in range(1, 9) for a in range(-3, 4)] + [(m, 0) for m in (0, 9, 10, 11, 12)]


@pytest.mark.parametrize("bsize", SIZES, ids=SIZE_IDS)
def test_every_block_size_every_intra_mode_and_angle(O, bsize):
    """K3 on every transform size: frames cut into blocks of ONE size (TX_MODE_LARGEST: the transform, and with it the prediction
    block, is the block), luma modes x angle deltas dealt round robin so that each of the 61 combinations occurs, chroma likewise
    (+ chroma from luma); 8 and 10 bit; one tile, so that edges cross superblocks"""
    bw, bh = B.BW4[bsize] * 4, B.BH4[bsize] * 4
    for bd, q, seed in ((8, 90, 1), (10, 150, 2)):
        rng = np.random.default_rng(1000 * bsize + seed)
        need = 64 if bd == 8 else 24
        cols = max(1, min(8, 512 // bw))
        rows = -(-need // cols)
        w, h = max(64, -(-cols * bw // 8) * 8), max(64, -(-rows * bh // 8) * 8)
        w, h = min(w, 512), min(h, 640)
        lay = B.Layout(w, h, 0, 0)
        seen = set()

        def symbols(i, b):
            if b["bsize"] != bsize:
                return
            k = len(seen) if len(seen) < len(MODE_ANGLES) else int(rng.integers(0, len(MODE_ANGLES)))
            b["y_mode"], b["angle_y"] = MODE_ANGLES[k]
            b["uv_mode"], b["angle_uv"] = MODE_ANGLES[(k + 17) % len(MODE_ANGLES)]
            if k % 7 == 3 and max(bw, bh) <= 32:
                b["uv_mode"] = 13
            seen.add(k)

        parts, blocks = B.random_frame(O, rng, lay, bd, q, chooser=B.uniform_chooser(bsize), symbols=symbols, p_skip=0.3)
        n = sum(b["bsize"] == bsize for b in blocks)
        assert n >= min(need, 20), (n, w, h)
        if bd == 8 and bsize >= B.BLOCK_8X8:      # (4xN / Nx4 below 8x8 code no angle delta)
            assert len(seen) == len(MODE_ANGLES) or n < len(MODE_ANGLES)
        tu = B.encode(lay, bd, q, parts, blocks)
        _same(D.decode(tu, inloop_filters=0)[0], B.decode(O, lay, bd, q, blocks), "%s %d-bit" % (B.BLOCK_NAMES[bsize], bd))


@pytest.mark.parametrize("bsize", SIZES, ids=SIZE_IDS)
def test_every_transform_size_with_every_type_of_its_sets(O, bsize):
    """K2 + K8: the block's largest transform (all 19 sizes) with every transform type of the intra set (key frame) and of the inter
    set (P frame, isolated inter blocks, one per superblock tile), no zero blocks; then reduced_tx_set.  The test checks that
    every (size, type) pair the sets allow was coded with coefficients."""
    tx = B.max_tx_rect(bsize)
    bd, q = (8, 40) if bsize % 2 else (10, 60)
    rng = np.random.default_rng(77 + bsize)
    for red in (0, 1):
        cyc = {}
        lay = B.Layout(128, 128)
        parts, blocks = B.random_frame(O, rng, lay, bd, q, chooser=B.uniform_chooser(bsize), p_skip=0.0, p_zero=0.0, cycle_types=cyc, reduced_tx_set=red)
        tu0 = B.encode(lay, bd, q, parts, blocks, reduced_tx_set=red)
        ref = B.decode(O, lay, bd, q, blocks, red)
        _same(D.decode(tu0, inloop_filters=0)[0], ref, "key frame")
        coded = {(t, ty) for b in blocks for ty, l in zip(b["tx_types"], b["levels"][0]) if l.any() for t in [b["tx"]]}
        want = set(B.TX_SETS[B.tx_set_of(tx, 0, red)])
        if len([b for b in blocks if b["bsize"] == bsize]) >= len(want):
            assert {ty for t, ty in coded if t == tx} == want, (B.BLOCK_NAMES[bsize], red, coded)
        # P frame: 16 superblock tiles, one inter block each
        lay2 = B.Layout(256, 256)
        cyc2 = {}
        want2 = set(B.TX_SETS[B.tx_set_of(tx, 1, red)])
        p1, b1 = B.random_frame(O, rng, lay2, bd, q, chooser=B.uniform_chooser(bsize), key=True)
        tuk = B.encode(lay2, bd, q, p1, b1)
        ref2 = B.decode(O, lay2, bd, q, b1)
        p2, b2 = B.random_frame(O, rng, lay2, bd, q, key=False, chooser=B.uniform_chooser(bsize), p_inter=0.6, p_skip=0.0, p_zero=0.0, cycle_types=cyc2,
                                reduced_tx_set=red)
        tup = B.encode(lay2, bd, q, p2, b2, key=False, with_sequence_header=False, reduced_tx_set=red)
        got = D.decode(tuk + tup, inloop_filters=0)
        assert len(got) == 2
        _same(got[1], B.decode(O, lay2, bd, q, b2, red, 0, ref2), "P frame")
        coded2 = {ty for b in b2 if b["is_inter"] and b["bsize"] == bsize for ty, l in zip(b["tx_types"], b["levels"][0]) if l.any()}
        n_inter = sum(b["is_inter"] and b["bsize"] == bsize for b in b2)
        assert n_inter >= 8, n_inter
        if n_inter >= len(want2):
            assert coded2 == want2, (B.BLOCK_NAMES[bsize], red, coded2)


@pytest.mark.parametrize("bsize", SIZES, ids=SIZE_IDS)
def test_every_inter_block_size_with_every_filter(O, bsize):
    """K4 on every block size: isolated inter blocks of ONE size (luma W x H, chroma W/2 x H/2: the 4-tap filters where a dimension
    is <= 4), eighth-sample luma vectors = every sixteenth-sample phase in chroma, some far outside the picture; the frame's
    filter regular / smooth / sharp / bilinear, and switchable with the filter coded per block"""
    bd, q = (10, 120) if bsize % 2 else (8, 80)
    rng = np.random.default_rng(4000 + bsize)
    lay = B.Layout(256, 192)
    pk, bk = B.random_frame(O, rng, lay, bd, q)
    tuk = B.encode(lay, bd, q, pk, bk)
    ref = B.decode(O, lay, bd, q, bk)
    phases = set()
    for filt in (0, 1, 2, 3, 4):
        p, b = B.random_frame(O, rng, lay, bd, q, key=False, chooser=B.uniform_chooser(bsize), p_inter=0.9, filters=(0, 1, 2), hp=1, p_skip=0.5)
        n = [x for x in b if x["is_inter"] and x["bsize"] == bsize]
        assert len(n) >= 7, len(n)
        phases |= {(x["mv"][0] & 15, x["mv"][1] & 15) for x in n}
        if filt == 4:
            assert {x["filt"] for x in n} == {0, 1, 2}
        tu = B.encode(lay, bd, q, p, b, key=False, with_sequence_header=False, interp_filter=filt, high_precision_mv=1)
        got = D.decode(tuk + tu, inloop_filters=0)
        _same(got[1], B.decode(O, lay, bd, q, b, 0, filt, ref), "%s filter %d" % (B.BLOCK_NAMES[bsize], filt))
    assert len(phases) >= 40


@pytest.mark.parametrize("w,h,bd,q,tiles,sel,red,filt,seed", [
    (128, 64, 8, 100, -1, 1, 0, 1, 41), (72, 88, 10, 128, 0, 0, 0, 2, 42), (192, 136, 8, 200, -1, 1, 0, 3, 43), (136, 200, 10, 60, -1, 0, 1, 4, 44),
    (192, 136, 10, 200, 0, 0, 1, 3, 48), (136, 200, 8, 60, -1, 1, 0, 4, 49)])
def test_random_trees_of_inter_and_intra_blocks(O, w, h, bd, q, tiles, sel, red, filt, seed):
    """P frames over random trees: inter blocks of any size between intra blocks (is_inter contexts, intra modes coded with the
    size-group context, chroma types derived from the luma type at the block's position), txfm_split trees"""
    rng = np.random.default_rng(seed)
    lay = B.Layout(w, h, tiles, tiles)
    tu0, b0, ref = _key(O, rng, lay, bd, q)
    p1, b1 = B.random_frame(O, rng, lay, bd, q, key=False, p_inter=0.5, tx_mode_select=sel, reduced_tx_set=red, filters=(0, 1, 2))
    tu1 = B.encode(lay, bd, q, p1, b1, key=False, with_sequence_header=False, tx_mode_select=sel, reduced_tx_set=red, interp_filter=filt)
    got = D.decode(tu0 + tu1, inloop_filters=0)
    assert len(got) == 2 and sum(b["is_inter"] for b in b1) >= 3
    _same(got[1], B.decode(O, lay, bd, q, b1, red, filt, ref), "P frame")


def test_the_block_writer_and_the_8x8_writer_agree_byte_for_byte(O):
    """two statements of the syntax, one stream: a key frame of 8x8 blocks described both ways"""
    import av1stream
    w, h, bd, q = 136, 72, 8, 120
    rng = np.random.default_rng(5)
    lay = B.Layout(w, h)
    parts, blocks = B.random_frame(O, rng, lay, bd, q, chooser=B.uniform_chooser(B.BLOCK_8X8), one_d_types=False)
    tu = B.encode(lay, bd, q, parts, blocks)
    nb = (w // 8) * (h // 8)
    arr = {k: np.zeros(nb, dt) for k, dt in (("y_mode", np.uint8), ("uv_mode", np.uint8), ("angle_y", np.int8), ("angle_uv", np.int8), ("skip", np.uint8),
                                             ("tx_type", np.uint8))}
    cfl, ly, lu, lv = np.zeros((nb, 2), np.int8), np.zeros((nb, 8, 8), np.int16), np.zeros((nb, 4, 4), np.int16), np.zeros((nb, 4, 4), np.int16)
    for b in blocks:
        i = (b["r"] // 2) * (w // 8) + b["c"] // 2
        for k in ("y_mode", "uv_mode", "angle_y", "angle_uv", "skip"):
            arr[k][i] = b[k]
        cfl[i] = b["cfl"]
        if not b["skip"]:
            arr["tx_type"][i] = b["tx_types"][0]
            ly[i], lu[i], lv[i] = b["levels"][0][0], b["levels"][1][0], b["levels"][2][0]
    assert tu == av1stream.temporal_unit(w, h, bd, q, cfl_alpha=cfl, lev_y=ly, lev_u=lu, lev_v=lv, **arr)


def test_the_block_writer_refuses_what_it_cannot_code(O):
    rng = np.random.default_rng(1)
    lay = B.Layout(64, 64)
    parts, blocks = B.random_frame(O, rng, lay, 8, 100, chooser=B.uniform_chooser(B.BLOCK_8X8))
    for b in blocks[:2]:
        b["is_inter"], b["mv"] = 1, (4, 4)
    with pytest.raises(ValueError, match="MV prediction reach"):
        B.encode(lay, 8, 100, parts, blocks, key=False)
    with pytest.raises(ValueError, match="partition"):
        B.encode(lay, 8, 100, parts[:-1], blocks)
    blocks[0]["is_inter"] = blocks[1]["is_inter"] = 0
    blocks[3]["y_mode"] = 13
    with pytest.raises(ValueError, match="intra mode"):
        B.encode(lay, 8, 100, parts, blocks)


@pytest.mark.parametrize("w,h,bd,q,tiles,sel,seed", [(128, 64, 8, 100, -1, 1, 71), (72, 88, 10, 128, 0, 0, 72), (192, 136, 8, 200, -1, 1, 73),
                                                     (136, 200, 10, 60, -1, 0, 74), (192, 136, 10, 200, 0, 0, 78), (136, 200, 8, 60, 0, 1, 79)])
def test_deblocking_and_cdef_over_general_block_structures(O, w, h, bd, q, tiles, sel, seed):
    """K5 with EVERY transform size on either side of an edge (the 4 / 6 / 8 / 14-sample filters, chosen from the smaller transform;
    transform edges inside skipped inter blocks left alone; prediction-block edges of every shape) and K6 with skip flags that come
    from 4x4 .. 64x64 blocks: the key frame's deblocked and CDEF planes and a P frame predicted from them == dav1d"""
    rng = np.random.default_rng(seed)
    lay = B.Layout(w, h, tiles, tiles)
    lv, sharp = [int(x) for x in rng.integers(1, 64, 4)], int(rng.integers(0, 8))
    cbits = int(rng.integers(0, 4))
    nset = 1 << cbits
    sets = np.stack([rng.integers(0, 16, nset), rng.integers(0, 4, nset), rng.integers(0, 16, nset), rng.integers(0, 4, nset)], 1).astype(np.uint8)
    idx, damping = rng.integers(0, nset, lay.sb_rows * lay.sb_cols).astype(np.uint8), int(rng.integers(3, 7))
    hdr = dict(lf_level=lv, lf_sharpness=sharp, cdef_damping=damping, cdef_bits=cbits, cdef_y=[int(a) << 2 | int(b) for a, b in sets[:, :2]],
               cdef_uv=[int(a) << 2 | int(b) for a, b in sets[:, 2:]], cdef_idx=idx, tx_mode_select=sel)

    def filtered(blocks, rec):
        mi, skip8 = B.loopfilter_maps(O, lay, blocks, lv)
        dbl = [O.deblock_plane(rec[p], bd, int(p > 0), mi[p], sharp) for p in range(3)]
        return dbl, list(O.cdef_frame(dbl[0], dbl[1], dbl[2], bd, damping, sets[idx], skip8))

    p0, b0 = B.random_frame(O, rng, lay, bd, q, tx_mode_select=sel, p_skip=0.3)
    tu0 = B.encode(lay, bd, q, p0, b0, **hdr)
    dbl, cdef = filtered(b0, B.decode(O, lay, bd, q, b0))
    _same(D.decode(tu0, inloop_filters=D.INLOOP_DEBLOCK)[0], dbl, "deblocked key frame")
    _same(D.decode(tu0, inloop_filters=D.INLOOP_DEBLOCK | D.INLOOP_CDEF)[0], cdef, "CDEF of the key frame")
    p1, b1 = B.random_frame(O, rng, lay, bd, q, key=False, p_inter=0.5, tx_mode_select=sel, p_skip=0.4)
    tu1 = B.encode(lay, bd, q, p1, b1, key=False, with_sequence_header=False, **hdr)
    dbl1, cdef1 = filtered(b1, B.decode(O, lay, bd, q, b1, 0, 0, cdef))
    got = D.decode(tu0 + tu1, inloop_filters=D.INLOOP_DEBLOCK | D.INLOOP_CDEF)
    _same(got[1], cdef1, "CDEF of the P frame")
