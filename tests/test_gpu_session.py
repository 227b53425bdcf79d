"""The GOP session (include/av1mi.h av1mi_gop_*): orchestration + policy + PCIe plumbing inside libav1mi.so.
 * its symbols and reference frames equal the oracle's closed-GOP chain built from the same policy numbers;
 * the AV1 stream the host writer makes of its symbols decodes in dav1d to exactly the session's own reference frames —
   at full 1080p 8-bit and 4K 10-bit sizes too (2 segments x 3 frames: lockstep, a P frame that references a P frame)."""
import numpy as np
import pytest

import dav1d_ref as D

pytestmark = pytest.mark.gpu


def _oracle_filters(O, r, bd, p, w, h, skip8, src, mi=None):
    """the oracle's filter chain with the session's parameters + its restoration ON / OFF decision against the source planes `src`:
    returns (the planes the next frame predicts from = what a decoder outputs, [on_y, on_u, on_v]); mi: other deblocking mode-info maps"""
    mi_y = np.full((h // 4, w // 4), int(O.lf_mi(3, 3, p.lf_level[0], p.lf_level[1])), np.uint32)
    mi_c = np.full((h // 8, w // 8), int(O.lf_mi(2, 2, p.lf_level[2], p.lf_level[2])), np.uint32)
    if mi is not None:
        mi_y, mi_c = mi
    dbl = [O.deblock_plane(r["rec_y"], bd, 0, mi_y, p.lf_sharpness), O.deblock_plane(r["rec_u"], bd, 1, mi_c, p.lf_sharpness),
           O.deblock_plane(r["rec_v"], bd, 1, mi_c, p.lf_sharpness)]
    nsb = ((h + 63) // 64) * ((w + 63) // 64)
    st = np.array([p.cdef_y >> 2, p.cdef_y & 3, p.cdef_uv >> 2, p.cdef_uv & 3], np.uint8)
    cdef = O.cdef_frame(dbl[0], dbl[1], dbl[2], bd, p.cdef_damping, np.tile(st, (nsb, 1)), skip8)
    ur = lambda n: max(1, (n + 32) // 64)
    uy = np.tile(np.array(list(p.lr_unit_y), np.int8), (ur(h), ur(w), 1))
    uc = np.tile(np.array(list(p.lr_unit_uv), np.int8), (ur(h // 2), ur(w // 2), 1))
    lr = [O.lr_plane(cdef[0], dbl[0], bd, 0, 64, uy), O.lr_plane(cdef[1], dbl[1], bd, 1, 64, uc), O.lr_plane(cdef[2], dbl[2], bd, 1, 64, uc)]
    return O.lr_select(src, cdef, lr, bd)


@pytest.mark.parametrize("w,h,bd,q", [(192, 128, 8, 110), (136, 72, 10, 150)])
def test_session_symbols_and_references_match_the_oracle_chain(ctx, av1mi, O, w, h, bd, q):
    gop, segs = 3, 2
    import synth
    Y, U, V = synth.frames(w, h, segs * gop, bd, 4)
    s = av1mi.GopSession(ctx, w, h, bd, q, gop, segs)
    try:
        ref = [None] * segs
        for t in range(gop):
            planes = s.input_planes()
            for sgi in range(segs):
                f = sgi * gop + t
                planes[0][sgi * h:(sgi + 1) * h] = Y[f]
                planes[1][sgi * h // 2:(sgi + 1) * h // 2] = U[f]
                planes[2][sgi * h // 2:(sgi + 1) * h // 2] = V[f]
            s.submit()
            fr = s.collect()
            gy, gu, gv = s.download_reference()
            p = fr["params"]
            assert p.frame_type == (0 if t == 0 else 1)
            pol = av1mi.policy_frame_params(q, bd, p.frame_type)
            assert list(pol.lf_level) == list(p.lf_level) and pol.cdef_y == p.cdef_y and pol.cdef_damping == p.cdef_damping
            for sgi in range(segs):
                f = sgi * gop + t
                if t == 0:
                    r = O.intra_encode_frame(Y[f], U[f], V[f], bd, 8, q)
                    assert (fr["y_mode"][sgi] == r["modes_y"]).all() and (fr["uv_mode"][sgi] == r["modes_uv"]).all()
                    skip8 = np.zeros((h // 8, w // 8), np.uint8)
                else:
                    r = O.inter_encode_frame((Y[f], U[f], V[f]), ref[sgi], bd, q, 8)
                    assert (fr["mv"][sgi] == r["mvs"]).all() and (fr["skip"][sgi] == r["skip"]).all()
                    skip8 = r["skip"].reshape(h // 8, w // 8)
                for k in ("lev_y", "lev_u", "lev_v"):
                    assert (fr[k][sgi] == r[k]).all(), (t, sgi, k)
                ref[sgi], on = _oracle_filters(O, r, bd, p, w, h, skip8, (Y[f], U[f], V[f]))
                assert fr["lr_on"][sgi].tolist() == on, (t, sgi, fr["lr_on"][sgi].tolist(), on)
                for got, exp, hh in ((gy, ref[sgi][0], h), (gu, ref[sgi][1], h // 2), (gv, ref[sgi][2], h // 2)):
                    assert (got[sgi * hh:(sgi + 1) * hh] == exp).all(), "frame %d segment %d: reference differs from the oracle chain" % (t, sgi)
    finally:
        s.close()


@pytest.mark.skipif(not D.available(), reason="no dav1d in this image")
@pytest.mark.parametrize("w,h,bd,q,gop,segs", [(192, 128, 8, 110, 4, 2), (1920, 1080, 8, 128, 3, 2), (3840, 2160, 10, 128, 3, 2)])
def test_session_stream_decodes_in_dav1d_to_the_gpu_reference(ctx, av1mi, w, h, bd, q, gop, segs):
    """GPU block pipeline + filters -> symbols -> host AV1 writer -> dav1d: the decoded frames equal the reference frames the GPU
    keeps (BASELINE configs[2] at 1080p 8-bit and configs[3] at 4K 10-bit, full size, two GOPs in lockstep, P referencing P)."""
    import av1stream
    import synth
    Y, U, V = synth.frames(w, h, segs * gop, bd, 4)
    s = av1mi.GopSession(ctx, w, h, bd, q, gop, segs)
    try:
        streams, refs = [b""] * segs, []
        for t in range(gop):
            planes = s.input_planes()
            for sgi in range(segs):
                f = sgi * gop + t
                planes[0][sgi * h:(sgi + 1) * h] = Y[f]
                planes[1][sgi * h // 2:(sgi + 1) * h // 2] = U[f]
                planes[2][sgi * h // 2:(sgi + 1) * h // 2] = V[f]
            s.submit()
            fr = s.collect()
            refs.append(s.download_reference())
            for sgi in range(segs):
                streams[sgi] += av1stream.session_frame_unit(w, h, bd, fr, sgi, threads=8)
        for sgi in range(segs):
            got = D.decode(streams[sgi])
            assert len(got) == gop
            for t in range(gop):
                for i, hh in ((0, h), (1, h // 2), (2, h // 2)):
                    assert (got[t][i] == refs[t][i][sgi * hh:(sgi + 1) * hh]).all(), "segment %d frame %d plane %d: dav1d differs from the GPU" % (sgi, t, i)
    finally:
        s.close()


def _oracle_key32(O, Y, U, V, bd, q):
    """the oracle's key frame the way a key_block_size 32 session codes it: 32x32 blocks over the complete superblock rows, 8x8 blocks
    in a last partial row (tiles are single superblocks: the two bands share nothing)"""
    h, w = Y.shape
    hA = h // 64 * 64
    a = O.intra_encode_frame(Y[:hA], U[:hA // 2], V[:hA // 2], bd, 32, q) if hA else None
    b = O.intra_encode_frame(Y[hA:], U[hA // 2:], V[hA // 2:], bd, 8, q) if hA < h else None
    rec = {k: np.concatenate([x[k] for x in (a, b) if x is not None]) for k in ("rec_y", "rec_u", "rec_v")}
    return a, b, rec


@pytest.mark.skipif(not D.available(), reason="no dav1d in this image")
@pytest.mark.parametrize("w,h,bd,q,segs", [(192, 128, 8, 110, 2), (256, 168, 10, 60, 2), (64, 64, 8, 1, 1), (128, 72, 10, 255, 1), (224, 136, 8, 100, 2), (1920, 1080, 8, 128, 1),
                                         (3840, 2160, 10, 23, 1)])
def test_key_frames_in_32x32_blocks(ctx, av1mi, O, w, h, bd, q, segs):
    """av1mi_gop_config.key_block_size = 32 (host entropy coding for now): key frames in 32x32 blocks over the complete
    superblock rows + 8x8 in a partial last row.  (1) symbols and reference frames equal the oracle's chain with the same rule,
    P frames included (they predict from the 32x32-coded key frame); (2) the stream av1mi_session_temporal_unit writes — the general
    block writer for the key frame, the 8x8 writer for the P frames — decodes in dav1d to the session's reference frames; (3) the
    key frame is smaller than the 8x8 session's at a PSNR-Y that is no lower than 0.3 dB below it."""
    import av1stream
    import synth
    gop = 3
    Y, U, V = synth.frames(w, h, segs * gop, bd, 4)
    hA = h // 64 * 64
    sizes, psnr = {}, {}
    for kbs in (32, 8):
        s = av1mi.GopSession(ctx, w, h, bd, q, gop, segs, key_block_size=kbs)
        try:
            ref, streams, refs = [None] * segs, [b""] * segs, []
            for t in range(gop):
                planes = s.input_planes()
                for sgi in range(segs):
                    f = sgi * gop + t
                    planes[0][sgi * h:(sgi + 1) * h] = Y[f]
                    planes[1][sgi * h // 2:(sgi + 1) * h // 2] = U[f]
                    planes[2][sgi * h // 2:(sgi + 1) * h // 2] = V[f]
                s.submit()
                fr = s.collect()
                got = s.download_reference()
                refs.append(got)
                p = fr["params"]
                for sgi in range(segs):
                    unit = av1stream.session_temporal_unit(w, h, bd, fr["raw"], sgi, with_sequence_header=(t == 0), threads=4)
                    streams[sgi] += unit
                    if t == 0:
                        sizes.setdefault(kbs, []).append(len(unit))
                        mse = np.mean((got[0][sgi * h:(sgi + 1) * h].astype(np.float64) - Y[sgi * gop]) ** 2)
                        psnr.setdefault(kbs, []).append(10 * np.log10(((1 << bd) - 1) ** 2 / mse))
                    if kbs == 8 or (w * h > 500000 and t > 0):
                        continue                  # (the oracle's P frames at 1080p take a minute each: the dav1d check below covers them)
                    f = sgi * gop + t
                    if t == 0:
                        assert fr["key_block_size"] == 32
                        a, b, rec = _oracle_key32(O, Y[f], U[f], V[f], bd, q)
                        if a is not None:
                            assert (fr["y_mode32"][sgi] == a["modes_y"]).all() and (fr["uv_mode32"][sgi] == a["modes_uv"]).all()
                            for k in ("lev_y", "lev_u", "lev_v"):
                                assert (fr[k + "32"][sgi] == a[k]).all(), (sgi, k)
                        if b is not None:
                            assert (fr["y_mode8"][sgi] == b["modes_y"]).all() and (fr["uv_mode8"][sgi] == b["modes_uv"]).all()
                            for k in ("lev_y", "lev_u", "lev_v"):
                                assert (fr[k + "8"][sgi] == b[k]).all(), (sgi, k)
                        r, skip8 = rec, np.zeros((h // 8, w // 8), np.uint8)
                        # the filters see 32x32 / 16x16 transform edges over the complete superblock rows
                        mi_y = np.full((h // 4, w // 4), int(O.lf_mi(3, 3, p.lf_level[0], p.lf_level[1])), np.uint32)
                        mi_c = np.full((h // 8, w // 8), int(O.lf_mi(2, 2, p.lf_level[2], p.lf_level[2])), np.uint32)
                        mi_y[:hA // 4] = int(O.lf_mi(5, 5, p.lf_level[0], p.lf_level[1]))
                        mi_c[:hA // 8] = int(O.lf_mi(4, 4, p.lf_level[2], p.lf_level[2]))
                        ref[sgi], on = _oracle_filters(O, r, bd, p, w, h, skip8, (Y[f], U[f], V[f]), mi=(mi_y, mi_c))
                    else:
                        r = O.inter_encode_frame((Y[f], U[f], V[f]), ref[sgi], bd, q, 8)
                        assert (fr["mv"][sgi] == r["mvs"]).all() and (fr["skip"][sgi] == r["skip"]).all()
                        for k in ("lev_y", "lev_u", "lev_v"):
                            assert (fr[k][sgi] == r[k]).all(), (t, sgi, k)
                        ref[sgi], on = _oracle_filters(O, r, bd, p, w, h, r["skip"].reshape(h // 8, w // 8), (Y[f], U[f], V[f]))
                    assert fr["lr_on"][sgi].tolist() == on
                    for g_, e_, hh in ((got[0], ref[sgi][0], h), (got[1], ref[sgi][1], h // 2), (got[2], ref[sgi][2], h // 2)):
                        assert (g_[sgi * hh:(sgi + 1) * hh] == e_).all(), "frame %d segment %d: reference differs from the oracle chain" % (t, sgi)
            if kbs == 32:
                key32_streams = list(streams)
            for sgi in range(segs):
                dec = D.decode(streams[sgi])
                assert len(dec) == gop
                for t in range(gop):
                    for i, hh in ((0, h), (1, h // 2), (2, h // 2)):
                        assert (dec[t][i] == refs[t][i][sgi * hh:(sgi + 1) * hh]).all(), "key block size %d, segment %d frame %d plane %d: dav1d differs from the GPU" % (kbs, sgi, t, i)
        finally:
            s.close()
    if 20 <= q <= 200:      # (at the ends of the quantiser range the two block sizes trade bytes for PSNR differently)
        assert sum(sizes[32]) < sum(sizes[8]) and min(a - b for a, b in zip(psnr[32], psnr[8])) > -0.3, (sizes, psnr)
    # (4) the GPU tile coder (k_av1_tokens32 for the 32x32 band's tiles, then the usual chains / range coder): the same bytes
    s = av1mi.GopSession(ctx, w, h, bd, q, gop, segs, key_block_size=32, gpu_entropy=1)
    try:
        coded = [b""] * segs
        for t in range(gop):
            planes = s.input_planes()
            for sgi in range(segs):
                f = sgi * gop + t
                planes[0][sgi * h:(sgi + 1) * h] = Y[f]
                planes[1][sgi * h // 2:(sgi + 1) * h // 2] = U[f]
                planes[2][sgi * h // 2:(sgi + 1) * h // 2] = V[f]
            s.submit()
            fr = s.collect()
            if q >= 10:      # (at q 1 every coefficient escapes to Golomb: the tile exceeds the coder's list and the batch comes back as symbols —
                assert "tile_size" in fr and s.entropy_fallbacks() == 0      # the fallback; the bytes below are then the host's)
            for sgi in range(segs):
                coded[sgi] += av1stream.session_temporal_unit(w, h, bd, fr["raw"], sgi, with_sequence_header=(t == 0), threads=4)
        assert coded == key32_streams
    finally:
        s.close()
    with pytest.raises(av1mi.Av1miError):
        av1mi.GopSession(ctx, 136, 72, bd, q, gop, segs, key_block_size=32)                    # width must be a multiple of 32


@pytest.mark.skipif(not D.available(), reason="no dav1d in this image")
@pytest.mark.parametrize("vw,vh,bd,q", [(190, 131, 8, 100), (252, 70, 10, 40)])
def test_key_frames_in_32x32_blocks_of_a_cropped_frame(ctx, av1mi, vw, vh, bd, q):
    """key_block_size 32 with a true size that is not a multiple of 8 (coded size rounded up, width a multiple of 64): the 32x32 band
    reaches over the cropped edge like the 8x8 blocks do; the GPU-coded and the host-coded streams are the same bytes and dav1d
    outputs vw x vh frames that equal the session's references cropped"""
    import av1stream
    import synth
    import test_av1_conformance as T
    gop, w, h = 3, (vw + 7) // 8 * 8, (vh + 7) // 8 * 8
    cvw, cvh = (vw + 1) // 2, (vh + 1) // 2
    Yc, Uc, Vc = synth.frames(w + 8, h + 8, gop, bd, 6)
    out = {}
    for mode in (1, 0):
        s = av1mi.GopSession(ctx, w, h, bd, q, gop, 1, gpu_entropy=mode, visible=(vw, vh), key_block_size=32)
        try:
            stream, refs = b"", []
            for t in range(gop):
                planes = s.input_planes()
                planes[0][:] = T._pad(Yc[t][:vh, :vw], h, w)
                planes[1][:] = T._pad(Uc[t][:cvh, :cvw], h // 2, w // 2)
                planes[2][:] = T._pad(Vc[t][:cvh, :cvw], h // 2, w // 2)
                s.submit()
                fr = s.collect()
                refs.append(s.download_reference())
                stream += av1stream.session_temporal_unit(w, h, bd, fr["raw"], 0, with_sequence_header=(t == 0), threads=4, visible=(vw, vh))
            assert s.entropy_fallbacks() == 0
            out[mode] = (stream, refs)
        finally:
            s.close()
    assert out[1][0] == out[0][0]
    got = D.decode(out[1][0])
    assert len(got) == gop
    for t in range(gop):
        for i, (ch, cw) in enumerate(((vh, vw), (cvh, cvw), (cvh, cvw))):
            assert got[t][i].shape == (ch, cw) and (got[t][i] == out[1][1][t][i][:ch, :cw]).all(), "frame %d plane %d: dav1d differs from the GPU" % (t, i)


@pytest.mark.skipif(not D.available(), reason="no dav1d in this image")
def test_the_bench_configuration_at_full_size(ctx, av1mi):
    """BASELINE configs[3] exactly as bench.py runs it: 3840x2160 10-bit, 12 closed GOPs of 30 frames in lockstep, q 128, key frames in
    32x32 blocks, GPU tile coder, three batches in flight.  No batch falls back; the first and the last segment's streams (60 frames)
    decode in dav1d to the reference frames the session kept at the time."""
    import av1stream
    import synth
    w, h, bd, q, gop, segs = 3840, 2160, 10, 128, 30, 12
    check = (0, segs - 1)
    Y, U, V = synth.frames(w, h, segs * gop, bd, 0)          # 9 GB of source, like the bench's
    s = av1mi.GopSession(ctx, w, h, bd, q, gop, segs, gpu_entropy=1, key_block_size=32)
    try:
        streams, refs, nbytes = {sg: b"" for sg in check}, {sg: [] for sg in check}, 0
        lag = s.max_in_flight() - 1

        def collect():
            nonlocal nbytes
            fr = s.collect()
            assert "tile_size" in fr
            nbytes += int(fr["tile_size"].sum(dtype=np.uint64))
            for sg in check:
                streams[sg] += av1stream.session_temporal_unit(w, h, bd, fr["raw"], sg, with_sequence_header=(fr["frame_type"] == 0))
        for t in range(gop):
            planes = s.input_planes()
            for sg in range(segs):
                f = sg * gop + t
                planes[0][sg * h:(sg + 1) * h] = Y[f]
                planes[1][sg * h // 2:(sg + 1) * h // 2] = U[f]
                planes[2][sg * h // 2:(sg + 1) * h // 2] = V[f]
            s.submit()
            got = s.download_reference()      # (waits for this batch's filters: the reference the NEXT frame predicts from)
            for sg in check:
                refs[sg].append([got[0][sg * h:(sg + 1) * h].copy(), got[1][sg * h // 2:(sg + 1) * h // 2].copy(), got[2][sg * h // 2:(sg + 1) * h // 2].copy()])
            if t >= lag:
                collect()
        while s.pending():
            collect()
        assert s.entropy_fallbacks() == 0
        assert 200e3 < nbytes / (gop * segs) < 800e3          # ~457 KB per frame at q 128
        for sg in check:
            dec = D.decode(streams[sg])
            assert len(dec) == gop
            for t in range(gop):
                for i in range(3):
                    assert (dec[t][i] == refs[sg][t][i]).all(), "segment %d frame %d plane %d: dav1d differs from the GPU" % (sg, t, i)
    finally:
        s.close()


@pytest.mark.skipif(not D.available(), reason="no dav1d in this image")
def test_key_frames_in_32x32_blocks_over_many_sizes(ctx, av1mi):
    """sizes around the band rule (no complete superblock row, exactly whole rows, one to seven 8x8 rows below), both depths, the whole
    quantiser range, one to three segments: the GPU-coded stream equals the host-coded one and dav1d decodes it to the session's references"""
    import av1stream
    import synth
    rng = np.random.default_rng(2024)
    for case in range(24):
        w = int(rng.choice([64, 96, 128, 160, 192, 320]))      # (96, 160: a last column of half superblocks)
        h = int(rng.choice([8, 40, 56, 64, 72, 120, 128, 136, 184, 200]))
        bd, q, segs, gop = int(rng.choice([8, 10])), int(rng.integers(12, 240)), int(rng.integers(1, 4)), 2
        Y, U, V = synth.frames(w, h, segs * gop, bd, case)
        out = {}
        for mode in (1, 0):
            s = av1mi.GopSession(ctx, w, h, bd, q, gop, segs, gpu_entropy=mode, key_block_size=32)
            try:
                streams, refs = [b""] * segs, []
                for t in range(gop):
                    planes = s.input_planes()
                    for sg in range(segs):
                        f = sg * gop + t
                        planes[0][sg * h:(sg + 1) * h] = Y[f]
                        planes[1][sg * h // 2:(sg + 1) * h // 2] = U[f]
                        planes[2][sg * h // 2:(sg + 1) * h // 2] = V[f]
                    s.submit()
                    fr = s.collect()
                    refs.append(s.download_reference())
                    for sg in range(segs):
                        streams[sg] += av1stream.session_temporal_unit(w, h, bd, fr["raw"], sg, with_sequence_header=(t == 0), threads=2)
                out[mode] = (streams, refs, s.entropy_fallbacks())
            finally:
                s.close()
        assert out[1][0] == out[0][0], (case, w, h, bd, q, segs)
        assert out[1][2] == 0, (case, w, h, bd, q)
        for sg in range(segs):
            dec = D.decode(out[1][0][sg])
            assert len(dec) == gop
            for t in range(gop):
                for i, hh in ((0, h), (1, h // 2), (2, h // 2)):
                    assert (dec[t][i] == out[1][1][t][i][sg * hh:(sg + 1) * hh]).all(), (case, w, h, bd, q, sg, t, i)


def test_session_api_misuse_is_reported(ctx, av1mi):
    s = av1mi.GopSession(ctx, 64, 64, 8, 100, 2, 1)
    try:
        with pytest.raises(av1mi.Av1miError):
            s.submit()                      # nothing acquired
        with pytest.raises(av1mi.Av1miError):
            s.collect()                     # nothing in flight
        s.input_planes()
        with pytest.raises(av1mi.Av1miError):
            s.submit(1)                     # a session starts with a key frame
        for _ in range(s.max_in_flight()):  # the pipeline holds max_in_flight batches, not one more
            s.input_planes()
            s.submit()
        assert s.pending() == s.max_in_flight() == 3
        with pytest.raises(av1mi.Av1miError):
            s.input_planes()
        while s.pending():
            s.collect()
    finally:
        s.close()
    with pytest.raises(av1mi.Av1miError):
        av1mi.GopSession(ctx, 60, 64, 8, 100, 2, 1)


@pytest.mark.parametrize("w,h,bd,q,gop,segs", [(192, 128, 8, 110, 4, 2), (136, 72, 10, 40, 3, 3), (64, 64, 8, 30, 2, 1), (328, 184, 8, 200, 2, 5),
                                                (640, 360, 10, 50, 2, 1), (1920, 1080, 8, 128, 3, 2), (3840, 2160, 10, 128, 3, 2),
                                                # the reference's own quality points (DetermineQuality, transcode.go:157-165: 24 at 1080p, 23 at 2160p)
                                                (1920, 1080, 8, 24, 2, 2), (3840, 2160, 10, 23, 2, 2)])
def test_gpu_tile_entropy_coder_bytes_equal_the_host_writer(ctx, av1mi, w, h, bd, q, gop, segs):
    """K9 for the real syntax: the AV1 tile entropy coder on the GPU (csrc/av1_entropy_kernels.hip).  With gpu_entropy = 2 the
    session hands out both the symbols and the GPU-coded tile payloads: the temporal unit assembled around the GPU's payloads
    must be byte-identical to the one the host writer makes of the symbols — key and inter frames, partial superblocks at the
    frame edge (1080 and 2160 are not multiples of 64), 8 and 10 bit, a single tile, fine (30 / 50) and coarse (200) quantisers, a
    tile count that is not a multiple of the coder's groups of 64, full 1080p / 4K sizes — and dav1d (when present) decodes it to
    the GPU's reference frames."""
    import av1stream
    import synth
    Y, U, V = synth.frames(w, h, segs * gop, bd, 4)
    s = av1mi.GopSession(ctx, w, h, bd, q, gop, segs, gpu_entropy=2)
    try:
        streams, refs = [b""] * segs, []
        for t in range(gop):
            planes = s.input_planes()
            for sgi in range(segs):
                f = sgi * gop + t
                planes[0][sgi * h:(sgi + 1) * h] = Y[f]
                planes[1][sgi * h // 2:(sgi + 1) * h // 2] = U[f]
                planes[2][sgi * h // 2:(sgi + 1) * h // 2] = V[f]
            s.submit()
            fr = s.collect()
            refs.append(s.download_reference())
            assert "tile_size" in fr, "the GPU coder gave the batch back (capacity) at q %d" % q
            assert fr["tiles_per_frame"] == ((w + 63) // 64) * ((h + 63) // 64) and (fr["tile_size"] > 0).all()
            for sgi in range(segs):
                host = av1stream.session_frame_unit(w, h, bd, fr, sgi, threads=8)
                gpu = av1stream.session_frame_unit_gpu(w, h, bd, fr, sgi)
                assert gpu == host, "frame %d segment %d: GPU-coded temporal unit differs from the host writer's (%d vs %d bytes)" % (t, sgi, len(gpu), len(host))
                streams[sgi] += gpu
        if D.available():
            for sgi in range(segs):
                got = D.decode(streams[sgi])
                assert len(got) == gop
                for t in range(gop):
                    for i, hh in ((0, h), (1, h // 2), (2, h // 2)):
                        assert (got[t][i] == refs[t][i][sgi * hh:(sgi + 1) * hh]).all()
    finally:
        s.close()


def test_gpu_entropy_only_mode_downloads_no_symbols(ctx, av1mi):
    import av1stream
    import synth
    w, h, bd, q = 192, 128, 8, 120
    Y, U, V = synth.frames(w, h, 2, bd, 1)
    a = av1mi.GopSession(ctx, w, h, bd, q, 2, 1, gpu_entropy=1)
    b = av1mi.GopSession(ctx, w, h, bd, q, 2, 1, gpu_entropy=0)
    try:
        for t in range(2):
            for s in (a, b):
                planes = s.input_planes()
                for dst, src in zip(planes, (Y[t], U[t], V[t])):
                    dst[:] = src
                s.submit()
            fa, fb = a.collect(), b.collect()
            assert "lev_y" not in fa and "mv" not in fa and "y_mode" not in fa and "tile_size" not in fb
            assert av1stream.session_frame_unit_gpu(w, h, bd, fa, 0) == av1stream.session_frame_unit(w, h, bd, fb, 0)
    finally:
        a.close()
        b.close()


@pytest.mark.skipif(not D.available(), reason="no dav1d in this image")
def test_gpu_coder_capacity_overflow_falls_back_to_the_host(ctx, av1mi):
    """white noise at base_q_idx 2: a 64x64 tile needs more ops than the GPU coder's per-tile list holds.  The session must not
    lose the batch: it hands the symbols out instead (tile_size absent), the host writer codes them, dav1d agrees with the GPU."""
    import av1stream
    w, h, bd, q = 128, 128, 8, 2
    rng = np.random.default_rng(0)
    s = av1mi.GopSession(ctx, w, h, bd, q, 2, 1, gpu_entropy=1)
    try:
        stream, refs = b"", []
        for t in range(2):
            planes = s.input_planes()
            planes[0][:] = rng.integers(0, 256, planes[0].shape)
            planes[1][:] = rng.integers(0, 256, planes[1].shape)
            planes[2][:] = rng.integers(0, 256, planes[2].shape)
            s.submit()
            fr = s.collect()
            refs.append(s.download_reference())
            if "tile_size" in fr:
                stream += av1stream.session_frame_unit_gpu(w, h, bd, fr, 0)
            else:
                assert "lev_y" in fr
                stream += av1stream.session_frame_unit(w, h, bd, fr, 0)
        assert s.entropy_fallbacks() >= 1
        got = D.decode(stream)
        assert len(got) == 2 and all((got[t][i] == refs[t][i]).all() for t in range(2) for i in range(3))
    finally:
        s.close()


@pytest.mark.skipif(not D.available(), reason="no dav1d in this image")
def test_fallback_with_three_batches_in_flight_at_the_reference_quality(ctx, av1mi):
    """white noise at the reference's quality 24 overflows the GPU coder's per-block records in EVERY batch.  Pipelined
    (submit t + 2 before collect t) the session hands every batch out as symbols — the first ones after the coder gave them back,
    later ones downloaded beside the filters (the session switches after three fallbacks) — and the host-coded stream equals the
    one of a host-mode session byte for byte and decodes in dav1d to the GPU's reference frames."""
    import av1stream
    w, h, bd, q, gop, segs, n = 256, 192, 8, 24, 3, 2, 9
    rng = np.random.default_rng(7)
    src = [rng.integers(0, 256, (n, segs * h // d, w // d)).astype(np.uint8) for d in (1, 2, 2)]

    def run(mode, lag):
        s = av1mi.GopSession(ctx, w, h, bd, q, gop, segs, gpu_entropy=mode)
        units, refs, gave_back = [[] for _ in range(segs)], [], 0
        try:
            for t in range(n):
                for dst, a in zip(s.input_planes(), src):
                    dst[:] = a[t]
                s.submit()
                if lag == 0:
                    refs.append(s.download_reference())
                while s.pending() > lag or (t == n - 1 and s.pending()):
                    fr = s.collect()
                    gave_back += "tile_size" not in fr
                    for sgi in range(segs):
                        units[sgi].append(av1stream.session_frame_unit_gpu(w, h, bd, fr, sgi) if "tile_size" in fr else av1stream.session_frame_unit(w, h, bd, fr, sgi, threads=4))
            return units, refs, gave_back, s.entropy_fallbacks()
        finally:
            s.close()

    host, refs, _, _ = run(0, 0)
    gpu, _, gave_back, fallbacks = run(1, 2)
    assert gave_back == n and fallbacks == n, (gave_back, fallbacks)
    assert gpu == host
    for sgi in range(segs):
        got = D.decode(b"".join(host[sgi]))
        assert len(got) == n
        for t in range(n):
            for i, hh in ((0, h), (1, h // 2), (2, h // 2)):
                assert (got[t][i] == refs[t][i][sgi * hh:(sgi + 1) * hh]).all()


@pytest.mark.parametrize("mode", [1, 0])
def test_three_batches_in_flight_give_the_bytes_of_lockstep(ctx, av1mi, mode):
    """The pipelined use (submit t + 2 before collect t: uploads, kernels, the two coder streams and the host overlap, every slot
    and both list sets of the coder are reused several times) must produce exactly the temporal units of the one-batch-at-a-time
    use — GPU tile coder (mode 1) and symbols for the host coder (mode 0), three GOPs per segment."""
    import av1stream
    import synth
    w, h, bd, q, gop, segs, n = 640, 360, 10, 120, 4, 3, 12
    Y, U, V = synth.frames(w, h, segs * n, bd, 2)

    def run(lag):
        s = av1mi.GopSession(ctx, w, h, bd, q, gop, segs, gpu_entropy=mode)
        units = []

        def take():
            fr = s.collect()
            for sgi in range(segs):
                units.append(av1stream.session_frame_unit_gpu(w, h, bd, fr, sgi) if mode else av1stream.session_frame_unit(w, h, bd, fr, sgi, threads=4))
        try:
            for t in range(n):
                planes = s.input_planes()
                for sgi in range(segs):
                    f = sgi * n + t
                    planes[0][sgi * h:(sgi + 1) * h] = Y[f]
                    planes[1][sgi * h // 2:(sgi + 1) * h // 2] = U[f]
                    planes[2][sgi * h // 2:(sgi + 1) * h // 2] = V[f]
                s.submit()
                if s.pending() > lag:
                    take()
            while s.pending():
                take()
        finally:
            s.close()
        return units

    a, b = run(0), run(2)
    assert len(a) == len(b) == n * segs
    for i, (x, y) in enumerate(zip(a, b)):
        assert x == y, "frame %d segment %d: pipelined session differs from lockstep (%d vs %d bytes)" % (i // segs, i % segs, len(y), len(x))
    if D.available():      # and the streams are real: segment 0 decodes to 12 frames
        got = D.decode(b"".join(b[i] for i in range(0, n * segs, segs)))
        assert len(got) == n


@pytest.mark.skipif(not D.available(), reason="no dav1d in this image")
def test_random_small_configurations_decode_to_the_gpu_frames(ctx, av1mi):
    """twenty random small configurations (sizes that are multiples of 8 but mostly not of 64, both bit depths, quantisers 20..235,
    GOP lengths 1..5, 1..4 segments, three batches in flight): the GPU-coded stream of every segment decodes in dav1d, frame by
    frame, to the planes the session reports as decoder output, and the host writer codes the same symbols to the same bytes"""
    import av1stream
    import synth
    rng = np.random.default_rng(2024)
    for case in range(20):
        w, h = int(rng.integers(8, 41)) * 8, int(rng.integers(8, 31)) * 8
        bd, q = int(rng.choice([8, 10])), int(rng.integers(20, 236))
        gop, segs, nfr = int(rng.integers(1, 6)), int(rng.integers(1, 5)), int(rng.integers(2, 7))
        Y, U, V = synth.frames(w, h, segs * nfr, bd, int(rng.integers(0, 50)))
        s = av1mi.GopSession(ctx, w, h, bd, q, gop, segs, gpu_entropy=2)
        streams, refs = [b""] * segs, []
        try:
            for t in range(nfr):
                planes = s.input_planes()
                for sgi in range(segs):
                    f = sgi * nfr + t
                    planes[0][sgi * h:(sgi + 1) * h] = Y[f]
                    planes[1][sgi * h // 2:(sgi + 1) * h // 2] = U[f]
                    planes[2][sgi * h // 2:(sgi + 1) * h // 2] = V[f]
                s.submit()
                fr = s.collect()
                refs.append(s.download_reference())
                assert "tile_size" in fr, (case, w, h, bd, q, "the GPU coder gave a batch back")
                for sgi in range(segs):
                    gpu = av1stream.session_frame_unit_gpu(w, h, bd, fr, sgi)
                    assert gpu == av1stream.session_frame_unit(w, h, bd, fr, sgi, threads=2), (case, w, h, bd, q, gop, segs, t, sgi)
                    streams[sgi] += gpu
        finally:
            s.close()
        for sgi in range(segs):
            got = D.decode(streams[sgi])
            assert len(got) == nfr, (case, w, h, bd, q, gop, segs)
            for t in range(nfr):
                for i, hh in ((0, h), (1, h // 2), (2, h // 2)):
                    assert (got[t][i] == refs[t][i][sgi * hh:(sgi + 1) * hh]).all(), (case, w, h, bd, q, gop, segs, t, sgi, i)


@pytest.mark.skipif(not D.available(), reason="no dav1d in this image")
@pytest.mark.parametrize("vw,vh,bd,q,gop,segs", [(100, 76, 8, 220, 3, 2), (61, 45, 8, 100, 3, 1), (130, 70, 10, 60, 3, 2), (132, 68, 10, 230, 2, 3),
                                                  (854, 480, 8, 128, 3, 2), (1366, 768, 10, 110, 2, 1), (17, 9, 8, 120, 2, 2), (9, 23, 10, 160, 2, 1)])
def test_sizes_that_are_not_multiples_of_8(ctx, av1mi, O, vw, vh, bd, q, gop, segs):
    """A vw x vh source coded at the size rounded up to 8 (include/av1mi.h av1mi_gop_config.visible_width): the stream announces
    the true size, the host writer and the GPU tile coder give the same bytes, dav1d outputs vw x vh frames that equal the
    session's reference frames cropped — P frames predicted across the replicated border included — and, for the small cases,
    the oracle's chain (tests/test_av1_conformance.py visible_gop, itself pinned to dav1d on the CPU)."""
    import av1stream
    import pipeline as P
    import synth
    import test_av1_conformance as T
    w, h = (vw + 7) // 8 * 8, (vh + 7) // 8 * 8
    cvw, cvh = (vw + 1) // 2, (vh + 1) // 2
    s = av1mi.GopSession(ctx, w, h, bd, q, gop, segs, gpu_entropy=2, visible=(vw, vh))
    try:
        streams, refs = [b""] * segs, []
        for t in range(gop):
            planes = s.input_planes()
            for sgi in range(segs):
                Yc, Uc, Vc = synth.frames(w + 8, h + 8, gop, bd, 3 + 5 * sgi)
                planes[0][sgi * h:(sgi + 1) * h] = T._pad(Yc[t][:vh, :vw], h, w)
                planes[1][sgi * h // 2:(sgi + 1) * h // 2] = T._pad(Uc[t][:cvh, :cvw], h // 2, w // 2)
                planes[2][sgi * h // 2:(sgi + 1) * h // 2] = T._pad(Vc[t][:cvh, :cvw], h // 2, w // 2)
            s.submit()
            fr = s.collect()
            refs.append(s.download_reference())
            for sgi in range(segs):
                host = av1stream.session_frame_unit(w, h, bd, fr, sgi, threads=8, visible=(vw, vh))
                gpu = av1stream.session_frame_unit_gpu(w, h, bd, fr, sgi, visible=(vw, vh))
                assert gpu == host, "frame %d segment %d: GPU-coded temporal unit differs from the host writer's" % (t, sgi)
                streams[sgi] += gpu
        for sgi in range(segs):
            got = D.decode(streams[sgi])
            assert len(got) == gop
            for t in range(gop):
                for i, (hh, ch, cw) in enumerate(((h, vh, vw), (h // 2, cvh, cvw), (h // 2, cvh, cvw))):
                    assert got[t][i].shape == (ch, cw)
                    assert (got[t][i] == refs[t][i][sgi * hh:sgi * hh + ch, :cw]).all(), "segment %d frame %d plane %d: dav1d differs from the GPU" % (sgi, t, i)
        if w * h <= 200 * 200:       # the oracle's chain of segment 0 (first frame 3): same stream, same reference frames
            stream, orefs, _ = T.visible_gop(O, P, vw, vh, bd, q, gop)
            assert stream == streams[0]
            for t in range(gop):
                for i, hh in ((0, h), (1, h // 2), (2, h // 2)):
                    assert (orefs[t][i] == refs[t][i][:hh]).all(), "frame %d plane %d: the GPU's reference differs from the oracle's (padding included)" % (t, i)
    finally:
        s.close()


def test_visible_size_must_be_within_the_last_block(ctx, av1mi):
    for vis in ((90, 80), (104, 72), (105, 80), (104, 81)):
        with pytest.raises(av1mi.Av1miError):
            av1mi.GopSession(ctx, 104, 80, 8, 100, 2, 1, visible=vis)
