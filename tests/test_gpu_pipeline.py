"""GPU end-to-end: the whole segment pipeline (intra coding + deblock + CDEF + loop restoration, 8 launches) against
the oracle chain, frame by frame, bit-exact; plus the quality sanity the metric needs."""
import numpy as np
import pytest

import synth

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


@pytest.mark.parametrize("w,h,bd,q,segs,gop,rng_", [
    (192, 128, 8, 90, 2, 4, 6),          # small: 2 segments x (1 key + 3 P)
    (1920, 1080, 8, 128, 1, 3, 8),       # BASELINE configs[2] size: 1080p 8-bit, key + 2 P
    (3840, 2160, 10, 128, 1, 2, 8),      # BASELINE configs[3] size = the bench's default workload: 4K 10-bit, key + P
])
def test_closed_gop_chain_matches_oracle(ctx, O, w, h, bd, q, segs, gop, rng_):
    """BASELINE configs 2/3 end to end: closed GOPs (1 key + P frames); every P frame predicts from the loop-filtered
    previous frame.  Reconstruction after the whole filter chain equals the oracle chain for every frame."""
    import pipeline
    gp = pipeline.GopPipeline(ctx, w, h, bd, segments=segs, gop=gop, qindex=q, first_frame=1, search_range=rng_)
    got = []
    dl = lambda bufs, shapes: [b.download(s.shape, s.dtype) for b, s in zip(bufs, shapes)]
    cd = [gp.key.d["cdef_" + p] for p in "yuv"]
    gp.step(on_frame=lambda t: got.append((dl(gp.d_ref, gp.src[t]), dl(cd, gp.src[t]))))
    lr_on = [gp.lr_on(t) for t in range(gop)]
    k = gp.key
    for s in range(segs):
        ref = None
        for t in range(gop):
            src = [gp.src[t][i][s] for i in range(3)]
            if t == 0:
                r = O.intra_encode_frame(src[0], src[1], src[2], bd, 8, q)
                skip8 = np.zeros((h // 8, w // 8), np.uint8)
            else:
                r = O.inter_encode_frame(src, ref, bd, q, rng_)
                skip8 = r["skip"].reshape(h // 8, w // 8)
            mi_y, mi_c, damping, cdef_sb, lr_unit, lr_y, lr_c = gp.oracle_filter_args(t)
            dbl = [O.deblock_plane(r["rec_y"], bd, 0, mi_y), O.deblock_plane(r["rec_u"], bd, 1, mi_c), O.deblock_plane(r["rec_v"], bd, 1, mi_c)]
            cdef = O.cdef_frame(dbl[0], dbl[1], dbl[2], bd, damping, cdef_sb, skip8)
            lr = [O.lr_plane(cdef[0], dbl[0], bd, 0, lr_unit, lr_y), O.lr_plane(cdef[1], dbl[1], bd, 1, lr_unit, lr_c),
                  O.lr_plane(cdef[2], dbl[2], bd, 1, lr_unit, lr_c)]
            # the restoration ON / OFF decision against the source: frame t + 1 predicts from the restored or the CDEF plane, and the
            # restored plane is complete exactly where it is kept (av1mi_lr_yuv_decide restores the other planes' sampled tiles only)
            ref, on = O.lr_select(src, cdef, lr, bd)
            assert lr_on[t][s].tolist() == on, (s, t, lr_on[t][s].tolist(), on)
            for i in range(3):
                assert (got[t][1][i][s] == cdef[i]).all(), (s, t, i)
                assert not on[i] or (got[t][0][i][s] == lr[i]).all(), (s, t, i)
    gp.close()


def test_job_edge_cases_and_error_behaviour(ctx, av1mi):
    """the boundary's error contract on the fused pipelines: an empty segment is a no-op that touches nothing, malformed jobs are
    refused with a negative code and a message (never a launch), 12-bit content is rejected (only 8 and 10 are built)"""
    w, h = 64, 64
    Y, U, V = synth.frames(w, h, 1, 8, first=0)
    src = [ctx.to_device(a) for a in (Y, U, V)]
    rec = [ctx.alloc(a.nbytes) for a in (Y, U, V)]
    lev = [ctx.alloc(a.size * 2) for a in (Y, U, V)]
    modes = [ctx.alloc(64), ctx.alloc(64)]
    for b in rec + lev + modes:
        ctx.memset(b, 0xAB, 64)
    def job(width=w, height=h, bd=8, nframes=1, q=100, bs=8):
        j = av1mi.IntraJob(width, height, bd, nframes, q, bs, width, width // 2)
        for k, b in zip(("src_y", "src_u", "src_v", "rec_y", "rec_u", "rec_v", "lev_y", "lev_u", "lev_v", "modes_y", "modes_uv"), src + rec + lev + modes):
            setattr(j, "d_" + k, b.ptr)
        return j
    ctx.intra_encode(job(nframes=0))                                   # empty segment: accepted, nothing written
    ctx.sync()
    assert (rec[0].download((64,), np.uint8) == 0xAB).all() and (modes[0].download((64,), np.uint8) == 0xAB).all()
    for bad, what in ((job(bd=12), "bit depth"), (job(width=60), "multiple"), (job(q=256), "qindex"), (job(bs=64), "block"), (job(nframes=-1), "nframes")):
        rc = ctx.lib.av1mi_intra_encode(ctx.h, __import__("ctypes").byref(bad))
        msg = ctx.lib.av1mi_last_error(ctx.h).decode()
        assert rc < 0 and msg, (what, rc, msg)
    j = job()
    j.d_rec_y = None
    assert ctx.lib.av1mi_intra_encode(ctx.h, __import__("ctypes").byref(j)) < 0            # null device pointer
    ctx.intra_encode(job())                                            # and the context still works afterwards
    ctx.sync()
    assert not (rec[0].download(Y.shape, np.uint8) == 0xAB).all()
    for b in src + rec + lev + modes:
        b.free()
