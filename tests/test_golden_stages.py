"""Committed golden vectors (tests/golden/stages_kat.npz, made by tools/gen_golden.py from the oracle): CPU test = the
oracle still reproduces them (drift guard; they are NOT reference-derived, parity stays unpinned); GPU test = the HIP
kernels reproduce them through the C ABI."""
import os
import sys

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden", "stages_kat.npz")
MVS = [(0, 0), (8, 8), (5, -11), (-37, 21), (255, 3), (-400, -400), (1500, 700), (15, 1)]
ST = np.array([[9, 2, 5, 3], [4, 1, 0, 2]], np.uint8)


def _intra_cases(bw, bh):
    for mode in range(13):
        for delta in ((-3, 0, 2) if 1 <= mode <= 8 else (0,)):
            for avail in ((bw, bw, bh, bh), (bw, 0, bh, 0)):
                yield mode, delta, avail


def test_oracle_reproduces_stage_goldens(O):
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "av1-go_amd"))
    import synth
    g = np.load(G)
    for bd in (8, 10):
        plane = g["intra_plane_%d" % bd]
        for (bw, bh) in ((4, 4), (8, 8), (16, 16), (32, 8)):
            exp = g["intra_%dx%d_%d" % (bw, bh, bd)]
            for i, (mode, delta, (nt, ntr, nl, nbl)) in enumerate(_intra_cases(bw, bh)):
                assert (O.intra_predict(plane, 40, 40, bw, bh, mode, delta, bd, nt, ntr, nl, nbl, 0, mode & 1) == exp[i]).all()
        assert (O.deblock_plane(g["lf_img_%d" % bd], bd, 0, g["lf_mi_%d" % bd], 2) == g["dbl_%d" % bd]).all()
        cy, cu, cv = O.cdef_frame(g["lf_img_%d" % bd], g["cdef_u_%d" % bd], g["cdef_v_%d" % bd], bd, 4, ST, g["cdef_skip_%d" % bd])
        assert (cy == g["cdef_out_y_%d" % bd]).all() and (cu == g["cdef_out_u_%d" % bd]).all() and (cv == g["cdef_out_v_%d" % bd]).all()
        assert (O.lr_plane(cy, g["dbl_%d" % bd], bd, 0, 64, g["lr_units"]) == g["lr_out_%d" % bd]).all()
    ref = g["mc_ref"]
    i = 0
    for f in range(4):
        for (mx, my) in MVS:
            assert (O.mc_block(ref, 10, 24, 16, 8, 8, mx, my, f, (f + 1) % 4) == g["mc_8x8"][i]).all()
            assert (O.mc_block(ref, 10, 32, 40, 16, 4, mx, my, f, f) == g["mc_16x4"][i]).all()
            i += 1
    Y, U, V = synth.frames(64, 48, 2, 8, 3)
    k = O.intra_encode_frame(Y[0], U[0], V[0], 8, 8, 96)
    p = O.inter_encode_frame((Y[1], U[1], V[1]), (k["rec_y"], k["rec_u"], k["rec_v"]), 8, 96, 6)
    for name, r in (("key", k), ("p", p)):
        for key, val in r.items():
            assert (val == g["enc_%s_%s" % (name, key)]).all(), (name, key)


@pytest.mark.gpu
def test_gpu_reproduces_stage_goldens(ctx, av1mi):
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "av1-go_amd"))
    import synth
    g = np.load(G)
    # K3
    for bd in (8, 10):
        plane = g["intra_plane_%d" % bd]
        d_ref = ctx.to_device(plane)
        for ts, (bw, bh) in ((0, (4, 4)), (1, (8, 8)), (2, (16, 16)), (16, (32, 8))):
            exp = g["intra_%dx%d_%d" % (bw, bh, bd)]
            for i, (mode, delta, (nt, ntr, nl, nbl)) in enumerate(_intra_cases(bw, bh)):
                lst = np.zeros(1, av1mi.INTRA_BLK_DTYPE)
                lst[0] = (40, 40, mode, delta, (mode & 1) << 1, nt, ntr, nl, nbl, 0)
                d_l, d_dst = ctx.to_device(lst), ctx.to_device(np.zeros_like(plane))
                ctx.intra_pred_list(ts, d_ref, 96, d_dst, 96, bd, d_l, 1)
                got = d_dst.download(plane.shape, plane.dtype)[40:40 + bh, 40:40 + bw]
                d_l.free(); d_dst.free()
                assert (got == exp[i]).all(), ((bw, bh), bd, mode, delta)
        d_ref.free()
    # K4
    ref = g["mc_ref"]
    d_ref = ctx.to_device(ref)
    for sid, (w, h, x, y, key) in ((1, (8, 8, 24, 16, "mc_8x8")), (14, (16, 4, 32, 40, "mc_16x4"))):
        i = 0
        for f in range(4):
            for (mx, my) in MVS:
                lst = np.zeros(1, av1mi.MC_BLK_DTYPE)
                lst[0] = (x, y, mx, my, f, (f + 1) % 4 if key == "mc_8x8" else f, 0)
                d_l, d_dst = ctx.to_device(lst), ctx.to_device(np.zeros_like(ref))
                ctx.mc_list(sid, d_ref, 80, 80, 64, d_dst, 80, 10, d_l, 1)
                got = d_dst.download(ref.shape, ref.dtype)[y:y + h, x:x + w]
                d_l.free(); d_dst.free()
                assert (got == g[key][i]).all(), (key, f, mx, my)
                i += 1
    d_ref.free()
    # K5, K6, K7
    for bd in (8, 10):
        img, mi = g["lf_img_%d" % bd], g["lf_mi_%d" % bd]
        d_s, d_m, d_o = ctx.to_device(img), ctx.to_device(mi), ctx.to_device(np.zeros_like(img))
        ctx.deblock_plane(d_s, 96, d_o, 96, 96, 64, bd, 0, d_m, 24, 2)
        assert (d_o.download(img.shape, img.dtype) == g["dbl_%d" % bd]).all()
        for b in (d_s, d_m, d_o):
            b.free()
        got = ctx.cdef_arrays(img[None], g["cdef_u_%d" % bd][None], g["cdef_v_%d" % bd][None], bd, 4, ST[None], g["cdef_skip_%d" % bd][None])
        for a, k in zip(got, ("y", "u", "v")):
            assert (a[0] == g["cdef_out_%s_%d" % (k, bd)]).all()
        d_c, d_d, d_u = ctx.to_device(g["cdef_out_y_%d" % bd]), ctx.to_device(g["dbl_%d" % bd]), ctx.to_device(g["lr_units"])
        d_o = ctx.alloc(img.nbytes)
        ctx.lr_frames(d_c, d_d, d_o, 96, 96, 64, bd, 0, 64, d_u, 0, 1)
        assert (d_o.download(img.shape, img.dtype) == g["lr_out_%d" % bd]).all()
        for b in (d_c, d_d, d_u, d_o):
            b.free()
    # encoder loops
    Y, U, V = synth.frames(64, 48, 2, 8, 3)
    k = ctx.intra_encode_arrays(Y[:1], U[:1], V[:1], 8, 8, 96)
    for key in ("rec_y", "rec_u", "rec_v", "lev_y", "lev_u", "lev_v", "modes_y", "modes_uv"):
        assert (k[key][0] == g["enc_key_" + key]).all(), key
    p = ctx.inter_encode_arrays((Y[1:], U[1:], V[1:]), (k["rec_y"], k["rec_u"], k["rec_v"]), 8, 96, 6)
    for key in ("rec_y", "rec_u", "rec_v", "lev_y", "lev_u", "lev_v", "mvs", "skip"):
        assert (p[key][0] == g["enc_p_" + key]).all(), key
