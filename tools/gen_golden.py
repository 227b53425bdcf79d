#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/liboracle.so).

These are regression pins of the ORACLE ITSELF (the reference holds no golden
vectors, "parity unpinned"); they are inputs + expected outputs only, seeded and small.
Run:  python tools/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def coef_cases(rng, ts, bd):
    ch, cw = O.coef_shape(ts)
    lim = 1 << (bd + 7)
    cases = []
    z = np.zeros((ch, cw), np.int32)
    dc = z.copy(); dc[0, 0] = 1000; cases.append(dc)
    ac = z.copy(); ac[min(1, ch - 1), min(2, cw - 1)] = -777; cases.append(ac)
    cases.append(np.full((ch, cw), lim - 1, np.int32))
    cases.append(np.full((ch, cw), -lim, np.int32))
    alt = ((np.indices((ch, cw)).sum(0) & 1) * 2 - 1).astype(np.int32) * (lim // 3); cases.append(alt)
    for _ in range(3):
        cases.append(rng.integers(-lim // 8, lim // 8, (ch, cw)).astype(np.int32))
    return cases


def gen_txfm():
    rng = np.random.default_rng(0xA51C0DE)
    d = {}
    for ts in range(19):
        for tt in range(16):
            if not O.txfm_valid(ts, tt):
                continue
            for bd in (8, 10):
                h, w = O.TX_H[ts], O.TX_W[ts]
                pred = rng.integers(0, 1 << bd, (h, w)).astype(np.uint16)
                cs = coef_cases(rng, ts, bd)
                # keep the file small: all cases for DCT_DCT, two for the other types
                if tt != 0:
                    cs = [cs[1], cs[5]]
                recs = np.stack([O.inv_txfm2d_add(c, pred, ts, tt, bd, 1) for c in cs]).astype(np.uint16)
                key = "inv_%d_%d_%d" % (ts, tt, bd)
                d[key + "_coef"] = np.stack(cs)
                d[key + "_pred"] = pred
                d[key + "_rec"] = recs
            res = rng.integers(-255, 256, (O.TX_H[ts], O.TX_W[ts])).astype(np.int16)
            d["fwd_%d_%d_res" % (ts, tt)] = res
            d["fwd_%d_%d_coef" % (ts, tt)] = O.fwd_txfm2d(res, ts, tt)
    np.savez_compressed(os.path.join(OUT, "txfm_kat.npz"), **d)
    print("txfm_kat.npz:", len(d), "arrays")


def gen_stages():
    """small seeded inputs + oracle outputs for K3 (intra), K4 (MC), K5 (deblock), K6 (CDEF), K7 (LR) and the two encoder loops"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "av1-go_amd"))
    from lf_util import random_mi, test_image
    import synth
    rng = np.random.default_rng(0xA51C0DE + 1)
    d = {}
    # K3: every mode x delta on 4x4 / 8x8 / 16x16 / 32x8, two availability patterns, 8- and 10-bit
    for bd in (8, 10):
        plane = rng.integers(0, 1 << bd, (96, 96)).astype(np.uint8 if bd == 8 else np.uint16)
        d["intra_plane_%d" % bd] = plane
        for (bw, bh) in ((4, 4), (8, 8), (16, 16), (32, 8)):
            outs = []
            for mode in range(13):
                for delta in ((-3, 0, 2) if 1 <= mode <= 8 else (0,)):
                    for (nt, ntr, nl, nbl) in ((bw, bw, bh, bh), (bw, 0, bh, 0)):
                        outs.append(O.intra_predict(plane, 40, 40, bw, bh, mode, delta, bd, nt, ntr, nl, nbl, 0, mode & 1))
            d["intra_%dx%d_%d" % (bw, bh, bd)] = np.stack(outs)
    # K4: 8x8 and 16x4 blocks, all three filter families, assorted phases incl. far-outside vectors
    ref = rng.integers(0, 1024, (64, 80)).astype(np.uint16)
    d["mc_ref"] = ref
    mvs = [(0, 0), (8, 8), (5, -11), (-37, 21), (255, 3), (-400, -400), (1500, 700), (15, 1)]
    d["mc_mvs"] = np.array(mvs, np.int32)
    d["mc_8x8"] = np.stack([O.mc_block(ref, 10, 24, 16, 8, 8, mx, my, f, (f + 1) % 4) for f in range(4) for (mx, my) in mvs])
    d["mc_16x4"] = np.stack([O.mc_block(ref, 10, 32, 40, 16, 4, mx, my, f, f) for f in range(4) for (mx, my) in mvs])
    # K5 / K6 / K7 on one 64x96 picture
    for bd in (8, 10):
        img = test_image(rng, 64, 96, bd)
        mi = random_mi(rng, O, 64, 96, 0)
        d["lf_img_%d" % bd], d["lf_mi_%d" % bd] = img, mi
        d["dbl_%d" % bd] = O.deblock_plane(img, bd, 0, mi, 2)
        u, v = test_image(rng, 32, 48, bd), test_image(rng, 32, 48, bd)
        st = np.array([[9, 2, 5, 3], [4, 1, 0, 2]], np.uint8)
        skip = (rng.random((8, 12)) < 0.2).astype(np.uint8)
        cy, cu, cv = O.cdef_frame(img, u, v, bd, 4, st, skip)
        d["cdef_u_%d" % bd], d["cdef_v_%d" % bd], d["cdef_st"], d["cdef_skip_%d" % bd] = u, v, st, skip
        d["cdef_out_y_%d" % bd], d["cdef_out_u_%d" % bd], d["cdef_out_v_%d" % bd] = cy, cu, cv
        units = np.zeros((1, 2, 8), np.int8)
        units[0, 0] = O.lr_unit_wiener((3, -7, 15), (-2, 6, 11)); units[0, 1] = O.lr_unit_sgr(4, -20, 60)
        d["lr_units"] = units
        d["lr_out_%d" % bd] = O.lr_plane(cy, d["dbl_%d" % bd], bd, 0, 64, units)
    # encoder loops on a 64x48 clip
    Y, U, V = synth.frames(64, 48, 2, 8, 3)
    k = O.intra_encode_frame(Y[0], U[0], V[0], 8, 8, 96)
    p = O.inter_encode_frame((Y[1], U[1], V[1]), (k["rec_y"], k["rec_u"], k["rec_v"]), 8, 96, 6)
    for name, r in (("key", k), ("p", p)):
        for key, val in r.items():
            d["enc_%s_%s" % (name, key)] = val
    np.savez_compressed(os.path.join(OUT, "stages_kat.npz"), **d)
    print("stages_kat.npz:", len(d), "arrays")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    O.build()
    gen_txfm()
    gen_stages()
