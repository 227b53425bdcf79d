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


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    O.build()
    gen_txfm()
