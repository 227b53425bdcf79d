"""experiment: open-loop intra mode decision — parity with the oracle's switch and speed against the closed loop (GPU box)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
av1mi = importlib.import_module("av1-go_amd.av1mi")
from oracle import oracle as O

ctx = av1mi.Context()
rng = np.random.default_rng(5)
for (w, h, bd, bs, q) in ((64, 48, 8, 8, 60), (96, 64, 10, 8, 128), (64, 64, 8, 16, 30), (128, 64, 10, 16, 200), (200, 136, 8, 8, 100), (1920, 1080, 10, 8, 90)):
    mx = (1 << bd) - 1
    yy, xx = np.mgrid[0:h, 0:w]
    Y = np.clip((xx * 3 + yy * 2) % (mx + 1) * 0.5 + rng.integers(0, mx // 4, (h, w)), 0, mx)
    U = np.clip(rng.integers(0, mx + 1, (h // 2, w // 2)), 0, mx); V = np.clip((yy[:h // 2, :w // 2] * 5) % (mx + 1), 0, mx)
    for ol in (False, True):
        g = ctx.intra_encode_arrays(Y[None], U[None], V[None], bd, bs, q, open_loop=ol)
        o = O.intra_encode_frame(Y, U, V, bd, bs, q, open_loop=ol)
        ok = all(np.array_equal(np.asarray(g[k])[0], o[k]) for k in o)
        print(w, h, bd, bs, q, "open" if ol else "closed", "OK" if ok else "MISMATCH " + str([k for k in o if not np.array_equal(np.asarray(g[k])[0], o[k])]), flush=True)
O.set_intra_open_loop(False)
