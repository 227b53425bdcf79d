#!/usr/bin/env python3
"""drives the GOP session alone (no bench harness) for profiling: python tools/prof_session.py [--gpu-entropy 1] [--segs 8] [--frames 10]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av1-go_amd"))
import av1mi   # noqa: E402
import synth   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpu-entropy", type=int, default=1)
ap.add_argument("--segs", type=int, default=8)
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--size", default="3840x2160")
ap.add_argument("--bd", type=int, default=10)
args = ap.parse_args()
w, h = (int(x) for x in args.size.split("x"))
Y, U, V = synth.frames(w, h, args.frames, args.bd, 3)
ctx = av1mi.Context(0)
s = av1mi.GopSession(ctx, w, h, args.bd, 128, 30, args.segs, gpu_entropy=args.gpu_entropy)
t0 = time.perf_counter()
for t in range(args.frames):
    planes = s.input_planes()
    for sg in range(args.segs):
        planes[0][sg * h:(sg + 1) * h] = Y[t]
        planes[1][sg * h // 2:(sg + 1) * h // 2] = U[t]
        planes[2][sg * h // 2:(sg + 1) * h // 2] = V[t]
    s.submit()
    if t >= 1:
        s.collect()
s.collect()
ctx.sync()
dt = time.perf_counter() - t0
print("%d frames in %.3f s = %.1f frames/s" % (args.frames * args.segs, dt, args.frames * args.segs / dt))
s.close()
ctx.close()
