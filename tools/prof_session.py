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
ap.add_argument("--sync", action="store_true", help="wait for the GPU after every submit: kernel durations without the next batch beside them")
ap.add_argument("--verbose", action="store_true", help="per-iteration host timing of fill / submit / collect")
ap.add_argument("--threads", type=int, default=16, help="host threads that fill the pinned input (like bench.py's end-to-end leg)")
args = ap.parse_args()
w, h = (int(x) for x in args.size.split("x"))
Y, U, V = synth.frames(w, h, args.frames, args.bd, 3)
ctx = av1mi.Context(0)
s = av1mi.GopSession(ctx, w, h, args.bd, 128, 30, args.segs, gpu_entropy=args.gpu_entropy)
from concurrent.futures import ThreadPoolExecutor
pool = ThreadPoolExecutor(args.threads)
t0 = time.perf_counter()
for t in range(args.frames):
    tf = time.perf_counter()
    planes = s.input_planes()
    jobs = []
    for sg in range(args.segs):
        jobs.append(pool.submit(np.copyto, planes[0][sg * h:(sg + 1) * h], Y[t]))
        jobs.append(pool.submit(np.copyto, planes[1][sg * h // 2:(sg + 1) * h // 2], U[t]))
        jobs.append(pool.submit(np.copyto, planes[2][sg * h // 2:(sg + 1) * h // 2], V[t]))
    for j in jobs:
        j.result()
    ta = time.perf_counter()
    s.submit()
    if args.sync:
        ctx.sync()
        time.sleep(0.01)     # the side stream's coder is not on the context's stream
    tb = time.perf_counter()
    if t >= s.max_in_flight() - 1:
        fr = s.collect()
    tc = time.perf_counter()
    if args.verbose:
        print("t %d fill %.2f submit %.2f collect %.2f ms" % (t, (ta - tf) * 1e3, (tb - ta) * 1e3, (tc - tb) * 1e3))
while s.pending():
    fr = s.collect()
if "tile_size" in fr:
    ts = fr["tile_size"].astype(np.int64)
    print("tile payload bytes (last batch): n %d mean %.0f p50 %.0f p90 %.0f p99 %.0f max %d" % (len(ts), ts.mean(), np.percentile(ts, 50), np.percentile(ts, 90), np.percentile(ts, 99), ts.max()))
ctx.sync()
dt = time.perf_counter() - t0
print("%d frames in %.3f s = %.1f frames/s" % (args.frames * args.segs, dt, args.frames * args.segs / dt))
s.close()
ctx.close()
