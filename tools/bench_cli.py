#!/usr/bin/env python3
"""The product's own command line end to end: a synthetic Y4M file on disk -> av1mi_transcode (RunTranscode: file reader, GOP
session, GPU tile entropy coder, Matroska muxer) -> an .mkv file; wall-clock frames/s, file read and write included.
    python tools/bench_cli.py [--size 3840x2160] [--bd 10] [--frames 120] [--segments 8] [--gpu-entropy 1]"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av1-go_amd"))
import synth   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="3840x2160")
ap.add_argument("--bd", type=int, default=10)
ap.add_argument("--frames", type=int, default=120)
ap.add_argument("--segments", type=int, default=8)
ap.add_argument("--gop", type=int, default=30)
ap.add_argument("--gpu-entropy", type=int, default=1)
ap.add_argument("--quality", type=int, default=128)
args = ap.parse_args()
w, h = (int(x) for x in args.size.split("x"))
exe = os.path.join(ROOT, "av1-go_amd", "host", "av1mi_transcode")
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    src, out = os.path.join(d, "clip.y4m"), os.path.join(d, "clip.av1-tmp.mkv")
    t0 = time.perf_counter()
    # 30 distinct synthetic frames, repeated: generating 4K frames takes ~0.4 s each, and the encoder does not care
    base = min(args.frames, args.gop)
    Y, U, V = synth.frames(w, h, base, args.bd, 0)
    with open(src, "wb") as f:
        f.write(("YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C%s\n" % (w, h, "420p10" if args.bd == 10 else "420jpeg")).encode())
        for t in range(args.frames):
            f.write(b"FRAME\n")
            for p in (Y, U, V):
                f.write(np.ascontiguousarray(p[t % base]).tobytes())
    t_gen = time.perf_counter() - t0
    cmd = [exe, "-hide_banner", "-i", src, "-global_quality:v:0", str(args.quality), "-g", str(args.gop), "-av1mi_segments", str(args.segments),
           "-av1mi_gpu_entropy", str(args.gpu_entropy), out]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    if r.returncode:
        print(r.stderr[-800:])
        sys.exit(r.returncode)
    print("av1mi_transcode %dx%d %d-bit, %d frames (source %.2f GB written in %.1f s): %.2f s wall = %.1f frames/s, output %.1f MB"
          % (w, h, args.bd, args.frames, os.path.getsize(src) / 1e9, t_gen, dt, args.frames / dt, os.path.getsize(out) / 1e6))
