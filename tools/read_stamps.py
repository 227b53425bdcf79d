#!/usr/bin/env python3
"""Diagnostic: run one segment through the -DAV1MI_STAMPS build (AV1MI_LIB=tools/_diag/libav1mi_stamps.so) and print the
share of wave time per phase of k_intra_pipe (lane 0 of one wave; shares, not absolute times)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av1-go_amd"))
import av1mi, pipeline
ctx = av1mi.Context(0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
p = pipeline.IntraPipeline(ctx, 1920, 1080, 8, frames, 128)
p.step(); ctx.sync()
out = (C.c_ulonglong * 8)()
rc = ctx.lib.av1mi_debug_read_stamps(out)
names = ["src load + setup", "edge build (2 LDS phases)", "11 mode predictions + SAD", "residual: fwd/quant/dequant/inv", "stores + context", "loop tail"]
tot = sum(out[:6])
print("rc", rc, "total cycles", tot, "per block-step", tot / 64)
for n, v in zip(names, out[:6]):
    print("  %-36s %10d  %5.1f%%" % (n, v, 100.0 * v / tot))
