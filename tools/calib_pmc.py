#!/usr/bin/env python3
"""PMC calibration workload (run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`): kernels with KNOWN byte counts and
the access widths the pipeline uses, on buffers larger than the 256 MiB Infinity Cache (MI355X_MICROARCH.md §HBM asks
for a calibration per access pattern):
   k_quantize    reads 16 B/lane (int4 of int32 coefficients), writes 8 B/lane (int16 levels)
   k_dequantize  reads  8 B/lane (int16 levels),               writes 16 B/lane (int32)
n = 128 Mi coefficients: quantize reads 512 MiB, writes 256 MiB; dequantize reads 256 MiB, writes 512 MiB."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av1-go_amd"))
import av1mi
ctx = av1mi.Context(0)
n = 128 << 20
d_coef, d_lev, d_dq = ctx.alloc(n * 4), ctx.alloc(n * 2), ctx.alloc(n * 4)
ctx.memset(d_coef, 1, n * 4)
for _ in range(3):
    ctx.quantize(d_coef, d_lev, None, n, 64, 100, 120, 0)
    ctx.dequantize(d_lev, d_dq, n, 64, 100, 120, 0, 8)
ctx.sync()
print("calibration launches done: n =", n)
