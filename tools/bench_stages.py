#!/usr/bin/env python3
"""Per-row measurement (SURVEY.md §8a rows K1-K9 + the two fused pipelines): every kernel ALONE on 1080p-sized inputs resident
in HBM, HIP events on the context's stream (the C ABI's per-kernel profile), algorithmic bytes per launch from SURVEY §8d /
DESIGN §3, against the 8 TB/s HBM3E peak.  bench.py reports the pipeline the metric is quoted on; this is the table behind
DESIGN.md §3's "bound today" column.  Usage (GPU box):  python tools/bench_stages.py [--frames 16] [--json profiles/x.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av1-go_amd"))
import av1mi      # noqa: E402
import pipeline   # noqa: E402

PEAK = 8000.0


def timed(ctx, fn, reps=5):
    fn()
    ctx.sync()
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(reps):
        fn()
    ctx.sync()
    ctx.prof_enable(False)
    p = ctx.prof_get()
    return {k: v[1] / v[0] for k, v in p.items()}      # avg ms per launch by kind


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    ctx = av1mi.Context(0)
    rows = []

    def add(row, kernel, ms, alg_bytes, what):
        gbps = alg_bytes / (ms * 1e-3) / 1e9
        rows.append({"row": row, "kernel": kernel, "workload": what, "avg_ms": ms, "algorithmic_bytes": alg_bytes,
                     "GBps": gbps, "frac_of_8TBps": gbps / PEAK})
        print("%-4s %-16s %8.3f ms  %8.1f GB/s  %5.1f %%   %s" % (row, kernel, ms, gbps, 100 * gbps / PEAK, what), flush=True)

    rng = np.random.default_rng(1)
    F, W, H = a.frames, 1920, 1088
    S = F * W * H                      # luma samples
    for bd in (8, 10):
        dt = np.uint8 if bd == 8 else np.uint16
        b = 1 if bd == 8 else 2
        nb8 = S // 64
        # K1 / K2 / K8 on 8x8 DCT_DCT blocks covering F luma planes
        res = rng.integers(-255, 256, (F * H, W)).astype(np.int16)
        d_res, d_coef = ctx.to_device(res), ctx.alloc(S * 4)
        if bd == 8:
            ms = timed(ctx, lambda: ctx.fwd_txfm_grid(1, d_res, W, d_coef, W // 8, nb8))["fwd_txfm"]
            add("K1", "k_fwd_txfm 8x8", ms, S * 6, "%d x 1080p residual int16 -> int32 coefficients" % F)
            d_lev, d_dq = ctx.alloc(S * 2), ctx.alloc(S * 4)
            ms = timed(ctx, lambda: ctx.quantize(d_coef, d_lev, None, S, 64, 60, 70, 0))["quantize"]
            add("K8", "k_quantize", ms, S * 6, "int32 -> int16 levels")
            ms = timed(ctx, lambda: ctx.dequantize(d_lev, d_dq, S, 64, 60, 70, 0, 8))["dequantize"]
            add("K8", "k_dequantize", ms, S * 6, "int16 levels -> int32")
            d_lev.free(); d_dq.free()
        pred = rng.integers(0, 1 << bd, (F * H, W)).astype(dt)
        d_plane = ctx.to_device(pred)
        ms = timed(ctx, lambda: ctx.inv_txfm_add_grid(1, d_coef, d_plane, W, bd, W // 8, nb8))["inv_txfm"]
        add("K2", "k_inv_txfm_add 8x8", ms, S * (4 + 2 * b), "%d-bit, coefficients + prediction -> reconstruction" % bd)
        for x in (d_res, d_coef):
            x.free()
        # K3: every 8x8 block of F planes, directional modes with deltas (list form)
        ys, xs = np.mgrid[8:F * H - 8:8, 8:W - 8:8]
        n = ys.size
        lst = np.zeros(n, av1mi.INTRA_BLK_DTYPE)
        lst["x"], lst["y"] = xs.ravel() % W, ys.ravel()
        lst["mode"] = rng.integers(1, 9, n)
        lst["angle_delta"] = rng.integers(-3, 4, n)
        lst["n_top"] = lst["n_topright"] = lst["n_left"] = lst["n_bottomleft"] = 8
        d_l, d_dst = ctx.to_device(lst), ctx.alloc(pred.nbytes)
        ms = timed(ctx, lambda: ctx.intra_pred_list(1, d_plane, W, d_dst, W, bd, d_l, n))["intra_pred"]
        add("K3", "k_intra_pred 8x8", ms, n * 64 * b + n * 33 * b, "%d-bit directional, random angle deltas, %d blocks" % (bd, n))
        # K4: every 8x8 block, random sub-pel vectors, regular filter
        mc = np.zeros(n, av1mi.MC_BLK_DTYPE)
        mc["x"], mc["y"] = lst["x"], lst["y"]
        mc["mvx"], mc["mvy"] = rng.integers(-64, 65, n), rng.integers(-64, 65, n)
        d_m = ctx.to_device(mc)
        ms = timed(ctx, lambda: ctx.mc_list(1, d_plane, W, W, F * H, d_dst, W, bd, d_m, n))["mc"]
        add("K4", "k_mc 8x8", ms, n * 64 * 2 * b, "%d-bit, 8-tap regular, random 1/16 vectors, %d blocks" % (bd, n))
        # K3+: chroma-from-luma on every 8x8 chroma block of F half-size planes (luma = the same random plane)
        cw, chh = W // 2, F * H // 2
        cys, cxs = np.mgrid[0:chh:8, 0:cw:8]
        nc = cys.size
        cfl = np.zeros(nc, av1mi.CFL_BLK_DTYPE)
        cfl["x"], cfl["y"], cfl["max_luma_w"], cfl["max_luma_h"] = cxs.ravel(), cys.ravel() % 65536, W, min(F * H, 65535)
        cfl["alpha_q3"] = rng.integers(-16, 17, nc)
        keep = cys.ravel() * 2 + 16 <= 65535           # the descriptor holds 16-bit coordinates
        cfl = cfl[keep]
        d_c, d_cd = ctx.to_device(cfl), ctx.alloc(cw * chh * b)
        ms = timed(ctx, lambda: ctx.cfl_pred_list(1, d_plane, W, d_cd, cw, bd, d_c, len(cfl)))["intra_pred"]
        add("K3+", "k_cfl_pred 8x8", ms, len(cfl) * 64 * (2 * b + 4 * b), "%d-bit chroma-from-luma, %d blocks" % (bd, len(cfl)))
        for x in (d_l, d_dst, d_m, d_plane, d_c, d_cd):
            x.free()
        # K5 / K6 / K7 / fused pipelines / K9 through the segment pipeline's buffers
        pipe = pipeline.IntraPipeline(ctx, 1920, 1080, bd, F, 128)
        t = timed(ctx, pipe.step)
        alg = pipe.algorithmic_bytes()
        add("pipe", "k_intra_pipe", t["intra_pipeline"], alg["intra_pipeline"], "%d-bit %d x 1080p intra coding loop" % (bd, F))
        add("K5", "k_deblock", t["deblock"], alg["deblock"], "%d-bit, avg of Y/U/V launches" % bd)
        add("K6", "k_cdef", t["cdef"], alg["cdef"], "%d-bit Y+U+V" % bd)
        add("K7", "k_lr (Wiener)", t["loop_restoration"], alg["loop_restoration"], "%d-bit, avg of Y/U/V launches" % bd)
        pipe.close()
        gp = pipeline.GopPipeline(ctx, 1920, 1080, bd, F, 2, 128)
        t = timed(ctx, gp.step, reps=3)
        add("pipe", "k_me_int", t["me_integer"], gp.algorithmic_bytes()["me_integer"], "%d-bit %d P frames, +-8 integer search" % (bd, F))
        add("pipe", "k_inter_pipe", t["inter_pipeline"], gp.algorithmic_bytes()["inter_pipeline"], "%d-bit %d P frames, sub-pel refinement + residual coding" % (bd, F))
        gp.close()
    ctx.close()
    if a.json:
        json.dump({"frames": F, "peak_GBps": PEAK, "rows": rows}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
