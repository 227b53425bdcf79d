"""Host-side orchestration of the device pipeline over one closed-GOP segment (plumbing only).

All frames of a segment are stacked into one tall plane per component so that every stage is ONE
launch per plane per segment (a 1080p frame is ~20 us of HBM traffic per stage: per-frame launches
would be launch-bound).  Everything stays resident in HBM between stages; only the int16 levels are
meant to leave the device (host entropy coding, SURVEY.md §8a row H1 — not built yet).
"""
import numpy as np

import av1mi
import synth

TX_8X8 = 1


class Plane:
    def __init__(self, ctx, src, bd):
        self.h, self.w = src.shape
        self.n = src.size
        self.src = src
        self.bps = 1 if bd == 8 else 2
        self.d_resid = ctx.to_device((src.astype(np.int32) - (1 << (bd - 1))).astype(np.int16))
        self.d_pred = ctx.alloc(self.n * self.bps)      # prediction in, reconstruction out
        self.d_coef = ctx.alloc(self.n * 4)
        self.d_levels = ctx.alloc(self.n * 2)
        self.d_dq = ctx.alloc(self.n * 4)

    def free(self):
        for b in (self.d_resid, self.d_pred, self.d_coef, self.d_levels, self.d_dq):
            b.free()


class IntraPipeline:
    """v0 of BASELINE config 2 (1080p 8-bit intra-only): per 8x8 block, flat (no-neighbour DC)
    prediction -> residual -> forward DCT -> quantise -> dequantise -> inverse DCT -> reconstruct.
    Directional intra prediction and the in-loop filters are not in this pipeline yet."""

    STAGES = ("k_fwd_txfm<8,8>", "k_quantize", "k_dequantize", "k_inv_txfm_add<8,8>")

    def __init__(self, ctx, width, height, bd, frames, qindex, first_frame=0):
        self.ctx, self.bd, self.frames = ctx, bd, frames
        self.width, self.height = width, height
        ch = (height + 15) // 16 * 16   # coded height: multiple of 16 (8 for chroma)
        Y, U, V = synth.frames(width, height, frames, bd, first_frame)

        def stack(p, hh):
            pad = hh - p.shape[1]
            if pad:
                p = np.concatenate([p, np.repeat(p[:, -1:, :], pad, axis=1)], axis=1)
            return p.reshape(-1, p.shape[2])
        self.planes = [Plane(ctx, stack(Y, ch), bd), Plane(ctx, stack(U, ch // 2), bd), Plane(ctx, stack(V, ch // 2), bd)]
        lib = ctx.lib
        self.dc_q, self.ac_q = lib.av1mi_dc_q(qindex, bd), lib.av1mi_ac_q(qindex, bd)
        self.samples_per_frame = width * height * 3 // 2
        self.coded_samples = sum(p.n for p in self.planes)

    def describe(self):
        return ("%dx%d %d-bit 4:2:0 intra-only, 8x8 blocks: flat pred -> fwd DCT -> quant -> dequant -> inv DCT + recon "
                "(K1+K8+K2; intra prediction / loop filters not in the loop yet)" % (self.width, self.height, self.bd))

    # ---- stages, each one launch per plane over the whole segment
    def _pred(self, p):
        self.ctx.memset(p.d_pred, 128 if self.bd == 8 else 2, p.n * p.bps)  # 10-bit: 0x0202 = 514

    def _fwd(self, p):
        self.ctx.fwd_txfm_grid(TX_8X8, p.d_resid, p.w, p.d_coef, p.w // 8, p.n // 64)

    def _quant(self, p):
        self.ctx.quantize(p.d_coef, p.d_levels, None, p.n, 64, self.dc_q, self.ac_q, 0)

    def _dequant(self, p):
        self.ctx.dequantize(p.d_levels, p.d_dq, p.n, 64, self.dc_q, self.ac_q, 0, self.bd)

    def _inv(self, p):
        self.ctx.inv_txfm_add_grid(TX_8X8, p.d_dq, p.d_pred, p.w, self.bd, p.w // 8, p.n // 64)

    def step(self):
        for p in self.planes:
            self._pred(p)
            self._fwd(p)
            self._quant(p)
            self._dequant(p)
            self._inv(p)

    # ---- measurement
    def _time(self, fn, reps=10):
        p = self.planes[0]
        fn(p)
        self.ctx.sync()
        self.ctx.timer_begin()
        for _ in range(reps):
            fn(p)
        return self.ctx.timer_end() / reps

    def stage_times(self):
        """ms per launch on the luma plane of the segment (HIP events on the pipeline's stream)."""
        fns = (self._fwd, self._quant, self._dequant, self._inv)
        return {name: self._time(fn) for name, fn in zip(self.STAGES, fns)}

    def roofline(self, peak_gbps):
        b = self.planes[0].bps
        S = self.planes[0].n
        alg = {self.STAGES[0]: 6 * S, self.STAGES[1]: 6 * S, self.STAGES[2]: 6 * S, self.STAGES[3]: (4 + 2 * b) * S}
        times = self.stage_times()
        dom = max(times, key=times.get)
        ach = alg[dom] / (times[dom] * 1e-3) / 1e9
        return {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": peak_gbps, "unit": "GB/s", "frac": ach / peak_gbps,
                "traffic": None, "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": times[dom],
                "samples_per_launch": S}

    def close(self):
        for p in self.planes:
            p.free()
