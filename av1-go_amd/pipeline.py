"""Host-side orchestration of the device pipeline over one closed-GOP segment (plumbing only).

All frames of a segment are stacked into one tall plane per component so that every stage is ONE launch per
segment (a 1080p frame is a few tens of microseconds of device work per stage: per-frame launches would be
launch-bound).  Everything stays resident in HBM between stages; only the int16 levels and the mode bytes are
meant to leave the device.  These classes drive the per-stage entry points of include/av1mi.h one by one: the parity tests
compare every stage with the oracle through them.  The PRODUCT's orchestration (and what bench.py times) is the GOP session in
the library (csrc/gop_session.hip, av1mi.GopSession), which adds the PCIe plumbing and the AV1 tile entropy coder.
"""
import numpy as np

import av1mi
import synth


def lf_mi_word(tx_w_log2, tx_h_log2, lvl_v, lvl_h, skip_inter=0, blk_left=1, blk_top=1):
    """one deblocking mode-info unit of av1mi_deblock_plane (include/av1mi.h): layout plumbing, not policy"""
    return (tx_w_log2 | (tx_h_log2 << 4) | (lvl_v << 8) | (lvl_h << 16) | (skip_inter << 24) | (blk_left << 25) | (blk_top << 26))


def frame_policy(qindex, bd, frame_type):
    """The encoder's filter-parameter policy for one frame.  It lives in libav1mi.so ONLY (csrc/gop_session.hip,
    av1mi_policy_frame_params): this module, host/backend.cpp and the tests all ask the library."""
    return av1mi.policy_frame_params(qindex, bd, frame_type)


def policy_arrays(qindex, bd, frame_type, width, height, block_size=8, visible=None):
    """host-side arrays of a frame's filter parameters in the layouts the stage entry points take.  visible: the true
    (width, height) when the coded size is rounded up to 8: deblocking units that start beyond it are marked "not filtered" (spec
    7.14.2 onScreen) and the restoration units tile the true frame."""
    p = frame_policy(qindex, bd, frame_type)
    l2y, l2c = int(np.log2(block_size)), int(np.log2(block_size // 2))
    nsb = ((height + 63) // 64) * ((width + 63) // 64)
    ur = lambda n: max(1, (n + p.lr_unit_size // 2) // p.lr_unit_size)
    vw, vh = visible if visible is not None else (width, height)
    mi_y = np.full((height // 4, width // 4), lf_mi_word(l2y, l2y, p.lf_level[0], p.lf_level[1]), np.uint32)
    mi_c = np.full((height // 8, width // 8), lf_mi_word(l2c, l2c, p.lf_level[2], p.lf_level[2]), np.uint32)
    off_y, off_c = lf_mi_word(l2y, l2y, 0, 0, 1, 0, 0), lf_mi_word(l2c, l2c, 0, 0, 1, 0, 0)     # skip && inter, no block edge: never filtered
    mi_y[(vh + 3) // 4:, :] = off_y; mi_y[:, (vw + 3) // 4:] = off_y                              # luma unit (r, c) starts at (4 r, 4 c)
    mi_c[(vh + 7) // 8:, :] = off_c; mi_c[:, (vw + 7) // 8:] = off_c                              # chroma unit (r, c) = luma (8 r, 8 c)
    return dict(params=p, mi_y=mi_y, mi_c=mi_c,
                cdef_damping=p.cdef_damping,
                cdef_sb=np.tile(np.array([p.cdef_y >> 2, p.cdef_y & 3, p.cdef_uv >> 2, p.cdef_uv & 3], np.uint8), (nsb, 1)),
                lr_unit=p.lr_unit_size,
                lr_units_y=np.tile(np.array(list(p.lr_unit_y), np.int8), (ur(height), ur(width), 1)),
                lr_units_c=np.tile(np.array(list(p.lr_unit_uv), np.int8), (ur(height // 2), ur(width // 2), 1)))


def extend_visible(plane, vw, vh):
    """the true last column / row of a plane replicated into the padding of the coded size (in place; plane: [..., h, w])"""
    plane[..., :, vw:] = plane[..., :, vw - 1:vw]
    plane[..., vh:, :] = plane[..., vh - 1:vh, :]
    return plane


class IntraPipeline:
    """BASELINE config 2 (intra-only, every frame a key frame) + the rest of the in-loop filter chain: per segment
         1 launch   k_intra_pipe  intra prediction + mode decision + fwd DCT + quant + dequant + inv DCT + recon
         3 launches k_deblock     deblocking of Y, U, V (both passes fused)         rec -> dbl
         1 launch   k_cdef        CDEF of the three planes                          dbl -> cdef
         3 launches k_lr          loop restoration (Wiener, fixed default taps)     cdef (+ dbl rows) -> out
    Filter parameters are fixed per segment by simple policies (no RD search)."""

    def __init__(self, ctx, width, height, bd, frames, qindex, first_frame=0, block_size=8):
        self.ctx, self.bd, self.frames, self.bs, self.qindex = ctx, bd, frames, block_size, qindex
        self.width, self.height = width, height
        Y, U, V = synth.frames(width, height, frames, bd, first_frame)
        self.src = (Y, U, V)
        self.bps = 1 if bd == 8 else 2
        nb = (height // block_size) * (width // block_size)
        self.d = {}
        for name, arr in (("src_y", Y), ("src_u", U), ("src_v", V)):
            self.d[name] = ctx.to_device(arr)
        for name, n in (("rec_y", Y.nbytes), ("rec_u", U.nbytes), ("rec_v", V.nbytes), ("dbl_y", Y.nbytes), ("dbl_u", U.nbytes),
                        ("dbl_v", V.nbytes), ("cdef_y", Y.nbytes), ("cdef_u", U.nbytes), ("cdef_v", V.nbytes),
                        ("out_y", Y.nbytes), ("out_u", U.nbytes), ("out_v", V.nbytes),
                        ("lev_y", Y.size * 2), ("lev_u", U.size * 2), ("lev_v", V.size * 2),
                        ("modes_y", frames * nb), ("modes_uv", frames * nb)):
            self.d[name] = ctx.alloc(n)
        job = av1mi.IntraJob(width, height, bd, frames, qindex, block_size, width, width // 2)
        for k in ("src_y", "src_u", "src_v", "rec_y", "rec_u", "rec_v", "lev_y", "lev_u", "lev_v", "modes_y", "modes_uv"):
            setattr(job, "d_" + k, self.d[k].ptr)
        self.job = job
        lib = ctx.lib
        self.dc_q, self.ac_q = lib.av1mi_dc_q(qindex, bd), lib.av1mi_ac_q(qindex, bd)
        # filter parameters: the library's policy (key-frame and inter-frame sets: the deblocking level differs for 8-bit)
        self.pol = [policy_arrays(qindex, bd, ft, width, height, block_size) for ft in (0, 1)]
        k = self.pol[0]
        self.lf_level = int(k["params"].lf_level[0])
        self.mi_y, self.mi_c = k["mi_y"], k["mi_c"]
        self.d["mi_y"], self.d["mi_c"] = ctx.to_device(self.mi_y), ctx.to_device(self.mi_c)
        self.d["mi_y_p"], self.d["mi_c_p"] = ctx.to_device(self.pol[1]["mi_y"]), ctx.to_device(self.pol[1]["mi_c"])
        self.d["cdef_sb_p"] = ctx.to_device(self.pol[1]["cdef_sb"])      # inter frames: their own CDEF strengths and damping
        self.samples = Y.size + U.size + V.size            # per step
        # CDEF: one strength set for every superblock, nothing skipped on key frames; LR: every unit Wiener with the default taps
        self.cdef_damping, self.cdef_sb = k["cdef_damping"], k["cdef_sb"]
        self.cdef_skip = np.zeros((height // 8, width // 8), np.uint8)
        self.lr_unit, self.lr_units_y, self.lr_units_c = k["lr_unit"], k["lr_units_y"], k["lr_units_c"]
        for name, a in (("cdef_sb", self.cdef_sb), ("cdef_skip", self.cdef_skip), ("lr_y", self.lr_units_y), ("lr_c", self.lr_units_c)):
            self.d[name] = ctx.to_device(a)
        d = self.d
        self.cdef_job = av1mi.CdefJob(width, height, bd, frames, self.cdef_damping, width, width // 2,
                                      d["dbl_y"].ptr, d["dbl_u"].ptr, d["dbl_v"].ptr, d["cdef_y"].ptr, d["cdef_u"].ptr,
                                      d["cdef_v"].ptr, d["cdef_sb"].ptr, 0, d["cdef_skip"].ptr, 0)

    def describe(self):
        return ("%dx%d %d-bit 4:2:0 intra-only (all key frames), tile = 64x64 superblock, %dx%d blocks: intra prediction "
                "(13 modes, SAD decision) + fwd DCT + quant + dequant + inv DCT + recon fused, then deblocking (level %d), "
                "CDEF (strengths %s, damping %d) and Wiener loop restoration (64x64 units, default taps); levels + modes stay in HBM"
                % (self.width, self.height, self.bd, self.bs, self.bs, self.lf_level, self.cdef_sb[0].tolist(), self.cdef_damping))

    def step(self):
        c, d = self.ctx, self.d
        w, h, f = self.width, self.height, self.frames
        c.intra_encode(self.job)
        c.deblock_frames(d["rec_y"], w, d["dbl_y"], w, w, h, self.bd, 0, d["mi_y"], w // 4, 0, 0, f)
        c.deblock_frames(d["rec_u"], w // 2, d["dbl_u"], w // 2, w // 2, h // 2, self.bd, 1, d["mi_c"], w // 8, 0, 0, f)
        c.deblock_frames(d["rec_v"], w // 2, d["dbl_v"], w // 2, w // 2, h // 2, self.bd, 1, d["mi_c"], w // 8, 0, 0, f)
        c.cdef_frames(self.cdef_job)
        c.lr_frames(d["cdef_y"], d["dbl_y"], d["out_y"], w, w, h, self.bd, 0, self.lr_unit, d["lr_y"], 0, f)
        c.lr_frames(d["cdef_u"], d["dbl_u"], d["out_u"], w // 2, w // 2, h // 2, self.bd, 1, self.lr_unit, d["lr_c"], 0, f)
        c.lr_frames(d["cdef_v"], d["dbl_v"], d["out_v"], w // 2, w // 2, h // 2, self.bd, 1, self.lr_unit, d["lr_c"], 0, f)

    def algorithmic_bytes(self):
        """per LAUNCH, by kernel kind: the fused coding kernel reads the source (b) and writes the
        reconstruction (b) and the int16 levels (2); deblocking reads and writes a plane (2b)."""
        b = self.bps
        return {"intra_pipeline": (2 * b + 2) * self.samples, "deblock": 2 * b * self.samples / 3.0,
                "cdef": 2 * b * self.samples, "loop_restoration": 2 * b * self.samples / 3.0}

    def download(self, frame=0):
        """outputs of one frame of the segment (tests / PSNR)"""
        out = {}
        for k, src in (("rec_y", 0), ("rec_u", 1), ("rec_v", 2), ("dbl_y", 0), ("dbl_u", 1), ("dbl_v", 2), ("cdef_y", 0),
                       ("cdef_u", 1), ("cdef_v", 2), ("out_y", 0), ("out_u", 1), ("out_v", 2)):
            a = self.src[src]
            out[k] = self.d[k].download(a.shape, a.dtype)[frame]
        return out

    def close(self):
        for b in self.d.values():
            b.free()


class GopPipeline:
    """BASELINE config 3: closed GOPs of `gop` frames (1 key frame + gop-1 P frames, single reference = the previous
    reconstructed, loop-filtered frame).  Frames inside a GOP are serially dependent, so the batch dimension is the
    SEGMENT: `segments` independent GOPs are coded in lockstep, the t-th frames of all of them stacked in one launch.
    Per step: gop x (coding launch(es) + 3 deblock + 1 CDEF + restoration with its ON / OFF decision)."""

    def __init__(self, ctx, width, height, bd, segments, gop, qindex, first_frame=0, search_range=8):
        self.ctx, self.bd, self.segments, self.gop, self.qindex, self.range = ctx, bd, segments, gop, qindex, search_range
        self.width, self.height, self.frames = width, height, segments * gop
        self.key = IntraPipeline(ctx, width, height, bd, segments, qindex, first_frame=first_frame)   # buffers + filter params
        # source: segment s holds frames first + s*gop .. ; re-stack as [t][s]
        Y, U, V = synth.frames(width, height, segments * gop, bd, first_frame)
        pick = lambda a, t: np.ascontiguousarray(a.reshape(segments, gop, *a.shape[1:])[:, t])
        self.src = [[pick(a, t) for a in (Y, U, V)] for t in range(gop)]
        k, d = self.key, self.key.d
        self.d_src = [[ctx.to_device(p) for p in self.src[t]] for t in range(gop)]
        nb = (height // 8) * (width // 8)
        self.d_mvs, self.d_skip = ctx.alloc(segments * nb * 4), ctx.alloc(segments * nb)
        self.sym = [dict(lev_y=d["lev_y"], lev_u=d["lev_u"], lev_v=d["lev_v"], modes_y=d["modes_y"], modes_uv=d["modes_uv"],
                         mvs=self.d_mvs, skip=self.d_skip)]
        self.d_ref = [ctx.alloc(self.src[0][i].nbytes) for i in range(3)]      # restored previous frame (see d_lr_on)
        # restoration ON / OFF per (segment, plane) of every frame index, decided on the GPU against the source (the session's policy:
        # av1mi_lr_frames_decide); frame t + 1 predicts from d_ref where ON and from the CDEF output where OFF
        self.d_lr_on = [ctx.to_device(np.ones(segments * 3 + 4, np.uint8)) for _ in range(gop)]      # (+ 4: read as aligned dwords)
        self.d_lr_scratch = ctx.alloc(ctx.lr_yuv_decide_scratch_bytes(height, segments))
        self.zero_skip = ctx.to_device(np.zeros(segments * nb, np.uint8))
        self.samples = sum(a.size for a in self.src[0]) * gop
        self.bps = k.bps

    def describe(self):
        return ("%dx%d %d-bit 4:2:0, %d closed GOPs of %d frames in lockstep (1 key + %d P frames, single reference, +-%d full "
                "search + half/quarter-pel refinement, 8x8 blocks), deblock + CDEF + Wiener LR + its on/off decision on every frame; "
                "symbols stay uncoded in HBM" % (self.width, self.height, self.bd, self.segments, self.gop, self.gop - 1, self.range))

    def _filters(self, skip_buf, skip_stride, key, t):
        c, k, d = self.ctx, self.key, self.key.d
        w, h, f = self.width, self.height, self.segments
        mi_y, mi_c = (d["mi_y"], d["mi_c"]) if key else (d["mi_y_p"], d["mi_c_p"])     # the policy's level depends on the frame type
        c.deblock_frames(d["rec_y"], w, d["dbl_y"], w, w, h, self.bd, 0, mi_y, w // 4, 0, 0, f)
        c.deblock_frames(d["rec_u"], w // 2, d["dbl_u"], w // 2, w // 2, h // 2, self.bd, 1, mi_c, w // 8, 0, 0, f)
        c.deblock_frames(d["rec_v"], w // 2, d["dbl_v"], w // 2, w // 2, h // 2, self.bd, 1, mi_c, w // 8, 0, 0, f)
        job = av1mi.CdefJob(w, h, self.bd, f, k.pol[0 if key else 1]["cdef_damping"], w, w // 2, d["dbl_y"].ptr, d["dbl_u"].ptr, d["dbl_v"].ptr,
                            d["cdef_y"].ptr, d["cdef_u"].ptr, d["cdef_v"].ptr, d["cdef_sb" if key else "cdef_sb_p"].ptr, 0, skip_buf.ptr, skip_stride)
        c.cdef_frames(job)
        s, on = self.d_src[t], self.d_lr_on[t]
        c.lr_yuv_decide(av1mi.LrDecideJob(w, h, self.bd, f, k.lr_unit, w, w // 2, d["cdef_y"].ptr, d["cdef_u"].ptr, d["cdef_v"].ptr,
                                          d["dbl_y"].ptr, d["dbl_u"].ptr, d["dbl_v"].ptr, self.d_ref[0].ptr, self.d_ref[1].ptr, self.d_ref[2].ptr,
                                          s[0].ptr, s[1].ptr, s[2].ptr, d["lr_y"].ptr, d["lr_c"].ptr, 0, 0, self.d_lr_scratch.ptr, on.ptr,
                                          int(k.lr_units_y[..., 0].max() < 2 and k.lr_units_c[..., 0].max() < 2)))

    def lr_on(self, t):
        """[segments, 3] restoration ON / OFF flags of the t-th frames after the last step"""
        self.ctx.sync()
        return self.d_lr_on[t].download((self.segments, 3), np.uint8)

    def step(self, on_frame=None):
        c, k, d = self.ctx, self.key, self.key.d
        w, h, f = self.width, self.height, self.segments
        nb = (h // 8) * (w // 8)
        for t in range(self.gop):
            s = self.d_src[t]
            y = self.sym[0]
            if t == 0:
                job = av1mi.IntraJob(w, h, self.bd, f, self.qindex, 8, w, w // 2, s[0].ptr, s[1].ptr, s[2].ptr, d["rec_y"].ptr,
                                     d["rec_u"].ptr, d["rec_v"].ptr, y["lev_y"].ptr, y["lev_u"].ptr, y["lev_v"].ptr, y["modes_y"].ptr,
                                     y["modes_uv"].ptr)
                c.intra_encode(job)
            else:
                job = av1mi.InterJob(w, h, self.bd, f, self.qindex, self.range, w, w // 2, s[0].ptr, s[1].ptr, s[2].ptr,
                                     self.d_ref[0].ptr, self.d_ref[1].ptr, self.d_ref[2].ptr, d["rec_y"].ptr, d["rec_u"].ptr,
                                     d["rec_v"].ptr, y["lev_y"].ptr, y["lev_u"].ptr, y["lev_v"].ptr, y["mvs"].ptr, y["skip"].ptr,
                                     d["cdef_y"].ptr, d["cdef_u"].ptr, d["cdef_v"].ptr, self.d_lr_on[t - 1].ptr)
                c.inter_encode(job)
            self._filters(self.zero_skip if t == 0 else y["skip"], 0 if t == 0 else nb, t == 0, t)
            if on_frame:
                on_frame(t)

    def oracle_filter_args(self, t):
        """(mi_y, mi_c, cdef_damping, cdef_sb, lr_unit, lr_units_y, lr_units_c) of the t-th frame of a GOP, for oracle chains"""
        p = self.key.pol[0 if t == 0 else 1]
        return p["mi_y"], p["mi_c"], p["cdef_damping"], p["cdef_sb"], p["lr_unit"], p["lr_units_y"], p["lr_units_c"]

    def algorithmic_bytes(self):
        b, per_frame = self.bps, self.samples / self.gop
        # k_me_int reads the luma source and the luma reference (2b per LUMA sample = 2/3 of the frame's samples); k_inter_pipe
        # reads source + reference, writes reconstruction + int16 levels ((3b + 2) per sample)
        return {"intra_pipeline": (2 * b + 2) * per_frame, "inter_pipeline": (3 * b + 2) * per_frame, "me_integer": 2 * b * per_frame * 2.0 / 3.0,
                "deblock": 2 * b * per_frame / 3.0,
                "cdef": 2 * b * per_frame, "loop_restoration": 2 * b * per_frame / 3.0}

    def close(self):
        for t in self.d_src:
            for b in t:
                b.free()
        for b in [self.d_mvs, self.d_skip, self.zero_skip] + self.d_ref + self.d_lr_on + [self.d_lr_scratch]:
            b.free()
        self.key.close()
