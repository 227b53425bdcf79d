"""ctypes binding of libav1mi.so — the same C ABI (include/av1mi.h) a cgo wrapper binds.

Used by tests/, bench.py and __graft_entry__.py.  It is plumbing, not a second implementation:
there is no CPU fallback here.  If the HIP library is missing or no GPU is present, loading /
opening fails loudly (reference behaviour for an unusable encoder: RunTranscode returns
(-1, err), internal/ffmpeg/transcode.go:311).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AV1MI_LIB") or os.path.join(_HERE, "libav1mi.so")   # AV1MI_LIB: diagnostic builds only

TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]


class Av1miError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("av1mi error %d: %s" % (code, text))
        self.code = code


class TxBlock(C.Structure):
    _fields_ = [("coef_off", C.c_uint32), ("x", C.c_uint16), ("y", C.c_uint16), ("tx_type", C.c_uint32),
                ("reserved", C.c_uint32)]


TXB_DTYPE = np.dtype([("coef_off", "<u4"), ("x", "<u2"), ("y", "<u2"), ("tx_type", "<u4"), ("reserved", "<u4")])

INTRA_BLK_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("mode", "u1"), ("angle_delta", "i1"), ("flags", "u1"),
                            ("n_top", "u1"), ("n_topright", "u1"), ("n_left", "u1"), ("n_bottomleft", "u1"),
                            ("reserved", "u1", (5,))])
assert INTRA_BLK_DTYPE.itemsize == 16 and TXB_DTYPE.itemsize == 16

CFL_BLK_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("max_luma_w", "<u2"), ("max_luma_h", "<u2"), ("alpha_q3", "i1"),
                          ("reserved", "u1", (7,))])
assert CFL_BLK_DTYPE.itemsize == 16

MC_BLK_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("mvx", "<i2"), ("mvy", "<i2"), ("filt_x", "u1"), ("filt_y", "u1"),
                         ("reserved", "u1", (6,))])
assert MC_BLK_DTYPE.itemsize == 16


class CdefJob(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("bit_depth", C.c_int), ("nframes", C.c_int), ("damping", C.c_int),
                ("stride_y", C.c_int), ("stride_uv", C.c_int),
                ("d_src_y", C.c_void_p), ("d_src_u", C.c_void_p), ("d_src_v", C.c_void_p),
                ("d_dst_y", C.c_void_p), ("d_dst_u", C.c_void_p), ("d_dst_v", C.c_void_p),
                ("d_sb_strength", C.c_void_p), ("sb_frame_stride", C.c_size_t),
                ("d_skip8", C.c_void_p), ("skip_frame_stride", C.c_size_t)]


class IntraJob(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("bit_depth", C.c_int), ("nframes", C.c_int), ("qindex", C.c_int),
                ("block_size", C.c_int), ("stride_y", C.c_int), ("stride_uv", C.c_int),
                ("d_src_y", C.c_void_p), ("d_src_u", C.c_void_p), ("d_src_v", C.c_void_p),
                ("d_rec_y", C.c_void_p), ("d_rec_u", C.c_void_p), ("d_rec_v", C.c_void_p),
                ("d_lev_y", C.c_void_p), ("d_lev_u", C.c_void_p), ("d_lev_v", C.c_void_p),
                ("d_modes_y", C.c_void_p), ("d_modes_uv", C.c_void_p), ("open_loop", C.c_int), ("frame_rows", C.c_int),
                ("modes_frame_stride", C.c_int)]


class InterJob(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("bit_depth", C.c_int), ("nframes", C.c_int), ("qindex", C.c_int),
                ("search_range", C.c_int), ("stride_y", C.c_int), ("stride_uv", C.c_int),
                ("d_src_y", C.c_void_p), ("d_src_u", C.c_void_p), ("d_src_v", C.c_void_p),
                ("d_ref_y", C.c_void_p), ("d_ref_u", C.c_void_p), ("d_ref_v", C.c_void_p),
                ("d_rec_y", C.c_void_p), ("d_rec_u", C.c_void_p), ("d_rec_v", C.c_void_p),
                ("d_lev_y", C.c_void_p), ("d_lev_u", C.c_void_p), ("d_lev_v", C.c_void_p),
                ("d_mvs", C.c_void_p), ("d_skip", C.c_void_p),
                ("d_ref_alt_y", C.c_void_p), ("d_ref_alt_u", C.c_void_p), ("d_ref_alt_v", C.c_void_p), ("d_ref_sel", C.c_void_p)]


class LrDecideJob(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("bit_depth", C.c_int), ("nframes", C.c_int), ("unit_size", C.c_int),
                ("stride_y", C.c_int), ("stride_uv", C.c_int),
                ("d_cdef_y", C.c_void_p), ("d_cdef_u", C.c_void_p), ("d_cdef_v", C.c_void_p),
                ("d_dbl_y", C.c_void_p), ("d_dbl_u", C.c_void_p), ("d_dbl_v", C.c_void_p),
                ("d_out_y", C.c_void_p), ("d_out_u", C.c_void_p), ("d_out_v", C.c_void_p),
                ("d_orig_y", C.c_void_p), ("d_orig_u", C.c_void_p), ("d_orig_v", C.c_void_p),
                ("d_units_y", C.c_void_p), ("d_units_uv", C.c_void_p), ("unit_frame_stride_y", C.c_size_t), ("unit_frame_stride_uv", C.c_size_t),
                ("d_scratch", C.c_void_p), ("d_on", C.c_void_p), ("no_self_guided_units", C.c_int)]


class GopConfig(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("bit_depth", C.c_int), ("base_q_idx", C.c_int), ("gop_length", C.c_int),
                ("segments", C.c_int), ("search_range", C.c_int), ("gpu_entropy", C.c_int), ("visible_width", C.c_int),
                ("visible_height", C.c_int), ("coder_streams", C.c_int), ("key_block_size", C.c_int)]


class FrameParams(C.Structure):
    _fields_ = [("frame_type", C.c_int), ("base_q_idx", C.c_int), ("lf_level", C.c_int * 4), ("lf_sharpness", C.c_int),
                ("cdef_damping", C.c_int), ("cdef_y", C.c_uint8), ("cdef_uv", C.c_uint8), ("lr_unit_size", C.c_int),
                ("lr_unit_y", C.c_int8 * 8), ("lr_unit_uv", C.c_int8 * 8)]


class GopFrame(C.Structure):
    _fields_ = [("params", FrameParams), ("segments", C.c_int), ("blocks_per_frame", C.c_size_t), ("y_mode", C.c_void_p),
                ("uv_mode", C.c_void_p), ("mv", C.c_void_p), ("skip", C.c_void_p), ("lev_y", C.c_void_p), ("lev_u", C.c_void_p),
                ("lev_v", C.c_void_p), ("tiles_per_frame", C.c_int), ("tile_size", C.c_void_p), ("tile_payload", C.c_void_p),
                ("payload_bytes", C.c_uint64), ("lr_on", C.c_void_p), ("key_block_size", C.c_int), ("key_modes_stride", C.c_int),
                ("key_modes_band", C.c_int)]


def policy_frame_params(base_q_idx, bit_depth, frame_type):
    """the session's filter-parameter policy for one frame (include/av1mi.h av1mi_policy_frame_params); no GPU needed"""
    p = FrameParams()
    rc = load().av1mi_policy_frame_params(int(base_q_idx), int(bit_depth), int(frame_type), C.byref(p))
    if rc:
        raise Av1miError(rc, "av1mi_policy_frame_params")
    return p


def _view(ptr, shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    return np.frombuffer((C.c_uint8 * n).from_address(ptr), dtype=dtype).reshape(shape)


class GopSession:
    """av1mi_gop_* (include/av1mi.h): closed GOPs in lockstep, policy and PCIe plumbing inside the library."""

    def __init__(self, ctx, width, height, bit_depth, base_q_idx, gop_length, segments=1, search_range=8, gpu_entropy=0, visible=None, coder_streams=0,
                 key_block_size=0):
        """visible: the true (width, height) when width x height is it rounded up to 8 (the caller replicates the source edge);
        key_block_size 32: key frames in 32x32 blocks (av1mi_gop_config.key_block_size)"""
        self.ctx, self.w, self.h, self.bd, self.segments = ctx, width, height, bit_depth, segments
        vw, vh = visible if visible is not None else (0, 0)
        self.cfg = GopConfig(width, height, bit_depth, base_q_idx, gop_length, segments, search_range, gpu_entropy, vw, vh, coder_streams, key_block_size)
        self.g = C.c_void_p()
        ctx.lib.av1mi_gop_open.argtypes = [C.c_void_p, C.POINTER(GopConfig), C.POINTER(C.c_void_p)]
        ctx._chk(ctx.lib.av1mi_gop_open(ctx.h, C.byref(self.cfg), C.byref(self.g)))
        for name in ("av1mi_gop_close", "av1mi_gop_acquire_input", "av1mi_gop_submit", "av1mi_gop_collect", "av1mi_gop_pending",
                     "av1mi_gop_download_reference"):
            getattr(ctx.lib, name).argtypes = None
        ctx.lib.av1mi_gop_close.restype = None
        self.dt = np.uint8 if bit_depth == 8 else np.uint16

    def input_planes(self):
        """numpy views of the pinned host planes of the next batch: shapes [segments * height, width] and the half-size chroma"""
        y, u, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.ctx._chk(self.ctx.lib.av1mi_gop_acquire_input(self.g, C.byref(y), C.byref(u), C.byref(v)))
        S, w, h = self.segments, self.w, self.h
        return (_view(y.value, (S * h, w), self.dt), _view(u.value, (S * h // 2, w // 2), self.dt), _view(v.value, (S * h // 2, w // 2), self.dt))

    def submit(self, frame_type=-1):
        self.ctx._chk(self.ctx.lib.av1mi_gop_submit(self.g, int(frame_type)))

    def submit_device(self, d_y, d_u, d_v, frame_type=-1):
        """a batch whose source planes (DevBuf) are already in device memory: no upload (av1mi_gop_submit_device)"""
        self.ctx.lib.av1mi_gop_submit_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.ctx._chk(self.ctx.lib.av1mi_gop_submit_device(self.g, d_y.ptr, d_u.ptr, d_v.ptr, int(frame_type)))

    def pending(self):
        return self.ctx.lib.av1mi_gop_pending(self.g)

    def max_in_flight(self):
        return self.ctx.lib.av1mi_gop_max_in_flight()

    def entropy_fallbacks(self):
        self.ctx.lib.av1mi_gop_entropy_fallbacks.restype = C.c_long
        return self.ctx.lib.av1mi_gop_entropy_fallbacks(self.g)

    def collect_raw(self):
        f = GopFrame()
        self.ctx._chk(self.ctx.lib.av1mi_gop_collect(self.g, C.byref(f)))
        return f

    def collect(self):
        """dict of numpy views (valid until the next submit) + params"""
        f = self.collect_raw()
        S, nb = f.segments, f.blocks_per_frame
        out = dict(params=f.params, frame_type=f.params.frame_type, lr_on=_view(f.lr_on, (S, 3), np.uint8), raw=f)      # restoration on / off per segment and plane
        if f.key_block_size == 32:
            out["key_block_size"] = 32
        if f.key_block_size == 32 and f.lev_y:
            # a key frame in 32x32 blocks: per segment the blocks of the complete superblock rows ("32": modes, levels [n, 32, 32] and the
            # 16x16 chroma), then the 8x8 blocks of a last partial row ("8")
            w, h = self.w, self.h
            hA = h // 64 * 64
            nA, nB = (hA // 32) * (w // 32), ((h - hA) // 8) * (w // 8)
            for name, ptr in (("y_mode", f.y_mode), ("uv_mode", f.uv_mode)):
                m = _view(ptr, (S, f.key_modes_stride), np.uint8)
                out[name + "32"], out[name + "8"] = m[:, :nA], m[:, f.key_modes_band:f.key_modes_band + nB]
            for name, ptr, d in (("lev_y", f.lev_y, 1), ("lev_u", f.lev_u, 2), ("lev_v", f.lev_v, 2)):
                pl = _view(ptr, (S, (h // d) * (w // d)), np.int16)
                cut = (hA // d) * (w // d)
                out[name + "32"] = pl[:, :cut].reshape(S, nA, 32 // d, 32 // d)
                out[name + "8"] = pl[:, cut:].reshape(S, nB, 8 // d, 8 // d)
            return out
        if f.lev_y:
            out.update(lev_y=_view(f.lev_y, (S, nb, 8, 8), np.int16), lev_u=_view(f.lev_u, (S, nb, 4, 4), np.int16),
                       lev_v=_view(f.lev_v, (S, nb, 4, 4), np.int16))
        if f.tile_size:
            out.update(tiles_per_frame=f.tiles_per_frame, tile_size=_view(f.tile_size, (S * f.tiles_per_frame,), np.uint32),
                       tile_payload=_view(f.tile_payload, (max(int(f.payload_bytes), 1),), np.uint8)[:int(f.payload_bytes)])
        # (absent when the tiles were coded on the GPU, gpu_entropy = 1: the host then gets the payloads only)
        if f.y_mode:
            out["y_mode"], out["uv_mode"] = _view(f.y_mode, (S, nb), np.uint8), _view(f.uv_mode, (S, nb), np.uint8)
        if f.mv:
            out["mv"], out["skip"] = _view(f.mv, (S, nb, 2), np.int16), _view(f.skip, (S, nb), np.uint8)
        return out

    def download_reference(self):
        S, w, h = self.segments, self.w, self.h
        y, u, v = np.empty((S * h, w), self.dt), np.empty((S * h // 2, w // 2), self.dt), np.empty((S * h // 2, w // 2), self.dt)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self.ctx._chk(self.ctx.lib.av1mi_gop_download_reference(self.g, vp(y), vp(u), vp(v)))
        return y, u, v

    def close(self):
        if self.g:
            self.ctx.lib.av1mi_gop_close(self.g)
            self.g = None


N_KERNEL_KINDS = 17   # enum av1mi_kernel_kind
_lib = None


def load():
    """dlopen libav1mi.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.av1mi_version.restype = C.c_char_p
        lib.av1mi_last_error.restype = C.c_char_p
        lib.av1mi_device_name.restype = C.c_char_p
        lib.av1mi_last_error.argtypes = [C.c_void_p]
        lib.av1mi_device_name.argtypes = [C.c_void_p]
        lib.av1mi_close.argtypes = [C.c_void_p]
        lib.av1mi_close.restype = None
        _lib = lib
    return _lib


def exported_symbols():
    """names declared in include/av1mi.h (parsed from the header)."""
    import re
    hdr = open(os.path.join(_HERE, "..", "include", "av1mi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(av1mi_[a-z0-9_]+)\s*\(", hdr)))


class DevBuf:
    """device allocation owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        ctx._chk(ctx.lib.av1mi_malloc(ctx.h, C.byref(p), C.c_size_t(self.nbytes)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._chk(self.ctx.lib.av1mi_upload(self.ctx.h, C.c_void_p(self.ptr), arr.ctypes.data_as(C.c_void_p),
                                                C.c_size_t(arr.nbytes)))
        return self

    def download(self, shape, dtype):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._chk(self.ctx.lib.av1mi_download(self.ctx.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr),
                                                  C.c_size_t(out.nbytes)))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.av1mi_free(self.ctx.h, C.c_void_p(self.ptr))
            self.ptr = None


class Context:
    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.av1mi_open(int(device), C.byref(h))
        if rc != 0:
            raise Av1miError(rc, "av1mi_open(device=%d) failed: no usable HIP device" % device)
        self.h = h

    def close(self):
        if self.h:
            self.lib.av1mi_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise Av1miError(rc, self.lib.av1mi_last_error(self.h).decode())

    @property
    def device_name(self):
        return self.lib.av1mi_device_name(self.h).decode()

    def alloc(self, nbytes):
        return DevBuf(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DevBuf(self, max(arr.nbytes, 16)).upload(arr)

    def sync(self):
        self._chk(self.lib.av1mi_sync(self.h))

    def memset(self, d_buf, value, nbytes):
        self._chk(self.lib.av1mi_memset(self.h, C.c_void_p(d_buf.ptr), int(value), C.c_size_t(nbytes)))

    def copy(self, d_dst, d_src, nbytes):
        self._chk(self.lib.av1mi_copy(self.h, C.c_void_p(d_dst.ptr), C.c_void_p(d_src.ptr), C.c_size_t(nbytes)))

    def timer_begin(self):
        self._chk(self.lib.av1mi_timer_begin(self.h))

    def timer_end(self):
        ms = C.c_float()
        self._chk(self.lib.av1mi_timer_end(self.h, C.byref(ms)))
        return ms.value

    # ---- K2 / K1 / K8, device-resident
    def inv_txfm_add_grid(self, tx_size, d_coef, d_plane, stride, bd, blocks_per_row, nblocks, d_types=None, uniform_type=0):
        self._chk(self.lib.av1mi_inv_txfm_add_grid(self.h, tx_size, C.c_void_p(d_coef.ptr), C.c_void_p(d_plane.ptr), stride, bd,
                                                   blocks_per_row, nblocks, C.c_void_p(d_types.ptr if d_types else None),
                                                   uniform_type))

    def inv_txfm_add_list(self, tx_size, d_coef, d_plane, stride, bd, d_list, nblocks):
        self._chk(self.lib.av1mi_inv_txfm_add_list(self.h, tx_size, C.c_void_p(d_coef.ptr), C.c_void_p(d_plane.ptr), stride, bd,
                                                   C.c_void_p(d_list.ptr), nblocks))

    def fwd_txfm_grid(self, tx_size, d_resid, stride, d_coef, blocks_per_row, nblocks, d_types=None, uniform_type=0):
        self._chk(self.lib.av1mi_fwd_txfm_grid(self.h, tx_size, C.c_void_p(d_resid.ptr), stride, C.c_void_p(d_coef.ptr),
                                               blocks_per_row, nblocks, C.c_void_p(d_types.ptr if d_types else None),
                                               uniform_type))

    def fwd_txfm_list(self, tx_size, d_resid, stride, d_coef, d_list, nblocks):
        self._chk(self.lib.av1mi_fwd_txfm_list(self.h, tx_size, C.c_void_p(d_resid.ptr), stride, C.c_void_p(d_coef.ptr),
                                               C.c_void_p(d_list.ptr), nblocks))

    def quantize(self, d_coef, d_levels, d_dq, n, coef_per_blk, dc_q, ac_q, log_scale):
        self._chk(self.lib.av1mi_quantize(self.h, C.c_void_p(d_coef.ptr), C.c_void_p(d_levels.ptr),
                                          C.c_void_p(d_dq.ptr if d_dq else None), C.c_size_t(n), coef_per_blk, dc_q, ac_q,
                                          log_scale))

    def dequantize(self, d_levels, d_dq, n, coef_per_blk, dc_q, ac_q, log_scale, bd):
        self._chk(self.lib.av1mi_dequantize(self.h, C.c_void_p(d_levels.ptr), C.c_void_p(d_dq.ptr), C.c_size_t(n), coef_per_blk,
                                            dc_q, ac_q, log_scale, bd))

    # ---- K3
    def intra_pred_list(self, tx_size, d_ref, ref_stride, d_dst, dst_stride, bd, d_list, nblocks):
        self._chk(self.lib.av1mi_intra_pred_list(self.h, tx_size, C.c_void_p(d_ref.ptr), ref_stride, C.c_void_p(d_dst.ptr),
                                                 dst_stride, bd, C.c_void_p(d_list.ptr), nblocks))

    def cfl_pred_list(self, tx_size, d_luma, luma_stride, d_dst, dst_stride, bd, d_list, nblocks):
        self._chk(self.lib.av1mi_cfl_pred_list(self.h, tx_size, C.c_void_p(d_luma.ptr), luma_stride, C.c_void_p(d_dst.ptr), dst_stride,
                                               bd, C.c_void_p(d_list.ptr), nblocks))

    # ---- K4
    def mc_list(self, size_id, d_ref, ref_stride, plane_w, plane_h, d_dst, dst_stride, bd, d_list, nblocks):
        self._chk(self.lib.av1mi_mc_list(self.h, size_id, C.c_void_p(d_ref.ptr), ref_stride, plane_w, plane_h, C.c_void_p(d_dst.ptr),
                                         dst_stride, bd, C.c_void_p(d_list.ptr), nblocks))

    # ---- K5
    def deblock_plane(self, d_src, src_stride, d_dst, dst_stride, w, h, bd, is_chroma, d_mi, mi_stride, sharpness):
        self._chk(self.lib.av1mi_deblock_plane(self.h, C.c_void_p(d_src.ptr), src_stride, C.c_void_p(d_dst.ptr), dst_stride, w, h, bd,
                                               int(is_chroma), C.c_void_p(d_mi.ptr), mi_stride, sharpness))

    def deblock_frames(self, d_src, src_stride, d_dst, dst_stride, w, h, bd, is_chroma, d_mi, mi_stride, mi_frame_stride,
                       sharpness, nframes):
        self._chk(self.lib.av1mi_deblock_frames(self.h, C.c_void_p(d_src.ptr), src_stride, C.c_void_p(d_dst.ptr), dst_stride, w, h,
                                                bd, int(is_chroma), C.c_void_p(d_mi.ptr), mi_stride, C.c_size_t(mi_frame_stride),
                                                sharpness, nframes))

    def prof_enable(self, on):
        self._chk(self.lib.av1mi_prof_enable(self.h, int(on)))

    def prof_reset(self):
        self._chk(self.lib.av1mi_prof_reset(self.h))

    def prof_get(self):
        """{kind name: (launches, total_ms)} for kinds that were launched"""
        self.lib.av1mi_kernel_kind_name.restype = C.c_char_p
        out = {}
        for k in range(N_KERNEL_KINDS):
            n, ms = C.c_int(), C.c_double()
            self._chk(self.lib.av1mi_prof_get(self.h, k, C.byref(n), C.byref(ms)))
            if n.value:
                out[self.lib.av1mi_kernel_kind_name(k).decode()] = (n.value, ms.value)
        return out

    # ---- K6
    def cdef_frames(self, job):
        self._chk(self.lib.av1mi_cdef_frames(self.h, C.byref(job)))

    def cdef_arrays(self, Y, U, V, bd, damping, sb_strength, skip8):
        """tests: Y/U/V [frames,h,w]; sb_strength [frames or 1, nsb, 4]; skip8 [frames or 1, h/8, w/8]"""
        dt = np.uint8 if bd == 8 else np.uint16
        Y, U, V = (np.ascontiguousarray(a, dt) for a in (Y, U, V))
        nf, h, w = Y.shape
        sb_strength = np.ascontiguousarray(sb_strength, np.uint8); skip8 = np.ascontiguousarray(skip8, np.uint8)
        bufs = [self.to_device(a) for a in (Y, U, V)] + [self.alloc(a.nbytes) for a in (Y, U, V)] + [self.to_device(sb_strength), self.to_device(skip8)]
        job = CdefJob(w, h, bd, nf, damping, w, w // 2, *[b.ptr for b in bufs[:6]], bufs[6].ptr,
                      0 if sb_strength.shape[0] == 1 else sb_strength.shape[1], bufs[7].ptr,
                      0 if skip8.shape[0] == 1 else skip8.shape[1] * skip8.shape[2])
        self.cdef_frames(job)
        out = (bufs[3].download(Y.shape, dt), bufs[4].download(U.shape, dt), bufs[5].download(V.shape, dt))
        for b in bufs:
            b.free()
        return out

    # ---- K7
    def lr_frames(self, d_cdef, d_dbl, d_out, stride, w, h, bd, ss, unit_size, d_units, unit_frame_stride, nframes):
        self._chk(self.lib.av1mi_lr_frames(self.h, C.c_void_p(d_cdef.ptr), C.c_void_p(d_dbl.ptr), C.c_void_p(d_out.ptr), stride, w, h, bd,
                                           int(ss), unit_size, C.c_void_p(d_units.ptr), C.c_size_t(unit_frame_stride), nframes))

    def lr_decide_scratch_bytes(self, h, ss, nframes):
        self.lib.av1mi_lr_decide_scratch_bytes.restype = C.c_size_t
        return int(self.lib.av1mi_lr_decide_scratch_bytes(h, int(ss), nframes))

    def lr_frames_decide(self, d_cdef, d_dbl, d_out, stride, w, h, bd, ss, unit_size, d_units, unit_frame_stride, nframes, d_orig, d_scratch, d_on,
                         on_offset=0, on_stride=1):
        """restoration + the per-frame ON / OFF decision against the source d_orig; d_on: uint8 buffer, entry on_offset + f * on_stride"""
        self._chk(self.lib.av1mi_lr_frames_decide(self.h, C.c_void_p(d_cdef.ptr), C.c_void_p(d_dbl.ptr), C.c_void_p(d_out.ptr), stride, w, h, bd,
                                                  int(ss), unit_size, C.c_void_p(d_units.ptr), C.c_size_t(unit_frame_stride), nframes,
                                                  C.c_void_p(d_orig.ptr), C.c_void_p(d_scratch.ptr), C.c_void_p(d_on.ptr + on_offset), int(on_stride)))

    def lr_yuv_decide_scratch_bytes(self, h, nframes):
        self.lib.av1mi_lr_yuv_decide_scratch_bytes.restype = C.c_size_t
        return int(self.lib.av1mi_lr_yuv_decide_scratch_bytes(int(h), int(nframes)))

    def lr_yuv_decide(self, job):
        """the three planes' restoration + ON / OFF decisions in one call (include/av1mi.h av1mi_lr_yuv_decide)"""
        self._chk(self.lib.av1mi_lr_yuv_decide(self.h, C.byref(job)))

    # ---- fused intra-only segment pipeline
    def extend_frames(self, d_plane, stride, w, h, visible_w, visible_h, bd, nframes):
        self._chk(self.lib.av1mi_extend_frames(self.h, C.c_void_p(d_plane.ptr), stride, w, h, visible_w, visible_h, bd, nframes))

    def intra_encode(self, job):
        self._chk(self.lib.av1mi_intra_encode(self.h, C.byref(job)))

    def intra_encode_arrays(self, Y, U, V, bd, bs, qindex, open_loop=False):
        """convenience for tests: Y/U/V are [frames, h, w] arrays; returns dict of outputs like the oracle's"""
        dt = np.uint8 if bd == 8 else np.uint16
        Y, U, V = (np.ascontiguousarray(a, dt) for a in (Y, U, V))
        nf, h, w = Y.shape
        nb = (h // bs) * (w // bs)
        bufs = {}
        job = IntraJob(w, h, bd, nf, qindex, bs, w, w // 2)
        job.open_loop = 1 if open_loop else 0
        for name, arr in (("src_y", Y), ("src_u", U), ("src_v", V)):
            bufs[name] = self.to_device(arr)
        for name, n in (("rec_y", Y.nbytes), ("rec_u", U.nbytes), ("rec_v", V.nbytes), ("lev_y", Y.size * 2), ("lev_u", U.size * 2),
                        ("lev_v", V.size * 2), ("modes_y", nf * nb), ("modes_uv", nf * nb)):
            bufs[name] = self.alloc(n)
            self.memset(bufs[name], 0, n)
        for name, b in bufs.items():
            setattr(job, "d_" + name, b.ptr)
        self.intra_encode(job)
        cs = bs // 2
        out = dict(rec_y=bufs["rec_y"].download(Y.shape, dt), rec_u=bufs["rec_u"].download(U.shape, dt),
                   rec_v=bufs["rec_v"].download(V.shape, dt), lev_y=bufs["lev_y"].download((nf, nb, bs, bs), np.int16),
                   lev_u=bufs["lev_u"].download((nf, nb, cs, cs), np.int16), lev_v=bufs["lev_v"].download((nf, nb, cs, cs), np.int16),
                   modes_y=bufs["modes_y"].download((nf, nb), np.uint8), modes_uv=bufs["modes_uv"].download((nf, nb), np.uint8))
        for b in bufs.values():
            b.free()
        return out

    # ---- inter (P-frame) pipeline
    def inter_encode(self, job):
        self._chk(self.lib.av1mi_inter_encode(self.h, C.byref(job)))

    def inter_encode_arrays(self, src, ref, bd, qindex, search_range=8):
        """tests: src / ref = (Y, U, V) with arrays [frames, h, w]; returns dict like the oracle's"""
        dt = np.uint8 if bd == 8 else np.uint16
        S = [np.ascontiguousarray(a, dt) for a in src]
        R = [np.ascontiguousarray(a, dt) for a in ref]
        nf, h, w = S[0].shape
        nb = (h // 8) * (w // 8)
        bufs = {}
        for i, p in enumerate("yuv"):
            bufs["src_" + p], bufs["ref_" + p] = self.to_device(S[i]), self.to_device(R[i])
            bufs["rec_" + p] = self.alloc(S[i].nbytes)
            bufs["lev_" + p] = self.alloc(S[i].size * 2)
        bufs["mvs"], bufs["skip"] = self.alloc(nf * nb * 4), self.alloc(nf * nb)
        job = InterJob(w, h, bd, nf, qindex, search_range, w, w // 2)
        for k, b in bufs.items():
            setattr(job, "d_" + k, b.ptr)
        self.inter_encode(job)
        out = dict(mvs=bufs["mvs"].download((nf, nb, 2), np.int16), skip=bufs["skip"].download((nf, nb), np.uint8))
        for i, p in enumerate("yuv"):
            bs = 8 if i == 0 else 4
            out["rec_" + p] = bufs["rec_" + p].download(S[i].shape, dt)
            out["lev_" + p] = bufs["lev_" + p].download((nf, nb, bs, bs), np.int16)
        for b in bufs.values():
            b.free()
        return out

    # ---- host-pointer single-block forms
    def inv_txfm2d_add(self, coef, pred, tx_size, tx_type, bd):
        coef = np.ascontiguousarray(coef, np.int32)
        dst = np.ascontiguousarray(pred, np.uint8 if bd == 8 else np.uint16).copy()
        self._chk(self.lib.av1mi_inv_txfm2d_add(self.h, coef.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p),
                                                dst.shape[1], tx_size, tx_type, bd))
        return dst

    def fwd_txfm2d(self, resid, tx_size, tx_type):
        resid = np.ascontiguousarray(resid, np.int16)
        coef = np.zeros((min(TX_H[tx_size], 32), min(TX_W[tx_size], 32)), np.int32)
        self._chk(self.lib.av1mi_fwd_txfm2d(self.h, resid.ctypes.data_as(C.c_void_p), resid.shape[1],
                                            coef.ctypes.data_as(C.c_void_p), tx_size, tx_type))
        return coef
