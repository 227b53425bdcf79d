"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from the product.  Pinned to dav1d, not to the reference — see
oracle/av1o_common.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
TX_NAMES = ["4x4", "8x8", "16x16", "32x32", "64x64", "4x8", "8x4", "8x16", "16x8", "16x32", "32x16",
            "32x64", "64x32", "4x16", "16x4", "8x32", "32x8", "16x64", "64x16"]
TX_TYPE_NAMES = ["DCT_DCT", "ADST_DCT", "DCT_ADST", "ADST_ADST", "FLIPADST_DCT", "DCT_FLIPADST",
                 "FLIPADST_FLIPADST", "ADST_FLIPADST", "FLIPADST_ADST", "IDTX", "V_DCT", "H_DCT",
                 "V_ADST", "H_ADST", "V_FLIPADST", "H_FLIPADST"]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def coef_shape(tx_size):
    return min(TX_H[tx_size], 32), min(TX_W[tx_size], 32)


def txfm_valid(tx_size, tx_type):
    return bool(lib().av1o_txfm_valid(tx_size, tx_type))


def idct(x, n, bit=12, rng=0):
    x = np.ascontiguousarray(x, np.int32)
    out = np.zeros(n, np.int32)
    lib().av1o_idct(_p(x, C.c_int32), _p(out, C.c_int32), n, bit, rng)
    return out


def fdct(x, n, bit=12):
    x = np.ascontiguousarray(x, np.int32)
    out = np.zeros(n, np.int32)
    lib().av1o_fdct(_p(x, C.c_int32), _p(out, C.c_int32), n, bit)
    return out


def idct_explicit(x, n, rng=0):
    x = np.ascontiguousarray(x, np.int32)
    out = np.zeros(n, np.int32)
    getattr(lib(), "av1o_idct%d_explicit" % n)(_p(x, C.c_int32), _p(out, C.c_int32), rng)
    return out


def iadst(x, n, bit=12, rng=0):
    x = np.ascontiguousarray(x, np.int32)
    out = np.zeros(n, np.int32)
    if n == 4:
        lib().av1o_iadst4(_p(x, C.c_int32), _p(out, C.c_int32), bit)
    else:
        getattr(lib(), "av1o_iadst%d" % n)(_p(x, C.c_int32), _p(out, C.c_int32), bit, rng)
    return out


def fadst(x, n, bit=12):
    x = np.ascontiguousarray(x, np.int32)
    out = np.zeros(n, np.int32)
    getattr(lib(), "av1o_fadst%d" % n)(_p(x, C.c_int32), _p(out, C.c_int32), bit)
    return out


def identity(x, n):
    x = np.ascontiguousarray(x, np.int32)
    out = np.zeros(n, np.int32)
    lib().av1o_identity(_p(x, C.c_int32), _p(out, C.c_int32), n)
    return out


def inv_txfm2d_add(coef, pred, tx_size, tx_type, bd, libaom_clamps=1):
    """coef: int32 [min(h,32), min(w,32)]; pred: [h, w] uint8/uint16.  returns the reconstruction."""
    coef = np.ascontiguousarray(coef, np.int32)
    dt = np.uint8 if bd == 8 else np.uint16
    dst = np.ascontiguousarray(pred, dt).copy()
    assert coef.shape == coef_shape(tx_size) and dst.shape == (TX_H[tx_size], TX_W[tx_size])
    rc = lib().av1o_inv_txfm2d_add(_p(coef, C.c_int32), dst.ctypes.data_as(C.c_void_p), dst.shape[1], tx_size,
                                   tx_type, bd, libaom_clamps)
    if rc:
        raise ValueError("av1o_inv_txfm2d_add rc=%d" % rc)
    return dst


def fwd_txfm2d(resid, tx_size, tx_type, bd=8):
    resid = np.ascontiguousarray(resid, np.int16)
    assert resid.shape == (TX_H[tx_size], TX_W[tx_size])
    coef = np.zeros(coef_shape(tx_size), np.int32)
    rc = lib().av1o_fwd_txfm2d(_p(resid, C.c_int16), resid.shape[1], _p(coef, C.c_int32), tx_size, tx_type, bd)
    if rc:
        raise ValueError("av1o_fwd_txfm2d rc=%d" % rc)
    return coef


def dc_q(qindex, bd=8, delta=0):
    return lib().av1o_dc_q(qindex, delta, bd)


def ac_q(qindex, bd=8, delta=0):
    return lib().av1o_ac_q(qindex, delta, bd)


def tx_scale(tx_size):
    return lib().av1o_tx_scale(tx_size)


def quantize(coef, dcq, acq, log_scale):
    coef = np.ascontiguousarray(coef, np.int32)
    lv = np.zeros(coef.shape, np.int16)
    dq = np.zeros(coef.shape, np.int32)
    nz = lib().av1o_quantize(_p(coef, C.c_int32), coef.size, dcq, acq, log_scale, _p(lv, C.c_int16),
                             _p(dq, C.c_int32))
    return lv, dq, nz


def dequantize(levels, dcq, acq, log_scale, bd):
    levels = np.ascontiguousarray(levels, np.int16)
    dq = np.zeros(levels.shape, np.int32)
    lib().av1o_dequantize(_p(levels, C.c_int16), levels.size, dcq, acq, log_scale, bd, _p(dq, C.c_int32))
    return dq


def txq_plane(resid, pred, tx_size, dcq, acq, bd, tx_types=None, uniform_type=0, rows=None):
    """whole-plane residual -> fwd -> quant -> dequant -> inv+recon; returns (recon, levels)."""
    resid = np.ascontiguousarray(resid, np.int16)
    dt = np.uint8 if bd == 8 else np.uint16
    rec = np.ascontiguousarray(pred, dt).copy()
    h, w = TX_H[tx_size], TX_W[tx_size]
    nby, nbx = resid.shape[0] // h, resid.shape[1] // w
    ch, cw = coef_shape(tx_size)
    levels = np.zeros((nby * nbx, ch, cw), np.int16)
    by0, by1 = rows if rows else (0, nby)
    tt = None if tx_types is None else np.ascontiguousarray(tx_types, np.uint8)
    rc = lib().av1o_txq_plane(_p(resid, C.c_int16), rec.ctypes.data_as(C.c_void_p), resid.shape[1], nbx, by0, by1, tx_size,
                              None if tt is None else _p(tt, C.c_uint8), uniform_type, dcq, acq, bd, _p(levels, C.c_int16))
    if rc:
        raise ValueError("av1o_txq_plane rc=%d" % rc)
    return rec, levels


INTRA_MODE_NAMES = ["DC", "V", "H", "D45", "D135", "D113", "D157", "D203", "D67", "SMOOTH", "SMOOTH_V", "SMOOTH_H", "PAETH"]


def intra_predict(plane, x, y, bw, bh, mode, angle_delta, bd, n_top, n_topright, n_left, n_bottomleft,
                  disable_edge_filter=0, filter_type=0):
    """plane: reconstructed plane (uint8 for bd 8, uint16 otherwise); returns the [bh, bw] prediction (uint16)."""
    dt = np.uint8 if bd == 8 else np.uint16
    plane = np.ascontiguousarray(plane, dt)
    pred = np.zeros((bh, bw), np.uint16)
    base = plane.ctypes.data + (y * plane.shape[1] + x) * plane.itemsize
    rc = lib().av1o_intra_predict(C.c_void_p(base), plane.shape[1], bd, bw, bh, mode, angle_delta, disable_edge_filter,
                                  filter_type, n_top, n_topright, n_left, n_bottomleft, _p(pred, C.c_uint16))
    if rc:
        raise ValueError("av1o_intra_predict rc=%d" % rc)
    return pred


def lf_mi(tx_w_log2, tx_h_log2, lvl_v, lvl_h, skip_inter=0, blk_left=1, blk_top=1):
    """pack one (or an array of) loop-filter mode-info unit(s) -> uint32"""
    return (np.uint32(tx_w_log2) | (np.uint32(tx_h_log2) << 4) | (np.uint32(lvl_v) << 8) | (np.uint32(lvl_h) << 16)
            | (np.uint32(skip_inter) << 24) | (np.uint32(blk_left) << 25) | (np.uint32(blk_top) << 26))


def deblock_plane(plane, bd, is_chroma, mi, sharpness=0, pass_mask=3):
    dt = np.uint8 if bd == 8 else np.uint16
    out = np.ascontiguousarray(plane, dt).copy()
    mi = np.ascontiguousarray(mi, np.uint32)
    h, w = out.shape
    assert mi.shape == (h // 4, w // 4)
    rc = lib().av1o_deblock_plane(out.ctypes.data_as(C.c_void_p), w, w, h, bd, int(is_chroma), mi.ctypes.data_as(C.c_void_p),
                                  mi.shape[1], sharpness, pass_mask)
    if rc:
        raise ValueError("av1o_deblock_plane rc=%d" % rc)
    return out


def set_intra_open_loop(on):
    """process-wide switch of the oracle's key-frame mode decision: open loop (candidates predicted from the SOURCE neighbours, edge
    filter type 0) or closed loop (from the reconstruction); also used by the GOP oracle's key frames"""
    lib().av1o_set_intra_open_loop(1 if on else 0)


def intra_encode_frame(Y, U, V, bd, bs, qindex, open_loop=False):
    """oracle intra-only encoder loop; returns dict(rec_y, rec_u, rec_v, lev_y, lev_u, lev_v, modes_y, modes_uv)"""
    set_intra_open_loop(open_loop)
    dt = np.uint8 if bd == 8 else np.uint16
    Y, U, V = (np.ascontiguousarray(a, dt) for a in (Y, U, V))
    h, w = Y.shape
    nb = (h // bs) * (w // bs)
    cs = bs // 2
    out = dict(rec_y=np.zeros_like(Y), rec_u=np.zeros_like(U), rec_v=np.zeros_like(V),
               lev_y=np.zeros((nb, bs, bs), np.int16), lev_u=np.zeros((nb, cs, cs), np.int16), lev_v=np.zeros((nb, cs, cs), np.int16),
               modes_y=np.zeros(nb, np.uint8), modes_uv=np.zeros(nb, np.uint8))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().av1o_intra_encode_frame(vp(Y), vp(U), vp(V), vp(out["rec_y"]), vp(out["rec_u"]), vp(out["rec_v"]), w, h, w, w // 2,
                                       bd, bs, qindex, vp(out["lev_y"]), vp(out["lev_u"]), vp(out["lev_v"]), vp(out["modes_y"]),
                                       vp(out["modes_uv"]))
    if rc:
        raise ValueError("av1o_intra_encode_frame rc=%d" % rc)
    return out


def subpel_filters():
    arr = (C.c_int16 * (6 * 16 * 8)).in_dll(lib(), "av1o_subpel_filters")
    return np.frombuffer(arr, np.int16).reshape(6, 16, 8).copy()


def mc_block(ref, bd, x, y, w, h, mvx, mvy, filt_x=0, filt_y=0):
    dt = np.uint8 if bd == 8 else np.uint16
    ref = np.ascontiguousarray(ref, dt)
    pred = np.zeros((h, w), np.uint16)
    rc = lib().av1o_mc_block(ref.ctypes.data_as(C.c_void_p), ref.shape[1], ref.shape[1], ref.shape[0], bd, x, y, w, h, mvx, mvy,
                             filt_x, filt_y, _p(pred, C.c_uint16))
    if rc:
        raise ValueError("av1o_mc_block rc=%d" % rc)
    return pred


def cdef_find_dir(block, bd):
    dt = np.uint8 if bd == 8 else np.uint16
    block = np.ascontiguousarray(block, dt)
    var = C.c_int()
    d = lib().av1o_cdef_find_dir(block.ctypes.data_as(C.c_void_p), block.shape[1], bd, C.byref(var))
    return d, var.value


def cdef_frame(Y, U, V, bd, damping, sb_strength, skip8):
    """sb_strength: [nsb, 4] uint8; skip8: [h/8, w/8] uint8; returns filtered (Y, U, V)"""
    dt = np.uint8 if bd == 8 else np.uint16
    Y, U, V = (np.ascontiguousarray(a, dt) for a in (Y, U, V))
    oy, ou, ov = np.zeros_like(Y), np.zeros_like(U), np.zeros_like(V)
    sb = np.ascontiguousarray(sb_strength, np.uint8)
    sk = np.ascontiguousarray(skip8, np.uint8)
    h, w = Y.shape
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().av1o_cdef_frame(vp(Y), vp(U), vp(V), vp(oy), vp(ou), vp(ov), w, h, w, w // 2, bd, damping, vp(sb), vp(sk))
    if rc:
        raise ValueError("av1o_cdef_frame rc=%d" % rc)
    return oy, ou, ov


def lr_units(unit_size, plane_size):
    return lib().av1o_lr_units(unit_size, plane_size)


def lr_unit_none():
    return np.zeros(8, np.int8)


def lr_unit_wiener(v, h):
    return np.array([1, v[0], v[1], v[2], h[0], h[1], h[2], 0], np.int8)


def lr_unit_sgr(sgr_set, xqd0, xqd1):
    return np.array([2, sgr_set, xqd0, xqd1, 0, 0, 0, 0], np.int8)


def lr_plane(cdef, dbl, bd, ss, unit_size, units):
    """cdef: CDEF output plane, dbl: deblocked (pre-CDEF) plane; units: [rows, cols, 8] int8.  Returns the restored plane."""
    dt = np.uint8 if bd == 8 else np.uint16
    cdef, dbl = np.ascontiguousarray(cdef, dt), np.ascontiguousarray(dbl, dt)
    out = np.zeros_like(cdef)
    h, w = cdef.shape
    units = np.ascontiguousarray(units, np.int8)
    assert units.shape == (lr_units(unit_size, h), lr_units(unit_size, w), 8)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().av1o_lr_plane(vp(cdef), vp(dbl), vp(out), w, w, h, bd, ss, unit_size, vp(units))
    if rc:
        raise ValueError("av1o_lr_plane rc=%d" % rc)
    return out


def lr_select(src, cdef, lr, bd, ss=None):
    """the restoration on/off policy per plane (av1o_lr_keep): src / cdef / lr are (Y, U, V) triples of one frame (ss: the planes' chroma
    flag when they are not a Y, U, V triple).
    Returns (planes the next frame predicts from, [on_y, on_u, on_v])"""
    dt = np.uint8 if bd == 8 else np.uint16
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    out, on = [], []
    for p, (s_, c_, l_) in enumerate(zip(src, cdef, lr)):
        s_, c_, l_ = (np.ascontiguousarray(a, dt) for a in (s_, c_, l_))
        h, w = s_.shape
        k = int(lib().av1o_lr_keep(vp(s_), vp(c_), vp(l_), w, w, h, bd, int(ss if ss is not None else p > 0)))
        on.append(k)
        out.append(l_ if k else c_)
    return out, on


def inter_encode_frame(src, ref, bd, qindex, search_range=8, bs=8):
    """src, ref: (Y, U, V) tuples; returns dict(rec_y/u/v, lev_y/u/v, mvs [nb,2], skip [nb])"""
    dt = np.uint8 if bd == 8 else np.uint16
    S = [np.ascontiguousarray(a, dt) for a in src]
    R = [np.ascontiguousarray(a, dt) for a in ref]
    h, w = S[0].shape
    nb = (h // bs) * (w // bs)
    cs = bs // 2
    out = dict(rec_y=np.zeros_like(S[0]), rec_u=np.zeros_like(S[1]), rec_v=np.zeros_like(S[2]),
               lev_y=np.zeros((nb, bs, bs), np.int16), lev_u=np.zeros((nb, cs, cs), np.int16), lev_v=np.zeros((nb, cs, cs), np.int16),
               mvs=np.zeros((nb, 2), np.int16), skip=np.zeros(nb, np.uint8))
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().av1o_inter_encode_frame(vp(S[0]), vp(S[1]), vp(S[2]), vp(R[0]), vp(R[1]), vp(R[2]), vp(out["rec_y"]), vp(out["rec_u"]),
                                       vp(out["rec_v"]), w, h, w, w // 2, bd, bs, qindex, search_range, vp(out["lev_y"]), vp(out["lev_u"]),
                                       vp(out["lev_v"]), vp(out["mvs"]), vp(out["skip"]))
    if rc:
        raise ValueError("av1o_inter_encode_frame rc=%d" % rc)
    return out


def cfl_predict(luma, dc_plane, bd, x, y, bw, bh, alpha_q3, max_luma_w=None, max_luma_h=None):
    """chroma-from-luma (4:2:0): returns a copy of dc_plane with the block at (x, y) replaced by the CfL prediction"""
    dt = np.uint8 if bd == 8 else np.uint16
    luma = np.ascontiguousarray(luma, dt)
    out = np.ascontiguousarray(dc_plane, dt).copy()
    rc = lib().av1o_cfl_predict(luma.ctypes.data_as(C.c_void_p), luma.shape[1], out.ctypes.data_as(C.c_void_p), out.shape[1], bd, x, y,
                                bw, bh, alpha_q3, max_luma_w or luma.shape[1], max_luma_h or luma.shape[0])
    if rc:
        raise ValueError("av1o_cfl_predict rc=%d" % rc)
    return out
