"""ctypes binding of the host AV1 bitstream writer (host/av1_bitstream.hpp, exported by libav1mi_host.so).

Plumbing only: builds the `av1mi_obu_frame` description from numpy arrays (the symbols the GPU pipeline produced) and
returns the Section-5 OBU bytes of one temporal unit.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HOST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "libav1mi_host.so")
_lib = None


class ObuFrame(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("bit_depth", C.c_int32), ("frame_type", C.c_int32),
                ("base_q_idx", C.c_int32), ("lf_level", C.c_int32 * 4), ("lf_sharpness", C.c_int32), ("cdef_damping", C.c_int32),
                ("cdef_bits", C.c_int32), ("cdef_y", C.c_uint8 * 8), ("cdef_uv", C.c_uint8 * 8), ("cdef_idx", C.c_void_p),
                ("lr_type", C.c_int32 * 3), ("lr_unit_shift", C.c_int32), ("lr_uv_shift", C.c_int32), ("lr_units", C.c_void_p * 3),
                ("reduced_tx_set", C.c_int32), ("disable_cdf_update", C.c_int32), ("tile_cols_log2", C.c_int32),
                ("tile_rows_log2", C.c_int32), ("y_mode", C.c_void_p), ("angle_y", C.c_void_p), ("uv_mode", C.c_void_p),
                ("angle_uv", C.c_void_p), ("cfl_alpha", C.c_void_p), ("skip", C.c_void_p), ("tx_type", C.c_void_p),
                ("is_inter", C.c_void_p), ("mv", C.c_void_p), ("lev_y", C.c_void_p), ("lev_u", C.c_void_p), ("lev_v", C.c_void_p),
                ("visible_width", C.c_int32), ("visible_height", C.c_int32)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_HOST):
            subprocess.check_call(["make", "-s", "-C", os.path.dirname(_HOST)])
        _lib = C.CDLL(_HOST)
        _lib.av1mi_obu_write_temporal_unit.restype = C.c_longlong
        _lib.av1mi_obu_write_temporal_unit.argtypes = [C.POINTER(ObuFrame), C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_char_p, C.c_int]
        _lib.av1mi_host_opstream_temporal_unit.restype = C.c_longlong
        _lib.av1mi_host_opstream_temporal_unit.argtypes = [C.POINTER(ObuFrame), C.c_int, C.c_void_p, C.c_longlong, C.c_char_p, C.c_int]
    return _lib


_PTR_FIELDS = {"cdef_idx": np.uint8, "y_mode": np.uint8, "angle_y": np.int8, "uv_mode": np.uint8, "angle_uv": np.int8, "cfl_alpha": np.int8,
               "skip": np.uint8, "tx_type": np.uint8, "is_inter": np.uint8, "mv": np.int16, "lev_y": np.int16, "lev_u": np.int16,
               "lev_v": np.int16}


def temporal_unit(width, height, bit_depth, base_q_idx, frame_type=0, with_sequence_header=True, threads=1, lf_level=(0, 0, 0, 0),
                  lf_sharpness=0, cdef_damping=3, cdef_bits=0, cdef_y=(0,), cdef_uv=(0,), lr_type=(0, 0, 0), lr_unit_shift=0, lr_uv_shift=0,
                  lr_units=(None, None, None), reduced_tx_set=0, disable_cdf_update=0, tile_cols_log2=-1, tile_rows_log2=-1, opstream=False,
                  visible=None, key_rows32=0, **arrays):
    """arrays: y_mode, angle_y, uv_mode, angle_uv, cfl_alpha, skip, tx_type, is_inter, mv, lev_y, lev_u, lev_v, cdef_idx (numpy, raster
    order over 8x8 blocks; see av1_bitstream.hpp).  visible: (width, height) a decoder outputs when the coded size is the source's rounded up to 8.
    Returns bytes."""
    f = ObuFrame()
    f.width, f.height, f.bit_depth, f.frame_type, f.base_q_idx = width, height, bit_depth, frame_type, base_q_idx
    if visible is not None:
        f.visible_width, f.visible_height = int(visible[0]), int(visible[1])
    for i in range(4):
        f.lf_level[i] = int(lf_level[i])
    f.lf_sharpness, f.cdef_damping, f.cdef_bits = lf_sharpness, cdef_damping, cdef_bits
    for i, v in enumerate(cdef_y):
        f.cdef_y[i] = int(v)
    for i, v in enumerate(cdef_uv):
        f.cdef_uv[i] = int(v)
    keep = []
    for p in range(3):
        f.lr_type[p] = int(lr_type[p])
        if lr_units[p] is not None:
            a = np.ascontiguousarray(lr_units[p], np.int8)
            keep.append(a)
            f.lr_units[p] = a.ctypes.data
    f.lr_unit_shift, f.lr_uv_shift = lr_unit_shift, lr_uv_shift
    f.reduced_tx_set, f.disable_cdf_update, f.tile_cols_log2, f.tile_rows_log2 = reduced_tx_set, disable_cdf_update, tile_cols_log2, tile_rows_log2
    for k, v in arrays.items():
        if k not in _PTR_FIELDS:
            raise TypeError("unknown field " + k)
        if v is None:
            continue
        a = np.ascontiguousarray(v, _PTR_FIELDS[k])
        keep.append(a)
        setattr(f, k, a.ctypes.data)
    cap = width * height * 4 + (1 << 16)
    out = np.empty(cap, np.uint8)
    err = C.create_string_buffer(256)
    if opstream:    # the op-stream formulation (the GPU tile coder's CPU twin, host/av1_opstream.cpp): must give the same bytes
        # key_rows32: a key frame whose first rows are coded in 32x32 blocks, arrays in the session's layout (host/av1_opstream.cpp)
        h = lib()
        h.av1mi_host_opstream_key32_temporal_unit.restype = C.c_longlong
        h.av1mi_host_opstream_key32_temporal_unit.argtypes = [C.POINTER(ObuFrame), C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_char_p, C.c_int]
        n = h.av1mi_host_opstream_key32_temporal_unit(C.byref(f), int(key_rows32), int(with_sequence_header), out.ctypes.data, cap, err, 256)
    else:
        n = lib().av1mi_obu_write_temporal_unit(C.byref(f), int(with_sequence_header), threads, out.ctypes.data, cap, err, 256)
    if n < 0:
        raise ValueError("av1 bitstream writer: " + err.value.decode())
    if n > cap:
        raise RuntimeError("temporal unit of %d bytes exceeds the buffer" % n)
    return out[:n].tobytes()


BLOCK_DTYPE = np.dtype([("mi_row", "<u2"), ("mi_col", "<u2"), ("bsize", "u1"), ("skip", "u1"), ("is_inter", "u1"), ("y_mode", "u1"), ("uv_mode", "u1"),
                        ("angle_y", "i1"), ("angle_uv", "i1"), ("cfl_alpha_u", "i1"), ("cfl_alpha_v", "i1"), ("tx_depth", "u1"), ("interp_filter", "u1"),
                        ("reserved", "u1"), ("mv_x", "<i2"), ("mv_y", "<i2"), ("tx_type_off", "<u4"), ("lev_off", "<u4", (3,))])
assert BLOCK_DTYPE.itemsize == 36      # struct av1mi_obu_block (include/av1mi_host.h)


class ObuBlocks(C.Structure):
    _fields_ = [("hdr", ObuFrame), ("tx_mode_select", C.c_int32), ("interp_filter", C.c_int32), ("high_precision_mv", C.c_int32), ("partition", C.c_void_p), ("n_partition", C.c_size_t),
                ("blocks", C.c_void_p), ("n_blocks", C.c_size_t), ("tx_type", C.c_void_p), ("levels", C.c_void_p)]


def blocks_temporal_unit(width, height, bit_depth, base_q_idx, partition, blocks, tx_type, levels, frame_type=0, with_sequence_header=True,
                         tx_mode_select=0, interp_filter=0, high_precision_mv=0, reduced_tx_set=0, disable_cdf_update=0, tile_cols_log2=-1,
                         tile_rows_log2=-1, lf_level=(0, 0, 0, 0), lf_sharpness=0, cdef_damping=3, cdef_bits=0, cdef_y=(0,), cdef_uv=(0,), cdef_idx=None,
                         lr_type=(0, 0, 0), lr_units=(None, None, None), lr_unit_shift=0, lr_uv_shift=0):
    """the general block-structured writer (av1mi_obu_write_blocks_temporal_unit): partition = uint8 partition types in decoding order,
    blocks = array of BLOCK_DTYPE in decoding order, tx_type = uint8 per luma transform block, levels = int16 (see include/av1mi_host.h)"""
    d = ObuBlocks()
    f = d.hdr
    f.width, f.height, f.bit_depth, f.frame_type, f.base_q_idx = width, height, bit_depth, frame_type, base_q_idx
    for i in range(4):
        f.lf_level[i] = int(lf_level[i])
    f.lf_sharpness, f.cdef_damping, f.cdef_bits = lf_sharpness, cdef_damping, cdef_bits
    for i, v in enumerate(cdef_y):
        f.cdef_y[i] = int(v)
    for i, v in enumerate(cdef_uv):
        f.cdef_uv[i] = int(v)
    f.reduced_tx_set, f.disable_cdf_update, f.tile_cols_log2, f.tile_rows_log2 = reduced_tx_set, disable_cdf_update, tile_cols_log2, tile_rows_log2
    keep = [np.ascontiguousarray(partition, np.uint8), np.ascontiguousarray(blocks, BLOCK_DTYPE), np.ascontiguousarray(tx_type, np.uint8),
            np.ascontiguousarray(levels, np.int16)]
    if cdef_idx is not None:
        keep.append(np.ascontiguousarray(cdef_idx, np.uint8))
        f.cdef_idx = keep[-1].ctypes.data
    for p in range(3):
        f.lr_type[p] = int(lr_type[p])
        if lr_units[p] is not None:
            keep.append(np.ascontiguousarray(lr_units[p], np.int8))
            f.lr_units[p] = keep[-1].ctypes.data
    f.lr_unit_shift, f.lr_uv_shift = lr_unit_shift, lr_uv_shift
    d.tx_mode_select, d.interp_filter, d.high_precision_mv = int(tx_mode_select), int(interp_filter), int(high_precision_mv)
    d.partition, d.n_partition, d.blocks, d.n_blocks = keep[0].ctypes.data, keep[0].size, keep[1].ctypes.data, keep[1].size
    d.tx_type, d.levels = keep[2].ctypes.data, keep[3].ctypes.data
    cap = width * height * 4 + (1 << 16)
    out = np.empty(cap, np.uint8)
    err = C.create_string_buffer(256)
    L = lib()
    L.av1mi_obu_write_blocks_temporal_unit.restype = C.c_longlong
    L.av1mi_obu_write_blocks_temporal_unit.argtypes = [C.POINTER(ObuBlocks), C.c_int, C.c_void_p, C.c_longlong, C.c_char_p, C.c_int]
    n = L.av1mi_obu_write_blocks_temporal_unit(C.byref(d), int(with_sequence_header), out.ctypes.data, cap, err, 256)
    if n < 0:
        raise ValueError("av1 block writer: " + err.value.decode())
    if n > cap:
        raise RuntimeError("temporal unit of %d bytes exceeds the buffer" % n)
    return out[:n].tobytes()


def assemble_temporal_unit(width, height, bit_depth, base_q_idx, payloads, sizes, with_sequence_header=True, **hdr):
    """temporal unit around tile payloads coded on the GPU (av1mi_obu_assemble_temporal_unit); hdr = header_from_params(...)"""
    f = ObuFrame()
    f.width, f.height, f.bit_depth, f.base_q_idx = width, height, bit_depth, base_q_idx
    if hdr.get("visible") is not None:
        f.visible_width, f.visible_height = int(hdr["visible"][0]), int(hdr["visible"][1])
    f.frame_type = hdr.get("frame_type", 0)
    for i in range(4):
        f.lf_level[i] = int(hdr.get("lf_level", (0,) * 4)[i])
    f.lf_sharpness, f.cdef_damping, f.cdef_bits = hdr.get("lf_sharpness", 0), hdr.get("cdef_damping", 3), 0
    f.cdef_y[0], f.cdef_uv[0] = int(hdr.get("cdef_y", (0,))[0]), int(hdr.get("cdef_uv", (0,))[0])
    keep = []
    for p in range(3):
        f.lr_type[p] = int(hdr.get("lr_type", (0, 0, 0))[p])
        u = hdr.get("lr_units", (None,) * 3)[p]
        if u is not None:
            a = np.ascontiguousarray(u, np.int8)
            keep.append(a)
            f.lr_units[p] = a.ctypes.data
    f.lr_unit_shift, f.lr_uv_shift = hdr.get("lr_unit_shift", 0), hdr.get("lr_uv_shift", 0)
    f.tile_cols_log2 = f.tile_rows_log2 = -1
    pay = np.ascontiguousarray(payloads, np.uint8)
    sz = np.ascontiguousarray(sizes, np.uint32)
    cap = int(pay.size) + (1 << 16) + 4 * int(sz.size)
    out = np.empty(cap, np.uint8)
    err = C.create_string_buffer(256)
    L = lib()
    L.av1mi_obu_assemble_temporal_unit.restype = C.c_longlong
    L.av1mi_obu_assemble_temporal_unit.argtypes = [C.POINTER(ObuFrame), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_char_p, C.c_int]
    n = L.av1mi_obu_assemble_temporal_unit(C.byref(f), pay.ctypes.data, sz.ctypes.data, int(sz.size), int(with_sequence_header), out.ctypes.data, cap, err, 256)
    if n < 0 or n > cap:
        raise ValueError("assemble: " + err.value.decode())
    return out[:n].tobytes()


def session_frame_unit_gpu(width, height, bit_depth, frame, seg, with_sequence_header=None, visible=None):
    """the temporal unit of segment `seg` of a collected batch whose tiles were entropy-coded on the GPU (gpu_entropy != 0)"""
    p = frame["params"]
    hdr = header_from_params(p, width, height, frame["lr_on"][seg], visible)
    nt = frame["tiles_per_frame"]
    sizes = frame["tile_size"][seg * nt:(seg + 1) * nt]
    start = int(frame["tile_size"][:seg * nt].sum(dtype=np.uint64))
    pay = frame["tile_payload"][start:start + int(sizes.sum(dtype=np.uint64))]
    sh = (p.frame_type == 0) if with_sequence_header is None else with_sequence_header
    return assemble_temporal_unit(width, height, bit_depth, p.base_q_idx, pay, sizes, with_sequence_header=sh, **hdr)


def session_temporal_unit(width, height, bit_depth, raw_frame, seg, with_sequence_header=True, threads=1, visible=None):
    """av1mi_session_temporal_unit (include/av1mi_host.h): the temporal unit of segment `seg` of a collected batch, from the raw
    av1mi_gop_frame (av1mi.GopSession.collect()["raw"]) — whichever coder produced it, whatever the key frames' block size"""
    h = lib()
    h.av1mi_session_temporal_unit.restype = C.c_longlong
    h.av1mi_session_temporal_unit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong,
                                              C.c_char_p, C.c_int]
    cap = width * height * 4 + (1 << 16)
    out = np.empty(cap, np.uint8)
    err = C.create_string_buffer(256)
    vw, vh = visible if visible is not None else (0, 0)
    n = h.av1mi_session_temporal_unit(C.addressof(raw_frame), int(seg), width, height, bit_depth, int(vw), int(vh), int(with_sequence_header), int(threads),
                                      out.ctypes.data, cap, err, 256)
    if n < 0:
        raise ValueError("av1mi_session_temporal_unit: " + err.value.decode())
    return out[:n].tobytes()


def header_from_params(p, width, height, lr_on=None, visible=None):
    """keyword arguments of temporal_unit() for a frame whose filter parameters are an av1mi_frame_params (GOP session policy);
    lr_on: the encoder's restoration ON / OFF decision per plane (the session's frame["lr_on"][segment]; None = the policy's types);
    visible: the true (width, height) when the coded size is rounded up to 8 (the restoration units tile the TRUE frame)"""
    ur = lambda n: max(1, (n + p.lr_unit_size // 2) // p.lr_unit_size)
    vw, vh = visible if visible is not None else (width, height)
    uy = np.tile(np.array(list(p.lr_unit_y), np.int8), (ur(vh), ur(vw), 1))
    uc = np.tile(np.array(list(p.lr_unit_uv), np.int8), (ur((vh + 1) // 2), ur((vw + 1) // 2), 1))
    shift = {64: 0, 128: 1, 256: 2}[p.lr_unit_size]
    types = [int(p.lr_unit_y[0]), int(p.lr_unit_uv[0]), int(p.lr_unit_uv[0])]
    if lr_on is not None:
        types = [t if int(k) else 0 for t, k in zip(types, lr_on)]
    return dict(frame_type=p.frame_type, lf_level=tuple(p.lf_level), lf_sharpness=p.lf_sharpness, cdef_damping=p.cdef_damping,
                cdef_y=(p.cdef_y,), cdef_uv=(p.cdef_uv,), lr_type=tuple(types), lr_unit_shift=shift, lr_uv_shift=0, lr_units=(uy, uc, uc),
                **({} if visible is None else {"visible": (int(vw), int(vh))}))


def session_frame_unit(width, height, bit_depth, frame, seg, with_sequence_header=None, threads=1, visible=None):
    """the temporal unit of segment `seg` of a collected GOP-session batch (av1mi.GopSession.collect())"""
    p = frame["params"]
    hdr = header_from_params(p, width, height, frame["lr_on"][seg], visible)
    if p.frame_type == 0:
        sym = dict(y_mode=frame["y_mode"][seg], uv_mode=frame["uv_mode"][seg])
    else:
        sym = dict(mv=frame["mv"][seg], skip=frame["skip"][seg])
    sh = (p.frame_type == 0) if with_sequence_header is None else with_sequence_header
    return temporal_unit(width, height, bit_depth, p.base_q_idx, with_sequence_header=sh, threads=threads, lev_y=frame["lev_y"][seg],
                         lev_u=frame["lev_u"][seg], lev_v=frame["lev_v"][seg], **sym, **hdr)
