"""dav1d (the conformant AV1 decoder) as an EXTERNAL checker, through ctypes.

TEST INFRASTRUCTURE ONLY.  The container image ships Pillow >= 11.3 whose bundled libavif (pillow.libs/libavif-*.so)
statically links dav1d 1.5.3 and EXPORTS dav1d's public API (dav1d_open / dav1d_send_data / dav1d_get_picture ...).
It is not part of the reference (IONIQ6000/av1-go) nor of this product; tests use it to decode the Section-5 OBU
streams the host bitstream writer emits and compare the decoded planes with the encoder's own reconstruction:
that pins the normative stages (dequantiser, inverse transforms, intra prediction, motion compensation,
deblocking, CDEF, loop restoration) to a conformant third-party decoder instead of to a self-made oracle.

Dav1dSettings.inloop_filters lets a test switch deblocking / CDEF / restoration off at decode time, so every
in-loop stage can be pinned on its own (pre-filter reconstruction, + deblock, + CDEF, + LR).
"""
import ctypes
import glob
import os

import numpy as np

INLOOP_DEBLOCK, INLOOP_CDEF, INLOOP_RESTORATION, INLOOP_ALL = 1, 2, 4, 7
_LIB = None


def find_library():
    try:
        import PIL
    except ImportError:
        return None
    libs = os.path.join(os.path.dirname(os.path.dirname(PIL.__file__)), "pillow.libs")
    for p in sorted(glob.glob(os.path.join(libs, "libavif-*.so*"))):
        return p
    return None


def load():
    global _LIB
    if _LIB is None:
        p = find_library()
        if p is None:
            return None
        lib = ctypes.CDLL(p)
        if not hasattr(lib, "dav1d_open"):
            return None
        lib.dav1d_version.restype = ctypes.c_char_p
        lib.dav1d_data_create.restype = ctypes.c_void_p
        lib.dav1d_data_create.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        lib.dav1d_open.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]
        lib.dav1d_send_data.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.dav1d_get_picture.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.dav1d_picture_unref.argtypes = [ctypes.c_void_p]
        lib.dav1d_close.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        _LIB = lib
    return _LIB


def available():
    return load() is not None


def version():
    return load().dav1d_version().decode()


class _Picture(ctypes.Structure):
    # dav1d/picture.h, stable over 1.x: seq_hdr, frame_hdr, data[3], stride[2], p{w,h,layout,bpc}, ...
    _fields_ = [("seq_hdr", ctypes.c_void_p), ("frame_hdr", ctypes.c_void_p), ("data", ctypes.c_void_p * 3),
                ("stride", ctypes.c_ssize_t * 2), ("w", ctypes.c_int), ("h", ctypes.c_int), ("layout", ctypes.c_int),
                ("bpc", ctypes.c_int), ("rest", ctypes.c_uint8 * 512)]


_EAGAIN = -11


def decode(obu_bytes, inloop_filters=INLOOP_ALL, strict=True):
    """decode a Section-5 (low-overhead) OBU stream; returns a list of frames [(Y, U, V)] as numpy arrays
    (uint8 for 8-bit, uint16 otherwise), in output order.  Raises RuntimeError when dav1d rejects the stream."""
    lib = load()
    settings = ctypes.create_string_buffer(512)
    lib.dav1d_default_settings(settings)
    ints = ctypes.cast(settings, ctypes.POINTER(ctypes.c_int))
    ints[0] = 1                     # n_threads
    ints[1] = 1                     # max_frame_delay
    ints[2] = 0                     # apply_grain
    assert ints[18] == INLOOP_ALL, "unexpected Dav1dSettings layout"
    ints[16] = 1 if strict else 0   # strict_std_compliance
    ints[18] = inloop_filters
    ctx = ctypes.c_void_p()
    rc = lib.dav1d_open(ctypes.byref(ctx), settings)
    if rc:
        raise RuntimeError("dav1d_open: %d" % rc)
    frames = []

    def drain():
        while True:
            pic = _Picture()
            rc = lib.dav1d_get_picture(ctx, ctypes.byref(pic))
            if rc == _EAGAIN:
                return
            if rc:
                raise RuntimeError("dav1d_get_picture: error %d" % rc)
            frames.append(_planes(pic))
            lib.dav1d_picture_unref(ctypes.byref(pic))

    try:
        data = ctypes.create_string_buffer(128)     # Dav1dData (72 bytes)
        ptr = lib.dav1d_data_create(data, len(obu_bytes))
        if not ptr:
            raise RuntimeError("dav1d_data_create failed")
        ctypes.memmove(ptr, obu_bytes, len(obu_bytes))
        szp = ctypes.cast(ctypes.addressof(data) + 8, ctypes.POINTER(ctypes.c_size_t))
        while szp[0] > 0:
            rc = lib.dav1d_send_data(ctx, data)
            if rc and rc != _EAGAIN:
                raise RuntimeError("dav1d_send_data: error %d (stream rejected)" % rc)
            drain()
        # flush: keep pulling until nothing comes
        for _ in range(4):
            drain()
    finally:
        lib.dav1d_close(ctypes.byref(ctx))
    return frames


def _planes(pic):
    bps = 1 if pic.bpc == 8 else 2
    dt = np.uint8 if bps == 1 else np.uint16
    ss = {0: None, 1: (1, 1), 2: (1, 0), 3: (0, 0)}[pic.layout]     # I400, I420, I422, I444
    out = []
    for i in range(3):
        if i and ss is None:
            break
        w = pic.w if i == 0 else (pic.w + ss[0]) >> ss[0]
        h = pic.h if i == 0 else (pic.h + ss[1]) >> ss[1]
        stride = pic.stride[0 if i == 0 else 1]
        buf = (ctypes.c_uint8 * (stride * h)).from_address(pic.data[i])
        a = np.frombuffer(buf, dtype=np.uint8).reshape(h, stride)[:, :w * bps]
        out.append(np.ascontiguousarray(a).view(dt).reshape(h, w).copy())
    return tuple(out)
