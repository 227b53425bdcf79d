#!/usr/bin/env python3
"""PSNR-Y / size of this encoder's key frames next to libaom's on the SAME frames (the second half of BASELINE.json's metric:
"PSNR-Y delta vs libaom").

libaom 3.13 is inside the image: Pillow's bundled libavif (pillow.libs/libavif-*.so) links it as the AVIF ENCODER and exports
the libavif C API.  This tool feeds it the synthetic 4:2:0 planes untouched (avifImage YUV planes, no RGB round trip), asks
for a still image at a fixed quantizer (libavif quantizer 0..63 -> libaom's quantizer_to_qindex: 32 -> base_q_idx 128, all-intra
usage, AOM_Q rate control), pulls the OBUs out of the AVIF's mdat box, decodes them with dav1d and measures PSNR-Y against
the source.  Our side: the oracle's key-frame chain (== the GPU's, tests/test_gpu_*) at base_q_idx 128, coded by the host
bitstream writer, decoded by the same dav1d.

TEST / MEASUREMENT INFRASTRUCTURE: nothing here is used by the product.  Prints one JSON object.

    python tools/libaom_compare.py [--size 1920x1080] [--bd 8] [--qindex 128] [--frames 2] [--speed 6]
"""
import argparse
import ctypes as C
import json
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "av1-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import dav1d_ref as D   # noqa: E402

AVIF_PIXEL_FORMAT_YUV420, AVIF_PLANES_YUV, AVIF_ADD_IMAGE_FLAG_SINGLE = 3, 1, 2


class RWData(C.Structure):
    _fields_ = [("data", C.c_void_p), ("size", C.c_size_t)]


def psnr(a, b, bd):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 10.0 * np.log10(((1 << bd) - 1) ** 2 / mse) if mse > 0 else float("inf")


_QINDEX = [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64, 68, 72, 76, 80, 84, 88, 92, 96, 100, 104, 108, 112, 116, 120, 124, 128, 132,
           136, 140, 144, 148, 152, 156, 160, 164, 168, 172, 176, 180, 184, 188, 192, 196, 200, 204, 208, 212, 216, 220, 224, 228, 232, 236, 240,
           244, 249, 255]      # libaom quantizer_to_qindex


def quantizer_for_qindex(qindex):
    return int(np.argmin([abs(q - qindex) for q in _QINDEX]))


def libaom_encode_still(Y, U, V, bd, quantizer, speed, threads=8):
    """one 4:2:0 frame -> (OBU bytes of the AVIF's mdat, seconds)"""
    L = C.CDLL(D.find_library())
    L.avifImageCreate.restype = C.c_void_p
    L.avifImageCreate.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.avifImageAllocatePlanes.argtypes = [C.c_void_p, C.c_int]
    L.avifImagePlane.restype = C.c_void_p
    L.avifImagePlane.argtypes = [C.c_void_p, C.c_int]
    L.avifImagePlaneRowBytes.argtypes = [C.c_void_p, C.c_int]
    L.avifEncoderCreate.restype = C.c_void_p
    L.avifEncoderAddImage.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
    L.avifEncoderFinish.argtypes = [C.c_void_p, C.POINTER(RWData)]
    L.avifEncoderDestroy.argtypes = [C.c_void_p]
    L.avifImageDestroy.argtypes = [C.c_void_p]
    L.avifRWDataFree.argtypes = [C.POINTER(RWData)]
    L.avifResultToString.restype = C.c_char_p
    h, w = Y.shape
    img = L.avifImageCreate(w, h, bd, AVIF_PIXEL_FORMAT_YUV420)
    assert L.avifImageAllocatePlanes(img, AVIF_PLANES_YUV) == 0
    for ch, a in enumerate((Y, U, V)):
        ptr, rb = L.avifImagePlane(img, ch), L.avifImagePlaneRowBytes(img, ch)
        a = np.ascontiguousarray(a)
        for r in range(a.shape[0]):
            C.memmove(ptr + r * rb, a[r].ctypes.data, a[r].nbytes)
    enc = L.avifEncoderCreate()
    f = C.cast(enc, C.POINTER(C.c_int32))
    # avifEncoder (libavif 1.4): codecChoice, maxThreads, speed, keyframeInterval, timescale (u64), repetitionCount, extraLayerCount,
    # quality, qualityAlpha, minQuantizer, maxQuantizer, ...  — checked against the defaults the constructor writes
    assert (f[1], f[2], f[10], f[11]) == (1, -1, 0, 63), "unexpected avifEncoder layout"
    f[1], f[2], f[10], f[11] = threads, speed, quantizer, quantizer
    t0 = time.perf_counter()
    rc = L.avifEncoderAddImage(enc, img, 1, AVIF_ADD_IMAGE_FLAG_SINGLE)
    out = RWData()
    if rc == 0:
        rc = L.avifEncoderFinish(enc, C.byref(out))
    dt = time.perf_counter() - t0
    if rc:
        raise RuntimeError("libavif: " + L.avifResultToString(rc).decode())
    data = C.string_at(out.data, out.size)
    L.avifRWDataFree(C.byref(out))
    L.avifEncoderDestroy(enc)
    L.avifImageDestroy(img)
    pos, mdat = 0, None
    while pos < len(data):
        sz, typ = struct.unpack(">I4s", data[pos:pos + 8])
        if typ == b"mdat":
            mdat = data[pos + 8:pos + sz]
        pos += sz
    return mdat, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--bd", type=int, default=8)
    ap.add_argument("--qindex", type=int, default=128)
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--speed", type=int, default=6)
    args = ap.parse_args()
    w, h = (int(x) for x in args.size.split("x"))
    import av1stream
    import pipeline as P
    import synth
    from oracle import oracle as O
    from test_av1_conformance import _chain
    O.build()
    Y, U, V = synth.frames(w, h, args.frames, args.bd, 3)
    quantizer = quantizer_for_qindex(args.qindex)
    ours, aom = [], {}
    for t in range(args.frames):
        r, hdr, stages = _chain(O, P, Y[t], U[t], V[t], args.bd, args.qindex)
        tu = av1stream.temporal_unit(w, h, args.bd, args.qindex, y_mode=r["modes_y"], uv_mode=r["modes_uv"], lev_y=r["lev_y"], lev_u=r["lev_u"],
                                     lev_v=r["lev_v"], **hdr)
        dec = D.decode(tu)[0]
        assert (dec[0] == stages[3][0]).all()
        ours.append((len(tu), psnr(dec[0], Y[t], args.bd)))
    # libaom at the matching quantizer and a sweep below it (our stream is the larger one): the equal-size comparison interpolates between them
    for qz in sorted({max(quantizer - d, 0) for d in (20, 16, 12, 8, 4, 0)} | {min(quantizer + 4, 63)}):
        pts = []
        for t in range(args.frames):
            obus, dt = libaom_encode_still(Y[t], U[t], V[t], args.bd, qz, args.speed)
            dec = D.decode(obus, strict=False)[0]
            pts.append((len(obus), psnr(dec[0], Y[t], args.bd), dt))
        aom[qz] = [float(np.mean([p[i] for p in pts])) for i in range(3)]
    ob, op = float(np.mean([o[0] for o in ours])), float(np.mean([o[1] for o in ours]))
    # libaom's PSNR at OUR size (log-size interpolation over its quantizer sweep)
    xs = np.log([aom[q][0] for q in sorted(aom)][::-1])
    ys = [aom[q][1] for q in sorted(aom)][::-1]
    at_equal_size = float(np.interp(np.log(ob), xs, ys)) if xs[0] <= np.log(ob) <= xs[-1] else None
    out = {"frame": "%dx%d %d-bit synthetic key frames (%d)" % (w, h, args.bd, args.frames), "qindex": args.qindex,
           "ours": {"bytes_per_frame": ob, "psnr_y_db": op, "tools": "8x8 blocks, 13 intra modes by SAD, DCT only, no RDO, filter policy from q + restoration on/off by squared error"},
           "libaom": {"version": "3.13 (bundled libavif %s), all-intra, AOM_Q, speed %d" % ("1.4", args.speed),
                      "by_quantizer": {str(q): {"bytes_per_frame": v[0], "psnr_y_db": v[1], "seconds_per_frame": v[2]} for q, v in aom.items()},
                      "matching_quantizer": quantizer},
           "psnr_y_delta_db_at_same_quantizer": op - aom[quantizer][1],
           "size_ratio_at_same_quantizer": ob / aom[quantizer][0],
           "psnr_y_delta_db_at_equal_size": None if at_equal_size is None else op - at_equal_size}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
