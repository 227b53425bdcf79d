#!/usr/bin/env python3
"""bench.py — encoded-frames/sec of the MI355X block-processing hot path (BASELINE.json metric).

One "step" = one closed-GOP segment of --frames synthetic frames through the device pipeline, inputs
already resident in HBM.  N ranks (one per GPU, launched by torch.distributed.run) each process their
own segments: no data-path collective, scaling is weak.  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 4k10-gop|1080p8-gop|4k10|1080p8] [--segments S | --frames F]

Default workload = the configuration BASELINE.json's metric ("encoded 4K30 frames/sec") is quoted on: configs[3], 4K 10-bit,
closed GOPs of 30 frames (1 key + 29 P), every stage of the block pipeline + the three in-loop filters on one GPU.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "av1-go_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="4k10-gop", choices=["1080p8", "4k10", "1080p8-gop", "4k10-gop"],
                    help="4k10-gop (default) = BASELINE configs[3], the 4K30 10-bit full pipeline the metric is quoted on: closed GOPs "
                         "of 30 frames, 1 key + 29 P; 1080p8-gop = configs[2]; 1080p8 = configs[1] (intra-only); 4k10 = 4K intra-only")
    ap.add_argument("--frames", type=int, default=0, help="intra-only workloads: frames per step (segment length); 0 = default")
    ap.add_argument("--segments", type=int, default=0, help="*-gop workloads: closed GOPs coded in lockstep per step; 0 = default")
    ap.add_argument("--qindex", type=int, default=128)
    ap.add_argument("--entropy", default="none", choices=["none", "gpu", "gpu-async"],
                    help="gpu = the tile entropy coder (K9) runs inside the timed step; none (default) = BASELINE config 2 as "
                         "written (transform + prediction + filters), with the entropy stage timed in a separate leg and "
                         "reported under \"entropy\"; gpu-async = the coder on the context's side stream, overlapping the next step")
    ap.add_argument("--entropy-tile", type=int, default=64, choices=[32, 64, 128])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (upload + pipeline + download + host AV1 entropy coding)")
    ap.add_argument("--e2e-segments", type=int, default=4)
    ap.add_argument("--e2e-steps", type=int, default=2)
    ap.add_argument("--e2e-gpu-segments", type=int, default=8, help="segments in lockstep of the end-to-end leg with GPU entropy coding")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="no GPU: exercise the rank/sharding/timing/aggregation plumbing with a stand-in step (gloo tests)")
    return ap.parse_args()


def cpu_baseline(pipe, seconds=12.0):
    """The oracle (own CPU restatement, kind "port" — the reference's CPU path does not exist in its tree) timed on
    this box's host cores on a bounded sample of the same workload: whole frames of the segment through the same
    stages as the GPU step (intra-only encoder loop + deblocking), one frame per worker thread."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.build()
    cores = min(os.cpu_count() or 1, 16)
    Y, U, V = pipe.src

    def one(f):
        r = O.intra_encode_frame(Y[f], U[f], V[f], pipe.bd, pipe.bs, pipe.qindex)
        dbl = [O.deblock_plane(r["rec_y"], pipe.bd, 0, pipe.mi_y), O.deblock_plane(r["rec_u"], pipe.bd, 1, pipe.mi_c),
               O.deblock_plane(r["rec_v"], pipe.bd, 1, pipe.mi_c)]
        cdef = O.cdef_frame(dbl[0], dbl[1], dbl[2], pipe.bd, pipe.cdef_damping, pipe.cdef_sb, pipe.cdef_skip)
        O.lr_plane(cdef[0], dbl[0], pipe.bd, 0, pipe.lr_unit, pipe.lr_units_y)
        O.lr_plane(cdef[1], dbl[1], pipe.bd, 1, pipe.lr_unit, pipe.lr_units_c)
        O.lr_plane(cdef[2], dbl[2], pipe.bd, 1, pipe.lr_unit, pipe.lr_units_c)
        return 1

    done, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        while time.perf_counter() - t0 < seconds:
            done += sum(ex.map(one, [(done + i) % pipe.frames for i in range(cores)]))
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d frames of the same segment through the oracle's intra encoder loop + deblock + CDEF + loop "
                      "restoration (same stages as the GPU step) in %.1f s" % (done, dt)}


def cpu_baseline_gop(pipe, p_frames=2):
    """Closed-GOP workloads: the oracle chain (kind "port") on this box's host cores, one GOP per worker thread, each coding
    the key frame and the first `p_frames` P frames of its GOP through the same stages as the GPU step (encoder loop +
    deblocking + CDEF + loop restoration; the P frames reference the worker's own loop-filtered reconstruction).  A GOP is
    serial in itself, so the whole-GOP rate follows from the two per-frame costs: 30 frames / (t_key + 29 t_P), times the
    number of workers — the sample is bounded to a few frames per worker because one 4K P frame is ~10 s of CPU work."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.build()
    cores = min(os.cpu_count() or 1, 16)
    k, gop = pipe.key, pipe.gop
    h, w = pipe.height, pipe.width

    def filters(r, skip8, t, src):
        mi_y, mi_c, damping, cdef_sb, lr_unit, lr_y, lr_c = pipe.oracle_filter_args(t)
        dbl = [O.deblock_plane(r["rec_y"], pipe.bd, 0, mi_y), O.deblock_plane(r["rec_u"], pipe.bd, 1, mi_c), O.deblock_plane(r["rec_v"], pipe.bd, 1, mi_c)]
        cdef = O.cdef_frame(dbl[0], dbl[1], dbl[2], pipe.bd, damping, cdef_sb, skip8)
        lr = [O.lr_plane(cdef[0], dbl[0], pipe.bd, 0, lr_unit, lr_y), O.lr_plane(cdef[1], dbl[1], pipe.bd, 1, lr_unit, lr_c),
              O.lr_plane(cdef[2], dbl[2], pipe.bd, 1, lr_unit, lr_c)]
        return O.lr_select(src, cdef, lr, pipe.bd)[0]      # the restoration ON / OFF decision against the source

    def one(worker):
        s = worker % pipe.segments
        t0 = time.perf_counter()
        src = [pipe.src[0][i][s] for i in range(3)]
        ref = filters(O.intra_encode_frame(src[0], src[1], src[2], pipe.bd, 8, pipe.qindex), np.zeros((h // 8, w // 8), np.uint8), 0, src)
        t1 = time.perf_counter()
        for t in range(1, 1 + p_frames):
            src = [pipe.src[t][i][s] for i in range(3)]
            r = O.inter_encode_frame(src, ref, pipe.bd, pipe.qindex, pipe.range)
            ref = filters(r, r["skip"].reshape(h // 8, w // 8), t, src)
        return t1 - t0, (time.perf_counter() - t1) / p_frames

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(one, range(cores)))
    wall = time.perf_counter() - t0
    t_key, t_p = float(np.mean([r[0] for r in res])), float(np.mean([r[1] for r in res]))
    return {"value": cores * gop / (t_key + (gop - 1) * t_p), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d workers x (key frame + %d P frames) of the same GOPs through the oracle's encoder loops + deblock + CDEF + loop "
                      "restoration in %.1f s wall: %.2f s per key frame, %.2f s per P frame per core; value = cores x %d / (t_key + %d t_P)"
                      % (cores, p_frames, wall, t_key, t_p, gop, gop - 1)}


def copy_bandwidth(ctx, nbytes=1 << 30, reps=10):
    """device-to-device copy rate measured on this box (report against the 8 TB/s spec AND a measured copy):
    GB/s moved = 2 x bytes (read + written) / time, HIP events on the context's stream"""
    a, b = ctx.alloc(nbytes), ctx.alloc(nbytes)
    ctx.memset(a, 1, nbytes)
    ctx.copy(b, a, nbytes)
    ctx.sync()
    ctx.timer_begin()
    for _ in range(reps):
        ctx.copy(b, a, nbytes)
    ms = ctx.timer_end()
    a.free()
    b.free()
    return 2.0 * nbytes * reps / (ms * 1e-3) / 1e9


def quality(pipe, bd):
    """PSNR-Y of the loop-filtered reconstruction against the source, on the frames the last step left in HBM (the metric's
    second half, "PSNR-Y delta vs libaom", needs libaom on the box: absent here, so only the absolute value is reported)"""
    peak = float((1 << bd) - 1)
    if hasattr(pipe, "d_ref"):                     # closed GOPs: the last P frames of all segments
        src = pipe.src[pipe.gop - 1][0]
        rec = pipe.d_ref[0].download(src.shape, src.dtype)
        # what a decoder outputs: the restored plane, or the CDEF output in the segments where restoration was switched off
        on, cdef = pipe.lr_on(pipe.gop - 1)[:, 0], pipe.key.d["cdef_y"].download(src.shape, src.dtype)
        rec = np.where(on[:, None, None] != 0, rec, cdef)
    else:
        src = pipe.src[0]
        rec = pipe.d["out_y"].download(src.shape, src.dtype)
    mse = float(np.mean((rec.astype(np.float64) - src.astype(np.float64)) ** 2))
    extra = {"restoration_on_in_segments": int(on.sum())} if hasattr(pipe, "d_ref") else {}
    return {"psnr_y_db": 10.0 * np.log10(peak * peak / mse) if mse > 0 else None, "frames": int(src.shape[0]), **extra,
            "note": "fixed qindex, no rate control; the comparison with libaom on the same key frame is under e2e.vs_libaom"}


def entropy_leg(ctx, pipe, args, launches=5, host_seconds=6.0):
    """The stage after the block pipeline, timed on the levels + modes the last step left in HBM: (a) the GPU tile entropy
    coder (K9), HIP events on the pipeline's stream; (b) the host coder of the same syntax (host/entropy.cpp, the stage
    BASELINE's north_star keeps on the host cores) on a bounded sample of the same frames, all host threads."""
    import ctypes as C
    ctx.entropy_encode(pipe.ent_job)          # warm: grows the context's scratch
    ctx.sync()
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(launches):
        ctx.entropy_encode(pipe.ent_job)
    ctx.sync()
    ctx.prof_enable(False)
    prof = ctx.prof_get()
    recs = pipe.coded_records()
    code_ms = prof["entropy_code"][1] / prof["entropy_code"][0]
    pack_ms = prof["entropy_pack"][1] / prof["entropy_pack"][0]
    tok_ms = prof["entropy_tokens"][1] / prof["entropy_tokens"][0]
    out = {"syntax": "own (AV1 range-coder arithmetic + CDF adaptation, spec 8.2.6); not an AV1 bitstream",
           "tile": pipe.entropy_tile, "bytes_per_frame": sum(len(r) for r in recs) / len(recs),
           "gpu": {"tokens_ms_per_launch": tok_ms, "code_ms_per_launch": code_ms, "pack_ms_per_launch": pack_ms,
                   "frames_per_launch": pipe.frames, "frames_per_s": pipe.frames / ((tok_ms + code_ms + pack_ms) * 1e-3),
                   "levels_GBps": 2 * pipe.samples / (code_ms * 1e-3) / 1e9}}
    host = os.path.join(ROOT, "av1-go_amd", "host", "libav1mi_host.so")
    if os.path.exists(host):
        lib = C.CDLL(host)
        P = C.c_void_p
        lib.av1mi_host_entropy_encode_stack.argtypes = [C.c_int] * 5 + [P] * 5
        lib.av1mi_host_entropy_encode_stack.restype = C.c_longlong
        threads = min(os.cpu_count() or 1, 16)
        n = min(pipe.frames, threads)
        w, h = pipe.width, pipe.height
        nb = (w // 8) * (h // 8)
        lv = [pipe.d["lev_" + p].download((pipe.frames, nb * (64 if p == "y" else 16)), np.int16)[:n].copy() for p in "yuv"]
        md = [pipe.d[k].download((pipe.frames, nb), np.uint8)[:n].copy() for k in ("modes_y", "modes_uv")]
        vp = lambda a: a.ctypes.data_as(P)
        done, tot, t0 = 0, 0, time.perf_counter()
        while time.perf_counter() - t0 < host_seconds:
            tot = lib.av1mi_host_entropy_encode_stack(w, h, n, threads, pipe.entropy_tile, *[vp(a) for a in lv + md])
            done += n
        dt = time.perf_counter() - t0
        out["host"] = {"frames_per_s": done / dt, "threads": threads, "sample": "%d frames in %.1f s" % (done, dt),
                       "bytes_match_gpu": tot == sum(len(r) for r in recs[:n])}
    return out


def probe_baseline_tools():
    """the preferred CPU baseline is FFmpeg + libsvtav1 on the same frames; record what this box actually has."""
    import shutil
    import subprocess
    tools = {n: shutil.which(n) is not None for n in ("ffmpeg", "ffprobe", "SvtAv1EncApp", "aomenc", "aomdec", "dav1d", "go")}
    tools["ffmpeg_has_libsvtav1"] = False
    if tools["ffmpeg"]:
        try:
            enc = subprocess.run(["ffmpeg", "-hide_banner", "-encoders"], capture_output=True, text=True, timeout=20).stdout
            tools["ffmpeg_has_libsvtav1"] = "libsvtav1" in enc
        except Exception:
            pass
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import dav1d_ref
        tools["bundled_libavif"] = dav1d_ref.find_library() is not None
        tools["bundled_dav1d"] = dav1d_ref.version() if dav1d_ref.available() else None
    except Exception:
        tools["bundled_libavif"], tools["bundled_dav1d"] = False, None
    return tools


def usable_cpus(cap=16):
    """host threads for the entropy stage: what this process may actually use (affinity mask, cgroup quota), capped at the GPU
    box's per-GPU CPU share (16) unless AV1MI_HOST_THREADS says otherwise"""
    if os.environ.get("AV1MI_HOST_THREADS"):
        return max(1, int(os.environ["AV1MI_HOST_THREADS"]))
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def libaom_leg(planes, bd, qindex, our_bytes, our_rec_y, threads):
    """"PSNR-Y delta vs libaom" (BASELINE.json's metric, second half) for the KEY frame the decoder check just coded: libaom 3.13
    (the AVIF encoder inside the image's bundled libavif; tools/libaom_compare.py) codes the same 4:2:0 planes as a still image at
    the quantizer that maps to the same base_q_idx and at three finer ones; dav1d decodes; PSNR-Y against the source.  Reports the
    delta at the same quantizer and at equal size (log-size interpolation), and libaom's own speed on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import libaom_compare as LC
    import dav1d_ref as D
    Y = planes[0]
    ours_psnr = LC.psnr(our_rec_y, Y, bd)
    qz = LC.quantizer_for_qindex(qindex)
    pts = {}
    for q in sorted({max(qz - d, 0) for d in (16, 8, 4, 0)}):
        obus, dt = LC.libaom_encode_still(planes[0], planes[1], planes[2], bd, q, 6, threads)
        dec = D.decode(obus, strict=False)[0]
        pts[q] = (len(obus), LC.psnr(dec[0], Y, bd), dt)
    xs = np.log([pts[q][0] for q in sorted(pts)][::-1])
    ys = [pts[q][1] for q in sorted(pts)][::-1]
    eq = float(np.interp(np.log(our_bytes), xs, ys)) if xs[0] <= np.log(our_bytes) <= xs[-1] else None
    return {"frame": "key frame 0 of the workload", "ours": {"bytes": our_bytes, "psnr_y_db": ours_psnr},
            "libaom": {"version": "3.13, all-intra still, AOM_Q, speed 6, %d threads" % threads, "matching_quantizer": qz,
                       "by_quantizer": {str(q): {"bytes": v[0], "psnr_y_db": v[1], "seconds": v[2]} for q, v in pts.items()}},
            "psnr_y_delta_db_at_same_quantizer": ours_psnr - pts[qz][1], "size_ratio_at_same_quantizer": our_bytes / pts[qz][0],
            "psnr_y_delta_db_at_equal_size": None if eq is None else ours_psnr - eq,
            "libaom_frames_per_s_on_host": 1.0 / pts[qz][2]}


def e2e_leg(ctx, W, H, bd, qindex, first_frame, segs=4, gop=30, steps=2, warmup_frames=2, gpu_entropy=0, compare_libaom=True):
    """END TO END: what the transcode job does per frame (reference: file in -> file out, internal/ffmpeg/transcode.go:194-203):
    source planes from host memory into the session's pinned buffers, H2D upload, block pipeline + in-loop filters on the GPU,
    D2H of the symbols, AV1 entropy coding + OBU packing on all host cores (north_star keeps that stage on the host).  Three
    batches in flight: the host codes frame t while the GPU works on frames t + 1 and t + 2.  Timed with the wall clock; the product is
    a decodable AV1 stream (its first frames are decoded with dav1d, when present, and compared with the GPU's reference)."""
    from concurrent.futures import ThreadPoolExecutor
    import av1mi
    import av1stream
    import synth
    threads = usable_cpus()
    t_gen = time.perf_counter()
    Y, U, V = synth.frames(W, H, segs * gop, bd, first_frame)
    src = [a.reshape(segs, gop, *a.shape[1:]) for a in (Y, U, V)]
    t_gen = time.perf_counter() - t_gen
    sess = av1mi.GopSession(ctx, W, H, bd, qindex, gop, segs, gpu_entropy=gpu_entropy)
    pool = ThreadPoolExecutor(min(threads, 3 * segs))
    coded = {"bytes": 0, "frames": 0, "t_fill": 0.0, "t_code": 0.0, "t_wait": 0.0}

    def fill(t):
        t0 = time.perf_counter()
        planes = sess.input_planes()
        jobs = []
        for p in range(3):
            hh = H if p == 0 else H // 2
            for sg in range(segs):
                jobs.append(pool.submit(np.copyto, planes[p][sg * hh:(sg + 1) * hh], src[p][sg, t]))
        for j in jobs:
            j.result()
        coded["t_fill"] += time.perf_counter() - t0

    def code(keep=None):
        t0 = time.perf_counter()
        fr = sess.collect()
        t1 = time.perf_counter()
        for sg in range(segs):
            if gpu_entropy and "tile_size" in fr:      # the tiles were coded on the GPU: the host only wraps them (frame header, tile-size fields)
                tu = av1stream.session_frame_unit_gpu(W, H, bd, fr, sg)
            else:
                tu = av1stream.session_frame_unit(W, H, bd, fr, sg, threads=threads)
            coded["bytes"] += len(tu)
            coded["frames"] += 1
            if keep is not None and sg == 0:
                keep.append(tu)
        coded["t_wait"] += t1 - t0
        coded["t_code"] += time.perf_counter() - t1

    lag = sess.max_in_flight() - 1       # batches the GPU holds while the host works on the oldest

    def run_gop(nframes, keep=None):
        for t in range(nframes):
            fill(t)
            sess.submit(0 if t == 0 else 1)
            if t >= lag:
                code(keep)
        while sess.pending():
            code(keep)

    run_gop(warmup_frames)
    for k in coded:
        coded[k] = 0
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_gop(gop)
    ctx.sync()
    dt = time.perf_counter() - t0
    frames = coded["frames"]
    out = {"entropy_fallbacks": int(sess.entropy_fallbacks()) if gpu_entropy else None, "frames_per_s": frames / dt, "frames": frames, "seconds": dt, "segments_in_lockstep": segs, "gop": gop, "host_threads": threads,
           "bytes_per_frame": coded["bytes"] / frames, "mbit_per_s_at_30fps": coded["bytes"] / frames * 8 * 30 / 1e6,
           "host_seconds": {"fill_pinned_input": coded["t_fill"], "wait_for_gpu": coded["t_wait"],
                            "assemble_obu" if gpu_entropy else "entropy_code": coded["t_code"]},
           "pcie_bytes_per_frame": {"up": W * H * 3 // 2 * (1 if bd == 8 else 2),
                                    "down": coded["bytes"] / frames if gpu_entropy else W * H * 3 + (W // 8) * (H // 8) * 5},
           "entropy_coding": "GPU (k_av1_*: AV1 tile syntax, one lane per tile, side stream)" if gpu_entropy else "host, %d threads" % threads,
           "what": ("pinned host source -> H2D -> block pipeline + deblock + CDEF + LR + AV1 tile entropy coder (GPU) -> D2H tile payloads -> "
                    "frame header + tile group assembly on the host" if gpu_entropy else
                    "pinned host source -> H2D -> block pipeline + deblock + CDEF + LR (GPU) -> D2H symbols -> AV1 entropy coding + OBU "
                    "packing on %d host threads" % threads) + "; synthetic source generated beforehand (%.1f s, not timed)" % t_gen}
    # the stream is real: decode the first frames of segment 0 and compare with the reference frames the GPU keeps
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import dav1d_ref as D
        if D.available():
            check = av1mi.GopSession(ctx, W, H, bd, qindex, gop, 1, gpu_entropy=gpu_entropy)
            units, refs = [], []
            for t in range(2):
                planes = check.input_planes()
                for p in range(3):
                    np.copyto(planes[p], src[p][0, t])
                check.submit(0 if t == 0 else 1)
                fr = check.collect()
                units.append(av1stream.session_frame_unit_gpu(W, H, bd, fr, 0) if gpu_entropy else av1stream.session_frame_unit(W, H, bd, fr, 0, threads=threads))
                refs.append(check.download_reference())
            check.close()
            dec = D.decode(b"".join(units))
            ok = len(dec) == 2 and all((dec[t][p] == refs[t][p]).all() for t in range(2) for p in range(3))
            out["decoder_check"] = {"decoder": "dav1d " + D.version(), "frames": 2, "bit_exact_vs_gpu_reference": bool(ok)}
            if compare_libaom:
                try:
                    if bd == 8:
                        out["vs_libaom"] = libaom_leg([src[p][0, 0] for p in range(3)], bd, qindex, len(units[0]), refs[0][0], threads)
                    else:
                        # the image's libaom is an 8-bit build (it refuses 10-bit planes): compare on the 8-bit rendition of the same
                        # synthetic frame, coded by a one-frame session of its own
                        Y8, U8, V8 = synth.frames(W, H, 1, 8, first_frame)
                        s8 = av1mi.GopSession(ctx, W, H, 8, qindex, 1, 1, gpu_entropy=gpu_entropy)
                        for dst, a in zip(s8.input_planes(), (Y8[0], U8[0], V8[0])):
                            np.copyto(dst, a)
                        s8.submit(0)
                        fr8 = s8.collect()
                        tu8 = av1stream.session_frame_unit_gpu(W, H, 8, fr8, 0) if gpu_entropy else av1stream.session_frame_unit(W, H, 8, fr8, 0, threads=threads)
                        rec8 = s8.download_reference()[0]
                        s8.close()
                        out["vs_libaom"] = libaom_leg([Y8[0], U8[0], V8[0]], 8, qindex, len(tu8), rec8, threads)
                        out["vs_libaom"]["note"] = "8-bit rendition of the workload's first frame: the bundled libaom is built without high bit depth"
                except Exception as e:
                    out["vs_libaom"] = {"error": repr(e)[:200]}
    except Exception as e:       # the check is a courtesy of the bench, the tests are the gate
        out["decoder_check"] = {"error": repr(e)[:200]}
    pool.shutdown()
    sess.close()
    return out


PMC_KERNEL = {"intra_pipeline": "k_intra_pipe", "deblock": "k_deblock", "cdef": "k_cdef", "loop_restoration": "k_lr",
              "inter_pipeline": "k_inter_pipe", "me_integer": "k_me_int"}


def pmc_traffic(kind, workload, frames):
    """HBM bytes per launch of the kernel behind `kind` from the committed rocprofv3 PMC passes (profiles/, made by
    tools/prof_pmc.sh + tools/pmc_to_json.py on the same command): reads = 2 x FETCH_SIZE (gfx950 tallies a 128-B read
    request as 64 B; confirmed on this box for 8- and 16-byte-per-lane streams, see "calibration" in the file) + WRITE_SIZE,
    scaled to this run's frames per LAUNCH (intra-only: the segment; closed GOPs: one frame of every GOP in lockstep).
    None when no matching profile is committed (PMC cannot be read live)."""
    tag = {"1080p8": "1080p8_intra", "4k10-gop": "4k10_gop"}.get(workload)
    if tag is None:
        return None
    path = os.path.join(ROOT, "profiles", "pmc_%s_latest.json" % tag)
    try:
        prof = json.load(open(path))
    except (OSError, ValueError):
        return None
    for name, d in prof["kernels"].items():
        if name.startswith(PMC_KERNEL.get(kind, "?")):
            corr = (prof.get("calibration") or {}).get("read_correction_8B_per_lane") or 2.0
            return (d["fetch_bytes_uncorrected"] * corr + d["write_bytes"]) * frames / prof["frames_per_step"]
    return None


def segment_of_rank(rank, frames_per_step):
    """closed-GOP sharding: rank r codes the r-th run of frames_per_step frames (no data exchanged between ranks)"""
    return rank * frames_per_step


def aggregate(dist, dt, device):
    """max over ranks of the timed region (the job is as slow as its slowest GPU)"""
    if dist is None:
        return dt
    import torch
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def dry_run(args, rank, world, dist):
    """CPU stand-in used by tests/test_distributed.py: same control flow as the GPU path, a sleep as the step."""
    frames = args.frames or 4
    first = segment_of_rank(rank, frames)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))        # uneven ranks: the aggregate must follow the slowest
    if dist is not None:
        dist.barrier()
    dt = aggregate(dist, time.perf_counter() - t0, "cpu")
    if rank == 0:
        print(json.dumps({"metric": "dry-run", "value": frames * args.steps * world / dt, "unit": "frames/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "scaling": "weak",
                          "first_frame_rank0": first, "frames_per_step": frames}))
    if dist is not None:
        dist.destroy_process_group()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU) as CHILD processes — before
    this process has imported torch or made any HIP call — with the environment torch.distributed.run would give them, relay
    rank 0's JSON line and exit with the first non-zero status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = procs[0].communicate()[0]
    codes = [p.wait() for p in procs]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [c for c in codes if c]
    if bad:
        sys.exit("bench.py: rank(s) failed with exit codes %s" % codes)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d does not match WORLD_SIZE %d of the launcher" % (args.gpus, world))
    dist = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch
        import torch.distributed as dist
        use_cuda = torch.cuda.is_available() and not args.dry_run_cpu
        if use_cuda:
            torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl" if use_cuda else "gloo")   # nccl == RCCL on ROCm; only barrier + max-reduce use it
        sync_t = torch.zeros(1, device="cuda" if use_cuda else "cpu")

    if args.dry_run_cpu:
        return dry_run(args, rank, world, dist)

    import av1mi
    import pipeline

    if args.workload.startswith("1080p8"):
        W, H, bd = 1920, 1080, 8
        frames = args.frames or 48   # 48 frames x 510 tiles / 8 tiles per wave = 12 waves per CU: one full generation
    else:
        W, H, bd = 3840, 2160, 10
        frames = args.frames or 8
    ctx = av1mi.Context(local_rank)
    if args.workload.endswith("-gop"):
        gop = 30
        # GOPs coded in lockstep per step (the t-th frames of all of them share a launch): 12 x 2040 tiles at 4K = 3 waves per
        # SIMD for the coding kernels, their occupancy limit; 24 x 510 at 1080p likewise.  --frames F keeps its old meaning (F/2 GOPs).
        segs = args.segments or (max(1, args.frames // 2) if args.frames else (24 if bd == 8 else 12))
        frames = segs * gop
        pipe = pipeline.GopPipeline(ctx, W, H, bd, segs, gop, args.qindex, first_frame=segment_of_rank(rank, frames),
                                    entropy_tile=args.entropy_tile if args.entropy != "none" else 0, entropy_async=args.entropy == "gpu-async")
    else:
        pipe = pipeline.IntraPipeline(ctx, W, H, bd, frames, args.qindex, first_frame=segment_of_rank(rank, frames),
                                      entropy_tile=args.entropy_tile, entropy_async=args.entropy == "gpu-async")
        pipe.entropy_in_step = args.entropy != "none"

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.all_reduce(sync_t)
            if sync_t.is_cuda:
                import torch
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.step()
    barrier()
    # every launch of the timed region is bracketed by its own HIP event pair on the pipeline's stream
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.step()
    barrier()
    dt = time.perf_counter() - t0
    ctx.prof_enable(False)
    prof = ctx.prof_get()
    dt = aggregate(dist, dt, sync_t.device if dist is not None else None)

    total_frames = frames * args.steps * world
    fps = total_frames / dt
    out = {
        "metric": "encoded 4K30 frames/sec (whole node) at fixed QP; PSNR-Y delta vs libaom",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8" if bd == 8 else "u16", "data": "synthetic",
        "config": {"workload": pipe.describe(), "value_is": "DEVICE-RESIDENT block pipeline + in-loop filters (source frames already in HBM, symbols left "
                   "in HBM: the bench contract's definition of `value`); the end-to-end encode rates (PCIe both ways + AV1 entropy coding, a "
                   "decodable stream verified by dav1d) are `e2e_frames_per_s` (entropy coding on the host cores, north_star's split) and "
                   "`e2e_gpu_entropy_frames_per_s` (the same bytes from the GPU tile entropy coder)", "frames_per_step": frames, "qindex": args.qindex,
                   "sharding": "closed-GOP segment per GPU, no collective", "device": ctx.device_name},
    }
    if rank == 0:
        # per-kernel roofline of the dominant kernel: HIP events on the pipeline's own stream
        alg = pipe.algorithmic_bytes()
        dom = max(prof, key=lambda k: prof[k][1])
        n, ms = prof[dom]
        ach = alg[dom] / (ms / n * 1e-3) / 1e9
        out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBPS, "traffic": pmc_traffic(dom, args.workload, pipe.segments if args.workload.endswith("-gop") else frames), "algorithmic_bytes_per_launch": alg[dom],
                           "avg_launch_ms": ms / n, "launches": n}
        copy_gbps = copy_bandwidth(ctx)
        out["roofline"]["copy_GBps_measured"] = copy_gbps
        out["roofline"]["frac_of_measured_copy"] = ach / copy_gbps
        # whole pipeline: (9b + 2) S per inter frame, (8b + 2) S per intra-only frame, times the job's frame rate
        b_, s_frame = (1 if bd == 8 else 2), W * H * 3 // 2
        per_frame = ((9 * b_ + 2) if args.workload.endswith("-gop") else (8 * b_ + 2)) * s_frame
        out["pipeline_roofline"] = {"algorithmic_bytes_per_frame": per_frame, "achieved": per_frame * fps / world / 1e9, "peak": HBM_PEAK_GBPS,
                                    "unit": "GB/s", "frac": per_frame * fps / world / 1e9 / HBM_PEAK_GBPS,
                                    "note": "per GPU: whole-job frames/s / n_gpus x the fused-pipeline bytes of BASELINE.md §3"}
        out["kernels"] = {k: {"launches": v[0], "avg_ms": v[1] / v[0],
                              "algorithmic_GBps": alg[k] / (v[1] / v[0] * 1e-3) / 1e9 if k in alg else None} for k, v in prof.items()}
        out["quality"] = quality(pipe, bd)
        out["baseline_tools"] = probe_baseline_tools()
        if world == 1 and not args.workload.endswith("-gop"):
            out["entropy"] = entropy_leg(ctx, pipe, args)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_gop(pipe) if args.workload.endswith("-gop") else cpu_baseline(pipe)
        else:
            out["cpu_baseline"] = None
    pipe.close()
    if rank == 0:
        if world == 1 and not args.no_e2e and args.workload.endswith("-gop"):
            e2e = e2e_leg(ctx, W, H, bd, args.qindex, segment_of_rank(rank, frames), segs=args.e2e_segments, steps=args.e2e_steps)
            out["e2e"] = e2e                      # north_star's split: entropy coding on the host cores
            out["e2e_frames_per_s"] = e2e["frames_per_s"]
            g = e2e_leg(ctx, W, H, bd, args.qindex, segment_of_rank(rank, frames), segs=args.e2e_gpu_segments, steps=args.e2e_steps, gpu_entropy=1,
                        compare_libaom=False)
            out["e2e_gpu_entropy"] = g            # the same stream, byte for byte, with the tile entropy coder on the GPU
            out["e2e_gpu_entropy_frames_per_s"] = g["frames_per_s"]
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
