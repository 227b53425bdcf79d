#!/usr/bin/env python3
"""bench.py — encoded frames/sec of the MI355X AV1 backend (BASELINE.json metric), measured on the PRODUCT's path: the GOP
session of libav1mi.so (csrc/gop_session.hip: block pipeline + in-loop filters + restoration decision + the AV1 tile entropy
coder on the GPU), fed with source frames that are already resident in HBM (av1mi_gop_submit_device), producing the coded tile
payloads of a decodable AV1 stream.  That is `value`.  What the timed region leaves out of a real transcode is the PCIe upload of
the source: the end-to-end legs (`e2e_gpu_entropy`, `e2e`) add it and are reported next to `value`, never as it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 4k10-gop|1080p8-gop|4k10|1080p8] [--segments S] [--qindex Q]

One "step" = one closed GOP of every segment (S segments in lockstep x 30 frames; intra-only workloads: 3 batches of S key frames).
N ranks (one per GPU; `--gpus N` alone spawns them, or torch.distributed.run does) each code their own segments: no data-path
collective, scaling is weak.  Rank 0 prints ONE JSON line.

Default workload = the configuration BASELINE.json's metric ("encoded 4K30 frames/sec") is quoted on: configs[3], 4K 10-bit,
closed GOPs of 30 frames (1 key + 29 P), every stage on one GPU.
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
METRIC = "encoded 4K30 frames/sec (whole node) at fixed QP; PSNR-Y delta vs libaom"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs; default: WORLD_SIZE of the launcher, else 1")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="4k10-gop", choices=["1080p8", "4k10", "1080p8-gop", "4k10-gop"],
                    help="4k10-gop (default) = BASELINE configs[3], the 4K30 10-bit full pipeline the metric is quoted on: closed GOPs "
                         "of 30 frames, 1 key + 29 P; 1080p8-gop = configs[2]; 1080p8 = configs[1] (intra-only); 4k10 = 4K intra-only")
    ap.add_argument("--segments", type=int, default=0, help="closed GOPs (intra-only: key frames) coded in lockstep per batch; 0 = default")
    ap.add_argument("--qindex", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end legs (source in pinned host memory: upload included)")
    ap.add_argument("--e2e-host-segments", type=int, default=4, help="segments of the end-to-end leg with entropy coding on the host cores")
    ap.add_argument("--e2e-steps", type=int, default=2)
    ap.add_argument("--e2e-segments", type=int, default=8, help="segments in lockstep of the end-to-end leg (GPU entropy coding)")
    ap.add_argument("--key-block-size", type=int, default=32, choices=[8, 32],
                    help="32 (default, what the product's command line uses): key frames in 32x32 blocks (av1mi_gop_config.key_block_size) where "
                         "the width is a multiple of 32; 8: every frame in 8x8 blocks")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="no GPU: exercise the rank/sharding/timing/aggregation plumbing with a stand-in step (gloo tests)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------ baselines
def cpu_baseline_gop(src, W, H, bd, qindex, gop, search_range=8, p_frames=2, key_block_size=0):
    """The oracle chain (own CPU restatement, kind "port": the reference's CPU path does not exist in its tree) on this box's host
    cores, one GOP per worker thread, each coding the key frame and the first `p_frames` P frames of its GOP through the same stages
    as the GPU step (encoder loop + deblocking + CDEF + loop restoration + its decision; the P frames reference the worker's own
    loop-filtered reconstruction).  A GOP is serial in itself, so the whole-GOP rate follows from the two per-frame costs:
    gop / (t_key + (gop - 1) t_P), times the number of workers — the sample is bounded to a few frames per worker because one 4K
    P frame is ~5 s of CPU work.  src[p][segment][t]."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    import pipeline
    O.build()
    cores = min(os.cpu_count() or 1, 16)
    segs = src[0].shape[0]
    pol = [pipeline.policy_arrays(qindex, bd, ft, W, H) for ft in (0, 1)]

    hA = H // 64 * 64 if key_block_size == 32 else 0
    mi32 = None
    if hA:       # key frames in 32x32 blocks over the complete superblock rows: 32x32 / 16x16 transform edges there
        mi32 = (pol[0]["mi_y"].copy(), pol[0]["mi_c"].copy())
        mi32[0][:hA // 4] = (mi32[0][:hA // 4] & ~np.uint32(0xFF)) | np.uint32(5 | (5 << 4))
        mi32[1][:hA // 8] = (mi32[1][:hA // 8] & ~np.uint32(0xFF)) | np.uint32(4 | (4 << 4))

    def key_frame(s):
        if not hA:
            return O.intra_encode_frame(s[0], s[1], s[2], bd, 8, qindex)
        a = O.intra_encode_frame(s[0][:hA], s[1][:hA // 2], s[2][:hA // 2], bd, 32, qindex)
        if hA == H:
            return a
        b = O.intra_encode_frame(s[0][hA:], s[1][hA // 2:], s[2][hA // 2:], bd, 8, qindex)
        return {k: np.concatenate([a[k], b[k]]) for k in ("rec_y", "rec_u", "rec_v")}

    def filters(r, skip8, ft, s):
        a = pol[ft]
        if ft == 0 and mi32 is not None:
            a = dict(a, mi_y=mi32[0], mi_c=mi32[1])
        dbl = [O.deblock_plane(r["rec_y"], bd, 0, a["mi_y"]), O.deblock_plane(r["rec_u"], bd, 1, a["mi_c"]), O.deblock_plane(r["rec_v"], bd, 1, a["mi_c"])]
        cdef = O.cdef_frame(dbl[0], dbl[1], dbl[2], bd, a["cdef_damping"], a["cdef_sb"], skip8)
        lr = [O.lr_plane(cdef[0], dbl[0], bd, 0, a["lr_unit"], a["lr_units_y"]), O.lr_plane(cdef[1], dbl[1], bd, 1, a["lr_unit"], a["lr_units_c"]),
              O.lr_plane(cdef[2], dbl[2], bd, 1, a["lr_unit"], a["lr_units_c"])]
        return O.lr_select(s, cdef, lr, bd)[0]

    def one(worker):
        sg = worker % segs
        t0 = time.perf_counter()
        s = [src[p][sg, 0] for p in range(3)]
        ref = filters(key_frame(s), np.zeros((H // 8, W // 8), np.uint8), 0, s)
        t1 = time.perf_counter()
        n = min(p_frames, gop - 1)
        for t in range(1, 1 + n):
            s = [src[p][sg, t] for p in range(3)]
            r = O.inter_encode_frame(s, ref, bd, qindex, search_range)
            ref = filters(r, r["skip"].reshape(H // 8, W // 8), 1, s)
        return t1 - t0, (time.perf_counter() - t1) / max(n, 1)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(one, range(cores)))
    wall = time.perf_counter() - t0
    t_key, t_p = float(np.mean([r[0] for r in res])), float(np.mean([r[1] for r in res]))
    return {"value": cores * gop / (t_key + (gop - 1) * t_p), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d workers x (key frame + %d P frames) of the same GOPs through the oracle's encoder loops + deblock + CDEF + loop "
                      "restoration in %.1f s wall: %.2f s per key frame, %.2f s per P frame per core; value = cores x %d / (t_key + %d t_P); "
                      "entropy coding not included (the GPU value includes it)" % (cores, min(p_frames, gop - 1), wall, t_key, t_p, gop, gop - 1)}


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


def usable_cpus(cap=16, share=1):
    """host threads for this rank: what the process may use (affinity mask, cgroup quota), divided among the `share` ranks of the
    node, capped at the GPU box's per-GPU CPU share (16) unless AV1MI_HOST_THREADS says otherwise"""
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
    return max(1, min(n // max(share, 1), cap))


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


# ------------------------------------------------------------------------------------------------------------------ the legs
def frame_unit(av1stream, W, H, bd, fr, sg, gpu_entropy, threads):
    if fr.get("key_block_size") == 32:         # a key frame in 32x32 blocks: the library's general block writer (host entropy coding only)
        return av1stream.session_temporal_unit(W, H, bd, fr["raw"], sg, with_sequence_header=True, threads=threads)
    if gpu_entropy and "tile_size" in fr:      # the tiles were coded on the GPU: the host only wraps them (frame header, tile-size fields)
        return av1stream.session_frame_unit_gpu(W, H, bd, fr, sg)
    return av1stream.session_frame_unit(W, H, bd, fr, sg, threads=threads)


def e2e_leg(ctx, src, W, H, bd, qindex, gop, steps=2, warmup_frames=2, gpu_entropy=1, threads=16, compare_libaom=False, check=True, barrier=None,
            key_block_size=0):
    """END TO END: what the transcode job does per frame (reference: file in -> file out, internal/ffmpeg/transcode.go:194-203):
    source planes from host memory into the session's pinned buffers, H2D upload, block pipeline + in-loop filters on the GPU, then
    either the GPU tile coder (gpu_entropy = 1: the payloads come back, the host adds frame header and tile sizes) or D2H of the
    symbols and AV1 entropy coding on the host cores (gpu_entropy = 0, north_star's split).  Three batches in flight.  Timed with the
    wall clock between `barrier`s; the product is a decodable AV1 stream (its first frames are decoded with dav1d, when present, and
    compared with the GPU's reference).  src[p][segment][t] (host arrays)."""
    from concurrent.futures import ThreadPoolExecutor
    import av1mi
    import av1stream
    segs = src[0].shape[0]
    sess = av1mi.GopSession(ctx, W, H, bd, qindex, gop, segs, gpu_entropy=gpu_entropy, key_block_size=key_block_size)
    pool = ThreadPoolExecutor(max(1, min(threads, 3 * segs)))
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

    def code():
        t0 = time.perf_counter()
        fr = sess.collect()
        t1 = time.perf_counter()
        for sg in range(segs):
            coded["bytes"] += len(frame_unit(av1stream, W, H, bd, fr, sg, gpu_entropy, threads))
            coded["frames"] += 1
        coded["t_wait"] += t1 - t0
        coded["t_code"] += time.perf_counter() - t1

    lag = sess.max_in_flight() - 1       # batches the GPU holds while the host works on the oldest

    def run_gop(nframes):
        for t in range(nframes):
            fill(t)
            sess.submit(0 if t == 0 else 1)
            if t >= lag:
                code()
        while sess.pending():
            code()

    run_gop(min(warmup_frames, gop))
    for k in coded:
        coded[k] = 0
    ctx.sync()
    if barrier:
        barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_gop(gop)
    ctx.sync()
    if barrier:
        barrier()
    dt = time.perf_counter() - t0
    frames = coded["frames"]
    out = {"entropy_fallbacks": int(sess.entropy_fallbacks()) if gpu_entropy else None, "frames_per_s": frames / dt, "frames": frames, "seconds": dt,
           "segments_in_lockstep": segs, "gop": gop, "host_threads": threads,
           "bytes_per_frame": coded["bytes"] / frames, "mbit_per_s_at_30fps": coded["bytes"] / frames * 8 * 30 / 1e6,
           "host_seconds": {"fill_pinned_input": coded["t_fill"], "wait_for_gpu": coded["t_wait"],
                            "assemble_obu" if gpu_entropy else "entropy_code": coded["t_code"]},
           "pcie_bytes_per_frame": {"up": W * H * 3 // 2 * (1 if bd == 8 else 2),
                                    "down": coded["bytes"] / frames if gpu_entropy else W * H * 3 + (W // 8) * (H // 8) * 5},
           "entropy_coding": "GPU (k_av1_*: AV1 tile syntax, side streams)" if gpu_entropy else "host, %d threads" % threads,
           "what": ("pinned host source -> H2D -> block pipeline + deblock + CDEF + LR + AV1 tile entropy coder (GPU) -> tile payloads written into "
                    "pinned memory -> frame header + tile group assembly on the host" if gpu_entropy else
                    "pinned host source -> H2D -> block pipeline + deblock + CDEF + LR (GPU) -> D2H symbols -> AV1 entropy coding + OBU "
                    "packing on %d host threads" % threads)}
    sess.close()
    pool.shutdown()
    if check:
        out.update(decoder_check(ctx, src, W, H, bd, qindex, gop, gpu_entropy, threads, compare_libaom, key_block_size))
    if key_block_size == 32:
        out["key_block_size"] = 32
        out["what"] += "; key frames in 32x32 blocks (av1mi_gop_config.key_block_size)" + ("" if gpu_entropy else ", written by the general block writer")
    return out


def decoder_check(ctx, src, W, H, bd, qindex, gop, gpu_entropy, threads, compare_libaom, key_block_size=0):
    """the stream is real: the first two frames of segment 0 decoded by dav1d == the reference frames the GPU keeps"""
    import av1mi
    import av1stream
    out = {}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import dav1d_ref as D
        if not D.available():
            return {"decoder_check": {"decoder": None}}
        chk = av1mi.GopSession(ctx, W, H, bd, qindex, gop, 1, gpu_entropy=gpu_entropy, key_block_size=key_block_size)
        units, refs = [], []
        for t in range(min(2, src[0].shape[1])):
            planes = chk.input_planes()
            for p in range(3):
                np.copyto(planes[p], src[p][0, t])
            chk.submit(0 if (t == 0 or gop == 1) else 1)
            units.append(frame_unit(av1stream, W, H, bd, chk.collect(), 0, gpu_entropy, threads))
            refs.append(chk.download_reference())
        chk.close()
        dec = D.decode(b"".join(units))
        ok = len(dec) == len(units) and all((dec[t][p] == refs[t][p]).all() for t in range(len(units)) for p in range(3))
        out["decoder_check"] = {"decoder": "dav1d " + D.version(), "frames": len(units), "bit_exact_vs_gpu_reference": bool(ok)}
        if compare_libaom:
            try:
                if bd == 8:
                    out["vs_libaom"] = libaom_leg([src[p][0, 0] for p in range(3)], bd, qindex, len(units[0]), refs[0][0], threads)
                else:
                    # the image's libaom is an 8-bit build (it refuses 10-bit planes): compare on the 8-bit rendition of the same frame
                    s8 = [np.clip((src[p][0, 0].astype(np.int32) + 2) >> 2, 0, 255).astype(np.uint8) for p in range(3)]
                    k8 = av1mi.GopSession(ctx, W, H, 8, qindex, 1, 1, gpu_entropy=gpu_entropy, key_block_size=key_block_size)
                    for dst, a in zip(k8.input_planes(), s8):
                        np.copyto(dst, a)
                    k8.submit(0)
                    tu8 = frame_unit(av1stream, W, H, 8, k8.collect(), 0, gpu_entropy, threads)
                    rec8 = k8.download_reference()[0]
                    k8.close()
                    out["vs_libaom"] = libaom_leg(s8, 8, qindex, len(tu8), rec8, threads)
                    out["vs_libaom"]["note"] = "8-bit rendition of the workload's first frame: the bundled libaom is built without high bit depth"
            except Exception as e:
                out["vs_libaom"] = {"error": repr(e)[:200]}
    except Exception as e:       # the check is a courtesy of the bench, the tests are the gate
        out["decoder_check"] = {"error": repr(e)[:200]}
    return out


PMC_KERNEL = {"intra_pipeline": "k_intra_pipe", "deblock": "k_deblock", "cdef": "k_cdef", "loop_restoration": "k_lr", "inter_pipeline": "k_inter_pipe",
              "me_integer": "k_me_int", "entropy_tokens": "k_av1_tokens", "entropy_chains": "k_av1_chains", "entropy_code": "k_av1_code"}


def pmc_numbers(kind, workload, frames_per_launch):
    """HBM bytes per launch of the kernel behind `kind`, and its VALU issue share, from the COMMITTED rocprofv3 PMC passes of this
    command (profiles/pmc_<workload>_latest.json, made by tools/prof_pmc.sh + tools/pmc_to_json.py): reads = 2 x FETCH_SIZE (gfx950
    tallies a 128-byte read request as 64 bytes; calibrated on this pool with kernels of known traffic, "calibration" in the file) +
    WRITE_SIZE, scaled to this run's frames per launch.  PMC counters cannot be read live next to the timing, so this is a
    builder-side constant of the profiled box, not a measurement of this run: `traffic_source` says so.  None when no matching
    profile is committed."""
    tag = {"1080p8": "1080p8_intra", "4k10-gop": "4k10_gop", "1080p8-gop": "1080p8_gop", "4k10": "4k10_intra"}.get(workload)
    path = os.path.join(ROOT, "profiles", "pmc_%s_latest.json" % tag)
    try:
        prof = json.load(open(path))
    except (OSError, ValueError):
        return None, None, None
    key = PMC_KERNEL.get(kind, "?")
    # the kernel of that name (not k_av1_tokens32 for k_av1_tokens); of a template's instantiations the one that moves the most bytes
    hits = sorted(((n, d) for n, d in prof["kernels"].items() if n == key or n.startswith(key + "<")), key=lambda nd: -(nd[1]["fetch_bytes_uncorrected"] + nd[1]["write_bytes"]))
    for name, d in hits[:1]:
        if True:
            corr = (prof.get("calibration") or {}).get("read_correction_8B_per_lane") or 2.0
            traffic = (d["fetch_bytes_uncorrected"] * corr + d["write_bytes"]) * frames_per_launch / prof["frames_per_step"]
            sq = d.get("sq_counters_per_launch") or {}
            # SQ_INSTS_VALU counts wave instructions over the whole chip; a SIMD issues one VALU instruction per 4 cycles at most.
            # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (k_inter_pipe: 14.6 M "cycles" for a 0.83 ms kernel): / 8 = kernel cycles
            valu = None
            if sq.get("SQ_INSTS_VALU") and sq.get("GRBM_GUI_ACTIVE"):
                valu = sq["SQ_INSTS_VALU"] * 4.0 / (1024.0 * sq["GRBM_GUI_ACTIVE"] / 8.0)
            return traffic, valu, os.path.relpath(path, ROOT)
    return None, None, None


def segment_of_rank(rank, frames_per_step):
    """closed-GOP sharding: rank r codes the r-th run of frames_per_step frames (no data exchanged between ranks)"""
    return rank * frames_per_step


def aggregate(dist, dt, device, op="max"):
    """max (or sum) over ranks: the job is as slow as its slowest GPU"""
    if dist is None:
        return dt
    import torch
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def dry_run(args, rank, world, dist):
    """CPU stand-in used by tests/test_distributed.py: same control flow as the GPU path (timed step between barriers, max over
    ranks; then the end-to-end leg on EVERY rank between barriers, frames summed over ranks, host threads shared), sleeps as work."""
    frames = args.segments or 4
    first = segment_of_rank(rank, frames)

    def barrier():
        if dist is not None:
            dist.barrier()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))        # uneven ranks: the aggregate must follow the slowest
    barrier()
    dt = aggregate(dist, time.perf_counter() - t0, "cpu")
    threads = usable_cpus(share=world)
    barrier()
    t0 = time.perf_counter()
    time.sleep(0.02 * (1 + rank))
    barrier()
    e2e_dt = aggregate(dist, time.perf_counter() - t0, "cpu")
    e2e_frames = aggregate(dist, float(frames), "cpu", "sum")
    if rank == 0:
        print(json.dumps({"metric": "dry-run", "value": frames * args.steps * world / dt, "unit": "frames/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "scaling": "weak",
                          "first_frame_rank0": first, "frames_per_step": frames,
                          "e2e_gpu_entropy_frames_per_s": e2e_frames / e2e_dt, "e2e_gpu_entropy": {"ranks": world, "host_threads_per_rank": threads}}))
    if dist is not None:
        dist.destroy_process_group()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU) as CHILD processes — before
    this process has imported torch or made any HIP call — with the environment torch.distributed.run would give them, relay
    rank 0's JSON line and exit with the first non-zero status.  A rank that dies takes the others down (they would wait in a
    barrier until the collective's timeout otherwise)."""
    import socket
    import subprocess
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("AV1MI_BENCH_TIMEOUT", "3000"))
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes) or time.time() > deadline:
            for i, p in enumerate(procs):      # the exact children started above, nothing else
                if codes[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=5)
    sys.stdout.write((out[0] if out else b"").decode())
    sys.stdout.flush()
    if any(codes):
        sys.exit("bench.py: rank(s) failed with exit codes %s" % codes)


# ------------------------------------------------------------------------------------------------------------------ main
def main():
    args = parse()
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is None:
        args.gpus = env_world if "RANK" in os.environ else 1      # an unspecified --gpus takes the launcher's world size
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    world = env_world if "RANK" in os.environ else 1
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d does not match WORLD_SIZE %d of the launcher" % (args.gpus, world))
    dist = None
    sync_t = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch
        import torch.distributed as dist
        use_cuda = torch.cuda.is_available() and not args.dry_run_cpu
        if use_cuda:
            torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl" if use_cuda else "gloo")   # nccl == RCCL on ROCm; only barrier + max / sum reductions use it
        sync_t = torch.zeros(1, device="cuda" if use_cuda else "cpu")

    if args.dry_run_cpu:
        return dry_run(args, rank, world, dist)

    import av1mi
    import synth

    gop_wl = args.workload.endswith("-gop")
    if args.workload.startswith("1080p8"):
        W, H, bd = 1920, 1080, 8
        segs = args.segments or (24 if gop_wl else 16)
    else:
        W, H, bd = 3840, 2160, 10
        segs = args.segments or (12 if gop_wl else 8)
    gop = 30 if gop_wl else 1
    batches = gop if gop_wl else 3                    # per step
    frames = segs * batches
    ctx = av1mi.Context(local_rank)
    first = segment_of_rank(rank, frames)
    t_gen = time.perf_counter()
    Y, U, V = synth.frames(W, H, frames, bd, first)
    # segment s holds frames first + s * batches ..: src[p][segment][t]; one batch = the t-th frames of all segments, stacked
    src = [a.reshape(segs, batches, *a.shape[1:]) for a in (Y, U, V)]
    d_src = [[ctx.to_device(np.ascontiguousarray(src[p][:, t])) for p in range(3)] for t in range(batches)]
    t_gen = time.perf_counter() - t_gen
    kbs = 32 if (args.key_block_size == 32 and W % 32 == 0) else 0
    sess = av1mi.GopSession(ctx, W, H, bd, args.qindex, gop, segs, gpu_entropy=1, key_block_size=kbs)
    stat = {"payload": 0, "frames": 0}

    def collect():
        fr = sess.collect()
        stat["frames"] += segs
        if "tile_size" in fr:
            stat["payload"] += int(fr["tile_payload"].size)

    lag = sess.max_in_flight() - 1

    def step():
        for t in range(batches):
            sess.submit_device(d_src[t][0], d_src[t][1], d_src[t][2], 0 if (t == 0 or not gop_wl) else 1)
            if sess.pending() > lag:
                collect()

    def drain():
        while sess.pending():
            collect()

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.all_reduce(sync_t)
            if sync_t.is_cuda:
                import torch
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    stat["payload"] = stat["frames"] = 0
    # every launch of the timed region is bracketed by its own HIP event pair on the stream it is launched on
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    barrier()
    dt = time.perf_counter() - t0
    ctx.prof_enable(False)
    prof = ctx.prof_get()
    dt = aggregate(dist, dt, sync_t.device if dist is not None else None)
    fallbacks = int(sess.entropy_fallbacks())
    words = av1mi.C.c_uint64()
    ctx.lib.av1mi_av1_entropy_last_list_words(ctx.h, av1mi.C.byref(words))      # of the last batch (a P-frame batch in GOP workloads)
    total_frames = frames * args.steps * world
    fps = total_frames / dt
    b_ = 1 if bd == 8 else 2
    fs = W * H * 3 // 2 * segs                       # samples per launch (one batch)
    payload_per_batch = stat["payload"] / max(args.steps * batches, 1)
    out = {
        "metric": METRIC, "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8" if bd == 8 else "u16", "data": "synthetic",
        "config": {"workload": "%dx%d %d-bit 4:2:0, %s, q index %d: block pipeline (intra: 13 modes; inter: +-8 full search + half / quarter-sample "
                               "refinement, 8x8 blocks) + deblocking + CDEF + Wiener restoration with its on/off decision + the AV1 tile entropy coder, "
                               "all on the GPU (the product's GOP session, libav1mi.so av1mi_gop_*)"
                               % (W, H, bd, ("%d closed GOPs of %d frames in lockstep (1 key + %d P frames, single reference)" % (segs, gop, gop - 1)) if gop_wl
                                  else "intra-only (all key frames), %d frames per batch" % segs, args.qindex),
                   "value_is": "ENCODED frames/s with the source frames already resident in HBM (av1mi_gop_submit_device) and the coded tile payloads of a "
                               "decodable AV1 stream written to pinned host memory: the GOP session the product runs, minus the PCIe upload of the source.  "
                               "`e2e_gpu_entropy_frames_per_s` adds the upload (pinned host source); `e2e_frames_per_s` is north_star's split with entropy "
                               "coding on the host cores; `block_pipeline_frames_per_s` is the round-1/2 definition of `value` (kernels of the main "
                               "stream only, no entropy coding), derived from this run's kernel times",
                   "frames_per_step": frames, "segments": segs, "qindex": args.qindex, "key_block_size": kbs or 8, "entropy_fallbacks": fallbacks,
                   "coded_bytes_per_frame": stat["payload"] / max(stat["frames"], 1),
                   "sharding": "closed-GOP segment per GPU, no collective", "device": ctx.device_name,
                   "source_generation_s": t_gen},
    }
    if rank == 0:
        nbatch = args.steps * batches
        # algorithmic bytes per BATCH (one launch, or the launches of one batch together), SURVEY.md §8d / DESIGN.md §3
        alg = {"intra_pipeline": (2 * b_ + 2) * fs, "inter_pipeline": (3 * b_ + 2) * fs, "me_integer": 2 * b_ * fs * 2.0 / 3.0,
               "deblock": 2 * b_ * fs, "cdef": 2 * b_ * fs, "loop_restoration": 2 * b_ * fs,
               # the coder: int16 levels in, list words (4 bytes each, written by the tokenizer, completed by the chains, read by the coder)
               "entropy_tokens": 2 * fs + 8 * words.value, "entropy_chains": 8 * words.value, "entropy_code": 4 * words.value + payload_per_batch,
               "entropy_pack": 2 * payload_per_batch}
        per_kind_batches = {"intra_pipeline": args.steps * (1 if gop_wl else batches), "inter_pipeline": args.steps * (batches - 1) if gop_wl else 0,
                            "me_integer": args.steps * (batches - 1) if gop_wl else 0}
        kernels = {}
        for k, (n, ms) in prof.items():
            nb = per_kind_batches.get(k, nbatch) or nbatch
            kernels[k] = {"launches": n, "total_ms": ms, "ms_per_batch": ms / nb, "launches_per_batch": n / nb,
                          "algorithmic_GBps": alg[k] / (ms / nb * 1e-3) / 1e9 if k in alg else None}
        out["kernels"] = kernels
        main_ms = sum(ms for k, (n, ms) in prof.items() if not k.startswith("entropy"))
        ent_ms = sum(ms for k, (n, ms) in prof.items() if k.startswith("entropy"))
        out["block_pipeline_frames_per_s"] = frames * args.steps / (main_ms * 1e-3) if main_ms else None
        out["entropy_coder_frames_per_s"] = frames * args.steps / (ent_ms * 1e-3) if ent_ms else None
        # roofline of the dominant kernel: the kind with the largest summed device time
        dom = max(prof, key=lambda k: prof[k][1])
        n, ms = prof[dom]
        nb = per_kind_batches.get(dom, nbatch) or nbatch
        per_launch_ms, per_launch_bytes = ms / n, alg[dom] * nb / n
        ach = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
        traffic, valu, tsrc = pmc_numbers(dom, args.workload, segs)
        out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                           "traffic": traffic,
                           "traffic_source": (tsrc + " (committed PMC passes of this command on the profiled box; not measured in this run)") if tsrc else None,
                           "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_ms": per_launch_ms, "launches": n,
                           "valu_issue_frac": valu,
                           "limiter": ("VALU issue" if (valu or 0) > 0.6 else "latency / dependent chains") +
                                      ": the HBM fraction says how far from the memory roofline the kernel runs, not what binds it"}
        copy_gbps = copy_bandwidth(ctx)
        out["roofline"]["copy_GBps_measured"] = copy_gbps
        out["roofline"]["frac_of_measured_copy"] = ach / copy_gbps
        out["roofline"]["note"] = ("measured in the timed region, where the tile coder's kernels run on their own streams BESIDE the block pipeline: a "
                                   "kernel's event span includes the time it shares the chip with them (the spans of all kinds add up to more than the "
                                   "wall time).  `kernels_isolated` / `roofline_isolated` are the same kernels with the coder serialised on the main "
                                   "stream (av1mi_gop_config.coder_streams = 2): every kernel alone on the GPU")
        # the same batches once more with every kernel ALONE on the GPU: the numbers to judge a kernel by
        iso = av1mi.GopSession(ctx, W, H, bd, args.qindex, gop, segs, gpu_entropy=1, coder_streams=2, key_block_size=kbs)
        for rep in range(2):
            if rep == 1:
                ctx.prof_reset()
                ctx.prof_enable(True)
            for t in range(batches):
                iso.submit_device(d_src[t][0], d_src[t][1], d_src[t][2], 0 if (t == 0 or not gop_wl) else 1)
                if iso.pending() > lag:
                    iso.collect()
            while iso.pending():
                iso.collect()
            ctx.sync()
        ctx.prof_enable(False)
        iprof = ctx.prof_get()
        iso.close()
        ikb = {"intra_pipeline": 1 if gop_wl else batches, "inter_pipeline": batches - 1, "me_integer": batches - 1}
        out["kernels_isolated"] = {k: {"launches": n, "ms_per_batch": ms / (ikb.get(k, batches) or batches),
                                       "algorithmic_GBps": alg[k] / (ms / (ikb.get(k, batches) or batches) * 1e-3) / 1e9 if k in alg else None}
                                   for k, (n, ms) in iprof.items()}
        idom = max(iprof, key=lambda k: iprof[k][1])
        n_i, ms_i = iprof[idom]
        nb_i = ikb.get(idom, batches) or batches
        ach_i = alg[idom] * nb_i / n_i / (ms_i / n_i * 1e-3) / 1e9
        tr_i, valu_i, tsrc_i = pmc_numbers(idom, args.workload, segs)
        out["roofline_isolated"] = {"kernel": idom, "bound": "hbm", "achieved": ach_i, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach_i / HBM_PEAK_GBPS,
                                    "traffic": tr_i, "traffic_source": tsrc_i, "algorithmic_bytes_per_launch": alg[idom] * nb_i / n_i,
                                    "avg_launch_ms": ms_i / n_i, "launches": n_i, "valu_issue_frac": valu_i,
                                    "serialised_ms_per_batch": sum(ms / (ikb.get(k, batches) or batches) for k, (n, ms) in iprof.items() if k != "intra_pipeline") ,
                                    "what": "one step with the coder serialised on the main stream: the dominant kernel by summed device time and its roofline"}
        # whole pipeline: (9b + 2) S per inter frame, (8b + 2) S per intra-only frame, times the job's frame rate
        per_frame = ((9 * b_ + 2) if gop_wl else (8 * b_ + 2)) * (W * H * 3 // 2)
        out["pipeline_roofline"] = {"algorithmic_bytes_per_frame": per_frame, "achieved": per_frame * fps / world / 1e9, "peak": HBM_PEAK_GBPS,
                                    "unit": "GB/s", "frac": per_frame * fps / world / 1e9 / HBM_PEAK_GBPS,
                                    "note": "per GPU: whole-job frames/s / n_gpus x the fused-pipeline bytes of BASELINE.md §3"}
        # PSNR-Y of what a decoder outputs for the last batch against its source
        rec = sess.download_reference()[0]
        last = np.ascontiguousarray(src[0][:, batches - 1]).reshape(rec.shape)
        mse = float(np.mean((rec.astype(np.float64) - last.astype(np.float64)) ** 2))
        out["quality"] = {"psnr_y_db": 10.0 * np.log10(((1 << bd) - 1) ** 2 / mse) if mse > 0 else None, "frames": segs,
                          "note": "last frames of the step; fixed qindex, no rate control; the comparison with libaom on the same key frame is under e2e.vs_libaom"}
        out["baseline_tools"] = probe_baseline_tools()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_gop(src, W, H, bd, args.qindex, gop, key_block_size=kbs)
        else:
            out["cpu_baseline"] = None
    sess.close()
    for t in d_src:
        for b in t:
            b.free()
    if not args.no_e2e:
        # END TO END on every rank, between barriers: the scaling risks SURVEY §8e names (host cores, PCIe) are in this leg
        threads = usable_cpus(share=world)
        es = min(args.e2e_segments, segs)
        esrc = [src[p][:es] for p in range(3)]
        g = e2e_leg(ctx, esrc, W, H, bd, args.qindex, gop, steps=args.e2e_steps if gop_wl else 8, warmup_frames=2 if gop_wl else 1, gpu_entropy=1,
                    threads=threads, check=rank == 0, barrier=barrier, key_block_size=kbs)
        e_dt = aggregate(dist, g["seconds"], sync_t.device if dist is not None else None)
        e_frames = aggregate(dist, float(g["frames"]), sync_t.device if dist is not None else None, "sum")
        if rank == 0:
            g["ranks"], g["host_threads_per_rank"] = world, threads
            g["frames_per_s_rank0"] = g["frames_per_s"]
            g["frames_per_s"] = e_frames / e_dt
            out["e2e_gpu_entropy"] = g
            out["e2e_gpu_entropy_frames_per_s"] = g["frames_per_s"]
            if world == 1 and gop_wl:
                hs = min(args.e2e_host_segments, segs)
                e = e2e_leg(ctx, [src[p][:hs] for p in range(3)], W, H, bd, args.qindex, gop, steps=1, gpu_entropy=0, threads=threads, compare_libaom=True,
                            key_block_size=kbs)
                out["e2e"] = e                      # north_star's split: entropy coding on the host cores
                out["e2e_frames_per_s"] = e["frames_per_s"]

    if rank == 0:
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
