"""Host-side mirror of the reference's transcode-job API (C++, av1-go_amd/host) pinned against values read off the
reference source: internal/ffmpeg/transcode.go:17-165, internal/daemon/daemon.go:18-21."""
import ctypes as C
import os

import numpy as np
import pytest

HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "av1-go_amd", "host", "libav1mi_host.so")


@pytest.fixture(scope="module")
def host():
    if not os.path.exists(HOST):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(HOST)])
    lib = C.CDLL(HOST)
    lib.av1mi_host_check_size_gate.argtypes = [C.c_longlong, C.c_longlong, C.c_double]
    lib.av1mi_host_process_job.argtypes = [C.c_char_p, C.c_longlong, C.c_double, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int]
    return lib


def _args(host, inp, out, has_video=1, index=0, height=1080, webrip=0):
    buf = C.create_string_buffer(8192)
    n = host.av1mi_host_transcode_args(inp.encode(), out.encode(), has_video, index, height, webrip, buf, 8192)
    return n, buf.value.decode().split("\n")


def test_determine_quality_thresholds(host):   # transcode.go:157-165
    for h, q in ((0, 25), (720, 25), (1079, 25), (1080, 24), (1439, 24), (1440, 23), (2160, 23)):
        assert host.av1mi_host_determine_quality(h) == q


def test_check_size_gate_edges(host):          # daemon.go:18-21
    assert host.av1mi_host_check_size_gate(1000, 900, 0.90) == 1
    assert host.av1mi_host_check_size_gate(1000, 901, 0.90) == 0
    assert host.av1mi_host_check_size_gate(0, 0, 0.90) == 1 and host.av1mi_host_check_size_gate(0, 1, 0.90) == 0


def test_transcode_args_match_the_reference_argv(host):
    n, a = _args(host, "/m/in.mkv", "/m/in.av1-tmp.mkv", index=2, height=1080, webrip=0)
    expect = ["-hide_banner", "-analyzeduration", "50M", "-probesize", "50M", "-init_hw_device", "vaapi=va", "-hwaccel", "vaapi",
              "-hwaccel_output_format", "vaapi", "-filter_hw_device", "va", "-i", "/m/in.mkv", "-map", "0", "-map", "-0:v", "-map", "-0:t",
              "-map", "0:v:2", "-map", "0:a?", "-map", "-0:a:m:language:rus", "-map", "-0:a:m:language:ru", "-map", "0:s?",
              "-map", "-0:s:m:language:rus", "-map", "-0:s:m:language:ru", "-map_chapters", "0", "-vf:v:0",
              "scale_vaapi=w=ceil(iw/2)*2:h=ceil(ih/2)*2,hwdownload,format=nv12,setsar=1,format=nv12,hwupload",
              "-c:v:0", "av1_vaapi", "-global_quality:v:0", "24", "-compression_level", "2", "-c:a", "copy", "-c:s", "copy",
              "-max_muxing_queue_size", "2048", "-map_metadata", "0", "-f", "matroska", "-movflags", "+faststart", "/m/in.av1-tmp.mkv"]
    assert n == len(expect) and a == expect
    n, w = _args(host, "a.mkv", "b.mkv", height=2160, webrip=1)
    assert w[13:17] == ["-fflags", "+genpts", "-copyts", "-start_at_zero"] and w[17:19] == ["-i", "a.mkv"]
    assert w[w.index("-vf:v:0") + 1].startswith("scale_vaapi=w='if(gt(iw,iw*sar),iw,iw*sar)':h='if(gt(iw,iw*sar),iw/sar,ih)',scale_vaapi=w=ceil")
    assert w[w.index("-global_quality:v:0") + 1] == "23"
    i = w.index("-compression_level")
    assert w[i + 2:i + 6] == ["-vsync", "0", "-avoid_negative_ts", "make_zero"] and w[-1] == "b.mkv" and n == len(w) == len(expect) + 8


def test_transcode_args_without_video_stream(host):   # transcode.go:18-20
    n, a = _args(host, "a", "b", has_video=0)
    assert n == -1 and a == ["no video stream found in probe result"]


def _write_y4m(path, w, h, n, bd=8):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HOST), ".."))
    import synth
    Y, U, V = synth.frames(w, h, n, bd)
    with open(path, "wb") as f:
        f.write(("YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C%s\n" % (w, h, "420jpeg" if bd == 8 else "420p10")).encode())
        for i in range(n):
            f.write(b"FRAME\n")
            for p in (Y[i], U[i], V[i]):
                f.write(p.astype("<u2" if bd == 10 else np.uint8).tobytes())


def _varint(data, pos):
    v = sh = 0
    while True:
        b = data[pos]
        pos += 1
        v |= (b & 127) << sh
        sh += 7
        if b < 128:
            return v, pos


def _check_first_segment_against_oracle(host, data, w, h, q, nframes):
    """the coded file, decoded by the host entropy decoder, carries exactly the symbols the oracle's encoder loop produces"""
    from oracle import oracle
    import synth
    oracle.build()
    P = C.c_void_p
    host.av1mi_host_entropy_decode.argtypes = [P, C.c_longlong, C.c_int, C.c_int, C.c_int, P, P, P, P, P, P, P]
    Y, U, V = synth.frames(w, h, nframes, 8)       # same call as _write_y4m: the texture depends on the clip length
    ok = oracle.intra_encode_frame(Y[0], U[0], V[0], 8, 8, q)
    nb = (w // 8) * (h // 8)
    pos = data.index(b"\n", data.index(b"SEG 3 ")) + 1
    vp = lambda a: a.ctypes.data_as(P)
    for t, key in ((0, 1), (1, 0)):
        assert data[pos:pos + 1] == (b"K" if key else b"P")
        n, pos = _varint(data, pos + 1)
        payload = np.frombuffer(data[pos:pos + n], np.uint8).copy()
        pos += n
        ly, lu, lv = np.zeros((nb, 8, 8), np.int16), np.zeros((nb, 4, 4), np.int16), np.zeros((nb, 4, 4), np.int16)
        my, muv, mvs, skip = np.zeros(nb, np.uint8), np.zeros(nb, np.uint8), np.zeros((nb, 2), np.int16), np.zeros(nb, np.uint8)
        assert host.av1mi_host_entropy_decode(vp(payload), n, w, h, key, vp(ly), vp(lu), vp(lv), vp(my), vp(muv), vp(mvs), vp(skip)) == 0
        if key:
            assert np.array_equal(ly, ok["lev_y"]) and np.array_equal(lu, ok["lev_u"]) and np.array_equal(lv, ok["lev_v"])
            assert np.array_equal(my, ok["modes_y"]) and np.array_equal(muv, ok["modes_uv"])
        else:
            assert skip.max() <= 1 and np.abs(mvs).max() <= 8 * 8 + 7     # +-8 integer search + sub-pel, 1/8 units
            assert (ly[skip == 1] == 0).all()


def test_run_transcode_fails_cleanly_without_gpu_or_input(host, tmp_path, av1mi):
    buf = C.create_string_buffer(1024)
    args = "\n".join(["-i", str(tmp_path / "missing.y4m"), "-global_quality:v:0", "25", str(tmp_path / "o.mkv")])
    rc = host.av1mi_host_run_transcode(args.encode(), buf, 1024)
    if av1mi.load().av1mi_device_count() == 0:
        assert rc == -1 and b"no usable HIP device" in buf.value      # "could not run": transcode.go:311
    else:
        assert rc == 1 and b"No such file or directory" in buf.value
    assert not (tmp_path / "o.mkv").exists()
    rc = host.av1mi_host_run_transcode(b"-x", buf, 1024)
    assert rc == 1 and b"Invalid argument" in buf.value


@pytest.mark.gpu
def test_run_transcode_and_process_job_on_gpu(host, tmp_path):
    src = tmp_path / "clip.y4m"
    _write_y4m(str(src), 192, 128, 7)
    out = tmp_path / "clip.av1-tmp.mkv"
    buf = C.create_string_buffer(1024)
    args = "\n".join(["-hide_banner", "-i", str(src), "-global_quality:v:0", "120", "-g", "3", str(out)])
    assert host.av1mi_host_run_transcode(args.encode(), buf, 1024) == 0 and buf.value == b""
    data = out.read_bytes()
    assert data.startswith(b"AV1MI2 W192 H128 B8 F30:1 Q120 G3\n") and data.count(b"SEG 3 ") == 2 and data.count(b"SEG 1 ") == 1
    # every segment starts with a key frame ('K' right after its header line), later frames are P frames
    first = data.index(b"SEG 3 ")
    assert data[data.index(b"\n", first) + 1:data.index(b"\n", first) + 2] == b"K"
    assert len(data) < src.stat().st_size            # coarse quantiser: packed levels are smaller than the raw input
    _check_first_segment_against_oracle(host, data, 192, 128, 120, 7)
    out.unlink()
    # lifecycle: generous ratio -> the source is replaced by the coded file; tight ratio -> skipped with markers
    status, reason = C.create_string_buffer(256), C.create_string_buffer(1024)
    orig = src.stat().st_size
    src2 = tmp_path / "other.y4m"
    src2.write_bytes(src.read_bytes())
    assert host.av1mi_host_process_job(str(src2).encode(), orig, 1e-4, str(tmp_path).encode(), 0, status, reason, 256) == 0
    assert status.value == b"skipped" and reason.value.startswith(b"size gate: new ") and (tmp_path / "other.av1qsvd-skip").exists()
    assert (tmp_path / "other.av1qsvd-why.txt").exists() and not (tmp_path / "other.av1-tmp.mkv").exists() and src2.stat().st_size == orig
    assert host.av1mi_host_process_job(str(src).encode(), orig, 5.0, str(tmp_path).encode(), 0, status, reason, 256) == 0
    assert status.value == b"success" and src.read_bytes().startswith(b"AV1MI2 ") and (tmp_path / "test.json").exists()


def _pool(host, paths, workers, ngpus, ratio, state):
    host.av1mi_host_job_pool.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_char_p, C.c_char_p, C.c_int]
    buf = C.create_string_buffer(8192)
    n = host.av1mi_host_job_pool("\n".join(paths).encode(), workers, ngpus, ratio, str(state).encode(), buf, 8192)
    return n, buf.value.decode().split("\n")


def test_job_pool_runs_every_job_even_when_the_backend_cannot_run(host, tmp_path, av1mi):
    """the worker pool that replaces the serial loop of cmd/av1d/main.go:291-349: every job is visited exactly once and
    its record written, whatever happens to it (here: no GPU / missing inputs)"""
    paths = [str(tmp_path / ("clip%d.y4m" % i)) for i in range(5)]
    n, status = _pool(host, paths, 3, 2, 0.9, tmp_path)
    assert n == 0 and len(status) == 5
    # missing sources: the reference's stability check fails first (daemon.go:59-62) -> error returned, job left pending
    assert all(s.startswith("pending: failed to check file stability") for s in status)


@pytest.mark.gpu
def test_job_pool_concurrent_contexts_are_deterministic(host, tmp_path):
    """BASELINE config 5 in miniature: 4 jobs on 3 workers (all on GPU 0 here, one context + stream each).  Concurrent
    contexts must not disturb each other: every output equals the one a lone job produces."""
    clips = []
    for i in range(4):
        p = tmp_path / ("c%d.y4m" % i)
        _write_y4m(str(p), 192, 128, 4)
        clips.append(p)
    solo = tmp_path / "solo.y4m"
    solo.write_bytes(clips[0].read_bytes())
    n, status = _pool(host, [str(solo)], 1, 1, 5.0, tmp_path)
    assert n == 1 and status == ["success"]
    ref = solo.read_bytes()
    assert ref.startswith(b"AV1MI2 ")
    n, status = _pool(host, [str(c) for c in clips], 3, 1, 5.0, tmp_path)
    assert n == 4 and status == ["success"] * 4
    for i, c in enumerate(clips):
        assert c.read_bytes() == ref, "job %d differs from the solo run" % i
        assert (tmp_path / ("pool%d.json" % i)).exists()
