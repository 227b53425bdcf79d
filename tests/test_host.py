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
    lib.av1mi_host_process_job.argtypes = [C.c_char_p, C.c_longlong, C.c_double, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_int]
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


def _obus(data):
    """split a Section-5 stream into (type, payload) pairs (obu_has_size_field = 1, no extension)"""
    out, pos = [], 0
    while pos < len(data):
        hdr = data[pos]
        assert hdr & 0x86 == 0x02, "unexpected OBU header byte %#x" % hdr
        size, sh, pos = 0, 0, pos + 1
        while True:
            b = data[pos]
            pos += 1
            size |= (b & 127) << sh
            sh += 7
            if b < 128:
                break
        out.append(((hdr >> 3) & 15, data[pos:pos + size]))
        pos += size
    return out


def _ebml_blocks(data):
    """SimpleBlock payloads (frames) of a Matroska file written by host/mux.cpp, as (is_key, bytes); tiny EBML walker"""
    def vint(pos, keep_marker):
        first = data[pos]
        n = 1
        while not first & (0x80 >> (n - 1)):
            n += 1
        v = first if keep_marker else first & (0xFF >> n)
        for k in range(1, n):
            v = (v << 8) | data[pos + k]
        return v, pos + n
    frames, codec_private = [], None

    def walk(pos, end):
        nonlocal codec_private
        while pos < end:
            eid, pos = vint(pos, True)
            size, pos = vint(pos, False)
            if eid in (0x18538067, 0x1F43B675, 0x1654AE6B, 0xAE):      # Segment, Cluster, Tracks, TrackEntry: descend
                walk(pos, pos + size)
            elif eid == 0xA3:
                assert data[pos] == 0x81
                frames.append((bool(data[pos + 3] & 0x80), data[pos + 4:pos + size]))
            elif eid == 0x63A2:
                codec_private = data[pos:pos + size]
            pos += size
    walk(0, len(data))
    return frames, codec_private


def _scan(host, path, group):
    host.av1mi_host_y4m_scan.restype = C.c_longlong
    host.av1mi_host_y4m_scan.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]
    s, seek, geo, err = C.c_ulonglong(), C.c_int(), (C.c_int * 5)(), C.create_string_buffer(512)
    n = host.av1mi_host_y4m_scan(str(path).encode(), group, C.byref(s), C.byref(seek), geo, err, 512)
    return n, s.value, seek.value, list(geo), err.value.decode()


def test_y4m_source_file_stream_and_header_variants(host, tmp_path):
    """the backend's input (host/y4m.cpp): a seekable file is read in place, a FIFO sequentially one group ahead — same frames; a
    header longer than any fixed buffer and FRAME lines with parameters are legal Y4M; a truncated frame is an error"""
    import threading
    w, h, n, bd = 70, 38, 7, 10
    plain = tmp_path / "plain.y4m"
    _write_y4m(str(plain), w, h, n, bd)
    n0, sum0, seek0, geo, _ = _scan(host, plain, 4)
    assert (n0, seek0, geo) == (n, 1, [w, h, bd, 30, 1])
    assert _scan(host, plain, 3)[:2] == (n, sum0) and _scan(host, plain, 64)[:2] == (n, sum0)      # the group size does not matter
    # the same frames with a 3000-byte header comment and per-frame parameters: not seekable in place, read sequentially
    raw = plain.read_bytes()
    hdr_end = raw.index(b"\n") + 1
    fb = (len(raw) - hdr_end) // n
    fancy = tmp_path / "fancy.y4m"
    with open(fancy, "wb") as f:
        f.write(raw[:hdr_end - 1] + b" X" + b"c" * 3000 + b" XYSCSS=420P10\n")
        for i in range(n):
            f.write(b"FRAME Ip\n" + raw[hdr_end + i * fb + 6:hdr_end + (i + 1) * fb])
    assert _scan(host, fancy, 4)[:3] == (n, sum0, 0)
    # a FIFO fed by another thread
    fifo = tmp_path / "in.fifo"
    os.mkfifo(fifo)
    t = threading.Thread(target=lambda: open(fifo, "wb").write(raw))
    t.start()
    assert _scan(host, fifo, 2)[:3] == (n, sum0, 0)
    t.join()
    # errors
    cut = tmp_path / "cut.y4m"
    cut.write_bytes(raw[:-100])
    nn, _, _, _, err = _scan(host, cut, 4)
    assert nn == -1 and "truncated" in err
    bad = tmp_path / "bad.y4m"
    bad.write_bytes(b"YUV4MPEG2 W64 H64 F30:1 C444\nFRAME\n")
    assert _scan(host, bad, 4)[0] == -1 and "colourspace" in _scan(host, bad, 4)[4]
    assert "No such file" in _scan(host, tmp_path / "nope.y4m", 4)[4]


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
def test_run_transcode_and_process_job_on_gpu(host, tmp_path, O):
    """RunTranscode end to end: Y4M in, AV1 out — the file decodes in dav1d, its frames equal what the oracle's closed-GOP
    chain reconstructs (key AND P frames, the policy included: the levels come from av1mi_policy_frame_params), in all three
    containers; then the ProcessJob lifecycle around it."""
    import av1mi
    import dav1d_ref as D
    sys_path_synth()
    import synth
    w, h, n, q, gop = 192, 128, 7, 120, 3
    src = tmp_path / "clip.y4m"
    _write_y4m(str(src), w, h, n)
    buf = C.create_string_buffer(1024)
    outs = {}
    for ext in ("obu", "ivf", "av1-tmp.mkv"):
        out = tmp_path / ("clip." + ext)
        args = "\n".join(["-hide_banner", "-i", str(src), "-global_quality:v:0", str(q), "-g", str(gop), "-av1mi_segments", "2", str(out)])
        assert host.av1mi_host_run_transcode(args.encode(), buf, 1024) == 0 and buf.value == b""
        outs[ext] = out.read_bytes()
    obu = outs["obu"]
    kinds = [t for t, _ in _obus(obu)]
    assert kinds.count(2) == n and kinds.count(1) == 3 and kinds.count(6) == n      # 7 temporal units, a sequence header per GOP
    assert len(obu) < src.stat().st_size
    # IVF and Matroska carry the same temporal units
    ivf = outs["ivf"]
    assert ivf[:4] == b"DKIF" and ivf[8:12] == b"AV01" and int.from_bytes(ivf[24:28], "little") == n
    pos, units = 32, []
    while pos < len(ivf):
        sz = int.from_bytes(ivf[pos:pos + 4], "little")
        units.append(ivf[pos + 12:pos + 12 + sz])
        pos += 12 + sz
    assert b"".join(units) == obu
    frames, priv = _ebml_blocks(outs["av1-tmp.mkv"])
    assert len(frames) == n and [k for k, _ in frames] == [i % gop == 0 for i in range(n)] and priv[0] == 0x81
    assert b"".join(b"\x12\x00" + f for _, f in frames) == obu
    if D.available():
        Y, U, V = synth.frames(w, h, n, 8)
        got = D.decode(obu)
        assert len(got) == n
        ref = None
        for i in range(n):
            key = i % gop == 0
            p = av1mi.policy_frame_params(q, 8, 0 if key else 1)
            from test_gpu_session import _oracle_filters, _oracle_key32
            mi = None
            if key and w % 64 == 0:
                # the command line's default (-av1mi_key_block_size 32): key frames in 32x32 blocks over the complete superblock rows
                _, _, r = _oracle_key32(O, Y[i], U[i], V[i], 8, q)
                hA = h // 64 * 64
                mi = (np.full((h // 4, w // 4), int(O.lf_mi(3, 3, p.lf_level[0], p.lf_level[1])), np.uint32),
                      np.full((h // 8, w // 8), int(O.lf_mi(2, 2, p.lf_level[2], p.lf_level[2])), np.uint32))
                mi[0][:hA // 4] = int(O.lf_mi(5, 5, p.lf_level[0], p.lf_level[1]))
                mi[1][:hA // 8] = int(O.lf_mi(4, 4, p.lf_level[2], p.lf_level[2]))
                skip8 = np.zeros((h // 8, w // 8), np.uint8)
            elif key:
                r = O.intra_encode_frame(Y[i], U[i], V[i], 8, 8, q)
                skip8 = np.zeros((h // 8, w // 8), np.uint8)
            else:
                r = O.inter_encode_frame((Y[i], U[i], V[i]), ref, 8, q, 8)
                skip8 = r["skip"].reshape(h // 8, w // 8)
            ref, _ = _oracle_filters(O, r, 8, p, w, h, skip8, (Y[i], U[i], V[i]), mi=mi)
            for pl in range(3):
                assert (got[i][pl] == ref[pl]).all(), "frame %d plane %d: the decoded file differs from the oracle chain" % (i, pl)
    # lifecycle: tight ratio -> skipped with markers; generous ratio -> success, and the SOURCE IS KEPT (video-only output)
    status, reason = C.create_string_buffer(256), C.create_string_buffer(1024)
    orig = src.stat().st_size
    src2 = tmp_path / "other.y4m"
    src2.write_bytes(src.read_bytes())
    assert host.av1mi_host_process_job(str(src2).encode(), orig, 1e-4, str(tmp_path).encode(), 0, 0, status, reason, 256) == 0
    assert status.value == b"skipped" and reason.value.startswith(b"size gate: new ") and (tmp_path / "other.av1qsvd-skip").exists()
    assert (tmp_path / "other.av1qsvd-why.txt").exists() and not (tmp_path / "other.av1-tmp.mkv").exists() and src2.stat().st_size == orig
    before = src.read_bytes()
    assert host.av1mi_host_process_job(str(src).encode(), orig, 5.0, str(tmp_path).encode(), 0, 0, status, reason, 256) == 0
    assert status.value == b"success" and src.read_bytes() == before and (tmp_path / "clip.av1mi.mkv").exists()
    assert not (tmp_path / "clip.av1-tmp.mkv").exists() and (tmp_path / "test.json").exists()
    # only on request does the coded file take the source's place (the reference's daemon.go:154 step)
    src3 = tmp_path / "third.y4m"
    src3.write_bytes(before)
    assert host.av1mi_host_process_job(str(src3).encode(), orig, 5.0, str(tmp_path).encode(), 0, 1, status, reason, 256) == 0
    assert status.value == b"success" and src3.read_bytes()[:4] == b"\x1a\x45\xdf\xa3"


@pytest.mark.gpu
def test_run_transcode_of_a_source_whose_size_is_not_a_multiple_of_8(host, tmp_path, O):
    """854x480-style sources: the backend pads to the coded size while reading, the file announces the true size, dav1d outputs it,
    and the frames equal the oracle's chain of the same source (tests/test_av1_conformance.py visible_gop) — host coder and GPU coder"""
    import dav1d_ref as D
    if not D.available():
        pytest.skip("no dav1d in this image")
    sys_path_synth()
    import pipeline as P
    import synth
    import test_av1_conformance as T
    for vw, vh, bd, q, n, gop in ((109, 75, 8, 140, 3, 3), (70, 61, 10, 90, 7, 3)):      # the second: three GOPs, two segments in lockstep
        w, h = (vw + 7) // 8 * 8, (vh + 7) // 8 * 8
        Yc, Uc, Vc = synth.frames(w + 8, h + 8, n, bd, 3)
        src = tmp_path / ("odd%d.y4m" % bd)
        with open(str(src), "wb") as f:
            f.write(("YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C%s\n" % (vw, vh, "420jpeg" if bd == 8 else "420p10")).encode())
            for i in range(n):
                f.write(b"FRAME\n")
                for pl in (Yc[i][:vh, :vw], Uc[i][:(vh + 1) // 2, :(vw + 1) // 2], Vc[i][:(vh + 1) // 2, :(vw + 1) // 2]):
                    f.write(np.ascontiguousarray(pl).astype("<u2" if bd == 10 else np.uint8).tobytes())
        stream, refs = b"", []
        for g0 in range(0, n, gop):
            st, rf, _ = T.visible_gop(O, P, vw, vh, bd, q, min(gop, n - g0), frames=(Yc[g0:], Uc[g0:], Vc[g0:]))
            stream += st
            refs += rf
        buf = C.create_string_buffer(1024)
        for gpu_entropy in ("0", "1"):
            out = tmp_path / ("odd%d_%s.obu" % (bd, gpu_entropy))
            args = "\n".join(["-i", str(src), "-global_quality:v:0", str(q), "-g", str(gop), "-av1mi_segments", "2", "-av1mi_gpu_entropy", gpu_entropy, str(out)])
            assert host.av1mi_host_run_transcode(args.encode(), buf, 1024) == 0 and buf.value == b"", buf.value
            obu = out.read_bytes()
            assert obu == stream, "the file differs from the oracle chain's stream"
            got = D.decode(obu)
            assert len(got) == n and got[0][0].shape == (vh, vw) and got[0][1].shape == ((vh + 1) // 2, (vw + 1) // 2)
            for t in range(n):
                for i, a in enumerate(T._crop(refs[t], vw, vh)):
                    assert (got[t][i] == a).all()
        mkv = tmp_path / ("odd%d.mkv" % bd)
        args = "\n".join(["-i", str(src), "-global_quality:v:0", str(q), "-g", str(gop), str(mkv)])
        assert host.av1mi_host_run_transcode(args.encode(), buf, 1024) == 0
        data = mkv.read_bytes()
        assert b"\xb0" + bytes([0x81, vw]) in data and b"\xba" + bytes([0x81, vh]) in data      # PixelWidth / PixelHeight = the true size


@pytest.mark.gpu
def test_the_public_drop_in_and_streamed_input(host, tmp_path):
    """(1) av1mi_run_transcode(argc, argv, err, cap) itself — the symbol the cgo shim of INTEGRATION.md binds in place of
    internal/ffmpeg/transcode.go:194 (call site internal/daemon/daemon.go:101) — with the argv TranscodeArgs builds; (2) the same
    clip piped into the command line (`cat clip.y4m | av1mi_transcode -i - ...`, transcode.go:68 `-i`) and through a FIFO gives the
    file the in-place reader gives, byte for byte (streams are read one group of GOPs ahead: y4m.cpp)"""
    import subprocess
    import threading
    w, h, n, q, gop = 136, 72, 11, 110, 4
    src = tmp_path / "clip.y4m"
    _write_y4m(str(src), w, h, n)
    host.av1mi_run_transcode.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_size_t]
    _, argv = _args(host, str(src), str(tmp_path / "clip.av1-tmp.mkv"), height=h)      # the reference's own argv
    argv = argv[:-1] + ["-g", str(gop), "-av1mi_segments", "2", argv[-1]]
    argv[argv.index("-global_quality:v:0") + 1] = str(q)
    arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
    err = C.create_string_buffer(1024)
    assert host.av1mi_run_transcode(len(argv), arr, err, 1024) == 0 and err.value == b""
    ref = (tmp_path / "clip.av1-tmp.mkv").read_bytes()
    assert len(_ebml_blocks(ref)[0]) == n
    bad = (C.c_char_p * 3)(b"-i", str(tmp_path / "missing.y4m").encode(), str(tmp_path / "x.mkv").encode())
    assert host.av1mi_run_transcode(3, bad, err, 1024) == 1 and err.value.startswith(b"av1mi failed with exit code 1: ") and not (tmp_path / "x.mkv").exists()
    cli = os.path.join(os.path.dirname(HOST), "av1mi_transcode")
    tail = ["-global_quality:v:0", str(q), "-g", str(gop), "-av1mi_segments", "2"]
    out = tmp_path / "piped.mkv"
    with open(src, "rb") as f:
        r = subprocess.run([cli, "-i", "-"] + tail + [str(out)], stdin=f, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == ref
    fifo, out2 = tmp_path / "in.fifo", tmp_path / "fifo.mkv"
    os.mkfifo(fifo)
    t = threading.Thread(target=lambda: open(fifo, "wb").write(src.read_bytes()))
    t.start()
    argv2 = ["-i", str(fifo)] + tail + [str(out2)]
    arr2 = (C.c_char_p * len(argv2))(*[a.encode() for a in argv2])
    assert host.av1mi_run_transcode(len(argv2), arr2, err, 1024) == 0, err.value
    t.join()
    assert out2.read_bytes() == ref


@pytest.mark.gpu
def test_run_transcode_of_a_cropped_source_with_key_frames_in_32x32_blocks(host, tmp_path):
    """a 250x70 source is coded at 256x72: the width is a multiple of 64, so the command line's default codes its key frames in 32x32
    blocks over the first superblock row (8x8 below) — the GPU coder and the host coder write the same file, dav1d outputs 250x70
    frames close to the source"""
    import dav1d_ref as D
    if not D.available():
        pytest.skip("no dav1d in this image")
    sys_path_synth()
    import synth
    vw, vh, n, q, gop = 250, 70, 5, 100, 3
    Yc, Uc, Vc = synth.frames(264, 80, n, 8, 3)
    src = tmp_path / "crop.y4m"
    with open(str(src), "wb") as f:
        f.write(("YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420jpeg\n" % (vw, vh)).encode())
        for i in range(n):
            f.write(b"FRAME\n")
            for pl in (Yc[i][:vh, :vw], Uc[i][:vh // 2, :vw // 2], Vc[i][:vh // 2, :vw // 2]):
                f.write(np.ascontiguousarray(pl).tobytes())
    buf = C.create_string_buffer(1024)
    outs = []
    for gpu_entropy, kbs in (("1", "32"), ("0", "32"), ("1", "8")):
        out = tmp_path / ("crop_%s_%s.obu" % (gpu_entropy, kbs))
        args = "\n".join(["-i", str(src), "-global_quality:v:0", str(q), "-g", str(gop), "-av1mi_segments", "2", "-av1mi_gpu_entropy", gpu_entropy,
                          "-av1mi_key_block_size", kbs, str(out)])
        assert host.av1mi_host_run_transcode(args.encode(), buf, 1024) == 0 and buf.value == b"", buf.value
        outs.append(out.read_bytes())
    assert outs[0] == outs[1] and outs[0] != outs[2] and len(outs[0]) < len(outs[2])
    got = D.decode(outs[0])
    assert len(got) == n and got[0][0].shape == (vh, vw)
    for i in range(n):
        mse = np.mean((got[i][0].astype(np.float64) - Yc[i][:vh, :vw]) ** 2)
        assert 10 * np.log10(255 ** 2 / mse) > 30


@pytest.mark.gpu
def test_run_transcode_copies_the_tracks_of_a_side_file(host, tmp_path):
    """`-c:a copy -c:s copy` (internal/ffmpeg/transcode.go:134-137) after an external demux: `-av1mi_tracks side.mka` puts the side
    file's audio and subtitle tracks next to the coded video; the video blocks are those of the video-only run (muxer details:
    tests/test_mux.py on the CPU)"""
    import struct
    import test_mux as M
    w, h, n, gop = 136, 72, 9, 4
    src = tmp_path / "clip.y4m"
    _write_y4m(str(src), w, h, n)
    side = tmp_path / "side.mka"
    audio = [(t, bytes([t % 251]) * 50) for t in range(0, 400, 20)]
    M.side_file(side, [(1, M.AUDIO), (2, M.SUBS)], [(0, [M.simple_block(1, t, d) for t, d in audio] + [M.block_group(2, 120, b"hello", duration=150)])])
    host.av1mi_run_transcode.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_size_t]
    err = C.create_string_buffer(1024)
    outs = []
    for extra, name in (([], "plain.mkv"), (["-av1mi_tracks", str(side)], "copied.mkv")):
        argv = ["-i", str(src), "-global_quality:v:0", "120", "-g", str(gop), "-av1mi_segments", "2"] + extra + [str(tmp_path / name)]
        arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
        assert host.av1mi_run_transcode(len(argv), arr, err, 1024) == 0, err.value
        outs.append(M.read_mkv(tmp_path / name))
    (_, t0, b0), (info, t1, b1) = outs
    assert sorted(t0) == [1] and sorted(t1) == [1, 2, 3] and t1[2][0x86] == b"A_OPUS" and t1[3][0x86] == b"S_TEXT/UTF8"
    assert [b for b in b1 if b["track"] == 1] == b0 and len(b0) == n
    assert [(b["t"], b["data"]) for b in b1 if b["track"] == 2] == audio
    assert [(b["t"], b["duration"], b["data"]) for b in b1 if b["track"] == 3] == [(120, 150, b"hello")]
    assert struct.unpack(">d", info[0x4489])[0] >= 380.0


@pytest.mark.gpu
def test_run_transcode_of_dense_content_at_the_reference_quality(host, tmp_path):
    """noise at quality 25 (DetermineQuality of a sub-1080p source) overflows the GPU coder; RunTranscode must still write
    the stream — the same bytes as with the host coder — also when the file holds fewer GOPs than segments x groups (absent segments
    are coded from flat planes and dropped), and dav1d must decode every frame"""
    import dav1d_ref as D
    w, h, n, gop = 192, 128, 8, 3
    rng = np.random.default_rng(3)
    src = tmp_path / "noise.y4m"
    with open(src, "wb") as f:
        f.write(("YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420jpeg\n" % (w, h)).encode())
        for i in range(n):
            f.write(b"FRAME\n" + rng.integers(0, 256, w * h * 3 // 2).astype(np.uint8).tobytes())
    buf = C.create_string_buffer(1024)
    outs = []
    for ge in ("1", "0"):
        out = tmp_path / ("noise%s.obu" % ge)
        args = "\n".join(["-i", str(src), "-global_quality:v:0", str(host.av1mi_host_determine_quality(h)), "-g", str(gop), "-av1mi_segments", "2",
                          "-av1mi_gpu_entropy", ge, str(out)])
        assert host.av1mi_host_run_transcode(args.encode(), buf, 1024) == 0, buf.value
        outs.append(out.read_bytes())
    assert outs[0] == outs[1]
    if D.available():
        assert len(D.decode(outs[0])) == n


def sys_path_synth():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HOST), ".."))


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
    ref = (tmp_path / "solo.av1mi.mkv").read_bytes()
    assert ref[:4] == b"\x1a\x45\xdf\xa3"            # EBML: a Matroska file
    n, status = _pool(host, [str(c) for c in clips], 3, 1, 5.0, tmp_path)
    assert n == 4 and status == ["success"] * 4
    for i, c in enumerate(clips):
        assert (tmp_path / ("c%d.av1mi.mkv" % i)).read_bytes() == ref, "job %d differs from the solo run" % i
        assert (tmp_path / ("pool%d.json" % i)).exists()


def test_containers_carry_the_temporal_units_unchanged(host, tmp_path, O):
    """host/mux.cpp on the CPU: an oracle-coded closed GOP written as .obu / .ivf / Matroska; the units read back from each
    container are the same bytes, and (where dav1d is present) decode to the oracle's frames"""
    import dav1d_ref as D
    import pipeline as P
    from test_av1_conformance import _gop
    w, h, bd, q, n = 128, 64, 8, 140, 3
    stream, refs = _gop(O, P, w, h, bd, q, n)
    units, cur = [], b""
    for t, payload in _obus(stream):
        if t == 2 and cur:
            units.append(cur)
            cur = b""
        hdr = bytes([t << 3 | 2])
        size, lebs = len(payload), b""
        while True:
            b7 = size & 127
            size >>= 7
            lebs += bytes([b7 | (128 if size else 0)])
            if not size:
                break
        cur += hdr + lebs + payload
    units.append(cur)
    assert len(units) == n and b"".join(units) == stream
    data = np.frombuffer(stream, np.uint8)
    sizes = np.array([len(u) for u in units], np.int64)
    keys = np.array([1, 0, 0], np.uint8)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    host.av1mi_host_mux_units.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    got = {}
    for ext in ("obu", "ivf", "mkv"):
        path = tmp_path / ("g." + ext)
        assert host.av1mi_host_mux_units(str(path).encode(), w, h, bd, 30000, 1001, vp(data), vp(sizes), vp(keys), n) == 0
        got[ext] = path.read_bytes()
    assert got["obu"] == stream
    ivf = got["ivf"]
    assert ivf[:4] == b"DKIF" and int.from_bytes(ivf[16:20], "little") == 30000 and int.from_bytes(ivf[24:28], "little") == n
    frames, priv = _ebml_blocks(got["mkv"])
    assert [k for k, _ in frames] == [True, False, False]
    assert priv[:4] == bytes([0x81, 0x1F, 0x0C, 0x00]) and priv[4] == 0x0A      # av1C, then the sequence header OBU
    rebuilt = b"".join(b"\x12\x00" + f for _, f in frames)
    assert rebuilt == stream
    if D.available():
        dec = D.decode(rebuilt)
        assert all((dec[t][i] == refs[t][i]).all() for t in range(n) for i in range(3))


def test_job_record_has_the_reference_fields_and_escapes_strings(host):
    """jobs.Job JSON (jobs.go:25-46): field names, omitempty, two-space indent; strings escaped (paths with quotes broke it)"""
    import json
    buf = C.create_string_buffer(4096)
    host.av1mi_host_job_json.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_longlong, C.c_longlong, C.c_char_p, C.c_int]
    host.av1mi_host_job_json(b"abc", b'/m/we "quoted"\\dir/a\tb.mkv', b"failed", b'ffmpeg exit code 1: x\ny', 1000, 0, buf, 4096)
    text = buf.value.decode()
    j = json.loads(text)
    assert j == {"id": "abc", "source_path": '/m/we "quoted"\\dir/a\tb.mkv', "created_at": "2026-01-02T03:04:05Z", "status": "failed",
                 "reason": "ffmpeg exit code 1: x\ny", "original_bytes": 1000, "is_webrip_like": False}
    assert text.startswith('{\n  "id": "abc",\n  "source_path": ') and list(j) == ["id", "source_path", "created_at", "status", "reason", "original_bytes",
                                                                                  "is_webrip_like"]


def test_amd_gpu_usage_probe(host, tmp_path):
    """the AMD twin of internal/tui/gpu.go getGPUUsage: amdgpu's gpu_busy_percent of the device-th card that has one"""
    host.av1mi_host_gpu_usage.restype = C.c_double
    host.av1mi_host_gpu_usage.argtypes = [C.c_int, C.c_char_p]
    assert host.av1mi_host_gpu_usage(0, str(tmp_path).encode()) == -1.0          # no card: "cannot determine"
    for card, val in ((0, None), (1, "37\n"), (3, "100\n")):                      # card0 = a display adapter without the file
        d = tmp_path / "class" / "drm" / ("card%d" % card) / "device"
        d.mkdir(parents=True)
        if val:
            (d / "gpu_busy_percent").write_text(val)
    assert host.av1mi_host_gpu_usage(0, str(tmp_path).encode()) == 37.0
    assert host.av1mi_host_gpu_usage(1, str(tmp_path).encode()) == 100.0
    assert host.av1mi_host_gpu_usage(2, str(tmp_path).encode()) == -1.0
