"""The Matroska muxer's STREAM COPY of demuxed tracks (host/mux.cpp; reference: `-map 0:a? -map 0:s? -c:a copy -c:s copy -f matroska`,
internal/ffmpeg/transcode.go:77-83,134-145).  The source's audio / subtitle tracks arrive as Matroska side files (what
`ffmpeg -i movie.mkv -map 0:a? -map 0:s? -c copy side.mka` or mkvextract leave); the muxer copies every track entry verbatim under a
new number and re-times every block into its own clusters, interleaved with the video.  CPU only: the muxer is driven through the
host library's test hook with stand-in temporal units, the files on both sides are built / read by the small EBML codec below."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "av1-go_amd", "host", "libav1mi_host.so")


# ------------------------------------------------------------------------------------------------ a small EBML writer / reader
def vint(v, n=None):
    if n is None:
        n = 1
        while v >= (1 << (7 * n)) - 1:
            n += 1
    return (v | (1 << (7 * n))).to_bytes(n, "big")


def eid(i):
    return i.to_bytes((i.bit_length() + 7) // 8, "big")


def el(i, body):
    return eid(i) + vint(len(body)) + body


def el_unknown(i, body):
    return eid(i) + b"\x01\xff\xff\xff\xff\xff\xff\xff" + body


def uint(v):
    return v.to_bytes(max((v.bit_length() + 7) // 8, 1), "big")


def simple_block(track, rel, data, key=True, lacing=0):
    return el(0xA3, vint(track) + struct.pack(">h", rel) + bytes([(0x80 if key else 0) | (lacing << 1)]) + data)


def block_group(track, rel, data, duration=None, ref=None):
    g = el(0xA1, vint(track) + struct.pack(">h", rel) + b"\x00" + data)
    if duration is not None:
        g += el(0x9B, uint(duration))
    if ref is not None:
        g += el(0xFB, struct.pack(">b", ref))
    return el(0xA0, g)


def side_file(path, entries, clusters, scale_ns=1000000, unknown_sizes=False, extra=b""):
    """entries: [(number, body without TrackNumber / TrackUID)]; clusters: [(timestamp, [block elements])]"""
    head = el(0x1A45DFA3, el(0x4286, uint(1)) + el(0x4282, b"matroska") + el(0x4287, uint(4)) + el(0x4285, uint(2)))
    info = el(0x1549A966, el(0x2AD7B1, uint(scale_ns)) + el(0x4D80, b"test"))
    tracks = el(0x1654AE6B, b"".join(el(0xAE, el(0xD7, uint(n)) + el(0x73C5, uint(1000 + n)) + body) for n, body in entries))
    cl = b"".join((el_unknown if unknown_sizes else el)(0x1F43B675, el(0xE7, uint(ts)) + b"".join(blocks)) for ts, blocks in clusters)
    body = el(0x114D9B74, b"") + info + tracks + cl + extra
    with open(path, "wb") as f:
        f.write(head + (el_unknown if unknown_sizes else el)(0x18538067, body))


def parse(buf, pos, end):
    """children of a master element: (id, payload start, payload end)"""
    out = []
    while pos < end:
        n = 1
        while not buf[pos] & (0x80 >> (n - 1)):
            n += 1
        i = int.from_bytes(buf[pos:pos + n], "big")
        pos += n
        k = 1
        while not buf[pos] & (0x80 >> (k - 1)):
            k += 1
        size = int.from_bytes(buf[pos:pos + k], "big") & ((1 << (7 * k)) - 1)
        pos += k
        out.append((i, pos, pos + size))
        pos += size
    return out


def read_mkv(path):
    buf = open(path, "rb").read()
    top = parse(buf, 0, len(buf))
    assert top[0][0] == 0x1A45DFA3 and top[1][0] == 0x18538067 and top[1][2] == len(buf)
    tracks, blocks, info = {}, [], {}
    for i, a, b in parse(buf, top[1][1], top[1][2]):
        if i == 0x1549A966:
            for j, c, d in parse(buf, a, b):
                info[j] = buf[c:d]
        elif i == 0x1654AE6B:
            for j, c, d in parse(buf, a, b):
                e = {k: buf[x:y] for k, x, y in parse(buf, c, d)}
                tracks[int.from_bytes(e[0xD7], "big")] = e
        elif i == 0x1F43B675:
            ts = None
            for j, c, d in parse(buf, a, b):
                if j == 0xE7:
                    ts = int.from_bytes(buf[c:d], "big")
                elif j == 0xA3:
                    rel = struct.unpack(">h", buf[c + 1:c + 3])[0]
                    blocks.append(dict(track=buf[c] & 0x7F, t=ts + rel, flags=buf[c + 3], data=buf[c + 4:d], duration=None, key=bool(buf[c + 3] & 0x80)))
                elif j == 0xA0:
                    g = {k: (x, y) for k, x, y in parse(buf, c, d)}
                    x, y = g[0xA1]
                    rel = struct.unpack(">h", buf[x + 1:x + 3])[0]
                    blocks.append(dict(track=buf[x] & 0x7F, t=ts + rel, flags=buf[x + 3], data=buf[x + 4:y], key=0xFB not in g,
                                       duration=int.from_bytes(buf[g[0x9B][0]:g[0x9B][1]], "big") if 0x9B in g else None))
    return info, tracks, blocks


# ------------------------------------------------------------------------------------------------ the muxer through the test hook
@pytest.fixture(scope="module")
def lib():
    h = C.CDLL(LIB)
    h.av1mi_host_mux_selftest.restype = C.c_int
    h.av1mi_host_mux_selftest.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_int]
    return h


def mux(lib, out, units, sides, fps=(30, 1), gop=30):
    data = np.frombuffer(b"".join(units), np.uint8)
    off = np.cumsum([0] + [len(u) for u in units]).astype(np.int64)
    err = C.create_string_buffer(512)
    rc = lib.av1mi_host_mux_selftest(str(out).encode(), 640, 360, 8, fps[0], fps[1], data.ctypes.data, off.ctypes.data, len(units), gop, "\n".join(map(str, sides)).encode(), err, 512)
    return rc, err.value.decode()


def video_units(n, seed=1):
    rng = np.random.default_rng(seed)
    return [b"\x12\x00" + rng.integers(0, 256, int(rng.integers(20, 400)), dtype=np.uint8).tobytes() for _ in range(n)]


AUDIO = el(0x83, uint(2)) + el(0x86, b"A_OPUS") + el(0x63A2, b"OpusHead\x01\x02\x38\x01\x80\xbb\x00\x00\x00\x00\x00") + el(0x22B59C, b"jpn") + \
        el(0xE1, el(0xB5, struct.pack(">d", 48000.0)) + el(0x9F, uint(2))) + el(0x56AA, uint(6500000))
SUBS = el(0x83, uint(17)) + el(0x86, b"S_TEXT/UTF8") + el(0x22B59C, b"eng") + el(0x536E, "Signs & songs".encode())


def test_tracks_of_a_side_file_are_copied_and_interleaved(lib, tmp_path):
    rng = np.random.default_rng(3)
    # 2.4 s of 20 ms audio frames in clusters of one second; three subtitle cues with durations, one of them beyond the video's end
    audio = [(t, rng.integers(0, 256, int(rng.integers(40, 160)), dtype=np.uint8).tobytes()) for t in range(0, 2400, 20)]
    cues = [(150, 1200, "first line".encode()), (900, 800, "zweite Zeile — ü".encode()), (2300, 500, b"after the last frame")]
    clusters = []
    for c0 in range(0, 3000, 1000):
        blocks = [(t, simple_block(1, t - c0, d)) for t, d in audio if c0 <= t < c0 + 1000] + \
                 [(t, block_group(2, t - c0, d, duration=dur)) for t, dur, d in cues if c0 <= t < c0 + 1000]
        clusters.append((c0, [b for _, b in sorted(blocks, key=lambda x: x[0])]))
    side = tmp_path / "side.mka"
    side_file(side, [(1, AUDIO), (2, SUBS)], clusters, extra=el(0x1C53BB6B, b"") + el(0x1254C367, b""))
    units = video_units(60)                      # 2 s of video at 30 fps, two closed GOPs
    out = tmp_path / "out.mkv"
    rc, err = mux(lib, out, units, [side])
    assert rc == 0, err
    info, tracks, blocks = read_mkv(out)
    assert sorted(tracks) == [1, 2, 3] and tracks[1][0x86] == b"V_AV1"
    # the entries are the side file's, verbatim, under new numbers / UIDs
    for number, body in ((2, AUDIO), (3, SUBS)):
        want = {k: body[a:b] for k, a, b in parse(body, 0, len(body))}
        got = {k: v for k, v in tracks[number].items() if k not in (0xD7, 0x73C5)}
        assert got == want
    v = [b for b in blocks if b["track"] == 1]
    assert [b["data"] for b in v] == [u[2:] for u in units] and [b["key"] for b in v] == [i % 30 == 0 for i in range(60)]
    assert [b["t"] for b in v] == [round(i * 1000 / 30) for i in range(60)]
    a = [b for b in blocks if b["track"] == 2]
    assert [(b["t"], b["data"]) for b in a] == audio and all(b["key"] and b["duration"] is None for b in a)
    s = [b for b in blocks if b["track"] == 3]
    assert [(b["t"], b["duration"], b["data"]) for b in s] == cues
    # interleaved: in file order no block is later than the video frame that follows it, and timestamps never step back by more than
    # one video frame
    last_video = -1
    for b in blocks:
        if b["track"] == 1:
            last_video = b["t"]
        else:
            assert b["t"] >= last_video - 34 or last_video < 0
    ts = [b["t"] for b in blocks]
    assert all(y >= x - 34 for x, y in zip(ts, ts[1:]))
    # the duration covers the last subtitle
    assert struct.unpack(">d", info[0x4489])[0] >= 2800.0


def test_two_side_files_other_timestamp_scale_unknown_sizes_lacing_and_non_key_blocks(lib, tmp_path):
    rng = np.random.default_rng(5)
    # file A: 0.1 ms timestamp scale, clusters and segment of unknown size (a live-muxed file), Xiph-laced blocks copied bit for bit
    laced = [(t, bytes([1, 10]) + rng.integers(0, 256, 30, dtype=np.uint8).tobytes()) for t in range(0, 9000, 400)]     # 40 ms apart in 0.1 ms units
    a = tmp_path / "a.mka"
    side_file(a, [(7, AUDIO)], [(0, [simple_block(7, t, d, lacing=1) for t, d in laced if t < 5000]),
                                (5000, [simple_block(7, t - 5000, d, lacing=1) for t, d in laced if t >= 5000])], scale_ns=100000, unknown_sizes=True)
    # file B: a track with non-key blocks (BlockGroup + ReferenceBlock) and a block of a track the file does not declare
    b = tmp_path / "b.mks"
    frames = [(t, rng.integers(0, 256, 25, dtype=np.uint8).tobytes(), t % 200 == 0) for t in range(0, 800, 100)]
    side_file(b, [(1, SUBS)], [(0, [simple_block(1, t, d) if k else block_group(1, t, d, ref=-100) for t, d, k in frames] + [simple_block(9, 5, b"stray")])])
    units = video_units(30, seed=2)
    out = tmp_path / "out.mkv"
    rc, err = mux(lib, out, units, [a, b], fps=(25, 1), gop=10)
    assert rc == 0, err
    _, tracks, blocks = read_mkv(out)
    assert sorted(tracks) == [1, 2, 3] and tracks[2][0x86] == b"A_OPUS" and tracks[3][0x86] == b"S_TEXT/UTF8"
    got = [(b_["t"], b_["flags"] & 0x06, b_["data"]) for b_ in blocks if b_["track"] == 2]
    assert got == [(t // 10, 2, d) for t, d in laced]                       # 0.1 ms units -> ms, lacing flag and laced payload untouched
    got = [(b_["t"], b_["data"], b_["key"]) for b_ in blocks if b_["track"] == 3]
    assert got == frames
    assert not any(b_["data"] == b"stray" for b_ in blocks)
    assert [b_["data"] for b_ in blocks if b_["track"] == 1] == [u[2:] for u in units]


def test_errors(lib, tmp_path):
    units = video_units(3)
    rc, err = mux(lib, tmp_path / "o.mkv", units, [tmp_path / "missing.mka"])
    assert rc == -1 and "No such file" in err
    bad = tmp_path / "bad.mka"
    bad.write_bytes(b"RIFF....WAVEfmt ")
    rc, err = mux(lib, tmp_path / "o.mkv", units, [bad])
    assert rc == -1 and "Invalid data" in err
    side = tmp_path / "s.mka"
    side_file(side, [(1, AUDIO)], [(0, [simple_block(1, 0, b"x")])])
    rc, err = mux(lib, tmp_path / "o.ivf", units, [side])
    assert rc == -1 and "Matroska" in err
    # a video-only file is what it was before
    rc, err = mux(lib, tmp_path / "plain.mkv", units, [])
    assert rc == 0
    _, tracks, blocks = read_mkv(tmp_path / "plain.mkv")
    assert sorted(tracks) == [1] and len(blocks) == 3
