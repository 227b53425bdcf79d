"""Measurement for DESIGN §7-1, CPU only: what would a per-TILE choice between 8x8, 16x16 and 32x32 blocks
(transform = block) buy on the synthetic key frames?  The oracle's intra loop codes the frame once per block size; tiles are single
superblocks and independent, so any per-tile mix of the three results is a valid encode.  Streams are written by the general block
writer (av1mi_obu_write_blocks_temporal_unit), tile sizes are parsed back out of them, the mixed stream is decoded by dav1d.
usage: python tools/rd_blocksize.py [width height bit_depth q ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "av1-go_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import synth
import av1_blocks as B
import dav1d_ref
from oracle import oracle as O

BSIZE = {8: 3, 16: 6, 32: 9, 64: 12}
SIZES = (8, 16, 32, 64)


def psnr(a, b, bd):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10 * np.log10(((1 << bd) - 1) ** 2 / mse)


def frame_blocks(lay, enc, choice):
    """choice[sb_row, sb_col] = block size of the tile; enc[bs] = oracle output of the whole frame at that size"""
    def chooser(r, c, bsize, allowed):
        bs = choice[r >> 4, c >> 4]
        return B.uniform_chooser(BSIZE[bs])(r, c, bsize, allowed)
    parts, tree = B.build_tree(lay, chooser)
    blocks = []
    for r, c, bsize, tb in tree:
        bs = B.BW4[bsize] * 4
        o = enc[bs]
        bw = lay.w // bs
        i = (r * 4 // bs) * bw + c * 4 // bs
        blocks.append(dict(r=r, c=c, bsize=bsize, tile=tb, skip=0, is_inter=0, y_mode=int(o["modes_y"][i]), uv_mode=int(o["modes_uv"][i]),
                           angle_y=0, angle_uv=0, cfl=(0, 0), tx_depth=0, filt=0, mv=(0, 0), tx=B.max_tx_rect(bsize), tx_types=[0],
                           levels=[[o["lev_y"][i] if bs < 64 else o["lev_y"][i].reshape(-1)[:1024].reshape(32, 32)], [o["lev_u"][i]], [o["lev_v"][i]]]))
    return parts, blocks


def tile_sizes(tu, ntiles):
    """sizes of the tiles of the (single) frame OBU of a temporal unit: the tile group's size fields are found by trying every start"""
    pos, frame = 0, None
    while pos < len(tu):
        hdr = tu[pos]; typ = (hdr >> 3) & 15; pos += 1
        size = 0; shift = 0
        while True:
            b = tu[pos]; pos += 1
            size |= (b & 127) << shift; shift += 7
            if not b & 128:
                break
        if typ == 6:
            frame = tu[pos:pos + size]
        pos += size
    for tsb in (1, 2, 3, 4):
        for start in range(4, 200):
            p, sizes = start, []
            ok = True
            for _ in range(ntiles - 1):
                if p + tsb > len(frame):
                    ok = False; break
                n = int.from_bytes(frame[p:p + tsb], "little") + 1
                p += tsb + n
                sizes.append(n)
                if p > len(frame):
                    ok = False; break
            if ok and p < len(frame) and max(sizes) >= (1 << (8 * (tsb - 1))) and (tsb == 1 or True):
                last = len(frame) - p
                if last > 0 and last < (1 << (8 * tsb)):
                    return np.array(sizes + [last])
    raise RuntimeError("tile sizes not found")


def run(W, H, bd, q, frame=0):
    Hc = (H + 63) // 64 * 64
    Y, U, V = (a[0] for a in synth.frames(W, H, 1, bd, frame))
    Y = np.pad(Y, ((0, Hc - H), (0, 0)), mode="edge"); U = np.pad(U, ((0, (Hc - H) // 2), (0, 0)), mode="edge"); V = np.pad(V, ((0, (Hc - H) // 2), (0, 0)), mode="edge")
    lay = B.Layout(W, Hc)
    sbr, sbc = Hc // 64, W // 64
    enc, sse, size, total = {}, {}, {}, {}
    for bs in SIZES:
        enc[bs] = O.intra_encode_frame(Y, U, V, bd, bs, q)
        d = (enc[bs]["rec_y"].astype(np.int64) - Y) ** 2
        sse[bs] = d.reshape(sbr, 64, sbc, 64).sum(axis=(1, 3)).astype(np.float64)
        parts, blocks = frame_blocks(lay, enc, np.full((sbr, sbc), bs))
        tu = B.encode(lay, bd, q, parts, blocks)
        size[bs] = tile_sizes(tu, sbr * sbc).reshape(sbr, sbc).astype(np.float64)
        total[bs] = (len(tu), psnr(enc[bs]["rec_y"], Y, bd))
        print("  uniform %2dx%-2d: %8d bytes  PSNR-Y %.3f dB" % (bs, bs, total[bs][0], total[bs][1]), flush=True)
    return lay, (Y, U, V), enc, sse, size, total


def mixed(lay, planes, enc, sse, size, bd, q, lam, check=False):
    sbr, sbc = sse[8].shape
    J = np.stack([sse[bs] + lam * 8 * size[bs] for bs in SIZES])
    choice = np.array(SIZES)[np.argmin(J, axis=0)]
    parts, blocks = frame_blocks(lay, enc, choice)
    tu = B.encode(lay, bd, q, parts, blocks)
    rec = np.zeros_like(planes[0])
    for r in range(sbr):
        for c in range(sbc):
            rec[r * 64:r * 64 + 64, c * 64:c * 64 + 64] = enc[int(choice[r, c])]["rec_y"][r * 64:r * 64 + 64, c * 64:c * 64 + 64]
    if check:
        pic = dav1d_ref.decode(bytes(tu), inloop_filters=0)[0]
        assert np.array_equal(np.asarray(pic[0])[:rec.shape[0], :rec.shape[1]], rec), "dav1d differs from the mixed reconstruction"
    share = {bs: float(np.mean(choice == bs)) for bs in SIZES}
    return len(tu), psnr(rec, planes[0], bd), share


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    W, H, bd = (a + [1920, 1080, 8])[:3] if len(a) >= 3 else (1920, 1080, 8)
    qs = a[3:] or [128]
    for q in qs:
        print("== %dx%d %d-bit key frame, q %d" % (W, H, bd, q), flush=True)
        t0 = time.time()
        lay, planes, enc, sse, size, total = run(W, H, bd, q)
        # the all-8x8 rate-distortion curve around q for the equal-size comparison
        curve = []
        for qq in sorted({max(q - 40, 1), max(q - 20, 1), q, min(q + 20, 255), min(q + 40, 255)}):
            if qq == q:
                curve.append((np.log(total[8][0]), total[8][1])); continue
            e = O.intra_encode_frame(planes[0], planes[1], planes[2], bd, 8, qq)
            parts, blocks = frame_blocks(lay, {8: e}, np.full(sse[8].shape, 8))
            curve.append((np.log(len(B.encode(lay, bd, qq, parts, blocks))), psnr(e["rec_y"], planes[0], bd)))
        curve.sort()
        at = lambda nbytes: float(np.interp(np.log(nbytes), [c[0] for c in curve], [c[1] for c in curve]))
        acq = O.ac_q(q, bd)
        for mult in (0.0, 0.0005, 0.001, 0.002, 0.005, 0.01):
            lam = mult * acq * acq / (16.0 if bd == 10 else 1.0) if False else mult * acq * acq
            n, p, share = mixed(lay, planes, enc, sse, size, bd, q, lam, check=(mult == 0.002))
            print("  lambda %.2f q^2: %8d bytes  PSNR-Y %.3f dB  vs all-8x8 at equal size %+.2f dB   tiles 8/16/32/64: %.2f %.2f %.2f %.2f" %
                  (mult, n, p, p - at(n), share[8], share[16], share[32], share[64]), flush=True)
        print("  (%.0f s)" % (time.time() - t0))
