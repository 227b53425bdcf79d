"""Symbol-level reconstruction model for the conformance tests: what a decoder makes of the symbols handed to the host
bitstream writer (8x8 blocks, one 64x64 superblock per tile), assembled in Python from the ORACLE's per-block primitives
(intra prediction, chroma from luma, motion compensation, dequantiser, inverse transforms).  TEST INFRASTRUCTURE ONLY.

It exists so that the tests can hand the writer ARBITRARY symbols — every intra mode with every angle delta, chroma from
luma, random vectors, random transform types — rather than only the encoder's own decisions, and compare dav1d's decode with
the oracle primitives on them (tests/test_av1_conformance.py)."""
import numpy as np

MODE_TO_TXFM = [0, 1, 2, 0, 3, 1, 2, 2, 1, 3, 1, 2, 3, 0]   # Mode_To_Txfm (spec 5.11.47): DCT_DCT 0, ADST_DCT 1, DCT_ADST 2, ADST_ADST 3
TX_4X4, TX_8X8 = 0, 1


def _morton(x, y):
    m = 0
    for i in range(3):
        m |= ((x >> i) & 1) << (2 * i) | ((y >> i) & 1) << (2 * i + 1)
    return m


def zorder_blocks(w8, h8):
    """(r8, c8) of all 8x8 blocks in decoding order: superblocks in raster order, z-order inside"""
    for sr in range(0, h8, 8):
        for sc in range(0, w8, 8):
            for k in range(64):
                bx = sum(((k >> (2 * i)) & 1) << i for i in range(3))
                by = sum(((k >> (2 * i + 1)) & 1) << i for i in range(3))
                if sr + by < h8 and sc + bx < w8:
                    yield sr + by, sc + bx


def _avail(r8, c8, w8, h8):
    """have_top, have_left, have_topright, have_bottomleft of an 8x8 block inside its 64x64 tile (spec 5.11.x BlockDecoded)"""
    by, bx = r8 & 7, c8 & 7
    top, left = by > 0, bx > 0
    tr = top and bx + 1 < 8 and c8 + 1 < w8 and _morton(bx + 1, by - 1) < _morton(bx, by)
    bl = left and by + 1 < 8 and r8 + 1 < h8 and _morton(bx - 1, by + 1) < _morton(bx, by)
    return top, left, tr, bl


def _is_smooth(m):
    return 9 <= m <= 11


def recon_frame(O, w, h, bd, q, y_mode, uv_mode, lev_y, lev_u, lev_v, angle_y=None, angle_uv=None, cfl_alpha=None, tx_type=None,
                is_inter=None, mv=None, skip=None, ref=None):
    """returns [Y, U, V] before the in-loop filters.  Intra blocks need y_mode / uv_mode; inter blocks (is_inter) need mv and
    ref = the reference planes."""
    w8, h8 = w // 8, h // 8
    dt = np.uint8 if bd == 8 else np.uint16
    rec = [np.zeros((h, w), dt), np.zeros((h // 2, w // 2), dt), np.zeros((h // 2, w // 2), dt)]
    dcq, acq = O.dc_q(q, bd), O.ac_q(q, bd)
    nb = w8 * h8
    inter = np.zeros(nb, np.uint8) if is_inter is None else np.asarray(is_inter)
    lev = [np.asarray(lev_y).reshape(nb, 8, 8), np.asarray(lev_u).reshape(nb, 4, 4), np.asarray(lev_v).reshape(nb, 4, 4)]

    def residual(p, b, pred, txt):
        if skip is not None and skip[b]:
            return pred.astype(dt)
        ts = TX_8X8 if p == 0 else TX_4X4
        return O.inv_txfm2d_add(O.dequantize(lev[p][b], dcq, acq, 0, bd), pred.astype(dt), ts, txt, bd)

    for r8, c8 in zorder_blocks(w8, h8):
        b = r8 * w8 + c8
        ltx = 0 if tx_type is None else int(tx_type[b])
        if inter[b]:
            mx, my = int(mv[2 * b]), int(mv[2 * b + 1])
            rec[0][r8 * 8:r8 * 8 + 8, c8 * 8:c8 * 8 + 8] = residual(0, b, O.mc_block(ref[0], bd, c8 * 8, r8 * 8, 8, 8, mx * 2, my * 2), ltx)
            for p in (1, 2):
                rec[p][r8 * 4:r8 * 4 + 4, c8 * 4:c8 * 4 + 4] = residual(p, b, O.mc_block(ref[p], bd, c8 * 4, r8 * 4, 4, 4, mx, my), ltx)
            continue
        top, left, tr, bl = _avail(r8, c8, w8, h8)
        # get_filter_type: a smooth-predicted intra neighbour inside the tile (inter neighbours do not count)
        nbrs = ([b - w8] if top else []) + ([b - 1] if left else [])
        fy = int(any(not inter[n] and _is_smooth(y_mode[n]) for n in nbrs))
        fc = int(any(not inter[n] and _is_smooth(uv_mode[n]) for n in nbrs))
        ym, da = int(y_mode[b]), 0 if angle_y is None else int(angle_y[b])
        pred = O.intra_predict(rec[0], c8 * 8, r8 * 8, 8, 8, ym, da, bd, 8 * top, 8 * tr, 8 * left, 8 * bl, 0, fy)
        rec[0][r8 * 8:r8 * 8 + 8, c8 * 8:c8 * 8 + 8] = residual(0, b, pred, ltx)
        um, du = int(uv_mode[b]), 0 if angle_uv is None else int(angle_uv[b])
        for p in (1, 2):
            if um == 13:    # chroma from luma: DC prediction corrected by the reconstructed luma's AC
                dc = O.intra_predict(rec[p], c8 * 4, r8 * 4, 4, 4, 0, 0, bd, 4 * top, 4 * tr, 4 * left, 4 * bl, 0, fc)
                tmp = rec[p].copy()
                tmp[r8 * 4:r8 * 4 + 4, c8 * 4:c8 * 4 + 4] = dc
                alpha = int(cfl_alpha[2 * b + p - 1])
                pred = O.cfl_predict(rec[0], tmp, bd, c8 * 4, r8 * 4, 4, 4, alpha, c8 * 8 + 8, r8 * 8 + 8)[r8 * 4:r8 * 4 + 4, c8 * 4:c8 * 4 + 4]
            else:
                pred = O.intra_predict(rec[p], c8 * 4, r8 * 4, 4, 4, um, du, bd, 4 * top, 4 * tr, 4 * left, 4 * bl, 0, fc)
            rec[p][r8 * 4:r8 * 4 + 4, c8 * 4:c8 * 4 + 4] = residual(p, b, pred, MODE_TO_TXFM[um])
    return rec
