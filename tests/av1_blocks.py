"""General block structures for the conformance tests (TEST INFRASTRUCTURE ONLY): random partition trees over every AV1 block size
4x4..64x64, blocks with arbitrary symbols, and the decoder-side model that applies the ORACLE's per-block primitives (intra
prediction incl. edge preparation, chroma from luma, motion compensation, dequantiser, inverse transforms) to them in decoding
order — what dav1d must reproduce from the stream av1-go_amd/host/av1_blockstream.cpp writes (tests/test_av1_blocks.py).

The structure follows the specification's decoding process: decode_partition / decode_block (5.11.4-5), residual / transform_block
(5.11.34-35: prediction per TRANSFORM block for intra blocks), the BlockDecoded flags behind haveAboveRt / haveBelowLft (5.11.3),
compute_tx_type (5.11.40), get_filter_type (7.11.2)."""
import numpy as np

import av1stream

BW4 = [1, 1, 2, 2, 2, 4, 4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 1, 4, 2, 8, 4, 16]
BH4 = [1, 2, 1, 2, 4, 2, 4, 8, 4, 8, 16, 8, 16, 32, 16, 32, 4, 1, 8, 2, 16, 4]
BLOCK_NAMES = ["4x4", "4x8", "8x4", "8x8", "8x16", "16x8", "16x16", "16x32", "32x16", "32x32", "32x64", "64x32", "64x64", "64x128", "128x64",
               "128x128", "4x16", "16x4", "8x32", "32x8", "16x64", "64x16"]
BLOCK_8X8, BLOCK_64X64 = 3, 12
TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
TX_SQR = [0, 1, 2, 3, 4, 0, 0, 1, 1, 2, 2, 3, 3, 0, 0, 1, 1, 2, 2]
TX_SQR_UP = [0, 1, 2, 3, 4, 1, 1, 2, 2, 3, 3, 4, 4, 2, 2, 3, 3, 4, 4]
SPLIT_TX = [0, 0, 1, 2, 3, 0, 0, 1, 1, 2, 2, 3, 3, 5, 6, 7, 8, 9, 10]
MAX_TX_DEPTH = [0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 4, 4, 2, 2, 3, 3, 4, 4]
SUBSAMPLED = [0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 1, 2, 16, 17, 18, 19]
MODE_TO_TXFM = [0, 1, 2, 0, 3, 1, 2, 2, 1, 3, 1, 2, 3, 0]
P_NONE, P_HORZ, P_VERT, P_SPLIT, P_HORZ_A, P_HORZ_B, P_VERT_A, P_VERT_B, P_HORZ_4, P_VERT_4 = range(10)
# transform types of each set (get_tx_set): 0 DCT only, 1 INTRA_1, 2 INTRA_2, 3 INTER_1, 4 INTER_2, 5 INTER_3
TX_SETS = {0: [0], 1: [9, 0, 10, 11, 3, 1, 2], 2: [9, 0, 3, 1, 2], 3: list(range(16)), 4: [9, 10, 11, 0, 1, 2, 4, 5, 3, 6, 7, 8], 5: [9, 0]}


def bsize_of(w4, h4):
    for b in range(22):
        if BW4[b] == w4 and BH4[b] == h4:
            return b
    raise ValueError((w4, h4))


def tx_of(w, h):
    for t in range(19):
        if TX_W[t] == w and TX_H[t] == h:
            return t
    raise ValueError((w, h))


def max_tx_rect(bsize):
    return tx_of(min(BW4[bsize] * 4, 64), min(BH4[bsize] * 4, 64))


def tx_set_of(tx, is_inter, reduced):
    sqr, up = TX_SQR[tx], TX_SQR_UP[tx]
    if up > 3:
        return 0
    if is_inter:
        return 5 if (reduced or up == 3) else 4 if sqr == 2 else 3
    if up == 3:
        return 0
    return 2 if (reduced or sqr == 2) else 1


def tx_scale(tx):
    n = TX_W[tx] * TX_H[tx]
    return int(n > 256) + int(n > 1024)


# ---------------------------------------------------------------------------------------------------------------------------------
class Layout:
    """frame geometry + uniform tiles as host/av1_bitstream_core.hpp frame_info() derives them (tile_cols_log2 / tile_rows_log2 = -1:
    one 64x64 superblock per tile)"""

    def __init__(self, w, h, tile_cols_log2=-1, tile_rows_log2=-1):
        assert w % 8 == 0 and h % 8 == 0
        self.w, self.h = w, h
        self.mi_cols, self.mi_rows = w // 4, h // 4
        self.sb_cols, self.sb_rows = (self.mi_cols + 15) >> 4, (self.mi_rows + 15) >> 4
        tl = lambda blk, target: next(k for k in range(32) if (blk << k) >= target)
        max_c, max_r, min_c = tl(1, min(self.sb_cols, 64)), tl(1, min(self.sb_rows, 64)), tl(64, self.sb_cols)
        cl = max_c if tile_cols_log2 < 0 else min(max(tile_cols_log2, min_c), max_c)
        self.tile_w_sb = (self.sb_cols + (1 << cl) - 1) >> cl
        self.tile_cols = (self.sb_cols + self.tile_w_sb - 1) // self.tile_w_sb
        min_tiles = max(min_c, tl(2304, self.sb_rows * self.sb_cols))
        min_r = max(min_tiles - tl(1, self.tile_cols), 0)
        rl = max_r if tile_rows_log2 < 0 else min(max(tile_rows_log2, min_r), max_r)
        self.tile_h_sb = (self.sb_rows + (1 << rl) - 1) >> rl
        self.tile_rows = (self.sb_rows + self.tile_h_sb - 1) // self.tile_h_sb
        self.args = dict(tile_cols_log2=tile_cols_log2, tile_rows_log2=tile_rows_log2)

    def tiles(self):
        for tr in range(self.tile_rows):
            for tc in range(self.tile_cols):
                r0, c0 = tr * self.tile_h_sb * 16, tc * self.tile_w_sb * 16
                yield r0, min(r0 + self.tile_h_sb * 16, self.mi_rows), c0, min(c0 + self.tile_w_sb * 16, self.mi_cols)


def build_tree(lay, choose):
    """walk every tile's superblocks like decode_partition does; choose(r, c, bsize, allowed) -> partition type.  Returns
    (partition symbols, blocks as (r, c, bsize, tile bounds)) in decoding order."""
    parts, blocks = [], []

    def node(r, c, bsize, tb):
        if r >= lay.mi_rows or c >= lay.mi_cols:
            return
        n4 = BW4[bsize]
        half, quarter = n4 >> 1, n4 >> 2
        has_rows, has_cols = r + half < lay.mi_rows, c + half < lay.mi_cols
        if has_rows and has_cols:
            allowed = [P_NONE, P_HORZ, P_VERT, P_SPLIT] + ([] if bsize == BLOCK_8X8 else [P_HORZ_A, P_HORZ_B, P_VERT_A, P_VERT_B, P_HORZ_4, P_VERT_4])
        elif has_cols:
            allowed = [P_HORZ, P_SPLIT]
        elif has_rows:
            allowed = [P_VERT, P_SPLIT]
        else:
            allowed = [P_SPLIT]
        p = choose(r, c, bsize, allowed)
        assert p in allowed
        parts.append(p)
        blk = lambda rr, cc, w4, h4: blocks.append((rr, cc, bsize_of(w4, h4), tb))
        sq = None if bsize == BLOCK_8X8 else bsize_of(half, half)
        if p == P_NONE:
            blk(r, c, n4, n4)
        elif p == P_HORZ:
            blk(r, c, n4, half)
            if has_rows:
                blk(r + half, c, n4, half)
        elif p == P_VERT:
            blk(r, c, half, n4)
            if has_cols:
                blk(r, c + half, half, n4)
        elif p == P_SPLIT:
            if bsize == BLOCK_8X8:
                for dr, dc in ((0, 0), (0, 1), (1, 0), (1, 1)):
                    blk(r + dr, c + dc, 1, 1)
            else:
                for dr, dc in ((0, 0), (0, half), (half, 0), (half, half)):
                    node(r + dr, c + dc, sq, tb)
        elif p == P_HORZ_A:
            blk(r, c, half, half); blk(r, c + half, half, half); blk(r + half, c, n4, half)
        elif p == P_HORZ_B:
            blk(r, c, n4, half); blk(r + half, c, half, half); blk(r + half, c + half, half, half)
        elif p == P_VERT_A:
            blk(r, c, half, half); blk(r + half, c, half, half); blk(r, c + half, half, n4)
        elif p == P_VERT_B:
            blk(r, c, half, n4); blk(r, c + half, half, half); blk(r + half, c + half, half, half)
        elif p == P_HORZ_4:
            for i in range(4):
                if i < 3 or r + quarter * 3 < lay.mi_rows:
                    blk(r + quarter * i, c, n4, quarter)
        else:
            for i in range(4):
                if i < 3 or c + quarter * 3 < lay.mi_cols:
                    blk(r, c + quarter * i, quarter, n4)

    for tb in lay.tiles():
        for r in range(tb[0], tb[1], 16):
            for c in range(tb[2], tb[3], 16):
                node(r, c, BLOCK_64X64, tb)
    return parts, blocks


def uniform_chooser(target):
    """partition choices that cut every superblock into blocks of BLOCK_* `target` wherever the frame edge allows"""
    tw, th = BW4[target], BH4[target]

    def choose(r, c, bsize, allowed):
        n4 = BW4[bsize]
        want = None
        if (tw, th) == (n4, n4):
            want = P_NONE
        elif (tw, th) == (n4, n4 // 2):
            want = P_HORZ
        elif (tw, th) == (n4 // 2, n4):
            want = P_VERT
        elif (tw, th) == (n4, n4 // 4) and bsize != BLOCK_8X8:
            want = P_HORZ_4
        elif (tw, th) == (n4 // 4, n4) and bsize != BLOCK_8X8:
            want = P_VERT_4
        if want is None or want not in allowed:
            want = P_SPLIT if P_SPLIT in allowed else allowed[0]
        return want
    return choose


def random_chooser(rng, p_split=None):
    def choose(r, c, bsize, allowed):
        ps = p_split if p_split is not None else {12: 0.7, 9: 0.55, 6: 0.4}.get(bsize, 0.25)      # large blocks split more often
        if P_SPLIT in allowed and rng.random() < ps:
            return P_SPLIT
        return int(rng.choice(allowed))
    return choose


# ---------------------------------------------------------------------------------------------------------------------------------
def tx_blocks(lay, r, c, bsize, plane, tx, is_inter):
    """(x4, y4, tx) of the plane's transform blocks in coding order (positions in units of 4 samples OF THE PLANE), those that
    start outside the frame left out: residual / transform_tree / transform_block (5.11.34-36)"""
    ss = 1 if plane else 0
    pbs = SUBSAMPLED[bsize] if plane else bsize
    n4w, n4h = BW4[pbs], BH4[pbs]
    bx, by = c >> ss, r >> ss
    mx, my = (lay.mi_cols + ss) >> ss, (lay.mi_rows + ss) >> ss
    out = []
    if plane == 0 and is_inter:
        lw, lh = TX_W[tx] // 4, TX_H[tx] // 4

        def tree(x, y, w4, h4):
            if x >= mx or y >= my:
                return
            if w4 <= lw and h4 <= lh:
                out.append((x, y, tx_of(w4 * 4, h4 * 4)))
            elif w4 > h4:
                tree(x, y, w4 // 2, h4); tree(x + w4 // 2, y, w4 // 2, h4)
            elif w4 < h4:
                tree(x, y, w4, h4 // 2); tree(x, y + h4 // 2, w4, h4 // 2)
            else:
                for dy, dx in ((0, 0), (0, w4 // 2), (h4 // 2, 0), (h4 // 2, w4 // 2)):
                    tree(x + dx, y + dy, w4 // 2, h4 // 2)
        tree(bx, by, n4w, n4h)
        return out
    ptx = max_tx_rect(pbs) if plane else tx
    for y in range(0, n4h, TX_H[ptx] // 4):
        for x in range(0, n4w, TX_W[ptx] // 4):
            if bx + x < mx and by + y < my:
                out.append((bx + x, by + y, ptx))
    return out


def has_chroma(r, c, bsize):
    return ((c & 1) or not (BW4[bsize] & 1)) and ((r & 1) or not (BH4[bsize] & 1))


def random_levels(O, rng, tx, tx_type, bd, q, p_zero=0.15):
    """levels of one transform block from a real residual (forward transform + quantiser): the stream stays inside the conformance
    limits on intermediate values"""
    w, h = TX_W[tx], TX_H[tx]
    if p_zero > 0:
        amp = int(rng.choice([0, 2, 8, 40, (1 << bd) - 1], p=[p_zero, 0.3, 0.55 - p_zero, 0.1, 0.05]))
    else:      # a block that certainly codes coefficients
        amp = int(rng.choice([40, 120, (1 << bd) - 1], p=[0.5, 0.35, 0.15]))
    shape = (min(h, 32), min(w, 32))
    if amp == 0:
        return np.zeros(shape, np.int16)
    res = rng.integers(-amp, amp + 1, (h, w)).astype(np.int32)
    if rng.random() < 0.5:
        res = (res * np.linspace(1, 0, w)[None, :]).astype(np.int32) + int(rng.integers(-amp, amp + 1)) // 2
    if rng.random() < 0.3:
        res = (res * np.linspace(1, 0.2, h)[:, None]).astype(np.int32)
    res = np.clip(res, -(1 << bd) + 1, (1 << bd) - 1).astype(np.int16)
    lev = O.quantize(O.fwd_txfm2d(res, tx, int(tx_type), bd), O.dc_q(q, bd), O.ac_q(q, bd), tx_scale(tx))[0]
    return lev.reshape(shape)


def random_frame(O, rng, lay, bd, q, key=True, chooser=None, tx_mode_select=0, reduced_tx_set=0, p_skip=0.15, p_inter=0.0, filters=(0,),
                 mv_range=40, one_d_types=True, hp=0, p_zero=0.15, symbols=None, cycle_types=None):
    """random symbols for every block of a random (or given) partition tree.  Inter blocks are kept ISOLATED (no other inter block of
    the tile within the MV prediction reach: the block writer's restriction).  Returns (partition list, list of block dicts)."""
    parts, tree = build_tree(lay, chooser or random_chooser(rng))
    blocks = []
    inter_at = []      # (tile bounds, r, c, bw4, bh4) of the inter blocks so far
    for r, c, bsize, tb in tree:
        bw4, bh4 = BW4[bsize], BH4[bsize]
        b = dict(r=r, c=c, bsize=bsize, tile=tb, skip=int(rng.random() < p_skip), is_inter=0, y_mode=int(rng.integers(0, 13)),
                 uv_mode=int(rng.integers(0, 14 if max(bw4, bh4) <= 8 else 13)), angle_y=int(rng.integers(-3, 4)), angle_uv=int(rng.integers(-3, 4)),
                 cfl=(0, 0), tx_depth=0, filt=0, mv=(0, 0))
        if not key and rng.random() < p_inter:
            far = all(tb != t or r + bh4 + 1 <= rr - 6 or rr + hh + 1 <= r - 6 or c + bw4 + 2 <= cc - 6 or cc + ww + 2 <= c - 6 for t, rr, cc, ww, hh in inter_at)
            if far:
                b["is_inter"] = 1
                b["filt"] = int(rng.choice(filters))
                step = 1 if hp else 2      # allow_high_precision_mv: eighth-sample vectors
                b["mv"] = (int(rng.integers(-mv_range, mv_range + 1)) * step, int(rng.integers(-mv_range, mv_range + 1)) * step)
                if rng.random() < 0.15:
                    b["mv"] = (int(rng.integers(-600, 601)) * step, int(rng.integers(-600, 601)) * step)      # far outside the picture
                inter_at.append((tb, r, c, bw4, bh4))
        if symbols is not None:
            symbols(len(blocks), b)      # the test's own choices (modes, angles ...), before anything is derived from them
        if b["uv_mode"] == 13:
            a = (int(rng.integers(-16, 17)), int(rng.integers(-16, 17)))
            b["cfl"] = a if a != (0, 0) else (5, 0)
        tx = max_tx_rect(bsize)
        if tx_mode_select and bsize > 0 and not (b["is_inter"] and b["skip"]):
            nsym = 3 if MAX_TX_DEPTH[bsize] > 1 else 2
            depth = int(rng.integers(0, nsym))
            if b["is_inter"]:      # txfm_split stops at 4x4 and at depth 2
                t, dmax = tx, 0
                while t != 0 and dmax < 2:
                    t, dmax = SPLIT_TX[t], dmax + 1
                depth = min(depth, dmax)
            b["tx_depth"] = depth
            for _ in range(depth):
                tx = SPLIT_TX[tx]
        b["tx"] = tx
        b["tx_types"], b["levels"] = [], [[], [], []]
        if not b["skip"]:
            for p in range(3 if has_chroma(r, c, bsize) else 1):
                for x4, y4, t in tx_blocks(lay, r, c, bsize, p, tx, b["is_inter"]):
                    if p == 0:
                        types = TX_SETS[tx_set_of(t, b["is_inter"], reduced_tx_set)]
                        if not one_d_types:
                            types = [k for k in types if k < 9] or [0]
                        if cycle_types is not None:      # every type of the set in turn, per transform size (a dict the caller keeps)
                            n = cycle_types.get((t, b["is_inter"]), 0)
                            cycle_types[(t, b["is_inter"])] = n + 1
                            ty = types[n % len(types)]
                        else:
                            ty = int(rng.choice(types))
                        b["tx_types"].append(ty)
                    else:
                        ty = 0      # (the levels only need to be plausible; the model derives the real chroma type)
                    b["levels"][p].append(random_levels(O, rng, t, ty if O.txfm_valid(t, ty) else 0, bd, q, p_zero))
        blocks.append(b)
    return parts, blocks


def encode(lay, bd, q, parts, blocks, key=True, **kw):
    """block dicts -> the arrays of av1mi_obu_blocks -> one temporal unit"""
    arr = np.zeros(len(blocks), av1stream.BLOCK_DTYPE)
    types, levels, nlev = [], [], 0
    for i, b in enumerate(blocks):
        a = arr[i]
        a["mi_row"], a["mi_col"], a["bsize"], a["skip"], a["is_inter"] = b["r"], b["c"], b["bsize"], b["skip"], b["is_inter"]
        a["y_mode"], a["uv_mode"], a["angle_y"], a["angle_uv"] = b["y_mode"], b["uv_mode"], b["angle_y"], b["angle_uv"]
        a["cfl_alpha_u"], a["cfl_alpha_v"], a["tx_depth"], a["interp_filter"] = b["cfl"][0], b["cfl"][1], b["tx_depth"], b["filt"]
        a["mv_x"], a["mv_y"] = b["mv"]
        a["tx_type_off"] = len(types)
        types += b["tx_types"]
        for p in range(3):
            a["lev_off"][p] = nlev
            for l in b["levels"][p]:
                levels.append(l.reshape(-1))
                nlev += l.size
    lev = np.concatenate(levels) if levels else np.zeros(1, np.int16)
    return av1stream.blocks_temporal_unit(lay.w, lay.h, bd, q, parts, arr, np.array(types + [0], np.uint8), lev, frame_type=0 if key else 1, **lay.args, **kw)


# ---------------------------------------------------------------------------------------------------------------------------------
class Decoder:
    """the decoding process on block dicts, built from the oracle's primitives; planes are padded to whole superblocks"""

    def __init__(self, O, lay, bd, q, reduced_tx_set=0, interp_filter=0, ref=None):
        self.O, self.lay, self.bd, self.q, self.reduced, self.frame_filter = O, lay, bd, q, reduced_tx_set, interp_filter
        dt = np.uint8 if bd == 8 else np.uint16
        ph, pw = lay.sb_rows * 64, lay.sb_cols * 64
        self.rec = [np.zeros((ph, pw), dt), np.zeros((ph // 2, pw // 2), dt), np.zeros((ph // 2, pw // 2), dt)]
        self.ref = ref
        self.dcq, self.acq = O.dc_q(q, bd), O.ac_q(q, bd)
        n = (lay.mi_rows + 32, lay.mi_cols + 32)
        self.y_mode = np.zeros(n, np.uint8); self.uv_mode = np.zeros(n, np.uint8); self.is_inter = np.zeros(n, np.uint8)
        self.txtype = np.zeros(n, np.uint8)
        self.sb = None

    # ---- 5.11.3 clear_block_decoded_flags
    def new_superblock(self, r, c, tb):
        self.sb = (r, c)
        self.dec = []
        for p in range(3):
            ss = 1 if p else 0
            n = 16 >> ss
            w4, h4 = (tb[3] - c) >> ss, (tb[1] - r) >> ss
            d = np.zeros((n + 3, n + 3), np.uint8)      # index + 1
            for y in range(-1, n + 1):
                for x in range(-1, n + 1):
                    if y < 0 and x < w4:
                        d[y + 1, x + 1] = 1
                    elif x < 0 and y < h4:
                        d[y + 1, x + 1] = 1
            d[n + 1, 0] = 0
            self.dec.append(d)

    def decoded(self, p, y4, x4):
        ss = 1 if p else 0
        return int(self.dec[p][y4 - (self.sb[0] >> ss) + 1, x4 - (self.sb[1] >> ss) + 1])

    def block(self, b):
        O, lay, bd = self.O, self.lay, self.bd
        r, c, bsize, tb = b["r"], b["c"], b["bsize"], b["tile"]
        if self.sb != (r & ~15, c & ~15):
            self.new_superblock(r & ~15, c & ~15, tb)
        bw4, bh4 = BW4[bsize], BH4[bsize]
        inside = lambda rr, cc: tb[0] <= rr < tb[1] and tb[2] <= cc < tb[3]
        avail_u, avail_l = inside(r - 1, c), inside(r, c - 1)
        chroma = has_chroma(r, c, bsize)
        avail_uc = inside(r - 2, c) if (chroma and bh4 == 1) else avail_u
        avail_lc = inside(r, c - 2) if (chroma and bw4 == 1) else avail_l
        inter = b["is_inter"]
        r1, c1 = min(r + bh4, lay.mi_rows), min(c + bw4, lay.mi_cols)
        self.is_inter[r:r1, c:c1] = inter
        self.y_mode[r:r1, c:c1] = 255 if inter else b["y_mode"]      # (an inter block's mode is never a smooth one)
        if chroma:
            self.uv_mode[r:r1, c:c1] = 255 if inter else b["uv_mode"]
        self.txtype[r:r1, c:c1] = 0
        planes = 3 if chroma else 1
        # ---- prediction of inter blocks (7.11.3): the whole block per plane; every other block the chroma area covers is intra here, so
        # chroma is predicted as one block with this block's vector
        if inter:
            filt = b["filt"] if self.frame_filter == 4 else self.frame_filter
            for p in range(planes):
                ss = 1 if p else 0
                pbs = SUBSAMPLED[bsize] if p else bsize
                x, y, w, h = (c >> ss) * 4, (r >> ss) * 4, BW4[pbs] * 4, BH4[pbs] * 4
                ref = self.ref[p]
                mvx, mvy = (b["mv"][0] * 2, b["mv"][1] * 2) if p == 0 else b["mv"]
                pred = O.mc_block(ref, bd, x, y, w, h, mvx, mvy, filt, filt)
                self.rec[p][y:y + h, x:x + w] = pred.astype(self.rec[p].dtype)
        max_luma = [c * 4, r * 4]
        for p in range(planes):
            ss = 1 if p else 0
            pbs = SUBSAMPLED[bsize] if p else bsize
            base_x4, base_y4 = c >> ss, r >> ss
            plane_w, plane_h = (lay.mi_cols * 4) >> ss, (lay.mi_rows * 4) >> ss
            tbs = tx_blocks(lay, r, c, bsize, p, b["tx"], inter)
            for k, (x4, y4, tx) in enumerate(tbs):
                tw, th = TX_W[tx], TX_H[tx]
                x, y = x4 * 4, y4 * 4
                if not inter:
                    mode = b["y_mode"] if p == 0 else b["uv_mode"]
                    is_cfl = p > 0 and mode == 13
                    have_left = (avail_lc if p else avail_l) if x4 == base_x4 else True
                    have_above = (avail_uc if p else avail_u) if y4 == base_y4 else True
                    have_ar = self.decoded(p, y4 - 1, x4 + tw // 4)
                    have_bl = self.decoded(p, y4 + th // 4, x4 - 1)
                    n_top = min(tw, plane_w - x) if have_above else 0
                    n_tr = min(tw, max(plane_w - (x + tw), 0)) if (have_above and have_ar) else 0
                    n_left = min(th, plane_h - y) if have_left else 0
                    n_bl = min(th, max(plane_h - (y + th), 0)) if (have_left and have_bl) else 0
                    ftype = self.filter_type(p, r, c, avail_uc if p else avail_u, avail_lc if p else avail_l)
                    angle = (b["angle_y"] if p == 0 else b["angle_uv"]) if bsize >= BLOCK_8X8 else 0
                    pred = O.intra_predict(self.rec[p], x, y, tw, th, 0 if is_cfl else mode, 0 if is_cfl else angle, bd, n_top, n_tr, n_left, n_bl, 0, ftype)
                    self.rec[p][y:y + th, x:x + tw] = pred.astype(self.rec[p].dtype)
                    if is_cfl:
                        self.rec[p] = O.cfl_predict(self.rec[0], self.rec[p], bd, x, y, tw, th, b["cfl"][p - 1], max_luma[0], max_luma[1])
                    if p == 0:
                        max_luma = [x + tw, y + th]      # MaxLumaW / MaxLumaH: the last luma transform block's far corner
                if not b["skip"]:
                    lev = b["levels"][p][k]
                    if lev.any():
                        if p == 0:
                            ty = b["tx_types"][k]
                            self.txtype[y4:min(y4 + th // 4, lay.mi_rows), x4:min(x4 + tw // 4, lay.mi_cols)] = ty
                        else:
                            ty = self.chroma_type(b, tx, x4, y4)
                        dq = O.dequantize(lev, self.dcq, self.acq, tx_scale(tx), bd)
                        self.rec[p][y:y + th, x:x + tw] = O.inv_txfm2d_add(dq, self.rec[p][y:y + th, x:x + tw], tx, ty, bd)
                d = self.dec[p]
                oy, ox = y4 - (self.sb[0] >> ss) + 1, x4 - (self.sb[1] >> ss) + 1
                d[oy:oy + th // 4, ox:ox + tw // 4] = 1
    def chroma_type(self, b, tx, x4, y4):      # compute_tx_type (5.11.40) for a chroma transform block
        if TX_SQR_UP[tx] > 3:
            return 0
        if b["is_inter"]:
            ty = int(self.txtype[max(b["r"], y4 << 1), max(b["c"], x4 << 1)])
        else:
            ty = MODE_TO_TXFM[b["uv_mode"]]
        return ty if ty in TX_SETS[tx_set_of(tx, b["is_inter"], self.reduced)] else 0

    def filter_type(self, p, r, c, avail_u, avail_l):      # get_filter_type (7.11.2): a smooth-predicted neighbour
        def smooth(rr, cc):
            if p == 0:
                return 9 <= self.y_mode[rr, cc] <= 11
            return (not self.is_inter[rr, cc]) and 9 <= self.uv_mode[rr, cc] <= 11
        a = l = False
        if avail_u:
            rr, cc = r - 1, c
            if p:
                if not (c & 1):
                    cc += 1
                if r & 1:
                    rr -= 1
            a = smooth(rr, cc)
        if avail_l:
            rr, cc = r, c - 1
            if p:
                if c & 1:
                    cc -= 1
                if not (r & 1):
                    rr += 1
            l = smooth(rr, cc)
        return int(a or l)

    def planes(self):
        h, w = self.lay.h, self.lay.w
        return [self.rec[0][:h, :w], self.rec[1][:h // 2, :w // 2], self.rec[2][:h // 2, :w // 2]]


def decode(O, lay, bd, q, blocks, reduced_tx_set=0, interp_filter=0, ref=None):
    d = Decoder(O, lay, bd, q, reduced_tx_set, interp_filter, ref)
    for b in blocks:
        d.block(b)
    return d.planes()


# ---------------------------------------------------------------------------------------------------------------------------------
def loopfilter_maps(O, lay, blocks, lf_level):
    """what deblocking (7.14.2) and CDEF (7.15) read of the mode info, per plane, from block dicts: the oracle's mode-info words per
    4x4 unit of each plane (transform size of the unit's transform block = LoopfilterTxSizes, the plane's filter levels, skip &&
    is_inter, prediction-block edges) and the per-8x8 skip flags of CDEF (all four 4x4 units skipped)"""
    mi = [np.zeros(((lay.mi_rows + s) >> s, (lay.mi_cols + s) >> s), np.uint32) for s in (0, 1, 1)]
    skip4 = np.zeros((lay.mi_rows, lay.mi_cols), np.uint8)
    for b in blocks:
        r, c, bsize = b["r"], b["c"], b["bsize"]
        skip4[r:r + BH4[bsize], c:c + BW4[bsize]] = b["skip"]
        for p in range(3 if has_chroma(r, c, bsize) else 1):
            ss = 1 if p else 0
            pbs = SUBSAMPLED[bsize] if p else bsize
            lv = (lf_level[0], lf_level[1]) if p == 0 else (lf_level[1 + p], lf_level[1 + p])
            for x4, y4, tx in tx_blocks(lay, r, c, bsize, p, b["tx"], b["is_inter"]):
                for y in range(y4, min(y4 + TX_H[tx] // 4, mi[p].shape[0])):
                    for x in range(x4, min(x4 + TX_W[tx] // 4, mi[p].shape[1])):
                        mi[p][y, x] = O.lf_mi(int(np.log2(TX_W[tx])), int(np.log2(TX_H[tx])), lv[0], lv[1], int(b["skip"] and b["is_inter"]),
                                              int((x - (c >> ss)) % BW4[pbs] == 0), int((y - (r >> ss)) % BH4[pbs] == 0))
    skip8 = (skip4.reshape(lay.mi_rows // 2, 2, lay.mi_cols // 2, 2).min(axis=(1, 3))).astype(np.uint8)
    return mi, skip8
