// av1_ops.hpp — the AV1 tile syntax of this encoder's tool set as an OP STREAM, shared by the GPU tile entropy coder
// (csrc/av1_entropy_kernels.hip) and its CPU twin (host/av1_opstream.cpp, tests): the same source compiled by hipcc for the
// device and by g++ for the host, so the logic is verified on the CPU byte for byte against the host bitstream writer
// (host/av1_bitstream.cpp, itself verified by dav1d) before it ever runs on a GPU.
//
// Why an op stream.  A range coder is serial in its state, but WHICH symbols a tile codes, and with which CDF, depends only on
// symbol values of the block and of its neighbours (AV1 spec 9.3: every context is a function of already-decoded syntax
// elements, never of the coder's state).  So the syntax splits into
//   tokenize   one thread per 8x8 block, all blocks in parallel: the block's syntax elements in decoding order as 16-bit ops
//              ("adaptive symbol s with CDF slot c" / "n literal bits"), with every context resolved from neighbour data;
//   code       one lane per TILE: a uniform loop over the tile's op list — CDF lookup, interval update, adaptation, byte output.
// Tool set = what the GPU block pipeline produces (host/av1_bitstream.hpp lists what the host writer can code beyond it): 8x8
// blocks, one 64x64 superblock per tile, TX_MODE_LARGEST, DCT_DCT luma, key frames with 13 intra modes (angle delta 0) or inter
// frames with single-reference blocks, cdef_bits = 0, Wiener restoration on 64x64 units, reduced_tx_set = 0, CDF update on.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AV1_HD __host__ __device__ inline
#else
#define AV1_HD inline
#endif

namespace av1ops {

// ------------------------------------------------------------------------------------------------ CDF slots of a tile
// Every adaptive CDF a tile of this tool set can touch has a slot.  A slot stores the N - 1 inverse-CDF values (32768 - cdf)
// followed by the adaptation counter, padded to a multiple of four uint16 (N <= 4: one 64-bit word).
enum Slot : int {
  S_SKIP = 0,                 // [3]
  S_PART8 = S_SKIP + 3,       // context 0 only (neighbours are 8x8 too)
  S_PART16 = S_PART8 + 1,     // [4]
  S_PART32 = S_PART16 + 4,    // [4]
  S_PART64 = S_PART32 + 4,    // [4]
  S_USE_WIENER = S_PART64 + 4,
  S_TXB_SKIP_Y = S_USE_WIENER + 1,      // all_zero, luma: context 0 (the transform covers the block)
  S_TXB_SKIP_C = S_TXB_SKIP_Y + 1,      // [3] chroma: contexts 7, 8, 9
  S_EOB64_Y = S_TXB_SKIP_C + 3,
  S_EOB16_C = S_EOB64_Y + 1,
  S_EOBX_Y = S_EOB16_C + 1,             // [5] eob_extra, eob_pt - 3 = 0..4
  S_EOBX_C = S_EOBX_Y + 5,              // [3]
  S_DC_SIGN_Y = S_EOBX_C + 3,           // [3]
  S_DC_SIGN_C = S_DC_SIGN_Y + 3,        // [3]
  S_BASE_EOB_Y = S_DC_SIGN_C + 3,       // [4]
  S_BASE_EOB_C = S_BASE_EOB_Y + 4,      // [4]
  S_BASE_Y = S_BASE_EOB_C + 4,          // [26] the 2-D class contexts
  S_BASE_C = S_BASE_Y + 26,             // [26]
  S_BR_Y = S_BASE_C + 26,               // [21]
  S_BR_C = S_BR_Y + 21,                 // [21]
  S_COMMON_END = S_BR_C + 21,
  // key frames
  S_KF_Y_MODE = S_COMMON_END,           // [5][5]
  S_UV_MODE = S_KF_Y_MODE + 25,         // [13] chroma-from-luma allowed
  S_ANGLE = S_UV_MODE + 13,             // [8]
  S_INTRA_TX = S_ANGLE + 8,             // [13] set 1, 8x8
  S_KEY_END = S_INTRA_TX + 13,
  // inter frames (they reuse the slot numbers after S_COMMON_END)
  S_IS_INTER = S_COMMON_END,            // [4]
  S_SINGLE_REF = S_IS_INTER + 4,        // [2 contexts: 1, 2][p1, p3, p4]
  S_NEW_MV = S_SINGLE_REF + 6,          // [6]
  S_ZERO_MV = S_NEW_MV + 6,             // context 0
  S_REF_MV = S_ZERO_MV + 1,             // [6]
  S_DRL = S_REF_MV + 6,                 // [3]
  S_MV_JOINT = S_DRL + 3,
  S_MV_COMP = S_MV_JOINT + 1,           // [2 components] x 16: class, class0, class0_fr[2], sign, bits[10], fr
  S_INTER_TX = S_MV_COMP + 32,          // set 1, 8x8
  S_INTER_END = S_INTER_TX + 1,
  S_MAX = S_KEY_END > S_INTER_END ? S_KEY_END : S_INTER_END
};
enum { MVC_CLASS = 0, MVC_CLASS0 = 1, MVC_CLASS0_FR = 2, MVC_SIGN = 4, MVC_BITS = 5, MVC_FR = 15 };

// alphabet size of a slot
AV1_HD int slot_nsym(int s, bool key) {
  if (s < S_PART8) return 2;
  if (s < S_PART16) return 4;
  if (s < S_USE_WIENER) return 10;
  if (s < S_EOB64_Y) return 2;
  if (s == S_EOB64_Y) return 7;
  if (s == S_EOB16_C) return 5;
  if (s < S_BASE_EOB_Y) return 2;
  if (s < S_BASE_Y) return 3;
  if (s < S_COMMON_END) return 4;
  if (key) {
    if (s < S_UV_MODE) return 13;
    if (s < S_ANGLE) return 14;
    if (s < S_INTRA_TX) return 7;
    return 7;
  }
  if (s < S_MV_JOINT) return 2;
  if (s == S_MV_JOINT) return 4;
  if (s < S_INTER_TX) {
    const int k = (s - S_MV_COMP) & 15;
    return k == MVC_CLASS ? 11 : (k == MVC_CLASS0_FR || k == MVC_CLASS0_FR + 1 || k == MVC_FR) ? 4 : 2;
  }
  return 16;
}
AV1_HD int slot_words(int nsym) { return nsym; }      // uint16 entries of a slot: nsym - 1 values + the counter, no padding (LDS is the coder's scarce resource)

// ------------------------------------------------------------------------------------------------ ops
// 32-bit ops, RESOLVED by the tokenizer (the slot's storage offset and alphabet size travel in the op, so the serial coder has no
// table lookup on its dependent chain):
// symbol op:  0 | offset of the slot in uint16 (12 bits, << 9) | alphabet size (5 bits, << 4) | symbol (4 bits)
// literal op: 1 | n (4 bits, 1..11, << 27) | value (11 bits)          n equiprobable bits, most significant first
// split op:   1 | n = 0 | kind (bit 26: 0 = split_or_horz, 1 = split_or_vert) | offset of the partition slot (12 bits): the bit
//             "split" coded with the probability gathered from the CURRENT partition CDF (frame edges, spec 9.3); no adaptation
typedef uint32_t op_t;
struct SlotTable { uint16_t off[S_MAX]; uint8_t nsym[(S_MAX + 3) & ~3]; int words; };     // a whole number of dwords
AV1_HD op_t op_sym(int off, int nsym, int s) { return (op_t)(((unsigned)off << 9) | ((unsigned)nsym << 4) | (unsigned)s); }
AV1_HD op_t op_lit(int n, unsigned v) { return (op_t)(0x80000000u | ((unsigned)n << 27) | (v & 0x7FFu)); }
AV1_HD op_t op_split(int kind, int off) { return (op_t)(0x80000000u | ((unsigned)kind << 26) | (unsigned)off); }

// where the ops of one block go: a counting pass (out == nullptr) and a writing pass share the code
struct Sink {
  op_t *out;
  int n;
  const SlotTable *tab;
  AV1_HD void put(op_t o) { if (out) out[n] = o; n++; }
  AV1_HD void sym(int slot, int s) { if (out) out[n] = op_sym(tab->off[slot], tab->nsym[slot], s); n++; }
  AV1_HD void split(int kind, int slot) { if (out) out[n] = op_split(kind, tab->off[slot]); n++; }
  AV1_HD void lit(unsigned v, int nbits) {                // most significant bits first, at most 11 per op
    while (nbits > 11) { nbits -= 11; put(op_lit(11, v >> nbits)); }
    if (nbits > 0) put(op_lit(nbits, v & ((1u << nbits) - 1u)));
  }
};

// ------------------------------------------------------------------------------------------------ frame view
// what the tokenizer reads: the block pipeline's outputs of ONE frame (device or host pointers) + per-block summaries
struct BlockInfo { uint8_t cul[3], dc[3], mode, flags; };    // cul = min(63, sum |level|), dc = 0 none / 1 negative / 2 positive per plane;
                                                             // inter frames: mode = 0 NEAREST 1 NEAR 2 GLOBAL 3 NEW | ref index << 2,
                                                             // flags bit 0 = coded with NEWMV
struct FrameView {
  int w8, h8;                   // frame size in 8x8 blocks
  int key;                      // 1 key frame, 0 inter frame
  const uint8_t *y_mode, *uv_mode;      // key
  const int16_t *mv;                    // inter: (x, y) per block, 1/8 samples
  const uint8_t *skip;                  // inter: 1 = no residual
  const int16_t *lev_y, *lev_u, *lev_v;
  const BlockInfo *info;
  int lr_on[3];                 // Wiener restoration per plane (unit size 64 in the plane's samples)
  int8_t lr_unit[2][8];         // the unit record (luma, chroma) every unit uses: type, v0 v1 v2, h0 h1 h2
  int lr_rows[3], lr_cols[3];   // units per plane
};

AV1_HD int iabs(int v) { return v < 0 ? -v : v; }
AV1_HD int imin(int a, int b) { return a < b ? a : b; }
AV1_HD int imax(int a, int b) { return a > b ? a : b; }
AV1_HD int ilog2(unsigned v) { return 31 - __builtin_clz(v | 1u); }      // floor(log2 v), v >= 1
AV1_HD unsigned morton8(unsigned x, unsigned y) {
  unsigned m = 0;
  for (int i = 0; i < 3; i++) m |= ((x >> i) & 1u) << (2 * i) | ((y >> i) & 1u) << (2 * i + 1);
  return m;
}
// z-order index inside a superblock -> (bx, by)
AV1_HD void demorton8(unsigned k, int *bx, int *by) {
  int x = 0, y = 0;
  for (int i = 0; i < 3; i++) { x |= ((k >> (2 * i)) & 1) << i; y |= ((k >> (2 * i + 1)) & 1) << i; }
  *bx = x; *by = y;
}

// per-block summary (pass 0): level contexts the neighbours will need
AV1_HD void block_summary(const FrameView &f, int b, BlockInfo *o) {
  const int16_t *p[3] = { f.lev_y + (long)b * 64, f.lev_u + (long)b * 16, f.lev_v + (long)b * 16 };
  const bool skip = !f.key && f.skip[b];
  for (int pl = 0; pl < 3; pl++) {
    int cul = 0;
    const int n = pl ? 16 : 64;
    for (int i = 0; i < n; i++) cul += iabs(p[pl][i]);
    o->cul[pl] = skip ? 0 : (uint8_t)imin(cul, 63);
    o->dc[pl] = skip ? 0 : (uint8_t)(p[pl][0] < 0 ? 1 : p[pl][0] > 0 ? 2 : 0);
  }
}

// ------------------------------------------------------------------------------------------------ MV prediction list (7.10.2)
// identical to TileEnc::mv_stack of the host writer (all blocks 8x8, single reference LAST_FRAME, every candidate weight 4)
struct MvCand { int16_t x, y; int weight; };
struct MvStack { MvCand st[8]; int num, new_ctx, ref_ctx; };

AV1_HD void mv_stack(const FrameView &f, int r8, int c8, bool use_flags, MvStack *S) {
  const int r0 = r8 & ~7, c0 = c8 & ~7, r1 = imin(r0 + 8, f.h8), c1 = imin(c0 + 8, f.w8);      // tile = superblock
  int num = 0, new_count = 0;
  bool found = false;
  MvCand *st = S->st;
  auto add = [&](int r, int c, bool count_new) {
    if (r < r0 || r >= r1 || c < c0 || c >= c1) return;
    const int nb = r * f.w8 + c;
    const int16_t mx = f.mv[2 * nb], my = f.mv[2 * nb + 1];
    if (count_new && use_flags && (f.info[nb].flags & 1)) new_count++;
    found = true;
    int i = 0;
    for (; i < num; i++) if (st[i].x == mx && st[i].y == my) break;
    if (i < num) st[i].weight += 4;
    else if (num < 8) { st[num].x = mx; st[num].y = my; st[num].weight = 4; num++; }
  };
  add(r8 - 1, c8, true);
  const bool above0 = found; found = false;
  add(r8, c8 - 1, true);
  const bool left0 = found; found = false;
  {
    const int tr = r8 - 1, tc = c8 + 1;
    if (tr >= r0 && tc < c1 && morton8(tc & 7, tr & 7) < morton8(c8 & 7, r8 & 7)) add(tr, tc, true);
  }
  bool above = above0 || found; found = false;
  const int close = (above ? 1 : 0) + (left0 ? 1 : 0);
  const int num_nearest = num, num_new = new_count;
  for (int i = 0; i < num_nearest; i++) st[i].weight += 640;
  add(r8 - 1, c8 - 1, false);
  above = above || found; found = false;
  bool left = left0;
  add(r8 - 2, c8, false); above = above || found; found = false;
  add(r8, c8 - 2, false); left = left || found; found = false;
  add(r8 - 3, c8, false); above = above || found; found = false;
  add(r8, c8 - 3, false); left = left || found; found = false;
  const int total = (above ? 1 : 0) + (left ? 1 : 0);
  for (int pass = 0; pass < 2; pass++) {          // stable bubble sorts: the nearest entries, then the rest
    const int a = pass ? num_nearest : 0, bnd = pass ? num : num_nearest;
    for (int len = bnd; len > a;) {
      int nr = a;
      for (int i = a + 1; i < len; i++)
        if (st[i - 1].weight < st[i].weight) { const MvCand t = st[i - 1]; st[i - 1] = st[i]; st[i] = t; nr = i; }
      len = nr;
    }
  }
  if (close == 0) { S->new_ctx = imin(total, 1); S->ref_ctx = total; }
  else if (close == 1) { S->new_ctx = 3 - imin(num_new, 1); S->ref_ctx = 2 + total; }
  else { S->new_ctx = 5 - imin(num_new, 1); S->ref_ctx = 5; }
  const int mi_r = r8 * 2, mi_c = c8 * 2, border = 128 + 2 * 4 * 8;
  const int top = -(mi_r * 32) - border, bottom = (f.h8 * 2 - 2 - mi_r) * 32 + border;
  const int lft = -(mi_c * 32) - border, right = (f.w8 * 2 - 2 - mi_c) * 32 + border;
  for (int i = 0; i < num; i++) {
    st[i].y = (int16_t)imin(imax(st[i].y, top), bottom);
    st[i].x = (int16_t)imin(imax(st[i].x, lft), right);
  }
  for (int i = num; i < 2; i++) { st[i].x = st[i].y = 0; st[i].weight = 0; }
  S->num = num;
}
// pass 1 of inter frames: the mode that codes the block's vector (the writer's rule) -> info.mode / info.flags
AV1_HD void inter_mode_decision(const FrameView &f, int r8, int c8, BlockInfo *o) {
  MvStack S;
  mv_stack(f, r8, c8, false, &S);
  const int b = r8 * f.w8 + c8, mx = f.mv[2 * b], my = f.mv[2 * b + 1];
  int mode = 3, ref_idx = 0;
  if (S.st[0].x == mx && S.st[0].y == my) mode = 0;
  else {
    for (int i = 1; i < imax(S.num, 2) && i < 4; i++)
      if (S.st[i].x == mx && S.st[i].y == my) { mode = 1; ref_idx = i; break; }
    if (mode == 3 && mx == 0 && my == 0) mode = 2;
  }
  if (mode == 3) {
    const int nsel = imin(S.num, 3);
    long best = -1;
    for (int i = 0; i < imax(nsel, 1); i++) {
      const long c = iabs(mx - S.st[i].x) + iabs(my - S.st[i].y);
      if (best < 0 || c < best) { best = c; ref_idx = i; }
    }
  }
  o->mode = (uint8_t)(mode | (ref_idx << 2));
  o->flags = (uint8_t)(mode == 3);
}

// ------------------------------------------------------------------------------------------------ tokenizer
AV1_HD void tok_subexp(Sink &k, int num_syms, int kk, int v) {        // decode_subexp_bool (5.11.58)
  int i = 0, mk = 0;
  for (;;) {
    const int b2 = i ? kk + i - 1 : kk, a = 1 << b2;
    if (num_syms <= mk + 3 * a) {
      const int n = num_syms - mk, x = v - mk, w = ilog2((unsigned)n) + 1, m = (1 << w) - n;     // NS(n)
      if (x < m) k.lit((unsigned)x, w - 1);
      else { k.lit((unsigned)((x + m) >> 1), w - 1); k.lit((unsigned)((x + m) & 1), 1); }
      return;
    }
    const int more = v >= mk + a;
    k.lit((unsigned)more, 1);
    if (!more) { k.lit((unsigned)(v - mk), b2); return; }
    i++; mk += a;
  }
}
AV1_HD int tok_recenter(int r, int v) { return v > 2 * r ? v : v >= r ? 2 * (v - r) : 2 * (r - v) - 1; }
AV1_HD void tok_signed_subexp_ref(Sink &k, int v, int low, int high, int kk, int r) {
  const int mx = high - low;
  v -= low; r -= low;
  tok_subexp(k, mx, kk, (r << 1) <= mx ? tok_recenter(r, v) : tok_recenter(mx - 1 - r, mx - 1 - v));
}
// read_lr for the superblock at (sbr, sbc): with one superblock per tile the reference taps are always the defaults
AV1_HD void tok_lr(const FrameView &f, Sink &k, int sbr, int sbc) {
  const int mi_r = sbr * 16, mi_c = sbc * 16;
  for (int p = 0; p < 3; p++) {
    if (!f.lr_on[p]) continue;
    const int ss = p ? 1 : 0, us = 64;
    const int row0 = (mi_r * (4 >> ss) + us - 1) / us, row1 = imin(((mi_r + 16) * (4 >> ss) + us - 1) / us, f.lr_rows[p]);
    const int col0 = (mi_c * (4 >> ss) + us - 1) / us, col1 = imin(((mi_c + 16) * (4 >> ss) + us - 1) / us, f.lr_cols[p]);
    const int8_t *u = f.lr_unit[p ? 1 : 0];
    for (int ur = row0; ur < row1; ur++)
      for (int uc = col0; uc < col1; uc++) {
        k.sym(S_USE_WIENER, u[0] == 1);
        if (u[0] != 1) continue;
        const int kmin[3] = { -5, -23, -17 }, kmax[3] = { 10, 8, 46 }, kk[3] = { 1, 2, 3 }, mid[3] = { 3, -7, 15 };
        for (int pass = 0; pass < 2; pass++)
          for (int j = p ? 1 : 0; j < 3; j++) tok_signed_subexp_ref(k, u[1 + pass * 3 + j], kmin[j], kmax[j] + 1, kk[j], mid[j]);
      }
  }
}
// the partition symbols that precede block (bx, by) of superblock (sbr, sbc): every level whose first block it is
AV1_HD void tok_partition_prefix(const FrameView &f, Sink &k, int sbr, int sbc, int bx, int by) {
  const int mi_rows = f.h8 * 2, mi_cols = f.w8 * 2;
  for (int bsize = 64; bsize >= 16; bsize >>= 1) {
    const int n8 = bsize >> 3;
    if ((bx & (n8 - 1)) || (by & (n8 - 1))) continue;      // not the first block of this level's square
    const int r = (sbr * 8 + by) * 2, c = (sbc * 8 + bx) * 2, half = bsize >> 3;
    const bool has_rows = r + half < mi_rows, has_cols = c + half < mi_cols;
    const bool au = by > 0, al = bx > 0;                   // tile = superblock
    const int slot = (bsize == 16 ? S_PART16 : bsize == 32 ? S_PART32 : S_PART64) + (al ? 2 : 0) + (au ? 1 : 0);
    if (has_rows && has_cols) k.sym(slot, 3);              // PARTITION_SPLIT
    else if (has_cols) k.split(0, slot);                   // split_or_horz = 1
    else if (has_rows) k.split(1, slot);                   // split_or_vert = 1
  }
  k.sym(S_PART8, 0);                                       // PARTITION_NONE
}

// coeffs (5.11.39) of an N x N block (N = 8 luma, 4 chroma)
template <int N> AV1_HD void tok_coeffs(Sink &k, int plane, const int16_t *lev, int above_cul, int above_dc, int left_cul, int left_dc, bool key, int y_mode) {
  const int nc = N * N, LG = N == 4 ? 2 : 3, MS = N + 4;
  // Default_Scan_4x4 / Default_Scan_8x8 for row-major blocks: diagonals, odd ones walked downwards from the top row
  static constexpr uint8_t kScan4[16] = { 0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15 };
  static constexpr uint8_t kScan8[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
  const uint8_t *scan = N == 4 ? kScan4 : kScan8;
  int eob = 0;
  for (int c = nc - 1; c >= 0; c--) if (lev[scan[c]]) { eob = c + 1; break; }
  const bool chroma = plane > 0;
  const int skip_slot = chroma ? S_TXB_SKIP_C + ((above_cul | above_dc) != 0) + ((left_cul | left_dc) != 0) : S_TXB_SKIP_Y;
  k.sym(skip_slot, eob == 0);
  if (!eob) return;
  if (!chroma) {
    if (key) k.sym(S_INTRA_TX + y_mode, 1);    // DCT_DCT in Tx_Type_Intra_Inv_Set1
    else k.sym(S_INTER_TX, 7);                 // DCT_DCT in the 16-type inter set
  }
  const int eob_pt = eob < 3 ? eob : ilog2((unsigned)(eob - 1)) + 2;
  k.sym(chroma ? S_EOB16_C : S_EOB64_Y, eob_pt - 1);
  if (eob_pt >= 3) {
    const int off = eob - ((1 << (eob_pt - 2)) + 1);
    int shift = eob_pt - 3;
    k.sym((chroma ? S_EOBX_C : S_EOBX_Y) + eob_pt - 3, (off >> shift) & 1);
    if (shift > 0) k.lit((unsigned)(off & ((1 << shift) - 1)), shift);
  }
  uint8_t mag[MS * MS];
  for (int i = 0; i < MS * MS; i++) mag[i] = 0;
  for (int c = 0; c < eob; c++) {
    const int pos = scan[c], a = iabs(lev[pos]);
    mag[(pos >> LG) * MS + (pos & (N - 1))] = (uint8_t)(a > 15 ? 15 : a);
  }
  const int base_eob = chroma ? S_BASE_EOB_C : S_BASE_EOB_Y, base = chroma ? S_BASE_C : S_BASE_Y, br = chroma ? S_BR_C : S_BR_Y;
  for (int c = eob - 1; c >= 0; c--) {
    const int pos = scan[c], row = pos >> LG, col = pos & (N - 1);
    const uint8_t *m = mag + row * MS + col;
    const int a = iabs(lev[pos]);
    if (c == eob - 1) {
      k.sym(base_eob + (c == 0 ? 0 : c <= nc / 8 ? 1 : c <= nc / 4 ? 2 : 3), imin(a, 3) - 1);
    } else {
      const int mm = imin(m[1], 3) + imin(m[MS], 3) + imin(m[MS + 1], 3) + imin(m[2], 3) + imin(m[2 * MS], 3);
      int bctx = imin((mm + 1) >> 1, 4);
      if (pos == 0) bctx = 0;
      else bctx += row + col < 2 ? 1 : row + col < 4 ? 6 : 21;
      k.sym(base + bctx, imin(a, 3));
    }
    if (a > 2) {
      int mm = m[1] + m[MS] + m[MS + 1];
      mm = imin((mm + 1) >> 1, 6);
      const int rctx = pos == 0 ? mm : (row < 2 && col < 2) ? mm + 7 : mm + 14;
      int rem = a - 3;
      for (int i = 0; i < 4; i++) {
        const int q = imin(rem, 3);
        k.sym(br + rctx, q);
        rem -= q;
        if (q < 3) break;
      }
    }
  }
  for (int c = 0; c < eob; c++) {
    const int v = lev[scan[c]];
    if (!v) continue;
    const int a = iabs(v);
    if (c == 0) {
      const int sg = (above_dc == 2) - (above_dc == 1) + (left_dc == 2) - (left_dc == 1);
      k.sym((chroma ? S_DC_SIGN_C : S_DC_SIGN_Y) + (sg < 0 ? 1 : sg > 0 ? 2 : 0), v < 0);
    } else {
      k.lit((unsigned)(v < 0), 1);
    }
    if (a > 14) {
      const unsigned x = (unsigned)(a - 14);
      const int len = ilog2(x) + 1;
      k.lit(0, len - 1);
      k.lit(x, len);
    }
  }
}

AV1_HD void tok_mv_comp(Sink &k, int comp, int diff) {            // read_mv_component (5.11.33), quarter-sample precision
  const int base = S_MV_COMP + comp * 16;
  k.sym(base + MVC_SIGN, diff < 0);
  const int off = iabs(diff) - 1, cls = (off >> 3) < 2 ? 0 : ilog2((unsigned)(off >> 3));
  k.sym(base + MVC_CLASS, cls);
  if (cls == 0) {
    k.sym(base + MVC_CLASS0, off >> 3);
    k.sym(base + MVC_CLASS0_FR + (off >> 3), (off >> 1) & 3);
  } else {
    const int o = off - (2 << (cls + 2)), d = o >> 3;
    for (int i = 0; i < cls; i++) k.sym(base + MVC_BITS + i, (d >> i) & 1);
    k.sym(base + MVC_FR, (o >> 1) & 3);
  }
}

// all ops of the block with z-order index `zi` of superblock (sbr, sbc); returns without ops for blocks outside the frame
AV1_HD void tok_block(const FrameView &f, Sink &k, int sbr, int sbc, int zi) {
  int bx, by;
  demorton8((unsigned)zi, &bx, &by);
  const int r8 = sbr * 8 + by, c8 = sbc * 8 + bx;
  if (r8 >= f.h8 || c8 >= f.w8) return;
  if (zi == 0) tok_lr(f, k, sbr, sbc);
  tok_partition_prefix(f, k, sbr, sbc, bx, by);
  const int b = r8 * f.w8 + c8;
  const bool au = by > 0, al = bx > 0;
  const BlockInfo zero = { { 0, 0, 0 }, { 0, 0, 0 }, 0, 0 };
  const BlockInfo ia = au ? f.info[b - f.w8] : zero, il = al ? f.info[b - 1] : zero;
  const int skip = f.key ? 0 : f.skip[b];
  k.sym(S_SKIP + (au && !f.key ? f.skip[b - f.w8] : 0) + (al && !f.key ? f.skip[b - 1] : 0), skip);
  int ym = 0;
  if (f.key) {
    static const uint8_t kCtx[13] = { 0, 1, 2, 3, 4, 4, 4, 4, 3, 0, 1, 2, 0 };     // Intra_Mode_Context
    ym = f.y_mode[b];
    k.sym(S_KF_Y_MODE + kCtx[au ? f.y_mode[b - f.w8] : 0] * 5 + kCtx[al ? f.y_mode[b - 1] : 0], ym);
    if (ym >= 1 && ym <= 8) k.sym(S_ANGLE + ym - 1, 3);
    const int uvm = f.uv_mode[b];
    k.sym(S_UV_MODE + ym, uvm);
    if (uvm >= 1 && uvm <= 8) k.sym(S_ANGLE + uvm - 1, 3);
  } else {
    k.sym(S_IS_INTER + 0, 1);                  // every neighbour is an inter block: context 0
    const int rctx = (au || al) ? 1 : 0;       // single_ref contexts 2 (an inter neighbour) / 1 (none)
    k.sym(S_SINGLE_REF + rctx * 3 + 0, 0);
    k.sym(S_SINGLE_REF + rctx * 3 + 1, 0);
    k.sym(S_SINGLE_REF + rctx * 3 + 2, 0);
    MvStack S;
    mv_stack(f, r8, c8, true, &S);
    const int mode = f.info[b].mode & 3, ref_idx = f.info[b].mode >> 2;
    const int mx = f.mv[2 * b], my = f.mv[2 * b + 1];
    k.sym(S_NEW_MV + S.new_ctx, mode != 3);
    if (mode != 3) {
      k.sym(S_ZERO_MV, mode != 2);
      if (mode != 2) k.sym(S_REF_MV + S.ref_ctx, mode == 1);
    }
    auto drl_ctx = [&](int i) {
      const bool a = S.st[i].weight >= 640, c = S.st[i + 1].weight >= 640;
      return a && c ? 0 : a ? 1 : !c ? 2 : 0;
    };
    if (mode == 3) {
      for (int i = 0; i < 2; i++)
        if (S.num > i + 1) {
          k.sym(S_DRL + drl_ctx(i), ref_idx != i);
          if (ref_idx == i) break;
        }
      const int pred = S.num <= 1 ? 0 : ref_idx;
      const int dx = mx - S.st[pred].x, dy = my - S.st[pred].y;
      k.sym(S_MV_JOINT, (dx ? 1 : 0) + (dy ? 2 : 0));
      if (dy) tok_mv_comp(k, 0, dy);
      if (dx) tok_mv_comp(k, 1, dx);
    } else if (mode == 1) {
      for (int i = 1; i < 3; i++)
        if (S.num > i + 1) {
          k.sym(S_DRL + drl_ctx(i), ref_idx != i);
          if (ref_idx == i) break;
        }
    }
  }
  if (skip) return;
  tok_coeffs<8>(k, 0, f.lev_y + (long)b * 64, ia.cul[0], ia.dc[0], il.cul[0], il.dc[0], f.key != 0, ym);
  tok_coeffs<4>(k, 1, f.lev_u + (long)b * 16, ia.cul[1], ia.dc[1], il.cul[1], il.dc[1], f.key != 0, ym);
  tok_coeffs<4>(k, 2, f.lev_v + (long)b * 16, ia.cul[2], ia.dc[2], il.cul[2], il.dc[2], f.key != 0, ym);
}

// ------------------------------------------------------------------------------------------------ op coder
// The range coder of the host writer (RangeEnc, av1_bitstream.cpp) over one tile's op list.  `cdf` = the tile's slot storage
// (offsets from SlotTable), bytes go to `out` (capacity `cap`); returns the payload size, or -1 on overflow.
AV1_HD void build_slot_table(bool key, SlotTable *t) {
  const int n = key ? S_KEY_END : S_INTER_END;
  int o = 0;
  for (int s = 0; s < S_MAX; s++) {
    if (s < n) { t->nsym[s] = (uint8_t)slot_nsym(s, key); t->off[s] = (uint16_t)o; o += slot_words(t->nsym[s]); }
    else { t->nsym[s] = 2; t->off[s] = 0; }
  }
  t->words = o;
}

struct Coder {
  uint64_t low;
  uint32_t rng;
  int nb;
  uint8_t *out;
  int pos, cap;
  bool overflow;
  AV1_HD void init(uint8_t *o, int c) { low = 0; rng = 0x8000; nb = -1; out = o; pos = 0; cap = c; overflow = false; }
  AV1_HD void carry_back() { for (int i = pos; i-- > 0;) if (++out[i] != 0) break; }
  AV1_HD void renorm() {
    const int d = 15 - ilog2(rng);
    rng <<= d; low <<= d; nb += d;
    while (nb >= 8) {
      nb -= 8;
      if (pos < cap) out[pos] = (uint8_t)(low >> (16 + nb)); else overflow = true;
      pos++;
      low &= ((uint64_t)1 << (16 + nb)) - 1;
    }
  }
  AV1_HD void encode(uint32_t fl, uint32_t fh, int s, int n) {
    const uint32_t r = rng, v = (((r >> 8) * (fh >> 6)) >> 1) + 4u * (uint32_t)(n - 1 - s);
    if (fl < 32768u) {
      const uint32_t u = (((r >> 8) * (fl >> 6)) >> 1) + 4u * (uint32_t)(n - s);
      low += r - u; rng = u - v;
    } else {
      rng = r - v;
    }
    if (nb >= 0 && (low >> (16 + nb))) { if (!overflow) carry_back(); low &= ((uint64_t)1 << (16 + nb)) - 1; }
    renorm();
  }
  AV1_HD void bit(int b) {
    const uint32_t r = rng, v = ((r >> 8) << 7) + 4;
    if (b) {
      low += r - v; rng = v;
      if (nb >= 0 && (low >> (16 + nb))) { if (!overflow) carry_back(); low &= ((uint64_t)1 << (16 + nb)) - 1; }
    } else {
      rng = r - v;
    }
    renorm();
  }
  AV1_HD int finish() {
    uint64_t e = ((low + 0x3FFF) & ~(uint64_t)0x3FFF) | 0x4000;
    if (nb >= 0 && (e >> (16 + nb))) { if (!overflow) carry_back(); e &= ((uint64_t)1 << (16 + nb)) - 1; }
    int top = 15 + nb;
    while (top >= 14) {
      uint8_t byte = 0;
      for (int q = 7; q >= 0 && top >= 14; q--, top--) byte |= (uint8_t)(((e >> top) & 1) << q);
      if (pos < cap) out[pos] = byte; else overflow = true;
      pos++;
    }
    return overflow ? -1 : pos;
  }
};

// one op on a tile's CDF storage
template <class CdfPtr> AV1_HD void code_op(Coder &c, CdfPtr cdf, op_t op) {
  if (op & 0x80000000u) {
    const int n = (op >> 27) & 15;
    if (n) {
      for (int i = n - 1; i >= 0; i--) c.bit((op >> i) & 1);
    } else {
      // split_or_horz / split_or_vert: "split" with the probability gathered from the partition CDF as it stands now
      const int kind = (op >> 26) & 1;
      const uint16_t *p = &cdf[op & 0xFFF];
      auto prob = [&](int q) { return (uint32_t)((q ? p[q - 1] : 32768) - (q == 9 ? 0 : p[q])); };
      const uint32_t psum = kind == 0 ? prob(2) + prob(3) + prob(4) + prob(6) + prob(7) + prob(9) : prob(1) + prob(3) + prob(4) + prob(5) + prob(6) + prob(8);
      c.encode(psum, 0, 1, 2);
    }
    return;
  }
  const int s = op & 15, n = (op >> 4) & 31;
  auto *v = &cdf[(op >> 9) & 0xFFF];
  c.encode(s ? v[s - 1] : 32768u, s == n - 1 ? 0u : v[s], s, n);
  const int count = v[n - 1];
  const int rate = 3 + (count > 15) + (count > 31) + (n >= 4 ? 2 : 1);
  for (int i = 0; i < n - 1; i++) {
    const int x = v[i];
    v[i] = (uint16_t)(i < s ? x + ((32768 - x) >> rate) : x - (x >> rate));
  }
  v[n - 1] = (uint16_t)(count + (count < 32));
}

}  // namespace av1ops
