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
#define AV1_UNROLL _Pragma("unroll")
#define AV1_NOUNROLL _Pragma("unroll 1")
#else
#define AV1_HD inline
#define AV1_UNROLL
#define AV1_NOUNROLL
#endif

namespace av1ops {

// uint16 / byte storage that is also read and written as 32- and 64-bit words
typedef uint32_t __attribute__((may_alias)) u32a;
typedef uint64_t __attribute__((may_alias)) u64a;

// ------------------------------------------------------------------------------------------------ CDF slots of a tile
// Every adaptive CDF a tile of this tool set can touch has a slot.  A slot stores the N - 1 inverse-CDF values (32768 - cdf)
// then zeros, the adaptation counter in the slot's last entry (slot_words: 4, 8 or 16 uint16).
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
// uint16 entries of a slot: N <= 4: four (ONE 64-bit word [v0 v1 v2 counter]: the coder's step is one LDS read and one LDS
// write), N <= 8: eight, else sixteen (128-bit words); unused entries are 0, the counter is the LAST entry
AV1_HD int slot_words(int nsym) { return nsym <= 4 ? 4 : nsym <= 8 ? 8 : 16; }

// ------------------------------------------------------------------------------------------------ ops, entries, tuples
// A tile's syntax leaves the tokenizer in two forms (32-bit words):
//   the tile's LIST, one word per syntax element in decoding order:
//     literal op: 1 | n (4 bits, 1..11, << 27) | value (11 bits)      n equiprobable bits, most significant first
//     (the words of the adaptive symbols are written later, by the chains, as tuples)
//   GROUPED ENTRIES, one per adaptive symbol, grouped by CDF slot and, inside a slot, in decoding order:
//     index of the element in the list (<< 4) | symbol (4 bits; in the partition slots 14 / 15 = the gathered bit
//     split_or_horz / split_or_vert of a frame edge, spec 9.3, which reads the slot's CDF and does not adapt it)
// What the range coder needs of an adaptive symbol is a TUPLE (the coder only ever uses icdf >> 6):
//     0 | n - s (5 bits, << 19) | icdf[s - 1] >> 6 (10 bits, << 9; 512 for s = 0) | icdf[s] >> 6 (9 bits; 0 for s = n - 1)
// A slot's CDF evolves with the symbols coded in THAT slot only, so a slot's entries are a chain that turns symbols into tuples
// on its own, in parallel with every other slot of the tile; the serial range coder that follows touches no CDF at all.
typedef uint32_t op_t;
struct SlotTable { uint16_t off[S_MAX]; uint8_t nsym[(S_MAX + 3) & ~3]; int words; };     // a whole number of dwords
AV1_HD op_t op_lit(int n, unsigned v) { return (op_t)(0x80000000u | ((unsigned)n << 27) | (v & 0x7FFu)); }
AV1_HD op_t make_tuple(uint32_t fl, uint32_t fh, int s, int n) { return (op_t)(((unsigned)(n - s) << 19) | ((fl >> 6) << 9) | (fh >> 6)); }
enum { kBlocksPerTile = 64, kSplitHorz = 14, kSplitVert = 15, kListAlign = 4, kBlockRecords = 1024 };    // a slot's entries start on 16 bytes

// The tokenizer runs ONCE per block and leaves 16-bit RECORDS, one per syntax element in decoding order:
//   literal: 1 | n (4 bits, 1..11, << 11) | value (11 bits);     adaptive symbol: 0 | slot (<< 4) | symbol (4 bits)
// while it counts, per (slot, block), the adaptive symbols (cnt[slot][block], uint8).  Once the counts of all 64 blocks of the tile
// are known every block's first entry of every slot has its place in the grouped entries (group_positions), and REPLAY turns
// the block's records into list words and grouped entries without looking at the frame again.
struct Sink {
  uint16_t *rec;            // the block's records (kBlockRecords of them)
  uint8_t *cnt;             // [S_MAX][64]
  int zi, nrec, n;          // n: list words so far (a literal of more than 11 bits is several)
  bool overflow;            // more records than the block's area holds, or 255 symbols of one slot in one block
  AV1_HD void put(unsigned r) { if (nrec < kBlockRecords) rec[nrec] = (uint16_t)r; else overflow = true; nrec++; n++; }
  AV1_HD void sym(int slot, int s) {
    put(((unsigned)slot << 4) | (unsigned)s);
    uint8_t &c = cnt[slot * kBlocksPerTile + zi];
    if (c == 255) overflow = true; else c++;
  }
  AV1_HD void split(int kind, int slot) { sym(slot, kind ? kSplitVert : kSplitHorz); }
  AV1_HD void lit(unsigned v, int nbits) {                // most significant bits first, at most 11 per record
    while (nbits > 11) { nbits -= 11; put(0x8000u | (11u << 11) | ((v >> nbits) & 0x7FFu)); }
    if (nbits > 0) put(0x8000u | ((unsigned)nbits << 11) | (v & ((1u << nbits) - 1u)));
  }
};
// counts of ONE slot over the tile's 64 blocks -> positions of the blocks' first entries; returns the slot's number of entries
AV1_HD int group_positions(const uint8_t *cnt_slot, uint16_t *pos_slot, int base) {
  int run = base;
  for (int b = 0; b < kBlocksPerTile; b++) { const int c = cnt_slot[b]; pos_slot[b] = (uint16_t)run; run += c; }
  return run - base;
}
// the block's records -> list words (from index `first`) and grouped entries; pos = [slots][64] running positions.  The GPU replays
// in two passes over half of the slots each (half the positions in LDS): a pass takes the symbols of slots [slot_lo, slot_hi), whose
// positions start at pos row 0, and the literals when `literals` is set
AV1_HD void replay_block(const uint16_t *rec, int nrec, uint16_t *pos, int zi, int first, op_t *list, uint32_t *grouped, int slot_lo = 0,
                         int slot_hi = 1 << 16, bool literals = true) {
  int n = first;
  for (int i0 = 0; i0 < nrec; i0 += 8) {
    struct alignas(16) R8 { uint32_t w[4]; } q = *reinterpret_cast<const R8 *>(rec + i0);
    AV1_UNROLL      // (fully unrolled the eight records are register halves; indexed at run time the array lived in scratch memory)
    for (int j = 0; j < 8; j++, n++) {
      if (i0 + j >= nrec) break;
      const unsigned r = (q.w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
      if (r & 0x8000u) { if (literals) list[n] = op_lit((int)((r >> 11) & 15), r & 0x7FFu); }
      else {
        const int sl = (int)(r >> 4);
        if (sl >= slot_lo && sl < slot_hi) { uint16_t &p = pos[(sl - slot_lo) * kBlocksPerTile + zi]; grouped[p] = ((uint32_t)n << 4) | (r & 15u); p++; }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ frame view
// what the tokenizer reads: the block pipeline's outputs of ONE frame (device or host pointers) + per-block summaries
// cul = min(63, sum |level|), dc = 0 none / 1 negative / 2 positive per plane.  Inter frames, from the block's MV prediction list
// (inter_mode_decision; the list is built ONCE per block): mode = 0 NEAREST 1 NEAR 2 GLOBAL 3 NEW | ref index << 2, flags bit 0 =
// coded with NEWMV, num = entries of the list, drl bit i = entry i has a nearest-class weight (>= 640), (px, py) = the entry a
// NEWMV difference is coded against
struct BlockInfo { uint8_t cul[3], dc[3], mode, flags; uint8_t num, drl; int16_t px, py; uint8_t pad[2]; };
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
  o->num = (uint8_t)S.num;
  int drl = 0;
  for (int i = 0; i < 4; i++) if (i < imax(S.num, 2) && S.st[i].weight >= 640) drl |= 1 << i;
  o->drl = (uint8_t)drl;
  const int pred = S.num <= 1 ? 0 : ref_idx;
  o->px = S.st[pred].x; o->py = S.st[pred].y;
}

// ------------------------------------------------------------------------------------------------ tokenizer
template <class K> AV1_HD void tok_subexp(K &k, int num_syms, int kk, int v) {        // decode_subexp_bool (5.11.58)
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
template <class K> AV1_HD void tok_signed_subexp_ref(K &k, int v, int low, int high, int kk, int r) {
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
// Default_Scan_4x4 / Default_Scan_8x8 for row-major blocks: diagonals, odd ones walked downwards from the top row
struct ScanTables { uint8_t s4[16], s8[64], i4[16], i8[64]; };      // scan order -> position, and position -> scan order
AV1_HD void fill_scan_tables(ScanTables *t) {
  for (int N = 4; N <= 8; N += 4) {
    uint8_t *o = N == 4 ? t->s4 : t->s8;
    int k = 0;
    for (int d = 0; d < 2 * N - 1; d++)
      for (int i = 0; i <= d; i++) {
        const int r = (d & 1) ? i : d - i, c = d - r;
        if (r < N && c < N) o[k++] = (uint8_t)(r * N + c);
      }
    uint8_t *inv = N == 4 ? t->i4 : t->i8;
    for (int i = 0; i < N * N; i++) inv[o[i]] = (uint8_t)i;
  }
}
// what a thread needs beside the frame to tokenize coefficients: its own scratch (LDS on the GPU: the magnitudes are read five
// at a time per coefficient) and the scan tables
enum { kMagStride = 10, kMagBytes = 100 };      // (8 + 2)^2: the context templates reach two columns / rows beyond a level; 25 dwords (odd) per thread
struct TokScratch { uint8_t *mag; const ScanTables *scan; };

// coeffs() (5.11.39) of one transform block: `lev` = its N x N levels, row-major (16-byte aligned)
template <int N> AV1_HD void tok_coeffs(Sink &k, const TokScratch &ts, int plane, const int16_t *lev, int above_cul, int above_dc, int left_cul, int left_dc, bool key,
                                        int y_mode) {
  const int nc = N * N, LG = N == 4 ? 2 : 3, MS = kMagStride;
  const uint8_t *scan = N == 4 ? ts.scan->s4 : ts.scan->s8;
  // min(|level|, 15) | sign << 7 of the whole block, zero-padded to the right and below: every later read is from this copy
  // (the exact value of the rare levels above 14 is re-read from `lev`)
  uint8_t *mag = ts.mag;
  const uint8_t *iscan = N == 4 ? ts.scan->i4 : ts.scan->i8;
  for (int i = 0; i < (N + 2) * MS / 4; i++) reinterpret_cast<u32a *>(mag)[i] = 0;      // rows 0 .. N + 1: all a context template reaches
  int eob = 0;      // 1 + the scan index of the last non-zero level, found while the block is copied (not by walking the scan backwards)
  for (int r = 0; r < nc / 8; r++) {
    struct alignas(16) L8 { int16_t v[8]; } q = *reinterpret_cast<const L8 *>(lev + 8 * r);
    for (int j = 0; j < 8; j++) {
      const int pos = 8 * r + j, v = q.v[j], a = iabs(v);
      if (v) eob = imax(eob, iscan[pos] + 1);
      mag[(pos >> LG) * MS + (pos & (N - 1))] = (uint8_t)((a > 15 ? 15 : a) | (v < 0 ? 128 : 0));
    }
  }
  auto at = [&](int pos) { return mag[(pos >> LG) * MS + (pos & (N - 1))]; };
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
  const int base_eob = chroma ? S_BASE_EOB_C : S_BASE_EOB_Y, base = chroma ? S_BASE_C : S_BASE_Y, br = chroma ? S_BR_C : S_BR_Y;
  for (int c = eob - 1; c >= 0; c--) {
    const int pos = scan[c], row = pos >> LG, col = pos & (N - 1);
    const uint8_t *m = mag + row * MS + col;
    const int m0 = m[0] & 15, m1 = m[1] & 15, m2 = m[2] & 15, mb = m[MS] & 15, md = m[MS + 1] & 15, mbb = m[2 * MS] & 15;
    int a = m0;
    if (a == 15) a = iabs(lev[pos]);
    if (c == eob - 1) {
      k.sym(base_eob + (c == 0 ? 0 : c <= nc / 8 ? 1 : c <= nc / 4 ? 2 : 3), imin(a, 3) - 1);
    } else {
      const int mm = imin(m1, 3) + imin(mb, 3) + imin(md, 3) + imin(m2, 3) + imin(mbb, 3);
      int bctx = imin((mm + 1) >> 1, 4);
      if (pos == 0) bctx = 0;
      else bctx += row + col < 2 ? 1 : row + col < 4 ? 6 : 21;
      k.sym(base + bctx, imin(a, 3));
    }
    if (a > 2) {
      int mm = m1 + mb + md;
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
    const int pos = scan[c], m = at(pos);
    if (!m) continue;
    const int neg = m >> 7;
    if (c == 0) {
      const int sg = (above_dc == 2) - (above_dc == 1) + (left_dc == 2) - (left_dc == 1);
      k.sym((chroma ? S_DC_SIGN_C : S_DC_SIGN_Y) + (sg < 0 ? 1 : sg > 0 ? 2 : 0), neg);
    } else {
      k.lit((unsigned)neg, 1);
    }
    if ((m & 15) == 15) {
      const int a = iabs(lev[pos]);
      if (a > 14) {
        const unsigned x = (unsigned)(a - 14);
        const int len = ilog2(x) + 1;
        k.lit(0, len - 1);
        k.lit(x, len);
      }
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
AV1_HD void tok_block(const FrameView &f, Sink &k, const TokScratch &ts, int sbr, int sbc, int zi) {
  int bx, by;
  demorton8((unsigned)zi, &bx, &by);
  const int r8 = sbr * 8 + by, c8 = sbc * 8 + bx;
  if (r8 >= f.h8 || c8 >= f.w8) return;
  if (zi == 0) tok_lr(f, k, sbr, sbc);
  tok_partition_prefix(f, k, sbr, sbc, bx, by);
  const int b = r8 * f.w8 + c8;
  const bool au = by > 0, al = bx > 0;
  const BlockInfo zero = {};
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
    // the prediction list was built by inter_mode_decision (info).  Its contexts: every candidate position above (left) of the
    // block lies inside the tile exactly when the row above (column to the left) does, so "close" and "total" matches are the same
    // count; the NEWMV neighbours are counted among above, left and above-right (7.10.2.7 with every block 8x8)
    const BlockInfo me = f.info[b];
    const int mode = me.mode & 3, ref_idx = me.mode >> 2, num = me.num;
    const int mx = f.mv[2 * b], my = f.mv[2 * b + 1];
    const int close = (au ? 1 : 0) + (al ? 1 : 0);
    int num_new = (au ? ia.flags & 1 : 0) + (al ? il.flags & 1 : 0);
    if (au && bx < 7 && c8 + 1 < f.w8 && morton8((unsigned)bx + 1, (unsigned)by - 1) < morton8((unsigned)bx, (unsigned)by)) num_new += f.info[b - f.w8 + 1].flags & 1;
    const int new_ctx = close == 0 ? 0 : close == 1 ? 3 - imin(num_new, 1) : 5 - imin(num_new, 1);
    const int ref_ctx = close == 0 ? 0 : close == 1 ? 3 : 5;
    k.sym(S_NEW_MV + new_ctx, mode != 3);
    if (mode != 3) {
      k.sym(S_ZERO_MV, mode != 2);
      if (mode != 2) k.sym(S_REF_MV + ref_ctx, mode == 1);
    }
    auto drl_ctx = [&](int i) {
      const bool a = (me.drl >> i) & 1, c = (me.drl >> (i + 1)) & 1;
      return a && c ? 0 : a ? 1 : !c ? 2 : 0;
    };
    if (mode == 3) {
      for (int i = 0; i < 2; i++)
        if (num > i + 1) {
          k.sym(S_DRL + drl_ctx(i), ref_idx != i);
          if (ref_idx == i) break;
        }
      const int dx = mx - me.px, dy = my - me.py;
      k.sym(S_MV_JOINT, (dx ? 1 : 0) + (dy ? 2 : 0));
      if (dy) tok_mv_comp(k, 0, dy);
      if (dx) tok_mv_comp(k, 1, dx);
    } else if (mode == 1) {
      for (int i = 1; i < 3; i++)
        if (num > i + 1) {
          k.sym(S_DRL + drl_ctx(i), ref_idx != i);
          if (ref_idx == i) break;
        }
    }
  }
  if (skip) return;
  tok_coeffs<8>(k, ts, 0, f.lev_y + (long)b * 64, ia.cul[0], ia.dc[0], il.cul[0], il.dc[0], f.key != 0, ym);
  AV1_NOUNROLL
  for (int p = 1; p < 3; p++)
    tok_coeffs<4>(k, ts, p, (p == 1 ? f.lev_u : f.lev_v) + (long)b * 16, ia.cul[p], ia.dc[p], il.cul[p], il.dc[p], f.key != 0, ym);
}

// ------------------------------------------------------------------------------------------------ op coder
// The range coder of the host writer (RangeEnc, av1_bitstream.cpp) over one tile's op list.  `cdf` = the tile's slot storage
// (offsets from SlotTable), bytes go to `out` (capacity `cap`); returns the payload size, or -1 on overflow.
AV1_HD void build_slot_table(bool key, SlotTable *t) {
  const int n = key ? S_KEY_END : S_INTER_END;
  int o = 0;
  for (int s = 0; s < S_MAX; s++) {
    if (s < n) {
      t->nsym[s] = (uint8_t)slot_nsym(s, key);
      const int w = slot_words(t->nsym[s]), al = w < 8 ? 4 : 8;      // 64-bit words on 8 bytes, 128-bit words on 16
      o = (o + al - 1) & ~(al - 1);
      t->off[s] = (uint16_t)o;
      o += w;
    } else { t->nsym[s] = 2; t->off[s] = 0; }
  }
  t->words = (o + 7) & ~7;
}

// The range coder (spec 8.2 mirrored; the arithmetic of the host writer's RangeEnc, av1_bitstream.cpp), built so that its steady
// state never waits for global memory (on gfx9 a wait for ANY outstanding vector-memory operation, stores included, is the only
// wait there is: one store per step puts a memory round trip on every step of the serial chain):
//  * bytes leave the window LAZILY: `low` keeps 16 + nb bits, sixteen of them are retired when nb reaches 32, so at least 16
//    finished bits always stay behind in the register and a carry reaches bytes already retired only through sixteen 1 bits in
//    a row (2^-16 per flush); the carry bit sits above the window (bit 16 + nb) until the next flush looks at it;
//  * retired bits go to a small STAGE (32 x 16 bits; LDS on the GPU), not to the output: spill() stores the stage to `out`
//    and is called by the loop that drives the coder — on the GPU in rare wave-uniform phases, all lanes at once.
#ifdef AV1_CODER_STATS
static long long coder_stat_carries = 0;
#endif
struct Coder {
  enum { kStage = 32 };
  uint64_t low;
  uint32_t rng;
  int nb;
  uint8_t *out;
  uint16_t *stage;         // kStage entries, most significant byte first as a NUMBER (swapped to memory order when spilled)
  int pos, cap, ns;        // bytes in `out`, its capacity, entries staged
  bool overflow;
  AV1_HD void init(uint8_t *o, int c, uint16_t *st) { low = 0; rng = 0x8000; nb = -1; out = o; stage = st; pos = 0; cap = c; ns = 0; overflow = false; }
  AV1_HD void carry_back() {        // + 1 on the bytes already retired: the staged ones first, then the stored ones
    for (int k = ns; k-- > 0;) { const uint16_t v = (uint16_t)(stage[k] + 1); stage[k] = v; if (v) return; }
    for (int i = pos; i-- > 0;) if (++out[i] != 0) return;
  }
  AV1_HD bool stage_full() const { return ns >= kStage - 1; }
  AV1_HD void spill() {
    if (pos + 2 * ns > cap) { overflow = true; pos += 2 * ns; ns = 0; return; }
    for (int k = 0; k < ns; k++) { const uint32_t v = stage[k]; *reinterpret_cast<uint16_t *>(out + pos + 2 * k) = (uint16_t)((v >> 8) | (v << 8)); }   // pos is even
    pos += 2 * ns; ns = 0;
  }
  AV1_HD void flush16() {           // nb >= 32; the driver keeps ns < kStage (stage_full -> spill)
    nb -= 16;
    const uint32_t w = (uint32_t)(low >> (16 + nb));          // 16 bits + the carry above them
    if ((w >> 16) && !overflow) {
#ifdef AV1_CODER_STATS
      coder_stat_carries++;          // host self-test builds only (host/coder_selftest.cpp)
#endif
      carry_back();
    }
    stage[ns++] = (uint16_t)w;
    low &= ((uint64_t)1 << (16 + nb)) - 1;
  }
  // symbol s of an alphabet of n with inverse CDF values fl = icdf[s - 1] (32768 for s = 0), fh = icdf[s] (0 for s = n - 1)
  AV1_HD void encode(uint32_t fl, uint32_t fh, int s, int n) {
    const uint32_t r = rng, v = (((r >> 8) * (fh >> 6)) >> 1) + 4u * (uint32_t)(n - 1 - s);
    const uint32_t u = fl < 32768u ? (((r >> 8) * (fl >> 6)) >> 1) + 4u * (uint32_t)(n - s) : r;
    low += r - u;
    const uint32_t nr = u - v;
    const int d = 15 - ilog2(nr);
    rng = nr << d; low <<= d; nb += d;
    if (nb >= 32) flush16();
  }
  AV1_HD void encode_tuple(op_t t) {      // make_tuple's fields: the same interval update
    const uint32_t fl6 = (t >> 9) & 1023u, fh6 = t & 511u, dl = (t >> 19) & 31u;
    const uint32_t r = rng, v = (((r >> 8) * fh6) >> 1) + 4u * (dl - 1u);
    const uint32_t u = fl6 < 512u ? (((r >> 8) * fl6) >> 1) + 4u * dl : r;
    low += r - u;
    const uint32_t nr = u - v;
    const int d = 15 - ilog2(nr);
    rng = nr << d; low <<= d; nb += d;
    if (nb >= 32) flush16();
  }
  AV1_HD void bit(int b) { encode(b ? 16384u : 32768u, b ? 0u : 16384u, b, 2); }      // an equiprobable bit (literals)
  AV1_HD int finish() {             // exit process (8.2.4)
    uint64_t e = ((low + 0x3FFF) & ~(uint64_t)0x3FFF) | 0x4000;
    if (nb >= 0 && (e >> (16 + nb))) { if (!overflow) carry_back(); e &= ((uint64_t)1 << (16 + nb)) - 1; }
    spill();
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

// 8.2.6 on inverse values: x + ((32768 - x) >> rate) for the entries below the symbol, x - (x >> rate) for the others; written
// without a branch (m = all ones below the symbol: 32768 - x = (x ^ m) + 32769, and the step changes sign with m)
AV1_HD uint32_t adapt(uint32_t x, bool below, int rate) {
  const uint32_t m = 0u - (uint32_t)below, d = ((x ^ m) + (m & 32769u)) >> rate;
  return x - ((d ^ m) - m);
}

// one link of a chain: the tuple of symbol s, and the slot's CDF adapted (8.2.6).  Alphabets of at most four: the slot is one
// 64-bit word [v0 v1 v2 counter] (a chain keeps it in registers)
AV1_HD op_t small_step(uint64_t &w, int s, int n) {
  const uint32_t x0 = (uint32_t)w & 0xFFFFu, x1 = (uint32_t)w >> 16, x2 = (uint32_t)(w >> 32) & 0xFFFFu, cnt = (uint32_t)(w >> 48);
  const uint32_t fl = s == 0 ? 32768u : s == 1 ? x0 : s == 2 ? x1 : x2;
  const uint32_t fh = s == n - 1 ? 0u : s == 0 ? x0 : s == 1 ? x1 : x2;
  const int rate = 3 + (cnt > 15) + (cnt > 31) + (n >= 4 ? 2 : 1);
  // unused entries are 0 and stay 0 (they are never "below" the symbol)
  const uint32_t y0 = adapt(x0, 0 < s, rate), y1 = adapt(x1, 1 < s, rate), y2 = adapt(x2, 2 < s, rate), ncnt = cnt + (cnt < 32);
  w = (uint64_t)(y0 | (y1 << 16)) | ((uint64_t)(y2 | (ncnt << 16)) << 32);
  return make_tuple(fl, fh, s, n);
}

// alphabets of 5..16: a slot of 8 (n <= 8) or 16 entries in memory (LDS on the GPU), read and written as dwords
AV1_HD op_t big_step(uint16_t *v, int s, int n) {
  const int nd = n > 8 ? 8 : 4;
  u32a *d = reinterpret_cast<u32a *>(v);
  uint32_t w[8];
  AV1_UNROLL
  for (int k = 0; k < 8; k++) w[k] = k < nd ? d[k] : 0u;
  const uint32_t fl = s ? v[s - 1] : 32768u, fh = s == n - 1 ? 0u : v[s];
  const uint32_t cnt = (nd == 8 ? w[7] : w[3]) >> 16;
  const int rate = 3 + (cnt > 15) + (cnt > 31) + 2;
  AV1_UNROLL
  for (int k = 0; k < 8; k++) {
    if (k >= nd) continue;
    uint32_t lo = adapt(w[k] & 0xFFFFu, 2 * k < s, rate), hi = adapt(w[k] >> 16, 2 * k + 1 < s, rate);
    if (k == nd - 1) hi = cnt + (cnt < 32);
    d[k] = lo | (hi << 16);
  }
  return make_tuple(fl, fh, s, n);
}

// split_or_horz / split_or_vert: "split" with the probability gathered from the partition CDF as it stands; no adaptation
AV1_HD op_t split_tuple(const uint16_t *p, int kind) {
  const uint32_t p1 = p[0] - p[1], p2 = p[1] - p[2], p3 = p[2] - p[3], p4 = p[3] - p[4], p5 = p[4] - p[5], p6 = p[5] - p[6], p7 = p[6] - p[7],
                 p8 = p[7] - p[8], p9 = p[8];
  return make_tuple(kind == kSplitHorz ? p2 + p3 + p4 + p6 + p7 + p9 : p1 + p3 + p4 + p5 + p6 + p8, 0, 1, 2);
}

// a whole chain, the plain way (the CPU twin; the GPU kernel runs the same steps with its own prefetching loops): `init` = the
// slot's default CDF (slot_words(n) entries), `g` = the slot's `cnt` grouped entries, tuples go to list[index]
AV1_HD void run_chain(const uint16_t *init, int n, const uint32_t *g, int cnt, op_t *list) {
  if (n <= 4) {
    uint64_t w = (uint64_t)init[0] | ((uint64_t)init[1] << 16) | ((uint64_t)init[2] << 32) | ((uint64_t)init[3] << 48);
    for (int k = 0; k < cnt; k++) list[g[k] >> 4] = small_step(w, (int)(g[k] & 15), n);
    return;
  }
  alignas(16) uint16_t v[16];
  for (int i = 0; i < 16; i++) v[i] = i < slot_words(n) ? init[i] : 0;
  for (int k = 0; k < cnt; k++) {
    const int s = (int)(g[k] & 15);
    list[g[k] >> 4] = s >= kSplitHorz ? split_tuple(v, s) : big_step(v, s, n);
  }
}

// one word of a tile's finished list through the range coder
AV1_HD void code_word(Coder &c, op_t e) {
  if (e >> 31) { for (int i = (int)((e >> 27) & 15) - 1; i >= 0; i--) { c.bit((e >> i) & 1); if (c.stage_full()) c.spill(); } }
  else { c.encode_tuple(e); if (c.stage_full()) c.spill(); }
}

}  // namespace av1ops
