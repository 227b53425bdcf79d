// av1_ops32.hpp — the tile syntax of KEY FRAMES IN 32x32 BLOCKS (av1mi_gop_config.key_block_size = 32) as the same op
// stream av1_ops.hpp makes of 8x8 blocks: records -> list words + grouped entries, after which the chains, the range coder and the
// gather of av1_entropy_kernels.hip (and their CPU twins) run unchanged.  A tile is one complete 64x64 superblock: PARTITION_SPLIT, four
// 32x32 blocks (PARTITION_NONE), luma transform 32x32 (DCT_DCT, not coded: the 32x32 set holds nothing else), chroma 16x16 with the
// transform type implied by the mode, TX_MODE_LARGEST.  ONE THREAD PER 32x32 BLOCK tokenizes its 1024 + 2 x 256 coefficients serially
// (key frames are one frame in a GOP; the unit of parallel work that would suit them better is a scan range of a transform block,
// DESIGN 7-1); what a block needs of its neighbours — their level summaries — is computed beforehand from the levels alone.
// Shared source: hipcc for the device, g++ for the CPU twin (host/av1_opstream.cpp), verified there byte for byte against the general
// block writer (host/av1_blockstream.cpp, itself verified by dav1d).
#pragma once
#include "av1_ops.hpp"

namespace av1ops {

enum SlotK32 : int {
  K_SKIP = 0,                    // context 0 (key frames code skip = 0)
  K_PART32 = 1,                  // context 0: the neighbours inside the tile are 32x32 blocks too
  K_PART64 = 2,                  // context 0: nothing of the tile lies above or to the left of a superblock
  K_USE_WIENER = 3,
  K_TXB_SKIP_Y = 4,              // all_zero, luma: context 0 (the transform covers the block)
  K_TXB_SKIP_C = 5,              // [3] chroma: contexts 7, 8, 9
  K_EOB_Y = 8,                   // eob_pt_1024
  K_EOB_C = 9,                   // eob_pt_256
  K_EOBX_Y = 10,                 // [9] eob_extra
  K_EOBX_C = 19,                 // [7]
  K_DC_SIGN_Y = 26,              // [3]
  K_DC_SIGN_C = 29,              // [3]
  K_BASE_EOB_Y = 32,             // [4]
  K_BASE_EOB_C = 36,             // [4]
  K_BASE_Y = 40,                 // [26]
  K_BASE_C = 66,                 // [26]
  K_BR_Y = 92,                   // [21]
  K_BR_C = 113,                  // [21]
  K_KF_Y_MODE = 134,             // [5][5]
  K_UV_MODE = 159,               // [13] chroma-from-luma allowed (blocks up to 32x32)
  K_ANGLE = 172,                 // [8]
  K_END = 180
};
static_assert((int)K_END <= (int)S_MAX, "the slot arrays of the coder are sized by S_MAX");

AV1_HD int slot_nsym_k32(int s) {
  if (s == K_SKIP) return 2;
  if (s == K_PART32 || s == K_PART64) return 10;
  if (s < K_EOB_Y) return 2;
  if (s == K_EOB_Y) return 11;
  if (s == K_EOB_C) return 9;
  if (s < K_BASE_EOB_Y) return 2;
  if (s < K_BASE_Y) return 3;
  if (s < K_KF_Y_MODE) return 4;
  if (s < K_UV_MODE) return 13;
  if (s < K_ANGLE) return 14;
  return 7;
}
AV1_HD void build_slot_table_k32(SlotTable *t) {
  int o = 0;
  for (int s = 0; s < S_MAX; s++) {
    if (s < K_END) {
      t->nsym[s] = (uint8_t)slot_nsym_k32(s);
      const int w = slot_words(t->nsym[s]), al = w < 8 ? 4 : 8;
      o = (o + al - 1) & ~(al - 1);
      t->off[s] = (uint16_t)o;
      o += w;
    } else { t->nsym[s] = 2; t->off[s] = 0; }
  }
  t->words = (o + 7) & ~7;
}

// Default_Scan_16x16 / Default_Scan_32x32 (zig-zag: odd diagonals downwards) and their inverses
struct ScanTables32 { uint16_t s32[1024], i32[1024]; uint8_t s16[256], i16[256]; };
// thread `tid` of `nthreads` fills its share: the scan index of a position has a closed form (the diagonals before its own, then its
// place on the diagonal), so no thread walks the scan
AV1_HD void fill_scan_tables32(ScanTables32 *t, int tid = 0, int nthreads = 1) {
  for (int N = 16; N <= 32; N += 16)
    for (int pos = tid; pos < N * N; pos += nthreads) {
      const int r = pos / N, c = pos % N, d = r + c;
      const int before = d < N ? d * (d + 1) / 2 : N * N - (2 * N - 1 - d) * (2 * N - d) / 2;
      const int rmin = imax(0, d - N + 1), rmax = imin(d, N - 1);
      const int k = before + ((d & 1) ? r - rmin : rmax - r);
      if (N == 16) { t->s16[k] = (uint8_t)pos; t->i16[pos] = (uint8_t)k; }
      else { t->s32[k] = (uint16_t)pos; t->i32[pos] = (uint16_t)k; }
    }
}
// a thread's scratch: min(|level|, 15) of the transform block at hand, FOUR BITS each, rows of 32 + 4 entries (18 bytes), 32 + 2 rows
// (the context templates reach two rows / columns beyond a level), then one sign bit per level.  (A byte per level was 1.2 KB per
// lane: with the counters 107 KB per 64 lanes, one wave per CU.)
enum { kMag32Stride = 36, kMag32Nibbles = 36 * 34 / 2, kMag32Bytes = kMag32Nibbles + 32 * 32 / 8 + 4 };      // 612 + 128 (+ 4: 16-byte multiples)
struct TokScratch32 { uint8_t *mag; const ScanTables32 *scan; };

// the sink of a block's tokenizer: records in one run, 16-bit symbol counts per (slot, block of the tile)
enum { kBlocks32 = 4, kBlockRecords32 = kBlocksPerTile * kBlockRecords / kBlocks32 };      // the tile's record area, a quarter per block
struct Sink32 {
  uint16_t *rec; uint16_t *cnt;      // cnt[K_END][kBlocks32]; this block's column is `blk`
  int blk, cap, nrec, n;
  bool overflow;
  uint32_t w0, w1, w2, w3;           // the records on their way out, eight per 16-byte store (rec is 16-byte aligned; call flush() at the end)
  AV1_HD void put(unsigned r) {
    const int j = nrec & 7;
    const uint32_t v = (r & 0xFFFFu) << (16 * (j & 1));
    if (j < 2) w0 = j ? w0 | v : v; else if (j < 4) w1 = j & 1 ? w1 | v : v; else if (j < 6) w2 = j & 1 ? w2 | v : v; else w3 = j & 1 ? w3 | v : v;
    nrec++; n++;
    if (j == 7) store8();
  }
  AV1_HD void store8() {              // the eight records before nrec (rounded up)
    const int at = (nrec - 1) & ~7;
    if (at + 8 <= cap) { struct alignas(16) R8 { uint32_t w[4]; } q = { { w0, w1, w2, w3 } }; *reinterpret_cast<R8 *>(rec + at) = q; }
    else overflow = true;
  }
  AV1_HD void flush() { if (nrec & 7) store8(); }
  AV1_HD void sym(int slot, int s) { put(((unsigned)slot << 4) | (unsigned)s); }      // counted afterwards: count_block32
  AV1_HD void split(int kind, int slot) { sym(slot, kind ? kSplitVert : kSplitHorz); }   // split_or_horz / split_or_vert = 1 at a frame edge
  AV1_HD void lit(unsigned v, int nbits) {
    while (nbits > 11) { nbits -= 11; put(0x8000u | (11u << 11) | ((v >> nbits) & 0x7FFu)); }
    if (nbits > 0) put(0x8000u | ((unsigned)nbits << 11) | (v & ((1u << nbits) - 1u)));
  }
};

// coeffs() (5.11.39) of one N x N transform block (N = 32 luma / 16 chroma), 2-D class; cul / dc: this block's level summary for its
// neighbours (min(63, sum |level|); 0 none / 1 negative / 2 positive)
template <int N> AV1_HD void tok_coeffs_big(Sink32 &k, const TokScratch32 &ts, bool chroma, const int16_t *lev, int above_cul, int above_dc, int left_cul,
                                            int left_dc) {
  const int nc = N * N, LG = N == 16 ? 4 : 5, MS = kMag32Stride;
  uint8_t *mag = ts.mag, *sgn = ts.mag + kMag32Nibbles;
  for (int i = 0; i < (N + 2) * MS / 8; i++) reinterpret_cast<u32a *>(mag)[i] = 0;        // (N + 2) rows of 18 bytes: a whole number of dwords for N = 16, 32
  auto nib = [&](int idx) { return (mag[idx >> 1] >> ((idx & 1) << 2)) & 15; };          // idx = row * MS + col
  int eob = 0;
  for (int r = 0; r < nc / 8; r++) {
    struct alignas(16) L8 { int16_t v[8]; } q = *reinterpret_cast<const L8 *>(lev + 8 * r);
    unsigned packed = 0, signs = 0;
    for (int j = 0; j < 8; j++) {
      const int pos = 8 * r + j, v = q.v[j], a = iabs(v);
      if (v) eob = imax(eob, (N == 16 ? (int)ts.scan->i16[pos] : (int)ts.scan->i32[pos]) + 1);
      packed |= (unsigned)(a > 15 ? 15 : a) << (4 * j);
      signs |= (unsigned)(v < 0) << j;
    }
    // eight levels of one row, from an even column on: two 16-bit stores (a row of 18 bytes starts on an even address only)
    const int at = (((8 * r) >> LG) * MS + ((8 * r) & (N - 1))) >> 1;
    *reinterpret_cast<uint16_t *>(mag + at) = (uint16_t)packed;
    *reinterpret_cast<uint16_t *>(mag + at + 2) = (uint16_t)(packed >> 16);
    sgn[r] = (uint8_t)signs;
  }
  k.sym(chroma ? K_TXB_SKIP_C + ((above_cul | above_dc) != 0) + ((left_cul | left_dc) != 0) : K_TXB_SKIP_Y, eob == 0);
  if (!eob) return;
  const int eob_pt = eob < 3 ? eob : ilog2((unsigned)(eob - 1)) + 2;
  k.sym(chroma ? K_EOB_C : K_EOB_Y, eob_pt - 1);
  if (eob_pt >= 3) {
    const int off = eob - ((1 << (eob_pt - 2)) + 1), shift = eob_pt - 3;
    k.sym((chroma ? K_EOBX_C : K_EOBX_Y) + eob_pt - 3, (off >> shift) & 1);
    if (shift > 0) k.lit((unsigned)(off & ((1 << shift) - 1)), shift);
  }
  const int base_eob = chroma ? K_BASE_EOB_C : K_BASE_EOB_Y, base = chroma ? K_BASE_C : K_BASE_Y, br = chroma ? K_BR_C : K_BR_Y;
  for (int c = eob - 1; c >= 0; c--) {
    const int pos = N == 16 ? (int)ts.scan->s16[c] : (int)ts.scan->s32[c], row = pos >> LG, col = pos & (N - 1);
    const int at = row * MS + col;
    const int m0 = nib(at), m1 = nib(at + 1), m2 = nib(at + 2), mb = nib(at + MS), md = nib(at + MS + 1), mbb = nib(at + 2 * MS);
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
    const int pos = N == 16 ? (int)ts.scan->s16[c] : (int)ts.scan->s32[c], m = nib((pos >> LG) * MS + (pos & (N - 1)));
    if (!m) continue;
    const int neg = (sgn[pos >> 3] >> (pos & 7)) & 1;
    if (c == 0) {
      const int sg = (above_dc == 2) - (above_dc == 1) + (left_dc == 2) - (left_dc == 1);
      k.sym((chroma ? K_DC_SIGN_C : K_DC_SIGN_Y) + (sg < 0 ? 1 : sg > 0 ? 2 : 0), neg);
    } else {
      k.lit((unsigned)neg, 1);
    }
    if (m == 15) {
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

// read_lr for the superblock (av1_ops.hpp tok_lr with this frame kind's slot)
AV1_HD void tok_lr32(const FrameView &f, Sink32 &k, int sbr, int sbc) {
  const int mi_r = sbr * 16, mi_c = sbc * 16;
  for (int p = 0; p < 3; p++) {
    if (!f.lr_on[p]) continue;
    const int ss = p ? 1 : 0, us = 64;
    const int row0 = (mi_r * (4 >> ss) + us - 1) / us, row1 = imin(((mi_r + 16) * (4 >> ss) + us - 1) / us, f.lr_rows[p]);
    const int col0 = (mi_c * (4 >> ss) + us - 1) / us, col1 = imin(((mi_c + 16) * (4 >> ss) + us - 1) / us, f.lr_cols[p]);
    const int8_t *u = f.lr_unit[p ? 1 : 0];
    for (int ur = row0; ur < row1; ur++)
      for (int uc = col0; uc < col1; uc++) {
        k.sym(K_USE_WIENER, u[0] == 1);
        if (u[0] != 1) continue;
        const int kmin[3] = { -5, -23, -17 }, kmax[3] = { 10, 8, 46 }, kk[3] = { 1, 2, 3 }, mid[3] = { 3, -7, 15 };
        for (int pass = 0; pass < 2; pass++)
          for (int j = p ? 1 : 0; j < 3; j++) tok_signed_subexp_ref(k, u[1 + pass * 3 + j], kmin[j], kmax[j] + 1, kk[j], mid[j]);
      }
  }
}

// what the neighbours of a transform block read of it: min(63, sum |level|) and the DC's sign class (0 none / 1 negative / 2 positive)
struct Sum32 { uint8_t cul[3], dc[3]; };
AV1_HD void block_sums32(const FrameView &f, long i, Sum32 *o) {
  for (int p = 0; p < 3; p++) {
    const int n = p ? 256 : 1024;
    const int16_t *lev = (p == 0 ? f.lev_y : p == 1 ? f.lev_u : f.lev_v) + i * n;
    int cul = 0;
    for (int r = 0; r < n / 8; r++) {
      struct alignas(16) L8 { int16_t v[8]; } q = *reinterpret_cast<const L8 *>(lev + 8 * r);
      for (int j = 0; j < 8; j++) cul += iabs(q.v[j]);
    }
    o->cul[p] = (uint8_t)imin(cul, 63);
    o->dc[p] = (uint8_t)(lev[0] < 0 ? 1 : lev[0] > 0 ? 2 : 0);
  }
}
AV1_HD long block_index32(const FrameView &f, int sbr, int sbc, int b) { return (long)(sbr * 2 + (b >> 1)) * (f.w8 / 4) + sbc * 2 + (b & 1); }

// all ops of block b (0..3, raster = decoding order) of the tile = superblock (sbr, sbc) of a key frame's 32x32 band; sums[4]: the
// tile's block summaries (of the blocks inside the frame).  f.y_mode / f.uv_mode: the band's modes, one per 32x32 block in raster order (w8 / 4 per row); f.lev_*:
// block-contiguous over the same grid (1024 luma, 256 + 256 chroma levels per block)
AV1_HD void tok_block32(const FrameView &f, Sink32 &k, const TokScratch32 &ts, int sbr, int sbc, int b, const Sum32 *sums) {
  static const uint8_t kCtx[13] = { 0, 1, 2, 3, 4, 4, 4, 4, 3, 0, 1, 2, 0 };     // Intra_Mode_Context
  const int w32 = f.w8 / 4, by = b >> 1, bx = b & 1;
  const long i = block_index32(f, sbr, sbc, b);
  const bool half = sbc * 8 + 4 >= f.w8;         // the frame ends after the superblock's left 32 columns (width % 64 == 32)
  if (b == 0) {
    tok_lr32(f, k, sbr, sbc);
    if (half) k.split(1, K_PART64);              // no room for a 64-wide block: split_or_vert = 1 (the band's rows are always complete)
    else k.sym(K_PART64, 3);                     // PARTITION_SPLIT
  }
  if (half && bx) return;                        // outside the frame: not coded
  k.sym(K_PART32, 0);                            // PARTITION_NONE
  k.sym(K_SKIP, 0);
  const int ym = f.y_mode[i], uvm = f.uv_mode[i];
  k.sym(K_KF_Y_MODE + kCtx[by ? f.y_mode[i - w32] : 0] * 5 + kCtx[bx ? f.y_mode[i - 1] : 0], ym);
  if (ym >= 1 && ym <= 8) k.sym(K_ANGLE + ym - 1, 3);
  k.sym(K_UV_MODE + ym, uvm);
  if (uvm >= 1 && uvm <= 8) k.sym(K_ANGLE + uvm - 1, 3);
  for (int p = 0; p < 3; p++) {
    const int ac = by ? sums[b - 2].cul[p] : 0, ad = by ? sums[b - 2].dc[p] : 0, lc = bx ? sums[b - 1].cul[p] : 0, ld = bx ? sums[b - 1].dc[p] : 0;
    if (p == 0) tok_coeffs_big<32>(k, ts, false, f.lev_y + i * 1024, ac, ad, lc, ld);
    else tok_coeffs_big<16>(k, ts, true, (p == 1 ? f.lev_u : f.lev_v) + i * 256, ac, ad, lc, ld);
  }
  k.flush();
}

// a block's records -> list words (from index `first`) and grouped entries; pos[K_END][kBlocks32]: the running positions of the
// tile's (slot, block) pairs
AV1_HD void replay_block32(const uint16_t *rec, int nrec, uint16_t *pos, int blk, int first, op_t *list, uint32_t *grouped) {
  int n = first;
  for (int i0 = 0; i0 < nrec; i0 += 8) {        // eight records per load (a dependent 2-byte load per record is a memory round trip each)
    struct alignas(16) R8 { uint32_t w[4]; } q = *reinterpret_cast<const R8 *>(rec + i0);
    AV1_UNROLL
    for (int j = 0; j < 8; j++, n++) {
      if (i0 + j >= nrec) break;
      const unsigned r = (q.w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
      if (r & 0x8000u) list[n] = op_lit((int)((r >> 11) & 15), r & 0x7FFu);
      else { uint16_t &p = pos[(int)(r >> 4) * kBlocks32 + blk]; grouped[p] = ((uint32_t)n << 4) | (r & 15u); p++; }
    }
  }
}
// a block's records -> its column of the tile's symbol counts (a pass of its own: the counters then share their LDS with the
// magnitude maps, which are dead by then)
AV1_HD bool count_block32(const uint16_t *rec, int nrec, uint16_t *cnt, int blk) {
  bool ok = true;
  for (int i0 = 0; i0 < nrec; i0 += 8) {
    struct alignas(16) R8 { uint32_t w[4]; } q = *reinterpret_cast<const R8 *>(rec + i0);
    AV1_UNROLL
    for (int j = 0; j < 8; j++) {
      if (i0 + j >= nrec) break;
      const unsigned r = (q.w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
      if (!(r & 0x8000u)) { uint16_t &c = cnt[(int)(r >> 4) * kBlocks32 + blk]; if (c == 65535) ok = false; else c++; }
    }
  }
  return ok;
}
// counts[K_END][kBlocks32] -> the positions of every (slot, block)'s first entry (in place), the slots' totals and bases; returns the
// entries incl. the slots' padding to kListAlign
AV1_HD int place_tile32(uint16_t *cnt, uint16_t *total, uint16_t *base) {
  int run = 0;
  for (int sl = 0; sl < K_END; sl++) {
    int p = run, n = 0;
    for (int b = 0; b < kBlocks32; b++) { const int c = cnt[sl * kBlocks32 + b]; cnt[sl * kBlocks32 + b] = (uint16_t)imin(p, 65535); p += c; n += c; }
    total[sl] = (uint16_t)imin(n, 65535); base[sl] = (uint16_t)imin(run, 65535);
    run += (n + kListAlign - 1) & ~(kListAlign - 1);
  }
  return run;
}

}  // namespace av1ops
