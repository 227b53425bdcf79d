// av1_blockstream.cpp — the GENERAL block-structured AV1 tile writer behind av1mi_obu_write_blocks_temporal_unit
// (include/av1mi_host.h): every block size 4x4..64x64, every partition type, TX_MODE_LARGEST / TX_MODE_SELECT with every
// transform size and all 16 transform types, the interpolation filters.  Same role as av1_bitstream.cpp (SURVEY.md §8a row H1:
// the stream the reference's FFmpeg child emits for `-c:v:0 av1_vaapi`, internal/ffmpeg/transcode.go:120); that file is the
// fast writer for the 8x8 tool set of today's GPU pipeline, this one states the whole block layer so that the dav1d pin of the
// oracle (tests/test_av1_blocks.py) reaches every size north_star names ("DCT/ADST 4x4-64x64", every intra / MC block size).
// Written from the AV1 Bitstream & Decoding Process Specification; section numbers in the comments are the specification's.
#include <map>
#include <thread>

#include "av1_bitstream_core.hpp"

namespace av1mi_host {
namespace av1 {
namespace {
using namespace core;

// ------------------------------------------------------------------------------------------------ block / transform size tables
enum { BLOCK_4X4, BLOCK_4X8, BLOCK_8X4, BLOCK_8X8, BLOCK_8X16, BLOCK_16X8, BLOCK_16X16, BLOCK_16X32, BLOCK_32X16, BLOCK_32X32, BLOCK_32X64,
       BLOCK_64X32, BLOCK_64X64, BLOCK_64X128, BLOCK_128X64, BLOCK_128X128, BLOCK_4X16, BLOCK_16X4, BLOCK_8X32, BLOCK_32X8, BLOCK_16X64,
       BLOCK_64X16, BLOCK_SIZES };
enum { TX_4X4, TX_8X8, TX_16X16, TX_32X32, TX_64X64, TX_4X8, TX_8X4, TX_8X16, TX_16X8, TX_16X32, TX_32X16, TX_32X64, TX_64X32, TX_4X16,
       TX_16X4, TX_8X32, TX_32X8, TX_16X64, TX_64X16, TX_SIZES_ALL };
enum { P_NONE, P_HORZ, P_VERT, P_SPLIT, P_HORZ_A, P_HORZ_B, P_VERT_A, P_VERT_B, P_HORZ_4, P_VERT_4 };
enum { T_V_DCT = 10, T_H_DCT, T_V_ADST, T_H_ADST, T_V_FLIPADST, T_H_FLIPADST };
enum { CLASS_2D, CLASS_HORIZ, CLASS_VERT };

const uint8_t kBW4[BLOCK_SIZES] = { 1, 1, 2, 2, 2, 4, 4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 1, 4, 2, 8, 4, 16 };     // Num_4x4_Blocks_Wide
const uint8_t kBH4[BLOCK_SIZES] = { 1, 2, 1, 2, 4, 2, 4, 8, 4, 8, 16, 8, 16, 32, 16, 32, 4, 1, 8, 2, 16, 4 };     // Num_4x4_Blocks_High
const uint8_t kMaxTxRect[BLOCK_SIZES] = { TX_4X4, TX_4X8, TX_8X4, TX_8X8, TX_8X16, TX_16X8, TX_16X16, TX_16X32, TX_32X16, TX_32X32, TX_32X64, TX_64X32,
                                          TX_64X64, TX_64X64, TX_64X64, TX_64X64, TX_4X16, TX_16X4, TX_8X32, TX_32X8, TX_16X64, TX_64X16 };
const uint8_t kMaxTxDepth[BLOCK_SIZES] = { 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 4, 4, 2, 2, 3, 3, 4, 4 };
const uint8_t kSizeGroup[BLOCK_SIZES] = { 0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 0, 0, 1, 1, 2, 2 };
const uint8_t kSubsampled[BLOCK_SIZES] = { BLOCK_4X4, BLOCK_4X4, BLOCK_4X4, BLOCK_4X4, BLOCK_4X8, BLOCK_8X4, BLOCK_8X8, BLOCK_8X16, BLOCK_16X8, BLOCK_16X16,
                                           BLOCK_16X32, BLOCK_32X16, BLOCK_32X32, BLOCK_32X64, BLOCK_64X32, BLOCK_64X64, BLOCK_4X8, BLOCK_8X4, BLOCK_4X16,
                                           BLOCK_16X4, BLOCK_8X32, BLOCK_32X8 };      // Subsampled_Size[..][1][1] (4:2:0)
const uint8_t kTxW[TX_SIZES_ALL] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
const uint8_t kTxH[TX_SIZES_ALL] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };
const uint8_t kSplitTx[TX_SIZES_ALL] = { TX_4X4, TX_4X4, TX_8X8, TX_16X16, TX_32X32, TX_4X4, TX_4X4, TX_8X8, TX_8X8, TX_16X16, TX_16X16, TX_32X32, TX_32X32,
                                         TX_4X8, TX_8X4, TX_8X16, TX_16X8, TX_16X32, TX_32X16 };
const uint8_t kTxSqr[TX_SIZES_ALL] = { 0, 1, 2, 3, 4, 0, 0, 1, 1, 2, 2, 3, 3, 0, 0, 1, 1, 2, 2 };       // Tx_Size_Sqr
const uint8_t kTxSqrUp[TX_SIZES_ALL] = { 0, 1, 2, 3, 4, 1, 1, 2, 2, 3, 3, 4, 4, 2, 2, 3, 3, 4, 4 };     // Tx_Size_Sqr_Up
// symbol of a transform type inside each set (inverse of Tx_Type_Intra_Inv_Set1/2, Tx_Type_Inter_Inv_Set1/2/3, spec 5.11.47); -1 = not in the set
const int8_t kInterSet2Sym[16] = { 3, 4, 5, 8, 6, 7, 9, 10, 11, 0, 1, 2, -1, -1, -1, -1 };
const int8_t kInterSet3Sym[16] = { 1, -1, -1, -1, -1, -1, -1, -1, -1, 0, -1, -1, -1, -1, -1, -1 };

inline int log2i(int v) { return floor_log2((uint32_t)v); }
int bsize_of(int w4, int h4) {
  for (int b = 0; b < BLOCK_SIZES; b++) if (kBW4[b] == w4 && kBH4[b] == h4) return b;
  return -1;
}
inline int tx_class_of(int t) {
  return (t == T_V_DCT || t == T_V_ADST || t == T_V_FLIPADST) ? CLASS_VERT : (t == T_H_DCT || t == T_H_ADST || t == T_H_FLIPADST) ? CLASS_HORIZ : CLASS_2D;
}
// get_tx_set (5.11.48): 0 DCT only, 1 INTRA_1, 2 INTRA_2, 3 INTER_1, 4 INTER_2, 5 INTER_3
int tx_set_of(int tx, bool is_inter, bool reduced) {
  const int sqr = kTxSqr[tx], up = kTxSqrUp[tx];
  if (up > TX_32X32) return 0;
  if (is_inter) return (reduced || up == TX_32X32) ? 5 : sqr == TX_16X16 ? 4 : 3;
  if (up == TX_32X32) return 0;
  return (reduced || sqr == TX_16X16) ? 2 : 1;
}

// get_scan (5.11.41): positions pos = row * tw + col of the (at most 32 x 32) coded area; kind 0 default, 1 row-major (mrow), 2 column-major (mcol)
const std::vector<uint16_t> &scan_of(int tw, int th, int kind) {
  static thread_local std::map<int, std::vector<uint16_t>> cache;      // (per thread: tiles are written by several)
  const int key = (tw << 16) | (th << 4) | kind;
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  std::vector<uint16_t> s;
  s.reserve((size_t)tw * th);
  if (kind == 1) {
    for (int i = 0; i < tw * th; i++) s.push_back((uint16_t)i);
  } else if (kind == 2) {
    for (int c = 0; c < tw; c++) for (int r = 0; r < th; r++) s.push_back((uint16_t)(r * tw + c));
  } else {
    for (int d = 0; d < tw + th - 1; d++) {
      // square: zig-zag (odd diagonals run downwards, even ones upwards); tall: every diagonal downwards; wide: every diagonal upwards
      const bool down = tw == th ? (d & 1) : th > tw;
      for (int i = 0; i <= d; i++) {
        const int r = down ? i : d - i, c = d - r;
        if (r < th && c < tw) s.push_back((uint16_t)(r * tw + c));
      }
    }
  }
  return cache.emplace(key, std::move(s)).first->second;
}

struct Mi {        // what later blocks' contexts read of a 4x4 unit
  uint8_t bsize, skip, is_inter, y_mode, tx, filt, decoded, txtype;
};

struct BlockWriter {
  const av1mi_obu_blocks &d;
  const av1mi_obu_frame &f;
  FrameInfo fi;
  std::string *err;
  std::vector<Mi> mi;
  size_t next_part = 0, next_block = 0;
  bool failed = false;
  // per tile
  RangeEnc ec;
  Cdfs cdf;
  bool adapt = true;
  int mi_r0 = 0, mi_r1 = 0, mi_c0 = 0, mi_c1 = 0;        // tile bounds in 4x4 units
  std::vector<uint8_t> a_lvl[3], a_dc[3];                // Above{Level,Dc}Context per plane: 4-sample units of the plane, tile relative
  uint8_t l_lvl[3][32], l_dc[3][32];                     // Left...: superblock relative
  int ref_wiener[3][2][3], ref_sgr[3][2];
  bool cdef_coded = false;

  BlockWriter(const av1mi_obu_blocks &d_, std::string *e) : d(d_), f(d_.hdr), fi(frame_info(d_.hdr)), err(e), cdf(default_cdfs(fi.qcat)) {
    fi.tx_mode_select = d.tx_mode_select;
    fi.interp_filter = d.interp_filter;
    fi.high_precision_mv = d.high_precision_mv;
    mi.assign((size_t)fi.mi_rows * fi.mi_cols, Mi{});
  }
  bool fail(const char *m) { if (!failed && err) *err = m; failed = true; return false; }
  inline void sym(uint16_t *icdf, int n, int s) { put_symbol(ec, icdf, n, s, adapt); }
  inline Mi &at(int r, int c) { return mi[(size_t)r * fi.mi_cols + c]; }
  inline bool inside(int r, int c) const { return r >= mi_r0 && r < mi_r1 && c >= mi_c0 && c < mi_c1; }      // is_inside (5.11.51)

  // ---- decode_tile (5.11.2)
  void tile(int tr, int tc) {
    mi_r0 = tr * fi.tile_h_sb * 16; mi_r1 = std::min(mi_r0 + fi.tile_h_sb * 16, fi.mi_rows);
    mi_c0 = tc * fi.tile_w_sb * 16; mi_c1 = std::min(mi_c0 + fi.tile_w_sb * 16, fi.mi_cols);
    ec = RangeEnc();
    cdf = default_cdfs(fi.qcat);
    adapt = !f.disable_cdf_update;
    for (int p = 0; p < 3; p++) {
      a_lvl[p].assign((size_t)(mi_c1 - mi_c0) + 48, 0); a_dc[p].assign((size_t)(mi_c1 - mi_c0) + 48, 0);      // clear_above_context
      for (int k = 0; k < 2; k++) { ref_wiener[p][k][0] = 3; ref_wiener[p][k][1] = -7; ref_wiener[p][k][2] = 15; }
      ref_sgr[p][0] = -32; ref_sgr[p][1] = 31;
    }
    for (int r = mi_r0; r < mi_r1 && !failed; r += 16) {
      memset(l_lvl, 0, sizeof(l_lvl)); memset(l_dc, 0, sizeof(l_dc));      // clear_left_context
      for (int c = mi_c0; c < mi_c1 && !failed; c += 16) {
        cdef_coded = false;
        write_lr(r, c);
        partition(r, c, BLOCK_64X64);
      }
    }
    ec.finish();
  }

  // ---- loop restoration units of a superblock (5.11.57, 5.11.58): as av1_bitstream.cpp
  void put_ns(int n, int v) {
    const int w = floor_log2((uint32_t)n) + 1, m = (1 << w) - n;
    if (v < m) ec.literal((uint32_t)v, w - 1);
    else { ec.literal((uint32_t)((v + m) >> 1), w - 1); ec.literal((uint32_t)((v + m) & 1), 1); }
  }
  void put_subexp(int num_syms, int k, int v) {
    int i = 0, mk = 0;
    for (;;) {
      const int b2 = i ? k + i - 1 : k, a = 1 << b2;
      if (num_syms <= mk + 3 * a) { put_ns(num_syms - mk, v - mk); return; }
      const int more = v >= mk + a;
      ec.literal((uint32_t)more, 1);
      if (!more) { ec.literal((uint32_t)(v - mk), b2); return; }
      i++; mk += a;
    }
  }
  static int recenter(int r, int v) { return v > 2 * r ? v : v >= r ? 2 * (v - r) : 2 * (r - v) - 1; }
  void put_signed_subexp_with_ref(int v, int low, int high, int k, int r) {
    const int mx = high - low; v -= low; r -= low;
    put_subexp(mx, k, (r << 1) <= mx ? recenter(r, v) : recenter(mx - 1 - r, mx - 1 - v));
  }
  void write_lr(int mi_r, int mi_c) {
    for (int p = 0; p < 3; p++) {
      if (!f.lr_type[p]) continue;
      const int ss = p ? 1 : 0, us = fi.lr_size[p];
      const int row0 = (mi_r * (4 >> ss) + us - 1) / us, row1 = std::min(((mi_r + 16) * (4 >> ss) + us - 1) / us, fi.lr_rows[p]);
      const int col0 = (mi_c * (4 >> ss) + us - 1) / us, col1 = std::min(((mi_c + 16) * (4 >> ss) + us - 1) / us, fi.lr_cols[p]);
      for (int ur = row0; ur < row1; ur++)
        for (int uc = col0; uc < col1; uc++) lr_unit(p, f.lr_units[p] + ((size_t)ur * fi.lr_cols[p] + uc) * 8);
    }
  }
  void lr_unit(int p, const int8_t *u) {
    const int type = u[0];
    if (f.lr_type[p] == 1) sym(cdf.use_wiener, 2, type == 1);
    else if (f.lr_type[p] == 2) sym(cdf.use_sgrproj, 2, type == 2);
    else sym(cdf.switchable_restore, 3, type);
    if (type == 1 && f.lr_type[p] != 2) {
      static const int kMin[3] = { -5, -23, -17 }, kMax[3] = { 10, 8, 46 }, kK[3] = { 1, 2, 3 };
      for (int pass = 0; pass < 2; pass++)
        for (int j = p ? 1 : 0; j < 3; j++) {
          const int v = u[1 + pass * 3 + j];
          put_signed_subexp_with_ref(v, kMin[j], kMax[j] + 1, kK[j], ref_wiener[p][pass][j]);
          ref_wiener[p][pass][j] = v;
        }
    } else if (type == 2 && f.lr_type[p] != 1) {
      static const int8_t kRadius[16][2] = { { 2, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 },
                                             { 0, 1 }, { 0, 1 }, { 0, 1 }, { 0, 1 }, { 2, 0 }, { 2, 0 } };
      static const int kMin[2] = { -96, -32 }, kMax[2] = { 31, 95 };
      const int set = u[1];
      ec.literal((uint32_t)set, 4);
      for (int i = 0; i < 2; i++) {
        int v = u[2 + i];
        if (kRadius[set][i]) put_signed_subexp_with_ref(v, kMin[i], kMax[i] + 1, 4, ref_sgr[p][i]);
        else v = i == 0 ? 0 : std::min(std::max(128 - ref_sgr[p][0], kMin[1]), kMax[1]);
        ref_sgr[p][i] = v;
      }
    }
  }

  // ---- decode_partition (5.11.4)
  void partition(int r, int c, int bsize) {
    if (failed || r >= fi.mi_rows || c >= fi.mi_cols) return;
    const int n4 = kBW4[bsize], half = n4 >> 1, quarter = half >> 1;
    const bool has_rows = r + half < fi.mi_rows, has_cols = c + half < fi.mi_cols;
    if (next_part >= d.n_partition) { fail("partition list too short"); return; }
    const int part = d.partition[next_part++];
    const bool au = inside(r - 1, c), al = inside(r, c - 1);
    if (bsize == BLOCK_8X8 ? part > P_SPLIT : (part > P_VERT_4)) { fail("partition type out of range"); return; }
    {
      // ctx (9.3): a neighbour narrower / lower than this block
      const int bsl = log2i(n4);
      const int above = au && log2i(kBW4[at(r - 1, c).bsize]) < bsl, left = al && log2i(kBH4[at(r, c - 1).bsize]) < bsl;
      const int pctx = left * 2 + above;
      uint16_t *pc = bsize == BLOCK_8X8 ? cdf.part8[pctx] : bsize == BLOCK_16X16 ? cdf.part16[pctx] : bsize == BLOCK_32X32 ? cdf.part32[pctx] : cdf.part64[pctx];
      const int nsym = bsize == BLOCK_8X8 ? 4 : 10;
      if (has_rows && has_cols) {
        sym(pc, nsym, part);
      } else if (has_rows || has_cols) {
        // split_or_horz / split_or_vert: the probability of "split" gathers every partition type that splits the missing way (9.3)
        auto prob = [&](int k) { return (uint32_t)((k ? pc[k - 1] : 32768) - pc[k]); };
        uint32_t psum;
        if (has_cols) {       // split_or_horz
          if (part != P_HORZ && part != P_SPLIT) { fail("partition at the bottom frame edge must be HORZ or SPLIT"); return; }
          psum = prob(P_VERT) + prob(P_SPLIT);
          if (bsize != BLOCK_8X8) psum += prob(P_HORZ_A) + prob(P_VERT_A) + prob(P_VERT_B) + prob(P_VERT_4);
        } else {              // split_or_vert
          if (part != P_VERT && part != P_SPLIT) { fail("partition at the right frame edge must be VERT or SPLIT"); return; }
          psum = prob(P_HORZ) + prob(P_SPLIT);
          if (bsize != BLOCK_8X8) psum += prob(P_HORZ_A) + prob(P_HORZ_B) + prob(P_VERT_A) + prob(P_HORZ_4);
        }
        // a two-symbol cdf { 32768 - psum, 32768 } without adaptation; symbol 1 = split
        if (part == P_SPLIT) ec.encode(psum, 0, 1, 2);
        else ec.encode(32768, psum, 0, 2);
      } else if (part != P_SPLIT) { fail("partition at the frame corner must be SPLIT"); return; }
    }
    const int sq = bsize_of(half, half);      // Partition_Subsize[PARTITION_SPLIT]
    auto blk = [&](int rr, int cc, int w4, int h4) { block(rr, cc, bsize_of(w4, h4)); };
    switch (part) {
      case P_NONE: blk(r, c, n4, n4); break;
      case P_HORZ: blk(r, c, n4, half); if (has_rows) blk(r + half, c, n4, half); break;
      case P_VERT: blk(r, c, half, n4); if (has_cols) blk(r, c + half, half, n4); break;
      case P_SPLIT:
        if (bsize == BLOCK_8X8) { blk(r, c, 1, 1); blk(r, c + 1, 1, 1); blk(r + 1, c, 1, 1); blk(r + 1, c + 1, 1, 1); }
        else { partition(r, c, sq); partition(r, c + half, sq); partition(r + half, c, sq); partition(r + half, c + half, sq); }
        break;
      case P_HORZ_A: blk(r, c, half, half); blk(r, c + half, half, half); blk(r + half, c, n4, half); break;
      case P_HORZ_B: blk(r, c, n4, half); blk(r + half, c, half, half); blk(r + half, c + half, half, half); break;
      case P_VERT_A: blk(r, c, half, half); blk(r + half, c, half, half); blk(r, c + half, half, n4); break;
      case P_VERT_B: blk(r, c, half, n4); blk(r, c + half, half, half); blk(r + half, c + half, half, half); break;
      case P_HORZ_4: for (int i = 0; i < 4; i++) if (i < 3 || r + quarter * 3 < fi.mi_rows) blk(r + quarter * i, c, n4, quarter); break;
      case P_VERT_4: for (int i = 0; i < 4; i++) if (i < 3 || c + quarter * 3 < fi.mi_cols) blk(r, c + quarter * i, quarter, n4); break;
    }
  }

  // ---- decode_block (5.11.5)
  void block(int r, int c, int bsize) {
    if (failed) return;
    if (bsize < 0) { fail("partition yields an invalid block size"); return; }
    if (next_block >= d.n_blocks) { fail("block list too short"); return; }
    const av1mi_obu_block &b = d.blocks[next_block++];
    if (b.mi_row != r || b.mi_col != c || b.bsize != bsize) { fail("block list does not follow the partition tree"); return; }
    const int bw4 = kBW4[bsize], bh4 = kBH4[bsize];
    const bool au = inside(r - 1, c), al = inside(r, c - 1);
    const bool has_chroma = ((c & 1) || !(bw4 & 1)) && ((r & 1) || !(bh4 & 1));
    const int is_inter = fi.key ? 0 : b.is_inter;
    if (b.skip > 1 || b.tx_depth > 2 || (!d.tx_mode_select && b.tx_depth)) { fail("bad skip / tx_depth"); return; }
    if (!is_inter && (b.y_mode > PAETH_PRED || b.uv_mode > UV_CFL_PRED)) { fail("intra mode out of range"); return; }
    // ---- mode info: intra_frame_mode_info (5.11.7) / inter_frame_mode_info (5.11.18)
    sym(cdf.skip[(au ? at(r - 1, c).skip : 0) + (al ? at(r, c - 1).skip : 0)], 2, b.skip);
    if (!b.skip && !cdef_coded) {      // read_cdef (5.11.56): with the first non-skipped block of the 64x64
      const int sb = (r >> 4) * fi.sb_cols + (c >> 4);
      ec.literal(f.cdef_idx ? f.cdef_idx[sb] : 0, f.cdef_bits);
      cdef_coded = true;
    }
    if (!fi.key) {
      const bool ai = au ? !at(r - 1, c).is_inter : false, li = al ? !at(r, c - 1).is_inter : false;
      int ctx;
      if (au && al) ctx = (li && ai) ? 3 : (li || ai);
      else if (au || al) ctx = 2 * (au ? ai : li);
      else ctx = 0;
      sym(cdf.is_inter[ctx], 2, is_inter);
    }
    if (!is_inter) {
      const int ym = b.y_mode;
      if (fi.key) {
        const int actx = kIntraModeContext[au ? at(r - 1, c).y_mode : (int)DC_PRED], lctx = kIntraModeContext[al ? at(r, c - 1).y_mode : (int)DC_PRED];
        sym(cdf.kf_y_mode[actx][lctx], 13, ym);
      } else {
        sym(cdf.y_mode[kSizeGroup[bsize]], 13, ym);      // intra_block_mode_info (5.11.22)
      }
      const bool angles = bsize >= BLOCK_8X8;            // (the enumeration's order: 4x16 and 16x4 code angle deltas too)
      if (angles && is_directional(ym)) sym(cdf.angle_delta[ym - V_PRED], 7, b.angle_y + 3);
      if (has_chroma) {
        const bool cfl_allowed = std::max(bw4, bh4) <= 8;
        const int uvm = b.uv_mode;
        if (uvm == UV_CFL_PRED && !cfl_allowed) { fail("chroma from luma needs a block of at most 32x32"); return; }
        if (cfl_allowed) sym(cdf.uv_mode_cfl[ym], 14, uvm);
        else sym(cdf.uv_mode_nocfl[ym], 13, uvm);
        if (uvm == UV_CFL_PRED) {
          const int au_ = b.cfl_alpha_u, av_ = b.cfl_alpha_v;
          if ((!au_ && !av_) || std::abs(au_) > 16 || std::abs(av_) > 16) { fail("chroma-from-luma alphas out of range"); return; }
          const int su = au_ == 0 ? 0 : au_ < 0 ? 1 : 2, sv = av_ == 0 ? 0 : av_ < 0 ? 1 : 2;
          sym(cdf.cfl_sign, 8, su * 3 + sv - 1);
          if (su) sym(cdf.cfl_alpha[(su - 1) * 3 + sv], 16, std::abs(au_) - 1);
          if (sv) sym(cdf.cfl_alpha[(sv - 1) * 3 + su], 16, std::abs(av_) - 1);
        } else if (angles && is_directional(uvm)) {
          sym(cdf.angle_delta[uvm - V_PRED], 7, b.angle_uv + 3);
        }
      }
      // palette: allow_screen_content_tools = 0; filter intra: off in the sequence header
    } else {
      inter_block_mode_info(b, r, c, bsize, au, al);
      if (failed) return;
    }
    // ---- read_block_tx_size (5.11.15)
    const int max_tx = kMaxTxRect[bsize];
    int tx = max_tx;
    if (d.tx_mode_select && bsize > BLOCK_4X4 && is_inter && !b.skip) {
      const int tw4 = kTxW[max_tx] >> 2, th4 = kTxH[max_tx] >> 2;
      for (int rr = r; rr < r + bh4; rr += th4)
        for (int cc = c; cc < c + bw4; cc += tw4) var_tx(b, bsize, rr, cc, max_tx, 0);
      for (int i = 0; i < b.tx_depth; i++) tx = kSplitTx[tx];
    } else {
      if (d.tx_mode_select && bsize > BLOCK_4X4 && (!b.skip || !is_inter)) {      // read_tx_size(allowSelect): tx_depth
        const int max_depth = kMaxTxDepth[bsize], nsym = max_depth > 1 ? 3 : 2;
        if (b.tx_depth >= nsym) { fail("tx_depth beyond what the block size allows"); return; }
        const int max_w = kTxW[max_tx], max_h = kTxH[max_tx];
        int above_w = 0, left_h = 0;
        if (au) above_w = at(r - 1, c).is_inter ? kBW4[at(r - 1, c).bsize] * 4 : above_tx_width(r, c, r);
        if (al) left_h = at(r, c - 1).is_inter ? kBH4[at(r, c - 1).bsize] * 4 : left_tx_height(r, c, c);
        const int ctx = (above_w >= max_w) + (left_h >= max_h);
        uint16_t *tc = max_depth == 4 ? cdf.tx64[ctx] : max_depth == 3 ? cdf.tx32[ctx] : max_depth == 2 ? cdf.tx16[ctx] : cdf.tx8[ctx];
        sym(tc, nsym, b.tx_depth);
        for (int i = 0; i < b.tx_depth; i++) tx = kSplitTx[tx];
      } else if (b.tx_depth) { fail("tx_depth on a block that cannot code it"); return; }
    }
    // the maps later blocks read
    for (int rr = r; rr < std::min(r + bh4, fi.mi_rows); rr++)
      for (int cc = c; cc < std::min(c + bw4, fi.mi_cols); cc++) {
        Mi &m = at(rr, cc);
        m.bsize = (uint8_t)bsize; m.skip = b.skip; m.is_inter = (uint8_t)is_inter; m.y_mode = is_inter ? (uint8_t)DC_PRED : b.y_mode; m.tx = (uint8_t)tx;
        m.filt = b.interp_filter; m.decoded = 1;
      }
    if (b.skip) { reset_block_context(r, c, bw4, bh4, has_chroma); return; }
    residual(b, r, c, bsize, tx, is_inter, has_chroma);
  }

  // get_above_tx_width / get_left_tx_height (9.3); r0 / c0: the block's first row / column
  int above_tx_width(int r, int c, int r0) {
    if (r == r0) {
      if (!inside(r - 1, c)) return 64;
      const Mi &m = at(r - 1, c);
      if (m.skip && m.is_inter) return kBW4[m.bsize] * 4;
    }
    return kTxW[at(r - 1, c).tx];
  }
  int left_tx_height(int r, int c, int c0) {
    if (c == c0) {
      if (!inside(r, c - 1)) return 64;
      const Mi &m = at(r, c - 1);
      if (m.skip && m.is_inter) return kBH4[m.bsize] * 4;
    }
    return kTxH[at(r, c - 1).tx];
  }
  // read_var_tx_size (5.11.17), every branch split down to the block's tx_depth
  void var_tx(const av1mi_obu_block &b, int bsize, int r, int c, int tx, int depth) {
    if (r >= fi.mi_rows || c >= fi.mi_cols) return;
    const int w4 = kTxW[tx] >> 2, h4 = kTxH[tx] >> 2;
    int split = 0;
    if (tx != TX_4X4 && depth < 2) {
      split = depth < b.tx_depth;
      const int above = above_tx_width(r, c, b.mi_row) < kTxW[tx], left = left_tx_height(r, c, b.mi_col) < kTxH[tx];
      const int max_sqr = kTxSqrUp[kMaxTxRect[bsize]];      // the square transform of the block's larger dimension
      const int cat = (kTxSqrUp[tx] != max_sqr && max_sqr > TX_8X8) + (TX_64X64 - max_sqr) * 2;
      sym(cdf.txfm_split[cat * 3 + above + left], 2, split);
    } else if (depth < b.tx_depth) { fail("tx_depth beyond what the transform size allows"); return; }
    if (split) {
      const int sub = kSplitTx[tx], sw = kTxW[sub] >> 2, sh = kTxH[sub] >> 2;
      for (int i = 0; i < h4; i += sh) for (int j = 0; j < w4; j += sw) var_tx(b, bsize, r + i, c + j, sub, depth + 1);
    } else {
      for (int i = r; i < std::min(r + h4, fi.mi_rows); i++) for (int j = c; j < std::min(c + w4, fi.mi_cols); j++) at(i, j).tx = (uint8_t)tx;      // InterTxSizes
    }
  }

  // ---- inter_block_mode_info (5.11.23): LAST_FRAME, NEWMV against an empty prediction list
  void inter_block_mode_info(const av1mi_obu_block &b, int r, int c, int bsize, bool au, bool al) {
    const int bw4 = kBW4[bsize], bh4 = kBH4[bsize];
    // the MV prediction scan (7.10.2) reaches rows r - 5 .. r + bh4 - 1 and columns c - 5 .. c + bw4 of the tile: no decoded inter block there
    for (int rr = std::max(r - 6, mi_r0); rr < std::min(r + bh4 + 1, mi_r1); rr++)
      for (int cc = std::max(c - 6, mi_c0); cc < std::min(c + bw4 + 2, mi_c1); cc++)
        if (at(rr, cc).decoded && at(rr, cc).is_inter) { fail("inter block within the MV prediction reach of another one (this writer codes isolated inter blocks)"); return; }
    // read_ref_frames (5.11.25): LAST_FRAME; no neighbour is inter, so every count is 0 and every context 1
    (void)au; (void)al;
    sym(cdf.single_ref[1][0], 2, 0);
    sym(cdf.single_ref[1][2], 2, 0);
    sym(cdf.single_ref[1][3], 2, 0);
    // empty list: NumMvFound = 0, NewMvContext = 0, the predictor is the global vector (0, 0)
    sym(cdf.new_mv[0], 2, 0);        // NEWMV
    const int dx = b.mv_x, dy = b.mv_y;
    if ((!d.high_precision_mv && ((dx & 1) || (dy & 1))) || std::abs(dx) >= (1 << 13) || std::abs(dy) >= (1 << 13)) { fail("vector must be below 2^13 and, without high_precision_mv, a multiple of 2"); return; }
    sym(cdf.mv_joint, 4, (dx ? 1 : 0) + (dy ? 2 : 0));
    if (dy) write_mv_comp(cdf.mv[0], dy);
    if (dx) write_mv_comp(cdf.mv[1], dx);
    if (d.interp_filter == 4) {      // interp_filter[0] (dual filter off): no inter neighbour, so both neighbour types are "none" -> ctx 3
      if (b.interp_filter > 2) { fail("interp_filter out of range"); return; }
      sym(cdf.interp_filter[3], 3, b.interp_filter);
    }
  }
  void write_mv_comp(MvCompCdf &m, int diff) {      // read_mv_component (5.11.33)
    sym(m.sign, 2, diff < 0);
    const int off = std::abs(diff) - 1;
    const int cls = (off >> 3) < 2 ? 0 : floor_log2((uint32_t)(off >> 3));
    sym(m.cls, 11, cls);
    if (cls == 0) {
      sym(m.class0, 2, off >> 3);
      sym(m.class0_fr[off >> 3], 4, (off >> 1) & 3);
      if (d.high_precision_mv) sym(m.class0_hp, 2, off & 1);
    } else {
      const int o = off - (2 << (cls + 2)), dd = o >> 3;
      for (int i = 0; i < cls; i++) sym(m.bits[i], 2, (dd >> i) & 1);
      sym(m.fr, 4, (o >> 1) & 3);
      if (d.high_precision_mv) sym(m.hp, 2, o & 1);
    }
    // (without allow_high_precision_mv the eighth-sample bit is implied 1: diff is even, off odd)
  }

  // ---- reset_block_context (5.11.6)
  void reset_block_context(int r, int c, int bw4, int bh4, bool has_chroma) {
    for (int p = 0; p < (has_chroma ? 3 : 1); p++) {
      const int ss = p ? 1 : 0;
      const int x0 = (c >> ss) - (mi_c0 >> ss), y0 = (r >> ss) & (15 >> ss), nw = (bw4 + ss) >> ss, nh = (bh4 + ss) >> ss;
      for (int i = 0; i < nw; i++) a_lvl[p][(size_t)x0 + i] = a_dc[p][(size_t)x0 + i] = 0;
      for (int i = 0; i < nh; i++) l_lvl[p][y0 + i] = l_dc[p][y0 + i] = 0;
    }
  }

  // ---- residual (5.11.34): blocks are at most 64x64, i.e. one chunk
  void residual(const av1mi_obu_block &b, int r, int c, int bsize, int tx, int is_inter, bool has_chroma) {
    size_t type_i = b.tx_type_off, lev_i[3] = { b.lev_off[0], b.lev_off[1], b.lev_off[2] };
    for (int p = 0; p < (has_chroma ? 3 : 1); p++) {
      const int ss = p ? 1 : 0;
      const int pbs = p ? kSubsampled[bsize] : bsize, ptx = p ? kMaxTxRect[pbs] : tx;      // get_tx_size (5.11.37): 4:2:0 chroma never reaches 64
      const int n4w = kBW4[pbs], n4h = kBH4[pbs], step_x = kTxW[ptx] >> 2, step_y = kTxH[ptx] >> 2;
      const int base_x4 = c >> ss, base_y4 = r >> ss;      // in units of 4 samples of the plane
      const int max_x4 = (fi.mi_cols + ss) >> ss, max_y4 = (fi.mi_rows + ss) >> ss;
      auto tb = [&](int x4, int y4, int t) {
        if (x4 >= max_x4 || y4 >= max_y4) return;        // transform_block (5.11.35): starts outside the frame
        int type = T_DCT_DCT;
        if (p == 0) {
          type = d.tx_type ? d.tx_type[type_i] : (int)T_DCT_DCT;
          type_i++;
        } else if (is_inter) {
          // compute_tx_type (5.11.40): the luma type at the transform block's position, where the chroma size's set holds it.  (Intra
          // chroma types follow the mode, Mode_To_Txfm: always of the 2-D class, which is all the tile syntax needs of them.)
          const int lt = at(std::max(r, y4 << 1), std::max(c, x4 << 1)).txtype, cs = tx_set_of(t, true, f.reduced_tx_set != 0);
          if (cs == 3 || (cs == 4 && kInterSet2Sym[lt] >= 0) || (cs == 5 && kInterSet3Sym[lt] >= 0)) type = lt;
        }
        const int tw = std::min<int>(kTxW[t], 32), th = std::min<int>(kTxH[t], 32);
        coeffs(p, x4, y4, t, pbs, d.levels + lev_i[p], is_inter, type, b.y_mode);
        lev_i[p] += (size_t)tw * th;
      };
      if (p == 0 && is_inter) {
        transform_tree(base_x4, base_y4, n4w, n4h, tb);
      } else {
        for (int y = 0; y < n4h; y += step_y) for (int x = 0; x < n4w; x += step_x) tb(base_x4 + x, base_y4 + y, ptx);
      }
    }
  }
  // transform_tree (5.11.36): the luma transform blocks of an inter block in quad-tree order
  template <class F> void transform_tree(int x4, int y4, int w4, int h4, F &tb) {
    if (x4 >= fi.mi_cols || y4 >= fi.mi_rows) return;
    const int ltx = at(y4, x4).tx, lw = kTxW[ltx] >> 2, lh = kTxH[ltx] >> 2;
    if (w4 <= lw && h4 <= lh) {
      int t = 0;
      for (; t < TX_SIZES_ALL; t++) if ((kTxW[t] >> 2) == w4 && (kTxH[t] >> 2) == h4) break;      // find_tx_size
      tb(x4, y4, t);
    } else if (w4 > h4) { transform_tree(x4, y4, w4 / 2, h4, tb); transform_tree(x4 + w4 / 2, y4, w4 / 2, h4, tb); }
    else if (w4 < h4) { transform_tree(x4, y4, w4, h4 / 2, tb); transform_tree(x4, y4 + h4 / 2, w4, h4 / 2, tb); }
    else {
      transform_tree(x4, y4, w4 / 2, h4 / 2, tb); transform_tree(x4 + w4 / 2, y4, w4 / 2, h4 / 2, tb);
      transform_tree(x4, y4 + h4 / 2, w4 / 2, h4 / 2, tb); transform_tree(x4 + w4 / 2, y4 + h4 / 2, w4 / 2, h4 / 2, tb);
    }
  }

  // ---- coeffs (5.11.39) of one transform block at (x4, y4) of the plane, in units of 4 samples (frame coordinates)
  void coeffs(int plane, int x4, int y4, int tx, int plane_bsize, const int16_t *lev, int is_inter, int tx_type, int y_mode) {
    const int ss = plane ? 1 : 0, ptype = plane > 0;
    const int w4 = kTxW[tx] >> 2, h4 = kTxH[tx] >> 2;
    const int tw = std::min<int>(kTxW[tx], 32), th = std::min<int>(kTxH[tx], 32), nc = tw * th, bwl = log2i(tw);
    const int txs_ctx = (kTxSqr[tx] + kTxSqrUp[tx] + 1) >> 1;
    const int max_x4 = (fi.mi_cols + ss) >> ss, max_y4 = (fi.mi_rows + ss) >> ss;
    const int ax = x4 - (mi_c0 >> ss), ly = y4 & (15 >> ss);      // tile / superblock relative positions of the context arrays
    int set = 0;
    if (plane == 0) {
      set = tx_set_of(tx, is_inter != 0, f.reduced_tx_set != 0);
      const int8_t *tab = set == 1 ? kIntraSet1Sym : set == 2 ? kIntraSet2Sym : set == 3 ? kInterSet1Sym : set == 4 ? kInterSet2Sym : kInterSet3Sym;
      if (tx_type < 0 || tx_type > 15 || (set == 0 ? tx_type != T_DCT_DCT : tab[tx_type] < 0)) { fail("transform type not in the set of this transform size"); return; }
    }
    const int cls = tx_class_of(tx_type);
    const std::vector<uint16_t> &scan = scan_of(tw, th, cls == CLASS_VERT ? 1 : cls == CLASS_HORIZ ? 2 : 0);
    int eob = 0;
    for (int k = nc - 1; k >= 0; k--) if (lev[scan[(size_t)k]]) { eob = k + 1; break; }
    // all_zero context (9.3)
    int ctx;
    {
      const int bw = kBW4[plane_bsize] * 4, bh = kBH4[plane_bsize] * 4, w = kTxW[tx], h = kTxH[tx];
      if (plane == 0) {
        int top = 0, left = 0;
        for (int k = 0; k < w4; k++) if (x4 + k < max_x4) top = std::max<int>(top, a_lvl[0][(size_t)ax + k]);
        for (int k = 0; k < h4; k++) if (y4 + k < max_y4) left = std::max<int>(left, l_lvl[0][ly + k]);
        if (bw == w && bh == h) ctx = 0;
        else if (top == 0 && left == 0) ctx = 1;
        else if (top == 0 || left == 0) ctx = 2 + (std::max(top, left) > 3);
        else if (std::max(top, left) <= 3) ctx = 4;
        else if (std::min(top, left) <= 3) ctx = 5;
        else ctx = 6;
      } else {
        int above = 0, left = 0;
        for (int k = 0; k < w4; k++) if (x4 + k < max_x4) above |= a_lvl[plane][(size_t)ax + k] | a_dc[plane][(size_t)ax + k];
        for (int k = 0; k < h4; k++) if (y4 + k < max_y4) left |= l_lvl[plane][ly + k] | l_dc[plane][ly + k];
        ctx = 7 + (above != 0) + (left != 0);
        if (bw * bh > w * h) ctx += 3;
      }
    }
    sym(cdf.txb_skip[txs_ctx][ctx], 2, eob == 0);
    if (plane == 0)      // TxTypes: what the chroma blocks of an inter block derive their type from (DCT_DCT where nothing is coded)
      for (int i = y4; i < std::min(y4 + h4, fi.mi_rows); i++) for (int j = x4; j < std::min(x4 + w4, fi.mi_cols); j++) at(i, j).txtype = (uint8_t)(eob ? tx_type : (int)T_DCT_DCT);
    int cul = 0, dc_cat = 0;
    if (eob) {
      if (plane == 0 && set > 0) {      // transform_type (5.11.47); base_q_idx > 0
        const int sq = kTxSqr[tx];
        switch (set) {
          case 1: sym(cdf.intra_tx1[sq][y_mode], 7, kIntraSet1Sym[tx_type]); break;
          case 2: sym(cdf.intra_tx2[sq][y_mode], 5, kIntraSet2Sym[tx_type]); break;
          case 3: sym(cdf.inter_tx1[sq], 16, kInterSet1Sym[tx_type]); break;
          case 4: sym(cdf.inter_tx2, 12, kInterSet2Sym[tx_type]); break;
          default: sym(cdf.inter_tx3[sq], 2, kInterSet3Sym[tx_type]); break;
        }
      }
      // eob_pt_*, eob_extra, eob_extra_bit
      const int eob_pt = eob < 3 ? eob : floor_log2((uint32_t)(eob - 1)) + 2;
      const int ectx2 = cls == CLASS_2D ? 0 : 1;
      switch (bwl + log2i(th) - 4) {      // eobMultisize
        case 0: sym(cdf.eob16[ptype][ectx2], 5, eob_pt - 1); break;
        case 1: sym(cdf.eob32[ptype][ectx2], 6, eob_pt - 1); break;
        case 2: sym(cdf.eob64[ptype][ectx2], 7, eob_pt - 1); break;
        case 3: sym(cdf.eob128[ptype][ectx2], 8, eob_pt - 1); break;
        case 4: sym(cdf.eob256[ptype][ectx2], 9, eob_pt - 1); break;
        case 5: sym(cdf.eob512[ptype][ectx2], 10, eob_pt - 1); break;
        default: sym(cdf.eob1024[ptype][ectx2], 11, eob_pt - 1); break;
      }
      if (eob_pt >= 3) {
        const int off = eob - ((1 << (eob_pt - 2)) + 1);
        int shift = eob_pt - 3;
        sym(cdf.eob_extra[txs_ctx][ptype][eob_pt - 3], 2, (off >> shift) & 1);
        for (shift--; shift >= 0; shift--) ec.bool_eq((off >> shift) & 1);
      }
      // magnitudes min(|level|, 15) with a zero border of 4 on the right and bottom
      const int MS = tw + 4;
      std::vector<uint8_t> mag((size_t)MS * (th + 4), 0);
      for (int k = 0; k < eob; k++) {
        const int pos = scan[(size_t)k], a = std::abs((int)lev[pos]);
        mag[(size_t)(pos >> bwl) * MS + (pos & (tw - 1))] = (uint8_t)(a > 15 ? 15 : a);
      }
      auto c3 = [](int v) { return v > 3 ? 3 : v; };
      for (int k = eob - 1; k >= 0; k--) {
        const int pos = scan[(size_t)k], row = pos >> bwl, col = pos & (tw - 1);
        const uint8_t *m = mag.data() + (size_t)row * MS + col;
        const int a = std::abs((int)lev[pos]);
        if (k == eob - 1) {
          const int ectx = k == 0 ? 0 : k <= nc / 8 ? 1 : k <= nc / 4 ? 2 : 3;
          sym(cdf.base_eob[txs_ctx][ptype][ectx], 3, (a > 3 ? 3 : a) - 1);
        } else {
          // get_coeff_base_ctx (9.3)
          int mm, bctx;
          if (cls == CLASS_2D) mm = c3(m[1]) + c3(m[MS]) + c3(m[MS + 1]) + c3(m[2]) + c3(m[2 * MS]);
          else if (cls == CLASS_HORIZ) mm = c3(m[1]) + c3(m[MS]) + c3(m[2]) + c3(m[3]) + c3(m[4]);
          else mm = c3(m[1]) + c3(m[MS]) + c3(m[2 * MS]) + c3(m[3 * MS]) + c3(m[4 * MS]);
          bctx = std::min((mm + 1) >> 1, 4);
          if (cls == CLASS_2D) {
            if (pos == 0) bctx = 0;
            else if (kTxW[tx] < kTxH[tx] && row < 2) bctx += 11;     // Coeff_Base_Ctx_Offset[txSz]: tall transforms (32x64 too), first two rows
            else if (kTxW[tx] > kTxH[tx] && col < 2) bctx += 16;     // wide transforms, first two columns
            else bctx += row + col < 2 ? 1 : row + col < 4 ? 6 : 21;
          } else {
            const int idx = cls == CLASS_VERT ? row : col;
            bctx += 26 + 5 * std::min(idx, 2);                       // Coeff_Base_Pos_Ctx_Offset
          }
          sym(cdf.base[txs_ctx][ptype][bctx], 4, a > 3 ? 3 : a);
        }
        if (a > 2) {     // coeff_br
          int mm = m[1] + m[MS] + (cls == CLASS_2D ? m[MS + 1] : cls == CLASS_HORIZ ? m[2] : m[2 * MS]);
          mm = std::min((mm + 1) >> 1, 6);
          int rctx;
          if (pos == 0) rctx = mm;
          else if (cls == CLASS_2D) rctx = (row < 2 && col < 2) ? mm + 7 : mm + 14;
          else if (cls == CLASS_HORIZ) rctx = col == 0 ? mm + 7 : mm + 14;
          else rctx = row == 0 ? mm + 7 : mm + 14;
          int rem = a - 3;
          for (int i = 0; i < 4; i++) {
            const int kk = rem > 3 ? 3 : rem;
            sym(cdf.br[std::min(txs_ctx, 3)][ptype][rctx], 4, kk);
            rem -= kk;
            if (kk < 3) break;
          }
        }
      }
      for (int k = 0; k < eob; k++) {      // signs and Golomb remainders
        const int pos = scan[(size_t)k], v = lev[pos];
        if (!v) continue;
        const int a = std::abs(v);
        if (k == 0) {
          int sg = 0;
          for (int i = 0; i < w4; i++) if (x4 + i < max_x4) { const int ad = a_dc[plane][(size_t)ax + i]; sg += (ad == 2) - (ad == 1); }
          for (int i = 0; i < h4; i++) if (y4 + i < max_y4) { const int ld = l_dc[plane][ly + i]; sg += (ld == 2) - (ld == 1); }
          sym(cdf.dc_sign[ptype][sg < 0 ? 1 : sg > 0 ? 2 : 0], 2, v < 0);
          dc_cat = v < 0 ? 1 : 2;
        } else {
          ec.bool_eq(v < 0);
        }
        if (a > 14) {
          const uint32_t x = (uint32_t)(a - 14);
          const int len = floor_log2(x) + 1;
          ec.literal(0, len - 1);
          ec.literal(x, len);
        }
        cul += a;
      }
      cul = std::min(cul, 63);
    }
    for (int k = 0; k < w4; k++) { a_lvl[plane][(size_t)ax + k] = (uint8_t)cul; a_dc[plane][(size_t)ax + k] = (uint8_t)dc_cat; }
    for (int k = 0; k < h4; k++) { l_lvl[plane][ly + k] = (uint8_t)cul; l_dc[plane][ly + k] = (uint8_t)dc_cat; }
  }
};

}  // namespace

// tile_start (optional): for every tile (+ one entry past the last) the index of its first block and of its first partition symbol —
// a caller that built the lists tile by tile knows them, and the tiles can then be written by `threads` threads (each with its own
// writer state; a tile reads nothing of another tile)
bool blocks_temporal_unit(const av1mi_obu_blocks &d, bool with_sequence_header, std::vector<uint8_t> *out, std::string *err, int threads,
                          const size_t (*tile_start)[2]) {
  using namespace core;
  if (!check(d.hdr, err, false)) return false;
  auto bad = [&](const char *m) { if (err) *err = m; return false; };
  if (!d.blocks || !d.partition || !d.levels) return bad("block description incomplete");
  if (d.tx_mode_select < 0 || d.tx_mode_select > 1 || d.interp_filter < 0 || d.interp_filter > 4) return bad("tx_mode_select / interp_filter out of range");
  BlockWriter bw(d, err);
  const int ntiles = bw.fi.tile_cols * bw.fi.tile_rows;
  std::vector<std::vector<uint8_t>> tiles((size_t)ntiles);
  if (tile_start && threads > 1 && ntiles > 1) {
    const int nt = std::min(threads, ntiles);
    std::vector<std::string> errs((size_t)nt);
    std::vector<char> ok((size_t)nt, 1);
    std::vector<std::thread> pool;
    for (int i = 0; i < nt; i++)
      pool.emplace_back([&, i]() {
        BlockWriter w(d, &errs[(size_t)i]);
        for (int t = i; t < ntiles && !w.failed; t += nt) {
          w.next_block = tile_start[t][0]; w.next_part = tile_start[t][1];
          w.tile(t / w.fi.tile_cols, t % w.fi.tile_cols);
          if (!w.failed && (w.next_block != tile_start[t + 1][0] || w.next_part != tile_start[t + 1][1])) w.fail("a tile's block / partition lists do not end where the next tile's begin");
          tiles[(size_t)t].swap(w.ec.out);
        }
        ok[(size_t)i] = !w.failed;
      });
    for (auto &th : pool) th.join();
    for (int i = 0; i < nt; i++) if (!ok[(size_t)i]) { if (err) *err = errs[(size_t)i]; return false; }
    bw.next_block = tile_start[ntiles][0]; bw.next_part = tile_start[ntiles][1];
  } else {
    for (int t = 0; t < ntiles && !bw.failed; t++) {
      bw.tile(t / bw.fi.tile_cols, t % bw.fi.tile_cols);
      tiles[(size_t)t].swap(bw.ec.out);
    }
  }
  if (bw.failed) return false;
  if (bw.next_block != d.n_blocks || bw.next_part != d.n_partition) return bad("block / partition list longer than the frame");
  std::vector<const uint8_t *> data((size_t)ntiles);
  std::vector<size_t> size((size_t)ntiles);
  for (int t = 0; t < ntiles; t++) { data[(size_t)t] = tiles[(size_t)t].data(); size[(size_t)t] = tiles[(size_t)t].size(); }
  std::vector<uint8_t> fr;
  if (!assemble_frame(bw.fi, data.data(), size.data(), &fr)) return false;
  *out = temporal_delimiter_obu();
  if (with_sequence_header) {
    const std::vector<uint8_t> sh = sequence_header_obu(sequence_params(d.hdr));
    out->insert(out->end(), sh.begin(), sh.end());
  }
  out->insert(out->end(), fr.begin(), fr.end());
  return true;
}

}  // namespace av1
}  // namespace av1mi_host
