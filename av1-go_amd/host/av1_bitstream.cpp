// av1_bitstream.cpp — see av1_bitstream.hpp.  Written from the AV1 Bitstream & Decoding Process Specification; section
// numbers in the comments are the specification's.  The default CDF tables come from av1_default_cdfs.inc (generated,
// tools/extract_av1_cdfs.py).  Conformance is checked by decoding with dav1d (tests/test_av1_conformance.py).
#include "av1_bitstream.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "av1_bitstream_core.hpp"

namespace av1mi_host {
namespace av1 {
namespace {
using namespace core;




struct Scans { uint8_t s4[16], s8[64]; };   // position = row * n + col
const Scans &scans() {   // Default_Scan_4x4 / Default_Scan_8x8 for row-major blocks (row = vertical frequency): 0, 1, 8, 16, 9, 2, 3, 10, ...
  static Scans sc = [] {
    Scans t;
    auto gen = [](int n, uint8_t *o) {
      int k = 0;
      for (int d = 0; d < 2 * n - 1; d++)
        for (int i = 0; i <= d; i++) {
          const int r = (d & 1) ? i : d - i, c = d - r;
          if (r < n && c < n) o[k++] = (uint8_t)(r * n + c);
        }
    };
    gen(4, t.s4); gen(8, t.s8);
    return t;
  }();
  return sc;
}


// ------------------------------------------------------------------------------------------------ one tile
struct MvCand { int16_t x, y; int weight; };

struct TileEnc {
  const FrameInfo &fi;
  const av1mi_obu_frame &f;
  int r8_0, r8_1, c8_0, c8_1;          // tile bounds in 8x8 blocks
  RangeEnc ec;
  Cdfs cdf;
  bool adapt;
  std::vector<uint8_t> a_lvl[3], a_dc[3];   // AboveLevelContext / AboveDcContext, 4-sample units of the plane, tile relative
  uint8_t l_lvl[3][16], l_dc[3][16];        // Left..., superblock relative
  int ref_wiener[3][2][3], ref_sgr[3][2];   // RefLrWiener / RefSgrXqd (5.11.58)
  bool cdef_coded = false;

  TileEnc(const FrameInfo &fi_, int tr, int tc) : fi(fi_), f(*fi_.f), cdf(default_cdfs(fi_.qcat)) {
    r8_0 = tr * fi.tile_h_sb * 8; r8_1 = std::min(r8_0 + fi.tile_h_sb * 8, fi.h8);
    c8_0 = tc * fi.tile_w_sb * 8; c8_1 = std::min(c8_0 + fi.tile_w_sb * 8, fi.w8);
    adapt = !f.disable_cdf_update;
    for (int p = 0; p < 3; p++) {
      const int n = (c8_1 - c8_0) * (p ? 1 : 2) + 4;
      a_lvl[p].assign(n, 0); a_dc[p].assign(n, 0);
      for (int k = 0; k < 2; k++) {
        ref_wiener[p][k][0] = 3; ref_wiener[p][k][1] = -7; ref_wiener[p][k][2] = 15;   // Wiener_Taps_Mid
      }
      ref_sgr[p][0] = -32; ref_sgr[p][1] = 31;                                          // Sgrproj_Xqd_Mid
    }
  }
  inline void sym(uint16_t *icdf, int n, int s) { put_symbol(ec, icdf, n, s, adapt); }
  inline int blk(int r8, int c8) const { return r8 * fi.w8 + c8; }
  inline bool avail_u(int r8) const { return r8 - 1 >= r8_0; }
  inline bool avail_l(int c8) const { return c8 - 1 >= c8_0; }
  inline int skip_of(int b) const { return f.skip ? f.skip[b] : 0; }
  inline int inter_of(int b) const { return fi.key ? 0 : (f.is_inter ? f.is_inter[b] : 1); }

  // ---- decode_tile (5.11.2)
  void run() {
    for (int r8 = r8_0; r8 < r8_1; r8 += 8) {
      memset(l_lvl, 0, sizeof(l_lvl)); memset(l_dc, 0, sizeof(l_dc));      // clear_left_context
      for (int c8 = c8_0; c8 < c8_1; c8 += 8) {
        cdef_coded = false;                                                 // clear_cdef
        write_lr(r8 * 2, c8 * 2);
        partition(r8 * 2, c8 * 2, 64);
      }
    }
    ec.finish();
  }

  // ---- subexponential codes with a reference, written with equiprobable bools (5.11.58, 4.10.10 structure)
  void put_ns(int n, int v) {               // NS(n) by literals
    const int w = floor_log2((uint32_t)n) + 1, m = (1 << w) - n;
    if (v < m) ec.literal((uint32_t)v, w - 1);
    else { ec.literal((uint32_t)((v + m) >> 1), w - 1); ec.literal((uint32_t)((v + m) & 1), 1); }
  }
  void put_subexp(int num_syms, int k, int v) {   // decode_subexp_bool
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
  static int recenter(int r, int v) { return v > 2 * r ? v : v >= r ? 2 * (v - r) : 2 * (r - v) - 1; }   // inverse of inverse_recenter
  void put_signed_subexp_with_ref(int v, int low, int high, int k, int r) {   // decode_signed_subexp_with_ref_bool
    const int mx = high - low; v -= low; r -= low;
    put_subexp(mx, k, (r << 1) <= mx ? recenter(r, v) : recenter(mx - 1 - r, mx - 1 - v));
  }

  // ---- read_lr / read_lr_unit (5.11.57, 5.11.58)
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
    const int type = u[0];   // 0 none, 1 Wiener, 2 self-guided
    if (f.lr_type[p] == 1) sym(cdf.use_wiener, 2, type == 1);
    else if (f.lr_type[p] == 2) sym(cdf.use_sgrproj, 2, type == 2);
    else sym(cdf.switchable_restore, 3, type);
    if (type == 1 && f.lr_type[p] != 2) {
      static const int kMin[3] = { -5, -23, -17 }, kMax[3] = { 10, 8, 46 }, kK[3] = { 1, 2, 3 };   // Wiener_Taps_Min / Max / K
      for (int pass = 0; pass < 2; pass++)
        for (int j = p ? 1 : 0; j < 3; j++) {
          const int v = u[1 + pass * 3 + j];
          put_signed_subexp_with_ref(v, kMin[j], kMax[j] + 1, kK[j], ref_wiener[p][pass][j]);
          ref_wiener[p][pass][j] = v;
        }
    } else if (type == 2 && f.lr_type[p] != 1) {
      static const int8_t kRadius[16][2] = { { 2, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 }, { 2, 1 },
                                             { 0, 1 }, { 0, 1 }, { 0, 1 }, { 0, 1 }, { 2, 0 }, { 2, 0 } };   // Sgr_Params radii
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

  // ---- decode_partition (5.11.4): always split down to 8x8
  void partition(int r, int c, int bsize) {
    if (r >= fi.mi_rows || c >= fi.mi_cols) return;
    const int half = bsize >> 3;      // halfBlock4x4
    const bool has_rows = r + half < fi.mi_rows, has_cols = c + half < fi.mi_cols;
    const bool au = (r >> 1) - 1 >= r8_0, al = (c >> 1) - 1 >= c8_0;
    if (bsize == 8) {
      sym(cdf.part8[0], 4, 0);        // PARTITION_NONE; the neighbours are 8x8 too, so the context is 0
      block(r >> 1, c >> 1);
      return;
    }
    // neighbours are 8x8 blocks, narrower than this one: ctx = left * 2 + above (9.3)
    uint16_t *pc = (bsize == 16 ? cdf.part16 : bsize == 32 ? cdf.part32 : cdf.part64)[(al ? 2 : 0) + (au ? 1 : 0)];
    if (has_rows && has_cols) {
      sym(pc, 10, 3);                 // PARTITION_SPLIT
    } else if (has_rows || has_cols) {
      // split_or_horz / split_or_vert: the probability of "split" gathers every partition type that splits the missing way
      auto prob = [&](int k) { return (uint32_t)((k ? pc[k - 1] : 32768) - pc[k]); };
      uint32_t psum;
      if (has_cols) psum = prob(2) + prob(3) + prob(4) + prob(6) + prob(7) + prob(9);   // VERT SPLIT HORZ_A VERT_A VERT_B VERT_4
      else psum = prob(1) + prob(3) + prob(4) + prob(5) + prob(6) + prob(8);            // HORZ SPLIT HORZ_A HORZ_B VERT_A HORZ_4
      ec.encode(psum, 0, 1, 2);       // the bit is 1 (split), cdf { 32768 - psum, 32768 }: no adaptation
    }
    const int q = bsize >> 1;
    partition(r, c, q); partition(r, c + half, q); partition(r + half, c, q); partition(r + half, c + half, q);
  }

  // ---- decode_block (5.11.5) of one 8x8 block
  void block(int r8, int c8) {
    const int b = blk(r8, c8);
    const bool au = avail_u(r8), al = avail_l(c8);
    const int skip = skip_of(b);
    if (fi.key) intra_frame_mode_info(r8, c8, b, au, al, skip);
    else inter_frame_mode_info(r8, c8, b, au, al, skip);
    // read_block_tx_size: TX_MODE_LARGEST, nothing coded.  residual (5.11.34)
    const int x4 = (c8 - c8_0) * 2, y4 = (r8 & 7) * 2, cx4 = c8 - c8_0, cy4 = r8 & 7;
    if (skip) {   // reset_block_context
      a_lvl[0][x4] = a_lvl[0][x4 + 1] = a_dc[0][x4] = a_dc[0][x4 + 1] = 0;
      l_lvl[0][y4] = l_lvl[0][y4 + 1] = l_dc[0][y4] = l_dc[0][y4 + 1] = 0;
      for (int p = 1; p < 3; p++) a_lvl[p][cx4] = a_dc[p][cx4] = l_lvl[p][cy4] = l_dc[p][cy4] = 0;
      return;
    }
    const int is_inter = inter_of(b);
    const int tx_type = f.tx_type ? f.tx_type[b] : (int)T_DCT_DCT;
    coeffs(0, x4, y4, 8, f.lev_y + (size_t)b * 64, is_inter, tx_type, f.y_mode ? f.y_mode[b] : 0);
    coeffs(1, cx4, cy4, 4, f.lev_u + (size_t)b * 16, is_inter, 0, 0);
    coeffs(2, cx4, cy4, 4, f.lev_v + (size_t)b * 16, is_inter, 0, 0);
  }

  void write_skip(int b, bool au, bool al, int skip) {
    const int ctx = (au ? skip_of(b - fi.w8) : 0) + (al ? skip_of(b - 1) : 0);
    sym(cdf.skip[ctx], 2, skip);
  }
  void write_cdef(int r8, int c8, int skip) {   // read_cdef (5.11.56): the index is coded with the first non-skipped block of a 64x64
    if (skip || cdef_coded) return;
    const int sb = (r8 >> 3) * fi.sb_cols + (c8 >> 3);
    ec.literal(f.cdef_idx ? f.cdef_idx[sb] : 0, f.cdef_bits);
    cdef_coded = true;
  }

  // ---- intra_frame_mode_info (5.11.7)
  void intra_frame_mode_info(int r8, int c8, int b, bool au, bool al, int skip) {
    write_skip(b, au, al, skip);
    write_cdef(r8, c8, skip);
    const int ym = f.y_mode[b];
    const int actx = kIntraModeContext[au ? f.y_mode[b - fi.w8] : (int)DC_PRED], lctx = kIntraModeContext[al ? f.y_mode[b - 1] : (int)DC_PRED];
    sym(cdf.kf_y_mode[actx][lctx], 13, ym);
    intra_tail(b, ym);
  }
  // intra_angle_info_y, uv_mode, read_cfl_alphas, intra_angle_info_uv (5.11.42 ..): shared by intra blocks of both frame types
  void intra_tail(int b, int ym) {
    if (is_directional(ym)) sym(cdf.angle_delta[ym - V_PRED], 7, (f.angle_y ? f.angle_y[b] : 0) + 3);
    const int uvm = f.uv_mode[b];
    sym(cdf.uv_mode_cfl[ym], 14, uvm);          // an 8x8 block allows chroma from luma
    if (uvm == UV_CFL_PRED) {
      const int au_ = f.cfl_alpha ? f.cfl_alpha[2 * b] : 0, av_ = f.cfl_alpha ? f.cfl_alpha[2 * b + 1] : 0;
      const int su = au_ == 0 ? 0 : au_ < 0 ? 1 : 2, sv = av_ == 0 ? 0 : av_ < 0 ? 1 : 2;   // CFL_SIGN_ZERO / NEG / POS
      sym(cdf.cfl_sign, 8, su * 3 + sv - 1);
      if (su) sym(cdf.cfl_alpha[(su - 1) * 3 + sv], 16, std::abs(au_) - 1);
      if (sv) sym(cdf.cfl_alpha[(sv - 1) * 3 + su], 16, std::abs(av_) - 1);
    } else if (is_directional(uvm)) {
      sym(cdf.angle_delta[uvm - V_PRED], 7, (f.angle_uv ? f.angle_uv[b] : 0) + 3);
    }
  }

  void inter_frame_mode_info(int r8, int c8, int b, bool au, bool al, int skip);   // below
  void mv_stack(int r8, int c8, MvCand *stack, int *num, int *new_ctx, int *ref_ctx);
  void write_mv_comp(MvCompCdf &m, int diff);

  // ---- coeffs (5.11.39) of one square transform block of N x N (4 or 8) at plane position (x4, y4), tile / superblock relative
  void coeffs(int plane, int x4, int y4, int n, const int16_t *lev, int is_inter, int tx_type, int y_mode) {
    if (n == 8) coeffs_n<8>(plane, x4, y4, lev, is_inter, tx_type, y_mode);
    else coeffs_n<4>(plane, x4, y4, lev, is_inter, tx_type, y_mode);
  }
  template <int N> void coeffs_n(int plane, int x4, int y4, const int16_t *lev, int is_inter, int tx_type, int y_mode) {
    constexpr int w4 = N >> 2, txs = N == 4 ? 0 : 1, nc = N * N, LG = N == 4 ? 2 : 3;      // txSzCtx == txSzSqr for square sizes
    constexpr int MS = N + 4;                                                                // stride of the magnitude array
    const int ptype = plane > 0;
    const uint8_t *scan = N == 4 ? scans().s4 : scans().s8;
    int eob = 0;
    for (int c = nc - 1; c >= 0; c--) if (lev[scan[c]]) { eob = c + 1; break; }
    // all_zero context (9.3)
    int ctx;
    if (plane == 0) {
      ctx = 0;      // the transform covers the whole block
    } else {
      int above = 0, left = 0;
      for (int k = 0; k < w4; k++) { above |= a_lvl[plane][x4 + k] | a_dc[plane][x4 + k]; left |= l_lvl[plane][y4 + k] | l_dc[plane][y4 + k]; }
      ctx = 7 + (above != 0) + (left != 0);
    }
    put_symbol_n<2>(ec, cdf.txb_skip[txs][ctx], eob == 0, adapt);
    int cul = 0, dc_cat = 0;
    if (eob) {
      if (plane == 0) {   // transform_type (5.11.47)
        if (is_inter) {
          if (f.reduced_tx_set) put_symbol_n<2>(ec, cdf.inter_tx3[txs], tx_type == T_IDTX ? 0 : 1, adapt);
          else put_symbol_n<16>(ec, cdf.inter_tx1[txs], kInterSet1Sym[tx_type], adapt);
        } else {
          if (f.reduced_tx_set) put_symbol_n<5>(ec, cdf.intra_tx2[txs][y_mode], kIntraSet2Sym[tx_type], adapt);
          else put_symbol_n<7>(ec, cdf.intra_tx1[txs][y_mode], kIntraSet1Sym[tx_type], adapt);
        }
      }
      // eob_pt_*, eob_extra, eob_extra_bit
      const int eob_pt = eob < 3 ? eob : floor_log2((uint32_t)(eob - 1)) + 2;   // eob in (2^(pt-2), 2^(pt-1)]
      if (N == 4) put_symbol_n<5>(ec, cdf.eob16[ptype][0], eob_pt - 1, adapt);
      else put_symbol_n<7>(ec, cdf.eob64[ptype][0], eob_pt - 1, adapt);
      if (eob_pt >= 3) {
        const int off = eob - ((1 << (eob_pt - 2)) + 1);
        int shift = eob_pt - 3;
        put_symbol_n<2>(ec, cdf.eob_extra[txs][ptype][eob_pt - 3], (off >> shift) & 1, adapt);
        for (shift--; shift >= 0; shift--) ec.bool_eq((off >> shift) & 1);
      }
      // levels, last to first.  mag = min(|level|, 15) with a zero border on the right and bottom; only positions below eob are
      // ever non-zero, so only those are written
      uint8_t mag[MS * MS];
      memset(mag, 0, sizeof(mag));
      for (int c = 0; c < eob; c++) {
        const int pos = scan[c];
        const int a = std::abs((int)lev[pos]);
        mag[(pos >> LG) * MS + (pos & (N - 1))] = (uint8_t)(a > 15 ? 15 : a);
      }
      uint16_t(*base_cdf)[5] = cdf.base[txs][ptype];
      uint16_t(*br_cdf)[5] = cdf.br[txs][ptype];
      for (int c = eob - 1; c >= 0; c--) {
        const int pos = scan[c], row = pos >> LG, col = pos & (N - 1);
        const uint8_t *m = mag + row * MS + col;
        const int a = std::abs((int)lev[pos]);
        if (c == eob - 1) {
          const int ectx = c == 0 ? 0 : c <= nc / 8 ? 1 : c <= nc / 4 ? 2 : 3;
          put_symbol_n<3>(ec, cdf.base_eob[txs][ptype][ectx], (a > 3 ? 3 : a) - 1, adapt);
        } else {
          // get_coeff_base_ctx, TX_CLASS_2D (9.3): neighbours (0,1) (1,0) (1,1) (0,2) (2,0), each capped at 3
          auto c3 = [](int v) { return v > 3 ? 3 : v; };
          const int mm = c3(m[1]) + c3(m[MS]) + c3(m[MS + 1]) + c3(m[2]) + c3(m[2 * MS]);
          int bctx = (mm + 1) >> 1;
          bctx = bctx > 4 ? 4 : bctx;
          if (pos == 0) bctx = 0;
          else bctx += row + col < 2 ? 1 : row + col < 4 ? 6 : 21;     // Coeff_Base_Ctx_Offset of the square sizes
          put_symbol_n<4>(ec, base_cdf[bctx], a > 3 ? 3 : a, adapt);
        }
        if (a > 2) {     // coeff_br: up to four increments of 0..3
          int mm = m[1] + m[MS] + m[MS + 1];
          mm = (mm + 1) >> 1;
          mm = mm > 6 ? 6 : mm;
          const int rctx = pos == 0 ? mm : (row < 2 && col < 2) ? mm + 7 : mm + 14;
          int rem = a - 3;
          for (int i = 0; i < 4; i++) {
            const int k = rem > 3 ? 3 : rem;
            put_symbol_n<4>(ec, br_cdf[rctx], k, adapt);
            rem -= k;
            if (k < 3) break;
          }
        }
      }
      // signs and Golomb remainders, first to last
      for (int c = 0; c < eob; c++) {
        const int pos = scan[c], v = lev[pos];
        if (!v) continue;
        const int a = std::abs(v);
        if (c == 0) {
          int sg = 0;
          for (int k = 0; k < w4; k++) {
            const int ad = a_dc[plane][x4 + k], ld = l_dc[plane][y4 + k];
            sg += (ad == 2) - (ad == 1) + (ld == 2) - (ld == 1);
          }
          put_symbol_n<2>(ec, cdf.dc_sign[ptype][sg < 0 ? 1 : sg > 0 ? 2 : 0], v < 0, adapt);
          dc_cat = v < 0 ? 1 : 2;
        } else {
          ec.bool_eq(v < 0);
        }
        if (a > 14) {    // read_golomb: x = a - 14 >= 1, length - 1 zeros then x in `length` bits
          const uint32_t x = (uint32_t)(a - 14);
          const int len = floor_log2(x) + 1;
          ec.literal(0, len - 1);
          ec.literal(x, len);
        }
        cul += a;
      }
      cul = std::min(cul, 63);
    }
    for (int k = 0; k < w4; k++) {
      a_lvl[plane][x4 + k] = (uint8_t)cul; a_dc[plane][x4 + k] = (uint8_t)dc_cat;
      l_lvl[plane][y4 + k] = (uint8_t)cul; l_dc[plane][y4 + k] = (uint8_t)dc_cat;
    }
  }
};

// ---- inter frames.  Tool set: single reference LAST_FRAME, modes NEWMV / NEARESTMV / NEARMV / GLOBALMV (whichever codes the
// encoder's vector), no segmentation, no skip mode, no compound, simple translation, fixed interpolation filter; intra blocks
// are allowed.  All blocks are 8x8, so every candidate of the MV prediction list has weight 2 * len = 4 (7.10.2.2 - 7.10.2.4).
inline unsigned morton8(unsigned x, unsigned y) {   // z-order index inside a superblock of 8 x 8 blocks
  unsigned m = 0;
  for (int i = 0; i < 3; i++) m |= ((x >> i) & 1u) << (2 * i) | ((y >> i) & 1u) << (2 * i + 1);
  return m;
}

// find_mv_stack (7.10.2) for a single-reference block: the list, its weights, and the mode contexts
void TileEnc::mv_stack(int r8, int c8, MvCand *stack, int *num_out, int *new_ctx, int *ref_ctx) {
  int num = 0, new_count = 0;
  bool found = false;
  auto inside = [&](int r, int c) { return r >= r8_0 && r < r8_1 && c >= c8_0 && c < c8_1; };
  auto add = [&](int r, int c, bool count_new) {     // add_ref_mv_candidate + search_stack_process (7.10.2.7, 7.10.2.8), weight 4
    if (!inside(r, c)) return;
    const int nb = blk(r, c);
    if (!inter_of(nb)) return;                        // intra neighbour: no candidate
    const int16_t mx = f.mv[2 * nb], my = f.mv[2 * nb + 1];   // already at quarter-sample precision: lower_mv_precision is the identity
    if (count_new && (*fi.newmv)[(size_t)nb]) new_count++;
    found = true;
    int i = 0;
    for (; i < num; i++) if (stack[i].x == mx && stack[i].y == my) break;
    if (i < num) stack[i].weight += 4;
    else if (num < 8) { stack[num].x = mx; stack[num].y = my; stack[num].weight = 4; num++; }
  };
  add(r8 - 1, c8, true);                              // scan_row(-1)
  const bool above0 = found; found = false;
  add(r8, c8 - 1, true);                              // scan_col(-1)
  const bool left0 = found; found = false;
  {   // scan_point(-1, bw4): the top-right block counts only if it has been decoded already
    const int tr = r8 - 1, tc = c8 + 1;
    bool decoded = false;
    if (inside(tr, tc)) {
      if ((tr >> 3) < (r8 >> 3)) decoded = true;                       // superblock row above
      else if ((tc >> 3) == (c8 >> 3)) decoded = morton8(tc & 7, tr & 7) < morton8(c8 & 7, r8 & 7);
    }
    if (decoded) add(tr, tc, true);
  }
  bool above = above0 || found; found = false;
  const int close = (above ? 1 : 0) + (left0 ? 1 : 0);
  const int num_nearest = num, num_new = new_count;
  for (int i = 0; i < num_nearest; i++) stack[i].weight += 640;   // REF_CAT_LEVEL
  // (no temporal candidates: use_ref_frame_mvs = 0, ZeroMvContext = 0)
  add(r8 - 1, c8 - 1, false);                         // scan_point(-1, -1)
  above = above || found; found = false;
  bool left = left0;
  add(r8 - 2, c8, false); above = above || found; found = false;      // scan_row(-3)
  add(r8, c8 - 2, false); left = left || found; found = false;        // scan_col(-3)
  add(r8 - 3, c8, false); above = above || found; found = false;      // scan_row(-5)
  add(r8, c8 - 3, false); left = left || found; found = false;        // scan_col(-5)
  const int total = (above ? 1 : 0) + (left ? 1 : 0);
  // sorting process (7.10.2.11): the nearest entries, then the rest, each by descending weight (stable bubble sort)
  auto sort_range = [&](int a, int b) {
    for (int len = b; len > a;) {
      int nr = a;
      for (int i = a + 1; i < len; i++)
        if (stack[i - 1].weight < stack[i].weight) { std::swap(stack[i - 1], stack[i]); nr = i; }
      len = nr;
    }
  };
  sort_range(0, num_nearest);
  sort_range(num_nearest, num);
  // extra search process (7.10.2.12): adds vectors of neighbours that use OTHER reference frames; every inter block here uses
  // LAST_FRAME, so its vector is in the list already.  Context and clamping process (7.10.2.14):
  if (close == 0) { *new_ctx = std::min(total, 1); *ref_ctx = total; }
  else if (close == 1) { *new_ctx = 3 - std::min(num_new, 1); *ref_ctx = 2 + total; }
  else { *new_ctx = 5 - std::min(num_new, 1); *ref_ctx = 5; }
  const int mi_r = r8 * 2, mi_c = c8 * 2;
  const int border = 128 + 2 * 4 * 8;                 // MV_BORDER + block size in 1/8 samples
  const int top = -(mi_r * 4 * 8) - border, bottom = (fi.mi_rows - 2 - mi_r) * 4 * 8 + border;
  const int lft = -(mi_c * 4 * 8) - border, right = (fi.mi_cols - 2 - mi_c) * 4 * 8 + border;
  for (int i = 0; i < num; i++) {
    stack[i].y = (int16_t)std::min(std::max<int>(stack[i].y, top), bottom);
    stack[i].x = (int16_t)std::min(std::max<int>(stack[i].x, lft), right);
  }
  for (int i = num; i < 2; i++) { stack[i].x = stack[i].y = 0; stack[i].weight = 0; }   // GlobalMvs: identity
  *num_out = num;
}

// read_mv_component (5.11.33), quarter-sample precision: diff is even and non-zero
void TileEnc::write_mv_comp(MvCompCdf &m, int diff) {
  sym(m.sign, 2, diff < 0);
  const int off = std::abs(diff) - 1;
  const int cls = (off >> 3) < 2 ? 0 : floor_log2((uint32_t)(off >> 3));
  sym(m.cls, 11, cls);
  if (cls == 0) {
    sym(m.class0, 2, off >> 3);
    sym(m.class0_fr[off >> 3], 4, (off >> 1) & 3);
  } else {
    const int o = off - (2 << (cls + 2)), d = o >> 3;
    for (int i = 0; i < cls; i++) sym(m.bits[i], 2, (d >> i) & 1);
    sym(m.fr, 4, (o >> 1) & 3);
  }
  // mv_class0_hp / mv_hp: allow_high_precision_mv = 0, implied 1
}

void TileEnc::inter_frame_mode_info(int r8, int c8, int b, bool au, bool al, int skip) {
  write_skip(b, au, al, skip);      // inter_segment_id, read_skip_mode: nothing to code
  write_cdef(r8, c8, skip);
  const int is_inter = inter_of(b);
  {   // is_inter (9.3)
    const bool ai = au ? !inter_of(b - fi.w8) : false, li = al ? !inter_of(b - 1) : false;   // AboveIntra / LeftIntra
    int ctx;
    if (au && al) ctx = (li && ai) ? 3 : (li || ai);
    else if (au || al) ctx = 2 * (au ? ai : li);
    else ctx = 0;
    sym(cdf.is_inter[ctx], 2, is_inter);
  }
  if (!is_inter) {   // intra_block_mode_info (5.11.22): y_mode with the size-group context (Size_Group[BLOCK_8X8] = 1)
    const int ym = f.y_mode[b];
    sym(cdf.y_mode[1], 13, ym);
    intra_tail(b, ym);
    (*fi.newmv)[(size_t)b] = 0;
    return;
  }
  // read_ref_frames (5.11.25): single_ref_p1 = 0, single_ref_p3 = 0, single_ref_p4 = 0 -> LAST_FRAME.  The contexts compare
  // counts of neighbouring references (9.3): only LAST_FRAME ever occurs, so each is 1 (no inter neighbour) or 2.
  const int n_last = (au && inter_of(b - fi.w8) ? 1 : 0) + (al && inter_of(b - 1) ? 1 : 0);
  const int rctx = n_last ? 2 : 1;
  sym(cdf.single_ref[rctx][0], 2, 0);
  sym(cdf.single_ref[rctx][2], 2, 0);
  sym(cdf.single_ref[rctx][3], 2, 0);
  MvCand st[8];
  int num, new_ctx, ref_ctx;
  mv_stack(r8, c8, st, &num, &new_ctx, &ref_ctx);
  const int mx = f.mv[2 * b], my = f.mv[2 * b + 1];
  // the cheapest mode that reproduces the encoder's vector: NEARESTMV, NEARMV (index 1..3), GLOBALMV, else NEWMV
  int mode = 3, ref_idx = 0;      // 0 NEARESTMV, 1 NEARMV, 2 GLOBALMV, 3 NEWMV
  if (st[0].x == mx && st[0].y == my) mode = 0;
  else {
    for (int i = 1; i < std::max(num, 2) && i < 4; i++)
      if (st[i].x == mx && st[i].y == my) { mode = 1; ref_idx = i; break; }
    if (mode == 3 && mx == 0 && my == 0) mode = 2;
  }
  sym(cdf.new_mv[new_ctx], 2, mode != 3);
  if (mode != 3) {
    sym(cdf.zero_mv[0], 2, mode != 2);
    if (mode != 2) sym(cdf.ref_mv[ref_ctx], 2, mode == 1);
  }
  auto drl_ctx = [&](int i) {     // drl_mode context from the weights of entries i and i + 1 (9.3)
    const bool a = st[i].weight >= 640, c = st[i + 1].weight >= 640;
    return a && c ? 0 : a ? 1 : !c ? 2 : 0;
  };
  if (mode == 3) {
    // NEWMV: the predictor is entry RefMvIdx of the list; pick the closest of the entries the syntax can name (0..2)
    const int nsel = std::min(num, 3);
    long best = -1;
    for (int i = 0; i < std::max(nsel, 1); i++) {
      const long c = std::abs(mx - st[i].x) + std::abs(my - st[i].y);
      if (best < 0 || c < best) { best = c; ref_idx = i; }
    }
    for (int i = 0; i < 2; i++)
      if (num > i + 1) {
        sym(cdf.drl[drl_ctx(i)], 2, ref_idx != i);
        if (ref_idx == i) break;
      }
    const int pred = num <= 1 ? 0 : ref_idx;            // assign_mv (5.11.26)
    const int dx = mx - st[pred].x, dy = my - st[pred].y;   // read_mv (5.11.32): component 0 is the row (vertical) difference
    sym(cdf.mv_joint, 4, (dx ? 1 : 0) + (dy ? 2 : 0));
    if (dy) write_mv_comp(cdf.mv[0], dy);
    if (dx) write_mv_comp(cdf.mv[1], dx);
  } else if (mode == 1) {
    for (int i = 1; i < 3; i++)
      if (num > i + 1) {
        sym(cdf.drl[drl_ctx(i)], 2, ref_idx != i);
        if (ref_idx == i) break;
      }
  }
  (*fi.newmv)[(size_t)b] = mode == 3;
  // read_interintra_mode, read_motion_mode, read_compound_type, interpolation filter: nothing to code with this tool set
}


}  // namespace

// ------------------------------------------------------------------------------------------------ public
std::vector<uint8_t> temporal_delimiter_obu() { return make_obu(2, {}); }

std::vector<uint8_t> sequence_header_obu(const SequenceParams &sp) {   // sequence_header_obu (5.5.1)
  BitWriter w;
  w.put(0, 3);      // seq_profile 0: 4:2:0, 8 / 10 bit
  w.put(0, 1);      // still_picture
  w.put(0, 1);      // reduced_still_picture_header
  w.put(0, 1);      // timing_info_present_flag
  w.put(0, 1);      // initial_display_delay_present_flag
  w.put(0, 5);      // operating_points_cnt_minus_1
  w.put(0, 12);     // operating_point_idc[0]
  w.put(31, 5);     // seq_level_idx[0] = 31: "maximum parameters" (one tile per superblock exceeds every numbered level's tile count)
  w.put(0, 1);      // seq_tier[0]
  const int wb = floor_log2((uint32_t)std::max(sp.width - 1, 1)) + 1, hb = floor_log2((uint32_t)std::max(sp.height - 1, 1)) + 1;
  w.put((uint32_t)(wb - 1), 4); w.put((uint32_t)(hb - 1), 4);
  w.put((uint32_t)(sp.width - 1), wb); w.put((uint32_t)(sp.height - 1), hb);
  w.put(0, 1);      // frame_id_numbers_present_flag
  w.put(0, 1);      // use_128x128_superblock
  w.put(0, 1);      // enable_filter_intra
  w.put(1, 1);      // enable_intra_edge_filter
  w.put(0, 1);      // enable_interintra_compound
  w.put(0, 1);      // enable_masked_compound
  w.put(0, 1);      // enable_warped_motion
  w.put(0, 1);      // enable_dual_filter
  w.put(0, 1);      // enable_order_hint
  w.put(0, 1);      // seq_choose_screen_content_tools
  w.put(0, 1);      // seq_force_screen_content_tools = 0 (so seq_force_integer_mv = SELECT_INTEGER_MV, not coded)
  w.put(0, 1);      // enable_superres
  w.put(1, 1);      // enable_cdef
  w.put(1, 1);      // enable_restoration
  write_color_config(w, sp.bit_depth);
  w.put(0, 1);      // film_grain_params_present
  w.trailing_bits();
  return make_obu(1, w.b);
}


bool frame_obu_from_tiles(const av1mi_obu_frame &f, const uint8_t *payloads, const uint32_t *sizes, int ntiles, std::vector<uint8_t> *out,
                          std::string *err) {
  if (!check(f, err, false)) return false;
  const FrameInfo fi = frame_info(f);
  if (ntiles != fi.tile_cols * fi.tile_rows) { if (err) *err = "tile count does not match the frame"; return false; }
  std::vector<const uint8_t *> data((size_t)ntiles);
  std::vector<size_t> size((size_t)ntiles);
  for (int t = 0; t < ntiles; t++) {
    if (sizes[t] == 0) { if (err) *err = "empty tile payload"; return false; }
    data[(size_t)t] = payloads; size[(size_t)t] = sizes[t]; payloads += sizes[t];
  }
  return assemble_frame(fi, data.data(), size.data(), out);
}

bool frame_obu(const av1mi_obu_frame &f, int threads, std::vector<uint8_t> *out, std::string *err) {
  if (!check(f, err)) return false;
  FrameInfo fi = frame_info(f);
  std::vector<uint8_t> newmv((size_t)fi.w8 * fi.h8, 0);
  fi.newmv = &newmv;
  const int ntiles = fi.tile_cols * fi.tile_rows;
  std::vector<std::vector<uint8_t>> tiles((size_t)ntiles);
  std::atomic<int> next(0);
  auto work = [&] {
    for (int t; (t = next.fetch_add(1)) < ntiles;) {
      TileEnc te(fi, t / fi.tile_cols, t % fi.tile_cols);
      te.run();
      tiles[(size_t)t].swap(te.ec.out);
    }
  };
  const int nt = std::max(1, std::min(threads, ntiles));
  if (nt == 1) {
    work();
  } else {
    std::vector<std::thread> pool;
    for (int i = 0; i < nt; i++) pool.emplace_back(work);
    for (auto &t : pool) t.join();
  }
  std::vector<const uint8_t *> data((size_t)ntiles);
  std::vector<size_t> size((size_t)ntiles);
  for (int t = 0; t < ntiles; t++) { data[(size_t)t] = tiles[(size_t)t].data(); size[(size_t)t] = tiles[(size_t)t].size(); }
  return assemble_frame(fi, data.data(), size.data(), out);
}

bool temporal_unit(const av1mi_obu_frame &f, bool with_sequence_header, int threads, std::vector<uint8_t> *out, std::string *err) {
  std::vector<uint8_t> fr;
  if (!frame_obu(f, threads, &fr, err)) return false;
  *out = temporal_delimiter_obu();
  if (with_sequence_header) {
    const SequenceParams sp = sequence_params(f);
    const std::vector<uint8_t> sh = sequence_header_obu(sp);
    out->insert(out->end(), sh.begin(), sh.end());
  }
  out->insert(out->end(), fr.begin(), fr.end());
  return true;
}

std::vector<uint8_t> range_code_raw(const uint32_t *fl, const uint32_t *fh, const uint8_t *sym, const uint8_t *nsym, size_t count) {
  RangeEnc ec;
  for (size_t i = 0; i < count; i++) ec.encode(fl[i], fh[i], sym[i], nsym[i]);
  ec.finish();
  return ec.out;
}

}  // namespace av1
}  // namespace av1mi_host
