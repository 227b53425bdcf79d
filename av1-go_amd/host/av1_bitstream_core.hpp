// av1_bitstream_core.hpp — shared pieces of the host AV1 bitstream writers (av1_bitstream.cpp: the 8x8-block writer the GPU
// pipeline feeds; av1_blockstream.cpp: the general block-structured writer): fixed-length bit writer, OBU framing, the symbol
// encoder (spec 8.2), the CDF context of a tile with its defaults (av1_default_cdfs.inc), syntax constants and the frame header
// (spec 5.9.2).  Internal to libav1mi_host.so.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "av1_bitstream.hpp"
#include "av1_default_cdfs.inc"

namespace av1mi_host {
namespace av1 {
namespace core {

// ------------------------------------------------------------------------------------------------ fixed-length bits
struct BitWriter {   // f(n): most significant bit first (spec 4.10.2)
  std::vector<uint8_t> b;
  int used = 0;      // bits used in the last byte (0 = byte aligned)
  void put(uint32_t v, int n) {
    for (int i = n - 1; i >= 0; i--) {
      if (!used) b.push_back(0);
      b.back() |= (uint8_t)(((v >> i) & 1u) << (7 - used));
      used = (used + 1) & 7;
    }
  }
  void byte_align() { used = 0; }                       // byte_alignment(): zero bits (spec 5.3.5)
  void trailing_bits() { put(1, 1); used = 0; }         // trailing_bits(): a one, then zeros (spec 5.3.4)
};
inline void put_leb128(std::vector<uint8_t> &o, uint64_t v) {  // spec 4.10.5
  do { uint8_t c = v & 0x7F; v >>= 7; if (v) c |= 0x80; o.push_back(c); } while (v);
}
inline std::vector<uint8_t> make_obu(int type, const std::vector<uint8_t> &payload) {   // obu_header (5.3.2) with obu_has_size_field = 1
  std::vector<uint8_t> o;
  o.push_back((uint8_t)((type << 3) | 2));
  put_leb128(o, payload.size());
  o.insert(o.end(), payload.begin(), payload.end());
  return o;
}
inline int tile_log2(int blk, int target) { int k = 0; while ((blk << k) < target) k++; return k; }   // spec 5.9.16
inline int floor_log2(uint32_t v) { return 31 - __builtin_clz(v); }

// ------------------------------------------------------------------------------------------------ symbol encoder (8.2)
// The dual of the spec's symbol decoder (8.2.2 init, 8.2.6 decode_symbol, 8.2.4 exit): the stream value x satisfies
// low <= x < low + rng at the current precision; symbol 0 sits at the BOTTOM of x-space (the decoder's SymbolValue is
// the complement).  CDFs are kept in inverse form, icdf[i] = 32768 - cdf[i], icdf[N-1] = 0, icdf[N] = adaptation counter.
struct RangeEnc {
  std::vector<uint8_t> out;
  uint64_t low = 0;
  uint32_t rng = 0x8000;
  int nb = -1;       // bits of x above the 16-bit window that are not in `out` yet (x has 15 + total shift bits)
  inline void carry() {
    if (nb >= 0 && (low >> (16 + nb))) {
      for (size_t i = out.size(); i-- > 0;) if (++out[i] != 0) break;
      low &= ((uint64_t)1 << (16 + nb)) - 1;
    }
  }
  inline void renorm() {
    const int d = 15 - floor_log2(rng);
    rng <<= d; low <<= d; nb += d;
    while (nb >= 8) {
      nb -= 8;
      out.push_back((uint8_t)(low >> (16 + nb)));
      low &= ((uint64_t)1 << (16 + nb)) - 1;
    }
  }
  // fl = icdf[s-1] (32768 for s = 0), fh = icdf[s]; n = number of symbols
  inline void encode(uint32_t fl, uint32_t fh, int s, int n) {
    const uint32_t r = rng;
    const uint32_t v = (((r >> 8) * (fh >> 6)) >> 1) + 4u * (uint32_t)(n - 1 - s);
    if (fl < 32768u) {
      const uint32_t u = (((r >> 8) * (fl >> 6)) >> 1) + 4u * (uint32_t)(n - s);
      low += r - u;
      rng = u - v;
    } else {
      rng = r - v;
    }
    carry();
    renorm();
  }
  inline void bool_eq(int bit) {   // read_bool(): cdf {1 << 14, 1 << 15, 0}: v = ((r >> 8) * 256 >> 1) + 4
    const uint32_t r = rng, v = ((r >> 8) << 7) + 4;
    if (bit) { low += r - v; rng = v; carry(); } else { rng = r - v; }
    renorm();
  }
  void literal(uint32_t v, int n) { for (int i = n - 1; i >= 0; i--) bool_eq((v >> i) & 1); }     // L(n), spec 4.10.7
  // exit process (8.2.4): the minimum number of bits, then the trailing one, then zero padding to a byte
  void finish() {
    uint64_t e = ((low + 0x3FFF) & ~(uint64_t)0x3FFF) | 0x4000;
    if (nb >= 0 && (e >> (16 + nb))) {
      for (size_t i = out.size(); i-- > 0;) if (++out[i] != 0) break;
      e &= ((uint64_t)1 << (16 + nb)) - 1;
    }
    int top = 15 + nb;                     // most significant pending bit of x inside e
    while (top >= 14) {
      uint8_t byte = 0;
      for (int k = 7; k >= 0 && top >= 14; k--, top--) byte |= (uint8_t)(((e >> top) & 1) << k);
      out.push_back(byte);
    }
  }
};

// adaptive symbol (8.2.6 + the CDF update of 8.2.6 / libaom update_cdf), icdf has n + 1 entries.  The alphabet size is a
// compile-time constant at every call site of the hot path (2, 3 and 4 symbols make up ~95 % of a frame's symbols): the update
// loop unrolls and the rate's log2 term folds.
template <int N> inline void put_symbol_n(RangeEnc &ec, uint16_t *icdf, int s, bool adapt) {
  ec.encode(s ? icdf[s - 1] : 32768u, icdf[s], s, N);
  if (adapt) {
    const int count = icdf[N];
    constexpr int lg = N >= 4 ? 2 : N >= 2 ? 1 : 0;
    const int rate = 3 + lg + (count > 15) + (count > 31);
#pragma GCC unroll 16
    for (int i = 0; i < N - 1; i++) {
      const int v = icdf[i];
      icdf[i] = (uint16_t)(i < s ? v + ((32768 - v) >> rate) : v - (v >> rate));
    }
    icdf[N] = (uint16_t)(count + (count < 32));
  }
}
inline void put_symbol(RangeEnc &ec, uint16_t *icdf, int n, int s, bool adapt) {
  switch (n) {
    case 2: put_symbol_n<2>(ec, icdf, s, adapt); break;
    case 3: put_symbol_n<3>(ec, icdf, s, adapt); break;
    case 4: put_symbol_n<4>(ec, icdf, s, adapt); break;
    case 5: put_symbol_n<5>(ec, icdf, s, adapt); break;
    case 7: put_symbol_n<7>(ec, icdf, s, adapt); break;
    case 8: put_symbol_n<8>(ec, icdf, s, adapt); break;
    case 10: put_symbol_n<10>(ec, icdf, s, adapt); break;
    case 11: put_symbol_n<11>(ec, icdf, s, adapt); break;
    case 13: put_symbol_n<13>(ec, icdf, s, adapt); break;
    case 14: put_symbol_n<14>(ec, icdf, s, adapt); break;
    case 16: put_symbol_n<16>(ec, icdf, s, adapt); break;
    default: {
      ec.encode(s ? icdf[s - 1] : 32768u, icdf[s], s, n);
      if (adapt) {
        const int count = icdf[n];
        const int rate = 3 + (count > 15) + (count > 31) + std::min(floor_log2((uint32_t)n), 2);
        for (int i = 0; i < n - 1; i++) {
          if (i < s) icdf[i] += (uint16_t)((32768 - icdf[i]) >> rate);
          else icdf[i] -= (uint16_t)(icdf[i] >> rate);
        }
        icdf[n] = (uint16_t)(count + (count < 32));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ CDF context of a tile
struct MvCompCdf { uint16_t cls[12], class0[3], class0_fr[2][5], class0_hp[3], sign[3], bits[10][3], fr[5], hp[3]; };
struct Cdfs {
  uint16_t skip[3][3];
  uint16_t kf_y_mode[5][5][14], y_mode[4][14], uv_mode_nocfl[13][14], uv_mode_cfl[13][15], angle_delta[8][8];
  uint16_t part8[4][5], part16[4][11], part32[4][11], part64[4][11];
  uint16_t cfl_sign[9], cfl_alpha[6][17];
  uint16_t intra_tx1[2][13][8], intra_tx2[3][13][6], inter_tx1[2][17], inter_tx2[13], inter_tx3[4][3];
  uint16_t use_wiener[3], use_sgrproj[3], switchable_restore[4];
  uint16_t tx8[3][3], tx16[3][4], tx32[3][4], tx64[3][4], txfm_split[21][3], interp_filter[16][4];   // general block sizes (av1_blockstream.cpp)
  uint16_t is_inter[4][3], single_ref[3][6][3], new_mv[6][3], zero_mv[2][3], ref_mv[6][3], drl[3][3];
  uint16_t mv_joint[5];
  MvCompCdf mv[2];
  uint16_t txb_skip[5][13][3], eob16[2][2][6], eob32[2][2][7], eob64[2][2][8], eob128[2][2][9], eob256[2][2][10],
      eob512[2][2][11], eob1024[2][2][12], eob_extra[5][2][9][3], dc_sign[2][3][3], base_eob[5][2][4][4], base[5][2][42][5],
      br[5][2][21][5];
};
// rows of (nsym - 1) spec-form values -> rows of dst_stride inverse-form entries
inline void load_rows(uint16_t *dst, int dst_stride, const uint16_t *src, int nsym, int rows) {
  for (int r = 0; r < rows; r++) {
    for (int i = 0; i < nsym - 1; i++) dst[r * dst_stride + i] = (uint16_t)(32768 - src[r * (nsym - 1) + i]);
    for (int i = nsym - 1; i < dst_stride; i++) dst[r * dst_stride + i] = 0;
  }
}
template <class D, class S> void load_tab(D &dst, const S &src, int nsym) {
  const int rows = (int)(sizeof(S) / sizeof(uint16_t)) / (nsym - 1);
  const int stride = (int)(sizeof(D) / sizeof(uint16_t)) / rows;
  load_rows((uint16_t *)&dst, stride, (const uint16_t *)&src, nsym, rows);
}
inline const Cdfs &default_cdfs(int qcat) {   // init_non_coeff_cdfs / init_coeff_cdfs (7.20 "setup past independence")
  static Cdfs tabs[4];
  static bool ready = [] {
    for (int q = 0; q < 4; q++) {
      Cdfs &c = tabs[q];
      memset(&c, 0, sizeof(c));
      load_tab(c.skip, Default_Skip_Cdf, 2);
      load_tab(c.kf_y_mode, Default_Intra_Frame_Y_Mode_Cdf, 13);
      load_tab(c.y_mode, Default_Y_Mode_Cdf, 13);
      load_tab(c.uv_mode_nocfl, Default_Uv_Mode_Cfl_Not_Allowed_Cdf, 13);
      load_tab(c.uv_mode_cfl, Default_Uv_Mode_Cfl_Allowed_Cdf, 14);
      load_tab(c.angle_delta, Default_Angle_Delta_Cdf, 7);
      load_tab(c.part8, Default_Partition_W8_Cdf, 4);
      load_tab(c.part16, Default_Partition_W16_Cdf, 10);
      load_tab(c.part32, Default_Partition_W32_Cdf, 10);
      load_tab(c.part64, Default_Partition_W64_Cdf, 10);
      load_tab(c.cfl_sign, Default_Cfl_Sign_Cdf, 8);
      load_tab(c.cfl_alpha, Default_Cfl_Alpha_Cdf, 16);
      load_tab(c.intra_tx1, Default_Intra_Tx_Type_Set1_Cdf, 7);
      load_tab(c.intra_tx2, Default_Intra_Tx_Type_Set2_Cdf, 5);
      load_tab(c.inter_tx1, Default_Inter_Tx_Type_Set1_Cdf, 16);
      load_tab(c.inter_tx2, Default_Inter_Tx_Type_Set2_Cdf, 12);
      load_tab(c.inter_tx3, Default_Inter_Tx_Type_Set3_Cdf, 2);
      load_tab(c.use_wiener, Default_Use_Wiener_Cdf, 2);
      load_tab(c.use_sgrproj, Default_Use_Sgrproj_Cdf, 2);
      load_tab(c.switchable_restore, Default_Switchable_Restore_Cdf, 3);
      load_tab(c.tx8, Default_Tx_8x8_Cdf, 2);
      load_tab(c.tx16, Default_Tx_16x16_Cdf, 3);
      load_tab(c.tx32, Default_Tx_32x32_Cdf, 3);
      load_tab(c.tx64, Default_Tx_64x64_Cdf, 3);
      load_tab(c.txfm_split, Default_Txfm_Split_Cdf, 2);
      load_tab(c.interp_filter, Default_Interp_Filter_Cdf, 3);
      load_tab(c.is_inter, Default_Is_Inter_Cdf, 2);
      load_tab(c.single_ref, Default_Single_Ref_Cdf, 2);
      load_tab(c.new_mv, Default_New_Mv_Cdf, 2);
      load_tab(c.zero_mv, Default_Zero_Mv_Cdf, 2);
      load_tab(c.ref_mv, Default_Ref_Mv_Cdf, 2);
      load_tab(c.drl, Default_Drl_Mode_Cdf, 2);
      load_tab(c.mv_joint, Default_Mv_Joint_Cdf, 4);
      for (int k = 0; k < 2; k++) {
        load_tab(c.mv[k].cls, Default_Mv_Class_Cdf, 11);
        load_tab(c.mv[k].class0, Default_Mv_Class0_Bit_Cdf, 2);
        load_tab(c.mv[k].class0_fr, Default_Mv_Class0_Fr_Cdf, 4);
        load_tab(c.mv[k].class0_hp, Default_Mv_Class0_Hp_Cdf, 2);
        load_tab(c.mv[k].sign, Default_Mv_Sign_Cdf, 2);
        load_tab(c.mv[k].bits, Default_Mv_Bit_Cdf, 2);
        load_tab(c.mv[k].fr, Default_Mv_Fr_Cdf, 4);
        load_tab(c.mv[k].hp, Default_Mv_Hp_Cdf, 2);
      }
      load_tab(c.txb_skip, Default_Txb_Skip_Cdf[q], 2);
      load_tab(c.eob16, Default_Eob_Pt_16_Cdf[q], 5);
      load_tab(c.eob32, Default_Eob_Pt_32_Cdf[q], 6);
      load_tab(c.eob64, Default_Eob_Pt_64_Cdf[q], 7);
      load_tab(c.eob128, Default_Eob_Pt_128_Cdf[q], 8);
      load_tab(c.eob256, Default_Eob_Pt_256_Cdf[q], 9);
      load_tab(c.eob512, Default_Eob_Pt_512_Cdf[q], 10);
      load_tab(c.eob1024, Default_Eob_Pt_1024_Cdf[q], 11);
      load_tab(c.eob_extra, Default_Eob_Extra_Cdf[q], 2);
      load_tab(c.dc_sign, Default_Dc_Sign_Cdf[q], 2);
      load_tab(c.base_eob, Default_Coeff_Base_Eob_Cdf[q], 3);
      load_tab(c.base, Default_Coeff_Base_Cdf[q], 4);
      load_tab(c.br, Default_Coeff_Br_Cdf[q], 4);
    }
    return true;
  }();
  (void)ready;
  return tabs[qcat];
}

// ------------------------------------------------------------------------------------------------ constants of the syntax
enum { DC_PRED, V_PRED, H_PRED, D45_PRED, D135_PRED, D113_PRED, D157_PRED, D203_PRED, D67_PRED, SMOOTH_PRED, SMOOTH_V_PRED,
       SMOOTH_H_PRED, PAETH_PRED, UV_CFL_PRED };
enum { T_DCT_DCT, T_ADST_DCT, T_DCT_ADST, T_ADST_ADST, T_FLIPADST_DCT, T_DCT_FLIPADST, T_FLIPADST_FLIPADST, T_ADST_FLIPADST,
       T_FLIPADST_ADST, T_IDTX };
static const uint8_t kIntraModeContext[13] = { 0, 1, 2, 3, 4, 4, 4, 4, 3, 0, 1, 2, 0 };   // Intra_Mode_Context (9.3)
// symbol of a 2-D-class transform type inside each set = inverse of Tx_Type_Intra_Inv_Set1/2, Tx_Type_Inter_Inv_Set1/3 (5.11.47)
static const int8_t kIntraSet1Sym[16] = { 1, 5, 6, 4, -1, -1, -1, -1, -1, 0, 2, 3, -1, -1, -1, -1 };
static const int8_t kIntraSet2Sym[16] = { 1, 3, 4, 2, -1, -1, -1, -1, -1, 0, -1, -1, -1, -1, -1, -1 };
static const int8_t kInterSet1Sym[16] = { 7, 8, 9, 12, 10, 11, 13, 14, 15, 0, 1, 2, 3, 4, 5, 6 };
inline bool is_directional(int m) { return m >= V_PRED && m <= D67_PRED; }
inline bool tx_class_2d(int t) { return t <= T_FLIPADST_ADST; }

// ------------------------------------------------------------------------------------------------ frame-level derived values
struct FrameInfo {
  const av1mi_obu_frame *f;
  int w8, h8;                 // frame size in 8x8 blocks
  int mi_rows, mi_cols;       // in 4x4 units
  int sb_rows, sb_cols;
  int tile_cols_log2, tile_rows_log2, tile_w_sb, tile_h_sb, tile_cols, tile_rows;
  int qcat;
  int lr_size[3], lr_rows[3], lr_cols[3];
  bool key;
  std::vector<uint8_t> *newmv;   // per block: coded with NEWMV (has_newmv of the MV prediction process); written by the tile coders
  int tx_mode_select = 0;        // 1 = TX_MODE_SELECT (av1_blockstream.cpp); the 8x8 writer codes TX_MODE_LARGEST
  int high_precision_mv = 0;     // inter frames: allow_high_precision_mv
  int interp_filter = 0;         // inter frames: 0 regular, 1 smooth, 2 sharp, 3 bilinear for the whole frame; 4 = switchable per block
};

inline bool check(const av1mi_obu_frame &f, std::string *err, bool need_symbols = true) {
  auto bad = [&](const char *m) { if (err) *err = m; return false; };
  if (f.width <= 0 || f.height <= 0 || (f.width & 7) || (f.height & 7) || f.width > 4096 || f.height > 4096)
    return bad("frame size must be a multiple of 8 and at most 4096x4096 (64 superblock tiles per dimension)");
  if (f.visible_width < 0 || f.visible_height < 0 || (f.visible_width && (f.visible_width > f.width || f.width - f.visible_width >= 8)) ||
      (f.visible_height && (f.visible_height > f.height || f.height - f.visible_height >= 8)))
    return bad("visible size must lie within 7 samples below the coded size");
  if (f.bit_depth != 8 && f.bit_depth != 10) return bad("bit depth must be 8 or 10");
  if (f.frame_type != 0 && f.frame_type != 1) return bad("frame_type must be 0 (key) or 1 (inter)");
  if (f.base_q_idx < 1 || f.base_q_idx > 255) return bad("base_q_idx must be 1..255 (0 is the lossless mode, not coded)");
  for (int i = 0; i < 4; i++) if (f.lf_level[i] < 0 || f.lf_level[i] > 63) return bad("loop filter level out of range");
  if (f.lf_sharpness < 0 || f.lf_sharpness > 7) return bad("loop filter sharpness out of range");
  if (f.cdef_damping < 3 || f.cdef_damping > 6 || f.cdef_bits < 0 || f.cdef_bits > 3) return bad("CDEF parameters out of range");
  for (int p = 0; p < 3; p++) {
    if (f.lr_type[p] < 0 || f.lr_type[p] > 3) return bad("restoration type out of range");
    if (f.lr_type[p] && !f.lr_units[p]) return bad("restoration units missing");
  }
  if (f.lr_unit_shift < 0 || f.lr_unit_shift > 2 || f.lr_uv_shift < 0 || f.lr_uv_shift > 1) return bad("restoration unit size out of range");
  if (!need_symbols) return true;      // header + tile payloads coded elsewhere (frame_obu_from_tiles)
  if (!f.lev_y || !f.lev_u || !f.lev_v) return bad("levels missing");
  if (f.frame_type == 0 && (!f.y_mode || !f.uv_mode)) return bad("key frame without prediction modes");
  if (f.frame_type == 1 && !f.mv) return bad("inter frame without motion vectors");
  return true;
}

inline FrameInfo frame_info(const av1mi_obu_frame &f) {
  FrameInfo fi;
  fi.f = &f;
  fi.key = f.frame_type == 0;
  fi.w8 = f.width / 8; fi.h8 = f.height / 8;
  fi.mi_cols = 2 * fi.w8; fi.mi_rows = 2 * fi.h8;
  fi.sb_cols = (fi.mi_cols + 15) >> 4; fi.sb_rows = (fi.mi_rows + 15) >> 4;
  // tile_info (5.9.15), uniform spacing
  const int max_log2_cols = tile_log2(1, std::min(fi.sb_cols, 64)), max_log2_rows = tile_log2(1, std::min(fi.sb_rows, 64));
  const int min_log2_cols = tile_log2(64, fi.sb_cols);
  fi.tile_cols_log2 = f.tile_cols_log2 < 0 ? max_log2_cols : std::min(std::max(f.tile_cols_log2, min_log2_cols), max_log2_cols);
  fi.tile_w_sb = (fi.sb_cols + (1 << fi.tile_cols_log2) - 1) >> fi.tile_cols_log2;
  fi.tile_cols = (fi.sb_cols + fi.tile_w_sb - 1) / fi.tile_w_sb;
  const int min_log2_tiles = std::max(min_log2_cols, tile_log2(2304, fi.sb_rows * fi.sb_cols));
  const int min_log2_rows = std::max(min_log2_tiles - tile_log2(1, fi.tile_cols), 0);
  fi.tile_rows_log2 = f.tile_rows_log2 < 0 ? max_log2_rows : std::min(std::max(f.tile_rows_log2, min_log2_rows), max_log2_rows);
  fi.tile_h_sb = (fi.sb_rows + (1 << fi.tile_rows_log2) - 1) >> fi.tile_rows_log2;
  fi.tile_rows = (fi.sb_rows + fi.tile_h_sb - 1) / fi.tile_h_sb;
  fi.qcat = f.base_q_idx <= 20 ? 0 : f.base_q_idx <= 60 ? 1 : f.base_q_idx <= 120 ? 2 : 3;   // init_coeff_cdfs index (7.20)
  for (int p = 0; p < 3; p++) {
    fi.lr_size[p] = (64 << f.lr_unit_shift) >> (p ? f.lr_uv_shift : 0);
    const int vh = visible_height(f), vw = visible_width(f);      // the restoration units tile the TRUE frame (spec 5.9.20 unitRows / unitCols)
    const int ph = p ? (vh + 1) >> 1 : vh, pw = p ? (vw + 1) >> 1 : vw;
    fi.lr_rows[p] = std::max((ph + (fi.lr_size[p] >> 1)) / fi.lr_size[p], 1);   // count_units_in_frame (5.11.57)
    fi.lr_cols[p] = std::max((pw + (fi.lr_size[p] >> 1)) / fi.lr_size[p], 1);
  }
  return fi;
}

// ------------------------------------------------------------------------------------------------ headers
inline void write_color_config(BitWriter &w, int bd) {   // color_config (5.5.2), profile 0
  w.put(bd == 10, 1);   // high_bitdepth
  w.put(0, 1);          // mono_chrome
  w.put(0, 1);          // color_description_present_flag
  w.put(0, 1);          // color_range: studio swing
  w.put(0, 2);          // chroma_sample_position: unknown
  w.put(0, 1);          // separate_uv_delta_q
}

inline void write_frame_header(BitWriter &w, const FrameInfo &fi, int tile_size_bytes) {   // uncompressed_header (5.9.2)
  const av1mi_obu_frame &f = *fi.f;
  w.put(0, 1);                       // show_existing_frame
  w.put(fi.key ? 0 : 1, 2);          // frame_type: KEY_FRAME / INTER_FRAME
  w.put(1, 1);                       // show_frame
  if (!fi.key) w.put(1, 1);          // error_resilient_mode = 1: every frame starts from the default CDFs (primary_ref_frame = NONE)
  w.put(f.disable_cdf_update ? 1 : 0, 1);
  // allow_screen_content_tools = seq_force_screen_content_tools = 0: not coded
  w.put(0, 1);                       // frame_size_override_flag
  // order_hint: 0 bits.  primary_ref_frame = PRIMARY_REF_NONE (intra frame or error resilient)
  if (!fi.key) {
    w.put(0x01, 8);                  // refresh_frame_flags: the frame replaces slot 0
    // error_resilient_mode && enable_order_hint would code ref_order_hint[]: order hints are off
    for (int i = 0; i < 7; i++) w.put(0, 3);   // ref_frame_idx[i] = 0: every reference name maps to slot 0 (the previous frame)
  }
  // frame_size(): sequence maximum; superres_params(): off; render_size():
  w.put(0, 1);                       // render_and_frame_size_different
  if (!fi.key) {
    w.put((uint32_t)(fi.high_precision_mv ? 1 : 0), 1);   // allow_high_precision_mv (force_integer_mv = 0)
    w.put(fi.interp_filter == 4, 1);                                   // is_filter_switchable
    if (fi.interp_filter != 4) w.put((uint32_t)fi.interp_filter, 2);   // interpolation_filter: EIGHTTAP, EIGHTTAP_SMOOTH, EIGHTTAP_SHARP, BILINEAR
    w.put(0, 1);                     // is_motion_mode_switchable
    // use_ref_frame_mvs = 0 (error resilient)
  }
  if (!f.disable_cdf_update) w.put(1, 1);   // disable_frame_end_update_cdf: nothing inherits this frame's CDFs
  // tile_info (5.9.15)
  w.put(1, 1);                       // uniform_tile_spacing_flag
  const int min_log2_cols = tile_log2(64, fi.sb_cols), max_log2_cols = tile_log2(1, std::min(fi.sb_cols, 64));
  for (int k = min_log2_cols; k < max_log2_cols; k++) {
    const int inc = k < fi.tile_cols_log2;
    w.put(inc, 1);                   // increment_tile_cols_log2
    if (!inc) break;
  }
  const int cols_log2 = tile_log2(1, fi.tile_cols);
  const int min_log2_tiles = std::max(min_log2_cols, tile_log2(2304, fi.sb_rows * fi.sb_cols));
  const int min_log2_rows = std::max(min_log2_tiles - cols_log2, 0), max_log2_rows = tile_log2(1, std::min(fi.sb_rows, 64));
  for (int k = min_log2_rows; k < max_log2_rows; k++) {
    const int inc = k < fi.tile_rows_log2;
    w.put(inc, 1);                   // increment_tile_rows_log2
    if (!inc) break;
  }
  const int rows_log2 = tile_log2(1, fi.tile_rows);
  if (cols_log2 || rows_log2) {
    w.put(0, cols_log2 + rows_log2); // context_update_tile_id
    w.put((uint32_t)(tile_size_bytes - 1), 2);   // tile_size_bytes_minus_1
  }
  // quantization_params (5.9.12)
  w.put((uint32_t)f.base_q_idx, 8);
  w.put(0, 1);                       // DeltaQYDc: delta_coded
  w.put(0, 1);                       // DeltaQUDc
  w.put(0, 1);                       // DeltaQUAc
  w.put(0, 1);                       // using_qmatrix
  w.put(0, 1);                       // segmentation_enabled (5.9.14)
  w.put(0, 1);                       // delta_q_present (5.9.17, base_q_idx > 0)
  // loop_filter_params (5.9.11)
  w.put((uint32_t)f.lf_level[0], 6); w.put((uint32_t)f.lf_level[1], 6);
  if (f.lf_level[0] || f.lf_level[1]) { w.put((uint32_t)f.lf_level[2], 6); w.put((uint32_t)f.lf_level[3], 6); }
  w.put((uint32_t)f.lf_sharpness, 3);
  w.put(0, 1);                       // loop_filter_delta_enabled
  // cdef_params (5.9.19)
  w.put((uint32_t)(f.cdef_damping - 3), 2);
  w.put((uint32_t)f.cdef_bits, 2);
  for (int i = 0; i < (1 << f.cdef_bits); i++) {
    w.put(f.cdef_y[i] >> 2, 4); w.put(f.cdef_y[i] & 3, 2);
    w.put(f.cdef_uv[i] >> 2, 4); w.put(f.cdef_uv[i] & 3, 2);
  }
  // lr_params (5.9.20)
  static const int kLrCode[4] = { 0, 2, 3, 1 };   // inverse of Remap_Lr_Type: none, Wiener, self-guided, switchable
  bool uses_lr = false, uses_chroma_lr = false;
  for (int p = 0; p < 3; p++) {
    w.put((uint32_t)kLrCode[f.lr_type[p]], 2);
    if (f.lr_type[p]) { uses_lr = true; if (p) uses_chroma_lr = true; }
  }
  if (uses_lr) {
    w.put(f.lr_unit_shift > 0, 1);                         // lr_unit_shift
    if (f.lr_unit_shift > 0) w.put(f.lr_unit_shift > 1, 1); // lr_unit_extra_shift
    if (uses_chroma_lr) w.put((uint32_t)f.lr_uv_shift, 1);
  }
  w.put((uint32_t)(fi.tx_mode_select ? 1 : 0), 1);   // tx_mode_select: TX_MODE_SELECT or TX_MODE_LARGEST (5.9.21)
  if (!fi.key) w.put(0, 1);          // reference_select (5.9.23): single reference
  // skip_mode_params: not allowed without order hints.  allow_warped_motion: off in the sequence
  w.put((uint32_t)(f.reduced_tx_set ? 1 : 0), 1);
  if (!fi.key) for (int i = 0; i < 7; i++) w.put(0, 1);   // global_motion_params (5.9.24): is_global = 0 for LAST..ALTREF
  // film_grain_params: not present in the sequence
}

// OBU_FRAME from finished tile payloads (raster order): frame header, tile group with the size prefixes
inline bool assemble_frame(const FrameInfo &fi, const uint8_t *const *tile_data, const size_t *tile_size, std::vector<uint8_t> *out) {
  const int ntiles = fi.tile_cols * fi.tile_rows;
  size_t largest = 0, total = 0;
  for (int t = 0; t + 1 < ntiles; t++) largest = std::max(largest, tile_size[t]);
  for (int t = 0; t < ntiles; t++) total += tile_size[t];
  const int tsb = largest <= 0x100 ? 1 : largest <= 0x10000 ? 2 : largest <= 0x1000000 ? 3 : 4;   // tile_size_minus_1 must fit
  BitWriter w;
  write_frame_header(w, fi, tsb);
  w.byte_align();                                 // frame_obu: byte_alignment after the header (5.10)
  // tile_group_obu (5.11.1)
  if (ntiles > 1) { w.put(0, 1); w.byte_align(); }   // tile_start_and_end_present_flag
  std::vector<uint8_t> payload;
  payload.swap(w.b);
  payload.reserve(payload.size() + total + (size_t)ntiles * 4);
  for (int t = 0; t < ntiles; t++) {
    if (t + 1 < ntiles) {
      const size_t sz = tile_size[t] - 1;         // tile_size_minus_1, little endian (le(TileSizeBytes))
      for (int k = 0; k < tsb; k++) payload.push_back((uint8_t)(sz >> (8 * k)));
    }
    payload.insert(payload.end(), tile_data[t], tile_data[t] + tile_size[t]);
  }
  *out = make_obu(6, payload);
  return true;
}

}  // namespace core
}  // namespace av1
}  // namespace av1mi_host
