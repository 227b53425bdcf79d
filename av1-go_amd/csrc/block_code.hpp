// block_code.hpp — the residual-coding tail shared by the fused intra and inter kernels: one B x B block held by B
// lanes of one wave (lane r owns row r): residual -> forward DCT (ADST where an intra chroma block's mode implies it: columns, rows; libaom fwd_txfm2d_c) -> quantise
// (libaom quantize_fp) -> levels to HBM -> dequantise (spec 7.12.3) -> inverse DCT (rows, columns; spec 7.13.3) ->
// reconstruction.  T is the group's B x (B+4) int32 LDS transpose buffer.  SURVEY.md §8a rows K1 + K8 + K2.
#pragma once
#include "intra.hpp"
#include "txfm_cfg.hpp"

namespace av1mi {

// a 24-bit multiply the compiler cannot turn back into a 32-bit one (it does when it can bound the operands: v_mul_lo_u32 is
// four passes on CDNA, v_mul_u32_u24 one); both operands non-negative and below 2^24
__device__ __forceinline__ int mul24_pinned(int a, int b) {
  int d;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

template <int N, typename Pix> __device__ __forceinline__ void load_row(const Pix *p, int *v) {
  if constexpr (sizeof(Pix) == 1) {
#pragma unroll
    for (int c = 0; c < N; c += 4) {
      const uint32_t u = *reinterpret_cast<const uint32_t *>(p + c);
      v[c] = u & 255; v[c + 1] = (u >> 8) & 255; v[c + 2] = (u >> 16) & 255; v[c + 3] = u >> 24;
    }
  } else {
#pragma unroll
    for (int c = 0; c < N; c += 4) {
      const uint2 u = *reinterpret_cast<const uint2 *>(p + c);
      v[c] = u.x & 0xffff; v[c + 1] = u.x >> 16; v[c + 2] = u.y & 0xffff; v[c + 3] = u.y >> 16;
    }
  }
}
template <int N, typename Pix> __device__ __forceinline__ void store_row(Pix *p, const int *v) {
  if constexpr (sizeof(Pix) == 1) {
#pragma unroll
    for (int c = 0; c < N; c += 4)
      *reinterpret_cast<uint32_t *>(p + c) = (uint32_t)v[c] | ((uint32_t)v[c + 1] << 8) | ((uint32_t)v[c + 2] << 16) | ((uint32_t)v[c + 3] << 24);
  } else {
#pragma unroll
    for (int c = 0; c < N; c += 4) {
      uint2 u; u.x = (uint32_t)v[c] | ((uint32_t)v[c + 1] << 16); u.y = (uint32_t)v[c + 2] | ((uint32_t)v[c + 3] << 16);
      *reinterpret_cast<uint2 *>(p + c) = u;
    }
  }
}

// 1-D pass that is a DCT or an ADST per BLOCK (lanes of different blocks share the wave): both are evaluated and the lane
// selects — for the 4-point (default) and 8-point chroma transforms that is cheaper than a divergent branch per group.
template <int B, int BIT> __device__ __forceinline__ void fwd_dct_or_adst(int32_t *x, bool adst) {
  static_assert(B == 4 || B == 8 || B == 16, "ADST is only selected for chroma blocks (4x4, or 8x8 / 16x16 with 16x16 / 32x32 luma blocks)");
  int32_t a[B];
#pragma unroll
  for (int i = 0; i < B; i++) a[i] = x[i];
  fdct<B, BIT>(x);
  if constexpr (B == 4) fadst4<BIT>(a); else if constexpr (B == 8) fadst8<BIT>(a); else fadst16<BIT>(a);
#pragma unroll
  for (int i = 0; i < B; i++) x[i] = adst ? a[i] : x[i];
}
template <int B, int RANGE> __device__ __forceinline__ void inv_dct_or_adst(int32_t *x, bool adst) {
  static_assert(B == 4 || B == 8 || B == 16, "see fwd_dct_or_adst");
  int32_t a[B];
#pragma unroll
  for (int i = 0; i < B; i++) a[i] = x[i];
  idct<B, RANGE>(x);
  if constexpr (B == 4) iadst4<12>(a); else if constexpr (B == 8) iadst8<12, RANGE>(a); else iadst16<12, RANGE>(a);
#pragma unroll
  for (int i = 0; i < B; i++) x[i] = adst ? a[i] : x[i];
}
// Mode_To_Txfm (AV1 spec 5.11.47 compute_tx_type): an intra CHROMA block's transform type follows from its prediction mode and is
// not coded.  Bit m of the masks: mode m uses the ADST vertically (column pass) / horizontally (row pass).
constexpr unsigned kModeVertAdst = 0x1732u;   // V D135 D113 D67 SMOOTH SMOOTH_V PAETH
constexpr unsigned kModeHorzAdst = 0x1AD4u;   // H D135 D157 D203 SMOOTH SMOOTH_H PAETH

// s[]: source row, bp[]: prediction row.  Writes this lane's row of levels to lev_row and returns its reconstruction in
// rec[]; the return value is non-zero when the row holds a non-zero level.  SEL = the block may use the ADST in either
// direction (vadst: column / vertical pass, hadst: row / horizontal pass); otherwise DCT_DCT.  AC_ROUND: the quantiser's rounding
// offset for AC coefficients in 1/128 of the step (txfm_cfg.hpp: one half in key frames, a dead zone in inter frames).
template <int B, typename Pix, bool SEL = false, int AC_ROUND = kAcRoundIntra>
__device__ __forceinline__ int code_residual(int32_t *T, int lane, const int *s, const int *bp, int dc_q, int ac_q, int dc_quant, int ac_quant, int16_t *lev_row,
                                             int *rec, bool vadst = false, bool hadst = false) {
  constexpr int RS = B + 4, bd = sizeof(Pix) == 1 ? 8 : 10;
  // forward transform (libaom fwd_txfm2d_c: columns, then rows), DCT_DCT
  {
    int4 *row = reinterpret_cast<int4 *>(T + lane * RS);
#pragma unroll
    for (int c = 0; c < B; c += 4) row[c / 4] = make_int4(s[c] - bp[c], s[c + 1] - bp[c + 1], s[c + 2] - bp[c + 2], s[c + 3] - bp[c + 3]);
  }
  AV1MI_GROUP_SYNC();
  int32_t xv[B];
#pragma unroll
  for (int r = 0; r < B; r++) xv[r] = T[r * RS + lane] << fwd_shift(B, B, 0);
  if constexpr (SEL) fwd_dct_or_adst<B, fwd_cos_bit_col(B, B)>(xv, vadst); else fdct<B, fwd_cos_bit_col(B, B)>(xv);
  AV1MI_GROUP_SYNC();
#pragma unroll
  for (int r = 0; r < B; r++) T[r * RS + lane] = round2(xv[r], -fwd_shift(B, B, 1));
  AV1MI_GROUP_SYNC();
#pragma unroll
  for (int c = 0; c < B; c += 4) {
    const int4 v = *reinterpret_cast<const int4 *>(T + lane * RS + c);
    xv[c] = v.x; xv[c + 1] = v.y; xv[c + 2] = v.z; xv[c + 3] = v.w;
  }
  if constexpr (SEL) fwd_dct_or_adst<B, fwd_cos_bit_row(B, B)>(xv, hadst); else fdct<B, fwd_cos_bit_row(B, B)>(xv);
  // quantise / dequantise this row (libaom quantize_fp; spec 7.12.3), log_scale LS: 0 for B <= 16, 1 for 32 x 32 (1024 coefficients)
  // dc_quant / ac_quant = (1 << 16) / step come from the host: written here as a division, the compiler sank the (loop-invariant)
  // division into the conditional block of every coefficient — ~25 scalar or ~30 vector instructions and a divergent branch, 8
  // times per row of every block
  constexpr int LS = B >= 32 ? 1 : 0;
  const int dc_rnd = round2((64 * dc_q) >> 7, LS), ac_rnd = round2((AC_ROUND * ac_q) >> 7, LS);
  const int maxv = (1 << (7 + bd)) - 1, minv = -(1 << (7 + bd));
  int lv[B];
#pragma unroll
  for (int c = 0; c < B; c++) {
    const bool dc = lane == 0 && c == 0;
    const int q = dc ? dc_q : ac_q, quant = dc ? dc_quant : ac_quant, rnd = dc ? dc_rnd : ac_rnd;
    const int v = round2(xv[c], -fwd_shift(B, B, 2));
    const bool neg = v < 0;
    int a = min(neg ? -v : v, 1 << 20), l = 0;
    l = mul24_pinned(min(a + rnd, 32767), quant) >> (16 - LS);    // 15 x 15 bits: a full-rate 24-bit multiply (a 32-bit one is four passes); < 2^15 at LS 0
    if constexpr (LS) l = min(l, 32767);
    l = (a << (1 + LS)) >= q ? l : 0;
    lv[c] = neg ? -l : l;
    const int d = (mul24_pinned(l, q) & 0xFFFFFF) >> LS;    // l < 2^15, q < 2^15
    xv[c] = min(max(neg ? -d : d, minv), maxv);
  }
#pragma unroll
  for (int c = 0; c < B; c += 4) {
    uint2 o;
    o.x = (uint32_t)(lv[c] & 0xffff) | ((uint32_t)lv[c + 1] << 16);
    o.y = (uint32_t)(lv[c + 2] & 0xffff) | ((uint32_t)lv[c + 3] << 16);
    *reinterpret_cast<uint2 *>(lev_row + c) = o;
  }
  // inverse transform (spec 7.13.3: rows, then columns) + reconstruction
  constexpr int ROW_RANGE = bd + 8;
#pragma unroll
  for (int c = 0; c < B; c++) xv[c] = clampr<ROW_RANGE>(xv[c]);
  if constexpr (SEL) inv_dct_or_adst<B, ROW_RANGE>(xv, hadst); else idct<B, ROW_RANGE>(xv);
  AV1MI_GROUP_SYNC();
#pragma unroll
  for (int c = 0; c < B; c += 4)
    *reinterpret_cast<int4 *>(T + lane * RS + c) = make_int4(round2(xv[c], inv_row_shift(B, B)), round2(xv[c + 1], inv_row_shift(B, B)),
                                                             round2(xv[c + 2], inv_row_shift(B, B)), round2(xv[c + 3], inv_row_shift(B, B)));
  AV1MI_GROUP_SYNC();
#pragma unroll
  for (int r = 0; r < B; r++) xv[r] = min(max(T[r * RS + lane], -32768), 32767);   // max(bd+6,16) = 16 bits for bd <= 10
  if constexpr (SEL) inv_dct_or_adst<B, 16>(xv, vadst); else idct<B, 16>(xv);
  AV1MI_GROUP_SYNC();
#pragma unroll
  for (int r = 0; r < B; r++) T[r * RS + lane] = round2(xv[r], 4);
  AV1MI_GROUP_SYNC();
  const int maxpix = (1 << bd) - 1;
#pragma unroll
  for (int c = 0; c < B; c += 4) {
    const int4 v = *reinterpret_cast<const int4 *>(T + lane * RS + c);
    rec[c] = min(max(bp[c] + v.x, 0), maxpix); rec[c + 1] = min(max(bp[c + 1] + v.y, 0), maxpix);
    rec[c + 2] = min(max(bp[c + 2] + v.z, 0), maxpix); rec[c + 3] = min(max(bp[c + 3] + v.w, 0), maxpix);
  }
  int nz = 0;
#pragma unroll
  for (int c = 0; c < B; c++) nz |= lv[c];
  return nz;
}

}  // namespace av1mi
