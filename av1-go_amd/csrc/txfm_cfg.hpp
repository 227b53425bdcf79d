// txfm_cfg.hpp — per-size constants of the 2-D transforms (spec Transform_Row_Shift; libaom
// av1_fwd_txfm_shift_ls / av1_fwd_cos_bit_col / av1_fwd_cos_bit_row), shared by the standalone K1/K2
// kernels and the fused pipelines.
#pragma once
#include "txfm1d.hpp"

namespace av1mi {

// The quantiser's rounding offset for AC coefficients in 1/128 of the step (DC: always 64 = one half, libaom's quantize_fp).  Key
// frames keep one half; inter frames use 51 = 0.4, a dead zone: -3.6 % BD-rate on the synthetic GOPs (0.336: -2.0 %, 0.25: +1.5 %),
// while on key frames alone it loses slightly against libaom's curve.  Non-normative; oracle/av1o_pipeline.c
// (AV1O_AC_ROUND_INTER) is the same constant.
constexpr int kAcRoundIntra = 64, kAcRoundInter = 51;
// the 32x32 blocks of key frames and their 16x16 chroma blocks (oracle: AV1O_AC_ROUND_KEY32; +0.26 dB at equal size at q 128, +0.05 at q 24)
constexpr int kAcRoundKey32 = 58;

__host__ __device__ constexpr int imin(int a, int b) { return a < b ? a : b; }
__host__ __device__ constexpr int imax(int a, int b) { return a > b ? a : b; }

// spec Transform_Row_Shift, indexed by (log2 w - 2, log2 h - 2)
__host__ __device__ constexpr int inv_row_shift(int w, int h) {
  // {4x4:0, 8x8:1, 16x16:2, 32x32:2, 64x64:2, 4x8:0, 8x4:0, 8x16:1,16x8:1,16x32:1,32x16:1,32x64:1,64x32:1,
  //  4x16:1,16x4:1, 8x32:2,32x8:2,16x64:2,64x16:2}
  const int a = imin(w, h), b = imax(w, h);
  if (a == b) return a == 4 ? 0 : a == 8 ? 1 : 2;
  if (b == 2 * a) return a == 4 ? 0 : 1;
  return a == 4 ? 1 : 2;  // 1:4
}
// libaom av1_fwd_txfm_shift_ls
__host__ __device__ constexpr int fwd_shift(int w, int h, int i) {
  const int a = imin(w, h), b = imax(w, h);
  int s0 = 2, s1 = 0, s2 = 0;
  if (a == b) { s1 = a == 4 ? 0 : a == 8 ? -1 : a == 16 ? -2 : a == 32 ? -4 : -2; if (a == 64) { s0 = 0; s2 = -2; } }
  else if (b == 2 * a) {
    s1 = a == 4 ? -1 : a == 8 ? -2 : a == 16 ? -4 : -2;
    if (a == 32) { if (w == 32) { s0 = 0; s1 = -2; s2 = -2; } else { s0 = 2; s1 = -4; s2 = -2; } }
  } else {
    s1 = a == 4 ? -1 : a == 8 ? -2 : -4;
    if (a == 16) { if (w == 16) { s0 = 0; s1 = -2; s2 = 0; } else { s0 = 2; s1 = -4; s2 = 0; } }
  }
  return i == 0 ? s0 : i == 1 ? s1 : s2;
}
__host__ __device__ constexpr int fwd_cos_bit_col(int w, int h) {
  constexpr int t[5][5] = { { 13, 13, 13, 0, 0 }, { 13, 13, 13, 12, 0 }, { 13, 13, 13, 12, 13 }, { 0, 13, 13, 12, 13 }, { 0, 0, 13, 12, 13 } };
  return t[ilog2c(w) - 2][ilog2c(h) - 2];
}
__host__ __device__ constexpr int fwd_cos_bit_row(int w, int h) {
  constexpr int t[5][5] = { { 13, 13, 12, 0, 0 }, { 13, 13, 13, 12, 0 }, { 13, 13, 12, 13, 12 }, { 0, 12, 13, 12, 11 }, { 0, 0, 12, 11, 10 } };
  return t[ilog2c(w) - 2][ilog2c(h) - 2];
}
__host__ __device__ constexpr bool is_rect2(int w, int h) { return w == 2 * h || h == 2 * w; }


}  // namespace av1mi
