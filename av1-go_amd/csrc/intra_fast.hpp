// intra_fast.hpp — the fused pipeline's intra predictor: ALL edge variants of a block in two LDS phases, then
// compile-time-specialised, branch-free prediction of every candidate mode.
//
// intra.hpp builds the edges of one (block, mode) pair with ~10 wave-level hand-offs and runtime mode dispatch; the
// mode search of the fused kernel would pay that 11 times per block, and a lone wave pays ~5 cycles for every scalar
// branch it takes.  Here the group fetches the raw neighbours once, then every lane computes a share of the entries
// of every derived edge array the candidate set needs (filtered / upsampled, with or without the above-right /
// bottom-left extension, corner-filtered top-left) straight from the raw arrays — filter taps and upsampling taps are
// evaluated on the fly, so no derived array depends on another — and after ONE more hand-off all candidates are
// predicted from LDS by code whose mode, angle, derivative and array slot are template constants and whose per-lane
// choices (availability, filter type, zone-2 above/left pick) are selects, not branches.
// The "flat" early-outs of libaom need no special case here: when an edge is unavailable the raw array is filled with
// exactly the value the early-out would return, and every filter / interpolation of a constant array is that constant.
// Same arithmetic as intra.hpp (AV1 spec §7.11.2; libaom build_intra_predictors and friends); the pipeline parity
// tests compare modes, levels and reconstruction with the oracle's per-mode builder bit for bit.
#pragma once
#include "intra.hpp"

namespace av1mi {

// raw edge arrays: index -1 = top-left, 0 .. 2B-1 samples; 2 entries of front padding (index -2 of upsampled arrays)
constexpr int kRawPad = 2;
__host__ __device__ constexpr int raw_len(int b) { return kRawPad + 2 * b + 2; }
// a derived array holds at most 2*(2B)+1 entries starting at index -2
__host__ __device__ constexpr int var_len(int b) { return kRawPad + 4 * b + 2; }
constexpr int kNumVariants = 9;   // D45a D67a D113a D135a D157a | D113l D135l D157l D203l
__host__ __device__ constexpr int fast_edge_len(int b) { return 2 * raw_len(b) + kNumVariants * var_len(b); }   // entries

__host__ __device__ constexpr int ct_mode_angle(int mode) {
  return mode == V_PRED ? 90 : mode == H_PRED ? 180 : mode == D45_PRED ? 45 : mode == D135_PRED ? 135 : mode == D113_PRED ? 113
       : mode == D157_PRED ? 157 : mode == D203_PRED ? 203 : mode == D67_PRED ? 67 : 0;
}
__host__ __device__ constexpr int ct_derivative(int angle) {   // spec Dr_Intra_Derivative at the nominal angles' complements
  return angle == 45 ? 64 : angle == 67 ? 27 : angle == 23 ? 151 : 0;
}
__host__ __device__ constexpr int ct_slot_above(int mode) { return mode == D45_PRED ? 0 : mode == D67_PRED ? 1 : mode == D113_PRED ? 2 : mode == D135_PRED ? 3 : mode == D157_PRED ? 4 : -1; }
__host__ __device__ constexpr int ct_slot_left(int mode) { return mode == D113_PRED ? 5 : mode == D135_PRED ? 6 : mode == D157_PRED ? 7 : mode == D203_PRED ? 8 : -1; }
// edge filter strength / upsampling switch for compile-time block size and angle delta, runtime filter type (select)
template <int B, int DELTA> __device__ __forceinline__ int ct_strength(int type) {
  constexpr int d = DELTA < 0 ? -DELTA : DELTA, wh = 2 * B;
  constexpr int s0 = wh <= 8 ? (d >= 56 ? 1 : 0) : wh <= 16 ? (d >= 40 ? 1 : 0) : wh <= 24 ? (d >= 32 ? 3 : d >= 16 ? 2 : d >= 8 ? 1 : 0)
                   : wh <= 32 ? (d >= 32 ? 3 : d >= 4 ? 2 : d >= 1 ? 1 : 0) : (d >= 1 ? 3 : 0);
  constexpr int s1 = wh <= 8 ? (d >= 64 ? 2 : d >= 40 ? 1 : 0) : wh <= 16 ? (d >= 48 ? 2 : d >= 20 ? 1 : 0) : wh <= 24 ? (d >= 4 ? 3 : 0) : (d >= 1 ? 3 : 0);
  return type ? s1 : s0;
}
template <int B, int DELTA> __device__ __forceinline__ int ct_upsample(int type) {
  constexpr int d = DELTA < 0 ? -DELTA : DELTA, wh = 2 * B;
  constexpr int u0 = (d == 0 || d >= 40) ? 0 : (wh <= 16 ? 1 : 0), u1 = (d == 0 || d >= 40) ? 0 : (wh <= 8 ? 1 : 0);
  return type ? u1 : u0;
}

// A lane's view of one raw edge: for each of its (up to 3) entries i = lane + it*B (p-space: 0 = top-left, k = sample k-1)
// the five raw values P(i-2) .. P(i+2), loaded ONCE from LDS (indices clamped into the array, never read out of range).
struct EdgeWin { int v[3][5]; };
template <int B, typename T>
__device__ __forceinline__ EdgeWin load_edge_win(const T *raw, int lane) {
  EdgeWin W;
#pragma unroll
  for (int it = 0; it < 3; it++)
#pragma unroll
    for (int d = -2; d <= 2; d++) W.v[it][d + 2] = raw[min(max(lane + it * B + d, 0), 2 * B) - 1];
  return W;
}

// one derived array (slot S) from the register windows; no LDS reads, no branches on per-lane state
template <int B, int S, typename T>
__device__ __forceinline__ void fast_build_slot(T *edge, int lane, int bd, int n_top, int n_left, int filter_type, int tl_raw, int tl_corner,
                                                const EdgeWin &WA, const EdgeWin &WL) {
  constexpr int RL = raw_len(B), VL = var_len(B);
  constexpr int angles[kNumVariants] = { 45, 67, 113, 135, 157, 113, 135, 157, 203 };
  constexpr int a = angles[S];
  constexpr bool is_above = S < 5, both = a > 90 && a < 180, ext = is_above ? a < 90 : a > 180;
  constexpr int n = B + (ext ? B : 0), delta = is_above ? a - 90 : a - 180;
  const int navail = is_above ? n_top : n_left;
  const int strength = navail > 0 ? ct_strength<B, delta>(filter_type) : 0;
  const int up = ct_upsample<B, delta>(filter_type);   // up != 0 implies strength == 0 (spec: d < 40 and small blocks only)
  const int sz = navail + 1 + (ext ? B : 0);
  const int tl = both ? tl_corner : tl_raw;
  const EdgeWin &W = is_above ? WA : WL;
  T *out = edge + 2 * RL + S * VL + kRawPad;
  const int maxv = (1 << bd) - 1;
  const int k0 = strength == 3 ? 2 : 0, k1 = strength == 2 ? 5 : 4, k2 = strength == 1 ? 8 : strength == 2 ? 6 : 4;
#pragma unroll
  for (int it = 0; it < (n + B) / B; it++) {
    const int i = lane + it * B;                    // p-space index of this lane's entry; edge index j = i - 1
    // P(i + d) with the top-left substituted at p-index 0
    const int pm2 = i - 2 <= 0 ? tl : W.v[it][0], pm1 = i - 1 <= 0 ? tl : W.v[it][1], p0 = i <= 0 ? tl : W.v[it][2];
    const int pp1 = W.v[it][3], pp2 = W.v[it][4];
    // 5-tap filter with taps clamped to [0, sz-1] (i < sz wherever the result is used)
    const int t1 = i + 1 <= sz - 1 ? pp1 : p0, t2 = i + 2 <= sz - 1 ? pp2 : t1;
    const int f = (k0 * (pm2 + t2) + k1 * (pm1 + t1) + k2 * p0 + 8) >> 4;
    const int v = (strength && i >= 1 && i < sz) ? f : p0;
    // upsampling taps (unfiltered there): entries j-2, j-1, j, j+1 clamped to [-1, n-1]  <=>  p-indices i-2.., max 0, min n
    const int u3 = i + 1 <= n ? pp1 : p0;
    const int h = min(max((-pm2 + 9 * pm1 + 9 * p0 - u3 + 8) >> 4, 0), maxv);
    const int j = i - 1;
    if (j < n) {
      if (!up) out[j] = (T)v;
      else if (j >= 0) { out[2 * j - 1] = (T)h; out[2 * j] = (T)p0; }
      else out[-2] = (T)p0;
    }
  }
}

// Build raw + derived arrays for a B x B block.  `edge`: fast_edge_len(B) entries of LDS owned by the group (B lanes).
template <int B, typename T, typename Fetch>
__device__ __forceinline__ void fast_build(T *edge, int lane, int bd, int n_top, int n_topright, int n_left, int n_bottomleft,
                                           int filter_type, Fetch fetch) {
  constexpr int RL = raw_len(B);
  T *ra = edge + kRawPad, *rl = edge + RL + kRawPad;
  const int base = 128 << (bd - 8);
  // phase 1: raw neighbours (above-right / bottom-left included when available, else replicated)
  {
    const int availa = n_top + n_topright, availl = n_left + n_bottomleft;
    const int fill_a = n_left > 0 ? fetch(0, -1) : base - 1, fill_l = n_top > 0 ? fetch(-1, 0) : base + 1;
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const int i = lane + it * B;
      const int va = fetch(-1, min(i, max(availa, 1) - 1)), vl = fetch(min(i, max(availl, 1) - 1), -1);
      ra[i] = (T)(n_top > 0 ? va : fill_a);
      rl[i] = (T)(n_left > 0 ? vl : fill_l);
    }
    if (lane == 0) {
      const int c = fetch(-1, -1);
      const int tl = (n_top > 0 && n_left > 0) ? c : n_top > 0 ? fill_l : n_left > 0 ? fill_a : base;
      ra[-1] = rl[-1] = (T)tl;
    }
  }
  AV1MI_GROUP_SYNC();
  // phase 2: the nine derived arrays from two register windows
  const EdgeWin WA = load_edge_win<B>(ra, lane), WL = load_edge_win<B>(rl, lane);
  const int tl_raw = ra[-1];
  const int tl_corner = (2 * B >= 24) ? (rl[0] * 5 + tl_raw * 6 + ra[0] * 5 + 8) >> 4 : tl_raw;
  fast_build_slot<B, 0>(edge, lane, bd, n_top, n_left, filter_type, tl_raw, tl_corner, WA, WL);
  fast_build_slot<B, 1>(edge, lane, bd, n_top, n_left, filter_type, tl_raw, tl_corner, WA, WL);
  fast_build_slot<B, 2>(edge, lane, bd, n_top, n_left, filter_type, tl_raw, tl_corner, WA, WL);
  fast_build_slot<B, 3>(edge, lane, bd, n_top, n_left, filter_type, tl_raw, tl_corner, WA, WL);
  fast_build_slot<B, 4>(edge, lane, bd, n_top, n_left, filter_type, tl_raw, tl_corner, WA, WL);
  fast_build_slot<B, 5>(edge, lane, bd, n_top, n_left, filter_type, tl_raw, tl_corner, WA, WL);
  fast_build_slot<B, 6>(edge, lane, bd, n_top, n_left, filter_type, tl_raw, tl_corner, WA, WL);
  fast_build_slot<B, 7>(edge, lane, bd, n_top, n_left, filter_type, tl_raw, tl_corner, WA, WL);
  fast_build_slot<B, 8>(edge, lane, bd, n_top, n_left, filter_type, tl_raw, tl_corner, WA, WL);
  AV1MI_GROUP_SYNC();
}

// prediction of row r of a B x B block for compile-time MODE; branch-free.  LW = lanes that hold the block (for the DC sums)
template <int MODE, int B, typename T>
__device__ __forceinline__ void fast_pred_row(const T *edge, int r, int bd, int n_top, int n_left, int filter_type, int *out) {
  constexpr int RL = raw_len(B), VL = var_len(B);
  constexpr int sa = ct_slot_above(MODE), sl = ct_slot_left(MODE);
  const T *above = sa >= 0 ? edge + 2 * RL + sa * VL + kRawPad : edge + kRawPad;
  const T *left = sl >= 0 ? edge + 2 * RL + sl * VL + kRawPad : edge + RL + kRawPad;
  constexpr int a = ct_mode_angle(MODE);
  if constexpr (MODE == V_PRED) {
#pragma unroll
    for (int c = 0; c < B; c++) out[c] = above[c];
  } else if constexpr (MODE == H_PRED) {
    const int v = left[r];
#pragma unroll
    for (int c = 0; c < B; c++) out[c] = v;
  } else if constexpr (MODE == D45_PRED || MODE == D67_PRED) {
    constexpr int dx = ct_derivative(a);
    const int up = ct_upsample<B, a - 90>(filter_type);
    const int max_base_x = (2 * B - 1) << up;
    const int x = (r + 1) * dx;
    const int b0 = x >> (6 - up), shift = ((x << up) & 0x3F) >> 1;
    const int vmax = above[max_base_x];
#pragma unroll
    for (int c = 0; c < B; c++) {
      const int bs = b0 + (c << up), bc = min(bs, max_base_x - 1);
      const int v = (above[bc] * (32 - shift) + above[bc + 1] * shift + 16) >> 5;
      out[c] = vmax ^ ((vmax ^ v) & ((bs - max_base_x) >> 31));      // bs < max_base_x ? v : vmax, as an integer mask
    }
  } else if constexpr (MODE == D113_PRED || MODE == D135_PRED || MODE == D157_PRED) {
    constexpr int dx = ct_derivative(180 - a), dy = ct_derivative(a - 90);
    const int upa = ct_upsample<B, a - 90>(filter_type), upl = ct_upsample<B, a - 180>(filter_type);
#pragma unroll
    for (int c = 0; c < B; c++) {
      const int x = (c << 6) - (r + 1) * dx, y = (r << 6) - (c + 1) * dy;
      const int base_x = x >> (6 - upa), base_y = y >> (6 - upl);
      // all ones where the sample projects onto the left edge: an integer mask, not a lane-mask register pair (the compiler
      // hoisted those out of the candidates and spilled them to VGPR lanes, paying v_readlane + hazard nops per use)
      const int use_left = (base_x + (1 << upa)) >> 31;
      const int ia = max(base_x, -2), il = max(base_y, -2);
      const int sha = ((x * (1 << upa)) & 0x3F) >> 1, shl = ((y * (1 << upl)) & 0x3F) >> 1;
      const int va = (above[ia] * (32 - sha) + above[ia + 1] * sha + 16) >> 5;
      const int vl = (left[il] * (32 - shl) + left[il + 1] * shl + 16) >> 5;
      out[c] = va ^ ((va ^ vl) & use_left);
    }
  } else if constexpr (MODE == D203_PRED) {
    constexpr int dy = ct_derivative(270 - a);
    const int up = ct_upsample<B, a - 180>(filter_type);
    const int max_base_y = (2 * B - 1) << up;
    const int vmax = left[max_base_y];
#pragma unroll
    for (int c = 0; c < B; c++) {
      const int y = (c + 1) * dy;
      const int bs = (y >> (6 - up)) + (r << up), shift = ((y << up) & 0x3F) >> 1, bc = min(bs, max_base_y - 1);
      const int v = (left[bc] * (32 - shift) + left[bc + 1] * shift + 16) >> 5;
      out[c] = vmax ^ ((vmax ^ v) & ((bs - max_base_y) >> 31));      // bs < max_base_y ? v : vmax
    }
  } else if constexpr (MODE == DC_PRED) {
    // lane r contributes above[r] + left[r]; the B lanes of the block sum by xor-shuffles
    const int sa_ = group_sum<B>(above[r]), sl_ = group_sum<B>(left[r]);
    constexpr int lg = B == 4 ? 2 : B == 8 ? 3 : B == 16 ? 4 : B == 32 ? 5 : 6;
    const bool ht = n_top > 0, hl = n_left > 0;
    const int both = (sa_ + sl_ + B) >> (lg + 1), one = ((ht ? sa_ : sl_) + B / 2) >> lg;
    const int v = (ht && hl) ? both : (ht || hl) ? one : (128 << (bd - 8));
#pragma unroll
    for (int c = 0; c < B; c++) out[c] = v;
  } else if constexpr (MODE == PAETH_PRED) {
    const int tl = above[-1], l = left[r];
#pragma unroll
    for (int c = 0; c < B; c++) {
      const int t = above[c], b = t + l - tl;
      const int pl = abs(b - l), pt = abs(b - t), ptl = abs(b - tl);
      out[c] = (pl <= pt && pl <= ptl) ? l : (pt <= ptl) ? t : tl;
    }
  } else if constexpr (MODE == SMOOTH_V_PRED) {
    const int below = left[B - 1], wh = sm_weight(B, r);
#pragma unroll
    for (int c = 0; c < B; c++) out[c] = (wh * above[c] + (256 - wh) * below + 128) >> 8;
  } else if constexpr (MODE == SMOOTH_H_PRED) {
    const int right = above[B - 1], l = left[r];
#pragma unroll
    for (int c = 0; c < B; c++) {
      const int ww = sm_weight(B, c);
      out[c] = (ww * l + (256 - ww) * right + 128) >> 8;
    }
  } else {   // SMOOTH_PRED
    const int below = left[B - 1], right = above[B - 1], l = left[r], wh = sm_weight(B, r);
#pragma unroll
    for (int c = 0; c < B; c++) {
      const int ww = sm_weight(B, c);
      out[c] = (wh * above[c] + (256 - wh) * below + ww * l + (256 - ww) * right + 256) >> 9;
    }
  }
}

}  // namespace av1mi
