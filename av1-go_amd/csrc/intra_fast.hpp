// intra_fast.hpp — the fused pipeline's edge builder: ALL edge variants of a block in two LDS phases.
//
// intra.hpp builds the edges of one (block, mode) pair and needs ~10 wave-level hand-offs to do it; the mode search
// of the fused kernel would pay that 11 times per block.  Here the group fetches the raw neighbours once, then every
// lane computes a share of the entries of every derived edge array the candidate set needs (filtered / upsampled,
// with or without the above-right / bottom-left extension, corner-filtered top-left) straight from the raw arrays —
// filter taps and upsampling taps are evaluated on the fly, so no derived array depends on another — and after ONE
// more hand-off all candidates are predicted from LDS without further synchronisation.
// Same arithmetic as intra.hpp (AV1 spec §7.11.2.7-12; libaom filter_intra_edge_corner, av1_filter_intra_edge_c,
// av1_upsample_intra_edge_c); the pipeline parity tests compare the result with the oracle's per-mode builder.
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

struct FastMode {          // per candidate, identical in all lanes of the group
  int mode, p_angle, is_dr, flat, flat_val, upsample_above, upsample_left;
  int a_slot, l_slot;      // derived array used for above / left (-1: raw)
};

__device__ __forceinline__ int variant_slot_above(int mode) { return mode == D45_PRED ? 0 : mode == D67_PRED ? 1 : mode == D113_PRED ? 2 : mode == D135_PRED ? 3 : mode == D157_PRED ? 4 : -1; }
__device__ __forceinline__ int variant_slot_left(int mode) { return mode == D113_PRED ? 5 : mode == D135_PRED ? 6 : mode == D157_PRED ? 7 : mode == D203_PRED ? 8 : -1; }

// value of p-space entry i (0 = top-left) of one edge after corner + edge filter, from the raw array
template <typename T>
__device__ __forceinline__ int filt_entry(const T *raw, int tl, int i, int sz, int strength) {
  auto P = [&](int k) -> int { return k == 0 ? tl : (int)raw[k - 1]; };
  if (!strength || i < 1 || i >= sz) return P(i);
  const int k0 = strength == 3 ? 2 : 0, k1 = strength == 2 ? 5 : 4, k2 = strength == 1 ? 8 : strength == 2 ? 6 : 4;
  const int im2 = max(i - 2, 0), im1 = max(i - 1, 0), ip1 = min(i + 1, sz - 1), ip2 = min(i + 2, sz - 1);
  return (k0 * (P(im2) + P(ip2)) + k1 * (P(im1) + P(ip1)) + k2 * P(i) + 8) >> 4;
}

// Build raw + derived arrays for a B x B block.  `edge` points at fast_edge_len(B) entries of LDS owned by the group
// (L = B lanes).  Returns nothing; use fast_mode_setup() + fast_arrays() to predict.
template <int B, typename T, typename Fetch>
__device__ __forceinline__ void fast_build(T *edge, int lane, int bd, int n_top, int n_topright, int n_left, int n_bottomleft,
                                           int filter_type, Fetch fetch) {
  constexpr int RL = raw_len(B), VL = var_len(B);
  T *ra = edge + kRawPad, *rl = edge + RL + kRawPad;
  const int base = 128 << (bd - 8);
  // phase 1: raw neighbours (above-right / bottom-left included when available, else replicated)
  {
    const int availa = n_top + n_topright, availl = n_left + n_bottomleft;
    for (int i = lane; i < 2 * B; i += B) {
      ra[i] = (T)(n_top > 0 ? fetch(-1, min(i, availa - 1)) : n_left > 0 ? fetch(0, -1) : base - 1);
      rl[i] = (T)(n_left > 0 ? fetch(min(i, availl - 1), -1) : n_top > 0 ? fetch(-1, 0) : base + 1);
    }
    if (lane == 0) {
      int tl = base;
      if (n_top > 0 && n_left > 0) tl = fetch(-1, -1);
      else if (n_top > 0) tl = fetch(-1, 0);
      else if (n_left > 0) tl = fetch(0, -1);
      ra[-1] = rl[-1] = (T)tl;
    }
  }
  AV1MI_GROUP_SYNC();
  // phase 2: derived arrays.  Slot s: modes {45, 67, 113, 135, 157} above, {113, 135, 157, 203} left.
  const int tl_raw = ra[-1];
  const int tl_corner = (2 * B >= 24) ? (rl[0] * 5 + tl_raw * 6 + ra[0] * 5 + 8) >> 4 : tl_raw;
  const int maxv = (1 << bd) - 1;
#pragma unroll
  for (int s = 0; s < kNumVariants; s++) {
    constexpr int angles[kNumVariants] = { 45, 67, 113, 135, 157, 113, 135, 157, 203 };
    const int a = angles[s];
    const bool is_above = s < 5;
    const bool both = a > 90 && a < 180;                         // zone 2: corner filter applies
    const bool ext = is_above ? a < 90 : a > 180;                 // need_right / need_bottom
    const int n = B + (ext ? B : 0);
    const int navail = is_above ? n_top : n_left;
    const int delta = is_above ? a - 90 : a - 180;
    const int strength = navail > 0 ? edge_filter_strength(B, B, delta, filter_type) : 0;
    const int up = use_edge_upsample(B, B, delta, filter_type);
    const int sz = navail + 1 + (ext ? B : 0);
    const T *raw = is_above ? ra : rl;
    const int tl = both ? tl_corner : tl_raw;
    T *out = edge + 2 * RL + s * VL + kRawPad;
    if (!up) {
      for (int j = lane - 1; j < n; j += B) out[j] = (T)filt_entry(raw, tl, j + 1, sz, strength);
    } else {
      for (int j = lane; j < n; j += B) {
        const int va = filt_entry(raw, tl, max(j - 2, -1) + 1, sz, strength), vb = filt_entry(raw, tl, j, sz, strength);
        const int vc = filt_entry(raw, tl, j + 1, sz, strength), vd = filt_entry(raw, tl, min(j + 1, n - 1) + 1, sz, strength);
        out[2 * j - 1] = (T)min(max((-va + 9 * vb + 9 * vc - vd + 8) >> 4, 0), maxv);
        out[2 * j] = (T)vc;
      }
      if (lane == 0) out[-2] = (T)filt_entry(raw, tl, 0, sz, strength);
    }
  }
  AV1MI_GROUP_SYNC();
}

// per-candidate description (scalar logic only, no LDS traffic)
template <int B, typename Fetch>
__device__ __forceinline__ FastMode fast_mode_setup(int mode, int bd, int n_top, int n_left, int filter_type, Fetch fetch) {
  FastMode M;
  const int base = 128 << (bd - 8);
  M.mode = mode;
  M.is_dr = mode >= V_PRED && mode <= D67_PRED;
  M.p_angle = M.is_dr ? mode_angle(mode) : 0;
  M.flat = 0; M.flat_val = 0; M.upsample_above = M.upsample_left = 0;
  const bool need_above = !M.is_dr || M.p_angle < 180, need_left = !M.is_dr || M.p_angle > 90;
  if ((!need_above && n_left == 0) || (!need_left && n_top == 0)) {
    M.flat = 1;
    M.flat_val = need_left ? (n_top > 0 ? fetch(-1, 0) : base + 1) : (n_left > 0 ? fetch(0, -1) : base - 1);
  }
  M.a_slot = variant_slot_above(mode); M.l_slot = variant_slot_left(mode);
  if (M.a_slot >= 0) M.upsample_above = use_edge_upsample(B, B, M.p_angle - 90, filter_type);
  if (M.l_slot >= 0) M.upsample_left = use_edge_upsample(B, B, M.p_angle - 180, filter_type);
  return M;
}
template <int B, typename T>
__device__ __forceinline__ void fast_arrays(const T *edge, const FastMode &M, const T *&above, const T *&left) {
  constexpr int RL = raw_len(B), VL = var_len(B);
  above = M.a_slot >= 0 ? edge + 2 * RL + M.a_slot * VL + kRawPad : edge + kRawPad;
  left = M.l_slot >= 0 ? edge + 2 * RL + M.l_slot * VL + kRawPad : edge + RL + kRawPad;
}

}  // namespace av1mi
