// intra.hpp — AV1 intra prediction of one transform block by a group of lanes of one wave.
//
// The group (L lanes, L divides 64, all in one wave) first builds the two edge arrays in LDS
// (above row incl. top-left and above-right, left column incl. bottom-left; unavailable-sample
// rules, corner filter, 5-tap edge filter, 2x edge upsampling), then every lane produces whole
// rows of the prediction from LDS.  Restates AV1 spec §7.11.2 and libaom build_intra_predictors /
// av1_dr_prediction_z1/z2/z3_c / av1_filter_intra_edge_c / av1_upsample_intra_edge_c (SURVEY.md §8a
// row K3); the reference tree has no counterpart (transcode.go:120).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace av1mi {

enum { DC_PRED, V_PRED, H_PRED, D45_PRED, D135_PRED, D113_PRED, D157_PRED, D203_PRED, D67_PRED,
       SMOOTH_PRED, SMOOTH_V_PRED, SMOOTH_H_PRED, PAETH_PRED, INTRA_MODES };

// lanes of one wave: LDS traffic is in order per wave, this only stops the compiler moving LDS
// accesses across the hand-off (and drains lgkmcnt).
#define AV1MI_GROUP_SYNC()                                   \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)

// Sum over the G (4, 8, 16 or 32) consecutive lanes of a group with DPP cross-lane VALU moves (no LDS round trip, unlike
// __shfl_xor's ds_bpermute): quad xor 1, quad xor 2, then half-row / row mirrors pair the quads.  Every lane gets the sum.
template <int G> __device__ __forceinline__ int group_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);                       // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);                       // quad_perm [2,3,0,1]
  if constexpr (G >= 8) v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);   // row_half_mirror: lane i <-> 7-i
  if constexpr (G >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);  // row_mirror: lane i <-> 15-i
  if constexpr (G >= 32) v += __shfl_xor(v, 16, 64);                                      // the other 16-lane row of the group (32x32 blocks: key frames only)
  return v;
}
template <int G> __device__ __forceinline__ int group_or(int v) {
  v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
  v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
  if constexpr (G >= 8) v |= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
  if constexpr (G >= 16) v |= __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
  if constexpr (G >= 32) v |= __shfl_xor(v, 16, 64);
  return v;
}

constexpr int kEdgePad = 16;             // entries in front of index 0 (top-left lives at -1, upsampling writes -2)
// entries of one edge array (uint16) for a BW x BH block: pad + (w+h, doubled where upsampling can apply) + slack
__host__ __device__ constexpr int edge_len(int bw, int bh) {
  return (kEdgePad + ((bw + bh) <= 16 ? 2 * (bw + bh) : (bw + bh)) + 8 + 7) / 8 * 8;
}

__device__ __forceinline__ int mode_angle(int mode) {
  // 0, 90, 180, 45, 135, 113, 157, 203, 67
  const int t[9] = { 0, 90, 180, 45, 135, 113, 157, 203, 67 };
  return mode <= D67_PRED ? t[mode] : 0;
}
// spec Dr_Intra_Derivative on the 3-degree lattice (angle in 3..87)
__device__ __forceinline__ int dr_derivative(int angle) {
  switch (angle) {
    case 3: return 1023; case 6: return 547; case 9: return 372; case 14: return 273; case 17: return 215;
    case 20: return 178; case 23: return 151; case 26: return 132; case 29: return 116; case 32: return 102;
    case 36: return 90; case 39: return 80; case 42: return 71; case 45: return 64; case 48: return 57;
    case 51: return 51; case 54: return 45; case 58: return 40; case 61: return 35; case 64: return 31;
    case 67: return 27; case 70: return 23; case 73: return 19; case 76: return 15; case 81: return 11;
    case 84: return 7; case 87: return 3; default: return 0;
  }
}
__device__ __forceinline__ int edge_filter_strength(int bs0, int bs1, int delta, int type) {
  const int d = abs(delta), wh = bs0 + bs1;
  int s = 0;
  if (type == 0) {
    if (wh <= 8) { if (d >= 56) s = 1; }
    else if (wh <= 16) { if (d >= 40) s = 1; }
    else if (wh <= 24) { if (d >= 8) s = 1; if (d >= 16) s = 2; if (d >= 32) s = 3; }
    else if (wh <= 32) { if (d >= 1) s = 1; if (d >= 4) s = 2; if (d >= 32) s = 3; }
    else { if (d >= 1) s = 3; }
  } else {
    if (wh <= 8) { if (d >= 40) s = 1; if (d >= 64) s = 2; }
    else if (wh <= 16) { if (d >= 20) s = 1; if (d >= 48) s = 2; }
    else if (wh <= 24) { if (d >= 4) s = 3; }
    else { if (d >= 1) s = 3; }
  }
  return s;
}
__device__ __forceinline__ int use_edge_upsample(int bs0, int bs1, int delta, int type) {
  const int d = abs(delta), wh = bs0 + bs1;
  if (d == 0 || d >= 40) return 0;
  return type ? (wh <= 8) : (wh <= 16);
}
__device__ __forceinline__ int sm_weight(int n, int i) {
  constexpr uint8_t w[4 + 8 + 16 + 32 + 64] = {
    255, 149, 85, 64,
    255, 197, 146, 105, 73, 50, 37, 32,
    255, 225, 196, 170, 145, 123, 102, 84, 68, 54, 43, 33, 26, 20, 17, 16,
    255, 240, 225, 210, 196, 182, 169, 157, 145, 133, 122, 111, 101, 92, 83, 74, 66, 59, 52, 45, 39, 34, 29, 25, 21, 17, 14, 12, 10, 9, 8, 8,
    255, 248, 240, 233, 225, 218, 210, 203, 196, 189, 182, 176, 169, 163, 156, 150, 144, 138, 133, 127, 121, 116, 111, 106, 101, 96, 91, 86, 82, 77, 73, 69,
    65, 61, 57, 54, 50, 47, 44, 41, 38, 35, 32, 29, 27, 25, 22, 20, 18, 16, 15, 13, 12, 10, 9, 8, 7, 6, 6, 5, 5, 4, 4, 4 };
  return w[n - 4 + i];   // offsets 0,4,12,28,60 == n-4 for n = 4,8,16,32,64
}

struct IntraBlk {       // what the caller knows about one block (libaom build_intra_predictors arguments)
  int mode, angle_delta, disable_edge_filter, filter_type;
  int n_top, n_topright, n_left, n_bottomleft;
};
struct IntraEdges {     // filled by intra_build_edges, consumed by intra_pred_row
  int p_angle, is_dr, flat, flat_val, upsample_above, upsample_left;
  int have_top, have_left;
};

// 5-tap edge filter, entry i of dst[0..sz-1] from src[] (src/dst index 0 == first entry of the filtered run)
__device__ __forceinline__ int edge_filter_tap(const uint16_t *src, int i, int sz, int strength) {
  const int k0 = strength == 3 ? 2 : 0, k1 = strength == 2 ? 5 : 4, k2 = strength == 1 ? 8 : strength == 2 ? 6 : 4;
  const int im2 = max(i - 2, 0), im1 = max(i - 1, 0), ip1 = min(i + 1, sz - 1), ip2 = min(i + 2, sz - 1);
  return (k0 * (src[im2] + src[ip2]) + k1 * (src[im1] + src[ip1]) + k2 * src[i] + 8) >> 4;
}

// Build the edges of one BW x BH block.  `lane` in [0,L).  above/left point at index 0 of LDS arrays of
// kEdgeLen entries each (so above[-1] is the top-left); tmpa/tmpl are scratch arrays of the same shape.
// fetch(y, x) returns the reconstructed sample at (y, x) relative to the block's top-left.
template <int BW, int BH, int L, typename Fetch>
__device__ __forceinline__ IntraEdges intra_build_edges(const IntraBlk &B, int bd, int lane, uint16_t *above, uint16_t *left,
                                                        uint16_t *tmpa, uint16_t *tmpl, Fetch fetch) {
  IntraEdges E;
  const int base = 128 << (bd - 8);
  E.is_dr = B.mode >= V_PRED && B.mode <= D67_PRED;
  E.p_angle = E.is_dr ? mode_angle(B.mode) + B.angle_delta * 3 : 0;
  E.have_top = B.n_top > 0; E.have_left = B.n_left > 0;
  E.upsample_above = E.upsample_left = 0; E.flat = 0; E.flat_val = 0;
  bool need_left, need_above, need_al;
  if (E.is_dr) { need_above = E.p_angle < 180; need_left = E.p_angle > 90; need_al = true; }
  else { need_above = need_left = true; need_al = B.mode == PAETH_PRED; }
  if ((!need_above && B.n_left == 0) || (!need_left && B.n_top == 0)) {
    E.flat = 1;
    if (need_left) E.flat_val = B.n_top > 0 ? fetch(-1, 0) : base + 1;
    else E.flat_val = B.n_left > 0 ? fetch(0, -1) : base - 1;
    return E;
  }
  const bool need_bottom = E.is_dr && E.p_angle > 180, need_right = E.is_dr && E.p_angle < 90;
  const int nl = BH + (need_bottom ? BW : 0), na = BW + (need_right ? BH : 0);
  // raw edges -> tmp arrays (index 0 == first sample, -1 == top-left)
  if (need_left) {
    const int avail = B.n_left > 0 ? B.n_left + (need_bottom ? B.n_bottomleft : 0) : 0;
    for (int i = lane; i < nl; i += L) {
      int v;
      if (B.n_left > 0) v = fetch(min(i, avail - 1), -1);
      else v = B.n_top > 0 ? fetch(-1, 0) : base + 1;
      tmpl[i] = (uint16_t)v;
    }
  }
  if (need_above) {
    const int avail = B.n_top > 0 ? B.n_top + (need_right ? B.n_topright : 0) : 0;
    for (int i = lane; i < na; i += L) {
      int v;
      if (B.n_top > 0) v = fetch(-1, min(i, avail - 1));
      else v = B.n_left > 0 ? fetch(0, -1) : base - 1;
      tmpa[i] = (uint16_t)v;
    }
  }
  if (lane == 0) {
    int tl = base;
    if (B.n_top > 0 && B.n_left > 0) tl = fetch(-1, -1);
    else if (B.n_top > 0) tl = fetch(-1, 0);
    else if (B.n_left > 0) tl = fetch(0, -1);
    tmpa[-1] = tmpl[-1] = (uint16_t)(need_al ? tl : 0);
  }
  AV1MI_GROUP_SYNC();
  int sa = 0, sl = 0;
  bool corner = false;
  if (E.is_dr && !B.disable_edge_filter) {
    if (E.p_angle != 90 && E.p_angle != 180) {
      corner = need_above && need_left && (BW + BH >= 24);
      if (need_above && B.n_top > 0) sa = edge_filter_strength(BW, BH, E.p_angle - 90, B.filter_type);
      if (need_left && B.n_left > 0) sl = edge_filter_strength(BH, BW, E.p_angle - 180, B.filter_type);
    }
    E.upsample_above = need_above ? use_edge_upsample(BW, BH, E.p_angle - 90, B.filter_type) : 0;
    E.upsample_left = need_left ? use_edge_upsample(BH, BW, E.p_angle - 180, B.filter_type) : 0;
  }
  // corner filter (reads raw neighbours, writes both top-left copies)
  if (corner) {
    const int s = (tmpl[0] * 5 + tmpa[-1] * 6 + tmpa[0] * 5 + 8) >> 4;
    AV1MI_GROUP_SYNC();
    if (lane == 0) { tmpa[-1] = (uint16_t)s; tmpl[-1] = (uint16_t)s; }
    AV1MI_GROUP_SYNC();
  }
  // edge filter: tmp (-1..n-1) -> above/left (-1..n-1).  The run filtered by libaom is
  // p = edge - 1 (top-left included as tap, never rewritten), length n_avail + 1 + extension.
  {
    const int sza = B.n_top + 1 + (need_right ? BH : 0);
    for (int i = lane; i < na + 1; i += L) {
      int v = tmpa[i - 1];
      if (need_above && sa && i >= 1 && i < sza) v = edge_filter_tap(tmpa - 1, i, sza, sa);
      if (need_above) above[i - 1] = (uint16_t)v;
    }
    const int szl = B.n_left + 1 + (need_bottom ? BW : 0);
    for (int i = lane; i < nl + 1; i += L) {
      int v = tmpl[i - 1];
      if (need_left && sl && i >= 1 && i < szl) v = edge_filter_tap(tmpl - 1, i, szl, sl);
      if (need_left) left[i - 1] = (uint16_t)v;
    }
  }
  AV1MI_GROUP_SYNC();
  // 2x upsampling: above/left (-1..n-1) -> tmp -> back (-2..2n-2)
  if (E.upsample_above || E.upsample_left) {
    const int maxv = (1 << bd) - 1;
    if (E.upsample_above) {
      for (int i = lane; i < na; i += L) {
        const int a = above[max(i - 2, -1)], b = above[i - 1], c = above[i], d = above[min(i + 1, na - 1)];
        tmpa[2 * i - 1] = (uint16_t)min(max((-a + 9 * b + 9 * c - d + 8) >> 4, 0), maxv);
        tmpa[2 * i] = (uint16_t)c;
      }
      if (lane == 0) tmpa[-2] = above[-1];
    }
    if (E.upsample_left) {
      for (int i = lane; i < nl; i += L) {
        const int a = left[max(i - 2, -1)], b = left[i - 1], c = left[i], d = left[min(i + 1, nl - 1)];
        tmpl[2 * i - 1] = (uint16_t)min(max((-a + 9 * b + 9 * c - d + 8) >> 4, 0), maxv);
        tmpl[2 * i] = (uint16_t)c;
      }
      if (lane == 0) tmpl[-2] = left[-1];
    }
    AV1MI_GROUP_SYNC();
    if (E.upsample_above) for (int i = lane; i < 2 * na + 1; i += L) above[i - 2] = tmpa[i - 2];
    if (E.upsample_left) for (int i = lane; i < 2 * nl + 1; i += L) left[i - 2] = tmpl[i - 2];
    AV1MI_GROUP_SYNC();
  }
  return E;
}

// prediction of row r (BW samples) of the block into out[]; T = sample type of the edge arrays (uint8_t / uint16_t)
template <int BW, int BH, typename T = uint16_t>
__device__ __forceinline__ void intra_pred_row(const IntraBlk &B, const IntraEdges &E, int bd, int r, const T *above,
                                               const T *left, int *out) {
  if (E.flat) {
#pragma unroll
    for (int c = 0; c < BW; c++) out[c] = E.flat_val;
    return;
  }
  if (E.is_dr) {
    const int a = E.p_angle;
    if (a == 90) {
#pragma unroll
      for (int c = 0; c < BW; c++) out[c] = above[c];
    } else if (a == 180) {
      const int v = left[r];
#pragma unroll
      for (int c = 0; c < BW; c++) out[c] = v;
    } else if (a < 90) {
      const int dx = dr_derivative(a), up = E.upsample_above;
      const int max_base_x = ((BW + BH) - 1) << up;
      const int x = (r + 1) * dx;
      int bs = x >> (6 - up);
      const int shift = ((x << up) & 0x3F) >> 1;
#pragma unroll
      for (int c = 0; c < BW; c++, bs += 1 << up)
        out[c] = bs < max_base_x ? (above[bs] * (32 - shift) + above[bs + 1] * shift + 16) >> 5 : above[max_base_x];
    } else if (a < 180) {
      const int dx = dr_derivative(180 - a), dy = dr_derivative(a - 90);
      const int upa = E.upsample_above, upl = E.upsample_left;
#pragma unroll
      for (int c = 0; c < BW; c++) {
        const int x = (c << 6) - (r + 1) * dx;
        const int base_x = x >> (6 - upa);
        int v;
        if (base_x >= -(1 << upa)) {
          const int shift = ((x * (1 << upa)) & 0x3F) >> 1;
          v = (above[base_x] * (32 - shift) + above[base_x + 1] * shift + 16) >> 5;
        } else {
          const int y = (r << 6) - (c + 1) * dy;
          const int base_y = y >> (6 - upl);
          const int shift = ((y * (1 << upl)) & 0x3F) >> 1;
          v = (left[base_y] * (32 - shift) + left[base_y + 1] * shift + 16) >> 5;
        }
        out[c] = v;
      }
    } else {
      const int dy = dr_derivative(270 - a), up = E.upsample_left;
      const int max_base_y = (BW + BH - 1) << up;
#pragma unroll
      for (int c = 0; c < BW; c++) {
        const int y = (c + 1) * dy;
        const int bs = (y >> (6 - up)) + (r << up);
        const int shift = ((y << up) & 0x3F) >> 1;
        out[c] = bs < max_base_y ? (left[bs] * (32 - shift) + left[bs + 1] * shift + 16) >> 5 : left[max_base_y];
      }
    }
    return;
  }
  if (B.mode == DC_PRED) {
    // every lane sums the (short) edges itself: BW + BH LDS reads, no cross-lane step
    int sum = 0, cnt = 0;
    if (E.have_top) { for (int i = 0; i < BW; i++) sum += above[i]; cnt += BW; }
    if (E.have_left) { for (int i = 0; i < BH; i++) sum += left[i]; cnt += BH; }
    const int v = cnt ? (sum + (cnt >> 1)) / cnt : 128 << (bd - 8);
#pragma unroll
    for (int c = 0; c < BW; c++) out[c] = v;
  } else if (B.mode == PAETH_PRED) {
    const int tl = above[-1], l = left[r];
#pragma unroll
    for (int c = 0; c < BW; c++) {
      const int t = above[c], b = t + l - tl;
      const int pl = abs(b - l), pt = abs(b - t), ptl = abs(b - tl);
      out[c] = (pl <= pt && pl <= ptl) ? l : (pt <= ptl) ? t : tl;
    }
  } else {
    const int below = left[BH - 1], right = above[BW - 1], l = left[r], wh = sm_weight(BH, r);
#pragma unroll
    for (int c = 0; c < BW; c++) {
      const int ww = sm_weight(BW, c), t = above[c];
      int v;
      if (B.mode == SMOOTH_PRED) v = (wh * t + (256 - wh) * below + ww * l + (256 - ww) * right + 256) >> 9;
      else if (B.mode == SMOOTH_V_PRED) v = (wh * t + (256 - wh) * below + 128) >> 8;
      else v = (ww * l + (256 - ww) * right + 128) >> 8;
      out[c] = v;
    }
  }
}

}  // namespace av1mi
