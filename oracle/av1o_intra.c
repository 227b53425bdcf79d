/*
 * av1o_intra.c — CPU oracle for SURVEY.md §8 row K3: intra prediction of one transform block
 * (directional modes incl. intra-edge filter / corner filter / edge upsampling; plus the adjacent
 * DC, Paeth and Smooth predictors the intra-only pipeline needs).
 *
 * TEST INFRASTRUCTURE ONLY; pinned to dav1d, not to the reference (see av1o_common.h).  Restates, from knowledge:
 *   av1o_intra_predict      AV1 spec §7.11.2 "intra prediction process" == libaom
 *                           build_intra_predictors() (av1/common/reconintra.c), incl. the
 *                           unavailable-edge rules (base-1 above / base+1 left)
 *   dr_z1/z2/z3             spec §7.11.2.4 == libaom av1_dr_prediction_z1/z2/z3_c
 *   edge_filter_strength    spec §7.11.2.9  == libaom intra_edge_filter_strength
 *   filter_edge / corner    spec §7.11.2.12 / §7.11.2.7 == av1_filter_intra_edge_c, filter_intra_edge_corner
 *   use_upsample / upsample spec §7.11.2.10/11 == av1_use_intra_edge_upsample, av1_upsample_intra_edge_c
 *   Dr_Intra_Derivative     spec §7.11.2.4 table == libaom dr_intra_derivative[]
 *   Sm_Weights_*            spec §7.11.2.6 == libaom sm_weight_arrays
 * The reference tree has nothing for this (transcode.go:120 names the external encoder only).
 */
#include "av1o_common.h"
#include <stdlib.h>
#include <string.h>

enum { DC_PRED, V_PRED, H_PRED, D45_PRED, D135_PRED, D113_PRED, D157_PRED, D203_PRED, D67_PRED,
       SMOOTH_PRED, SMOOTH_V_PRED, SMOOTH_H_PRED, PAETH_PRED, INTRA_MODES };

static const int mode_to_angle[INTRA_MODES] = { 0, 90, 180, 45, 135, 113, 157, 203, 67, 0, 0, 0, 0 };

/* indexed by angle (degrees, 0..89); non-zero only at the 3-degree lattice AV1 uses */
static int dr_derivative(int angle) {
  switch (angle) {
    case 3: return 1023; case 6: return 547; case 9: return 372; case 14: return 273; case 17: return 215;
    case 20: return 178; case 23: return 151; case 26: return 132; case 29: return 116; case 32: return 102;
    case 36: return 90; case 39: return 80; case 42: return 71; case 45: return 64; case 48: return 57;
    case 51: return 51; case 54: return 45; case 58: return 40; case 61: return 35; case 64: return 31;
    case 67: return 27; case 70: return 23; case 73: return 19; case 76: return 15; case 81: return 11;
    case 84: return 7; case 87: return 3; default: return 0;
  }
}
static int get_dx(int angle) {
  if (angle > 0 && angle < 90) return dr_derivative(angle);
  if (angle > 90 && angle < 180) return dr_derivative(180 - angle);
  return 1;
}
static int get_dy(int angle) {
  if (angle > 90 && angle < 180) return dr_derivative(angle - 90);
  if (angle > 180 && angle < 270) return dr_derivative(270 - angle);
  return 1;
}

static const uint8_t sm_w4[4] = { 255, 149, 85, 64 };
static const uint8_t sm_w8[8] = { 255, 197, 146, 105, 73, 50, 37, 32 };
static const uint8_t sm_w16[16] = { 255, 225, 196, 170, 145, 123, 102, 84, 68, 54, 43, 33, 26, 20, 17, 16 };
static const uint8_t sm_w32[32] = { 255, 240, 225, 210, 196, 182, 169, 157, 145, 133, 122, 111, 101, 92, 83, 74,
                                    66, 59, 52, 45, 39, 34, 29, 25, 21, 17, 14, 12, 10, 9, 8, 8 };
static const uint8_t sm_w64[64] = { 255, 248, 240, 233, 225, 218, 210, 203, 196, 189, 182, 176, 169, 163, 156, 150,
                                    144, 138, 133, 127, 121, 116, 111, 106, 101, 96, 91, 86, 82, 77, 73, 69,
                                    65, 61, 57, 54, 50, 47, 44, 41, 38, 35, 32, 29, 27, 25, 22, 20,
                                    18, 16, 15, 13, 12, 10, 9, 8, 7, 6, 6, 5, 5, 4, 4, 4 };
static const uint8_t *sm_weights(int n) {
  return n == 4 ? sm_w4 : n == 8 ? sm_w8 : n == 16 ? sm_w16 : n == 32 ? sm_w32 : sm_w64;
}
const uint8_t *av1o_sm_weights(int n) { return sm_weights(n); }

int av1o_intra_edge_filter_strength(int bs0, int bs1, int delta, int type) {
  const int d = abs(delta), blk_wh = bs0 + bs1;
  int strength = 0;
  if (type == 0) {
    if (blk_wh <= 8) { if (d >= 56) strength = 1; }
    else if (blk_wh <= 12) { if (d >= 40) strength = 1; }
    else if (blk_wh <= 16) { if (d >= 40) strength = 1; }
    else if (blk_wh <= 24) { if (d >= 8) strength = 1; if (d >= 16) strength = 2; if (d >= 32) strength = 3; }
    else if (blk_wh <= 32) { if (d >= 1) strength = 1; if (d >= 4) strength = 2; if (d >= 32) strength = 3; }
    else { if (d >= 1) strength = 3; }
  } else {
    if (blk_wh <= 8) { if (d >= 40) strength = 1; if (d >= 64) strength = 2; }
    else if (blk_wh <= 16) { if (d >= 20) strength = 1; if (d >= 48) strength = 2; }
    else if (blk_wh <= 24) { if (d >= 4) strength = 3; }
    else { if (d >= 1) strength = 3; }
  }
  return strength;
}
int av1o_use_intra_edge_upsample(int bs0, int bs1, int delta, int type) {
  const int d = abs(delta), blk_wh = bs0 + bs1;
  if (d == 0 || d >= 40) return 0;
  return type ? (blk_wh <= 8) : (blk_wh <= 16);
}
/* p[0..sz-1]; p[0] is kept, p[1..] filtered with edge clamping */
void av1o_filter_intra_edge(uint16_t *p, int sz, int strength) {
  static const int kernel[3][5] = { { 0, 4, 8, 4, 0 }, { 0, 5, 6, 5, 0 }, { 2, 4, 4, 4, 2 } };
  uint16_t edge[160];
  if (!strength) return;
  memcpy(edge, p, sz * sizeof(*p));
  for (int i = 1; i < sz; i++) {
    int s = 0;
    for (int j = 0; j < 5; j++) {
      int k = i - 2 + j;
      k = k < 0 ? 0 : k;
      k = k > sz - 1 ? sz - 1 : k;
      s += edge[k] * kernel[strength - 1][j];
    }
    p[i] = (uint16_t)((s + 8) >> 4);
  }
}
/* p[-2..2*sz-2] written; p[-1..sz-1] read */
void av1o_upsample_intra_edge(uint16_t *p, int sz, int bd) {
  uint16_t in[16 + 3];
  const int maxv = (1 << bd) - 1;
  in[0] = p[-1]; in[1] = p[-1];
  for (int i = 0; i < sz; i++) in[i + 2] = p[i];
  in[sz + 2] = p[sz - 1];
  p[-2] = in[0];
  for (int i = 0; i < sz; i++) {
    int s = -in[i] + 9 * in[i + 1] + 9 * in[i + 2] - in[i + 3];
    s = av1o_clampi((s + 8) >> 4, 0, maxv);
    p[2 * i - 1] = (uint16_t)s;
    p[2 * i] = in[i + 2];
  }
}

static void dr_z1(uint16_t *dst, int stride, int bw, int bh, const uint16_t *above, int upsample_above, int dx) {
  const int max_base_x = ((bw + bh) - 1) << upsample_above;
  const int frac_bits = 6 - upsample_above, base_inc = 1 << upsample_above;
  int x = dx;
  for (int r = 0; r < bh; ++r, dst += stride, x += dx) {
    int base = x >> frac_bits;
    const int shift = ((x << upsample_above) & 0x3F) >> 1;
    if (base >= max_base_x) {
      for (int i = r; i < bh; ++i, dst += stride)
        for (int c = 0; c < bw; c++) dst[c] = above[max_base_x];
      return;
    }
    for (int c = 0; c < bw; ++c, base += base_inc) {
      if (base < max_base_x) dst[c] = (uint16_t)av1o_round2(above[base] * (32 - shift) + above[base + 1] * shift, 5);
      else dst[c] = above[max_base_x];
    }
  }
}
static void dr_z2(uint16_t *dst, int stride, int bw, int bh, const uint16_t *above, const uint16_t *left,
                  int upsample_above, int upsample_left, int dx, int dy) {
  const int min_base_x = -(1 << upsample_above);
  const int frac_bits_x = 6 - upsample_above, frac_bits_y = 6 - upsample_left;
  for (int r = 0; r < bh; ++r, dst += stride)
    for (int c = 0; c < bw; ++c) {
      int val, y = r + 1, x = (c << 6) - y * dx;
      const int base_x = x >> frac_bits_x;
      if (base_x >= min_base_x) {
        const int shift = ((x * (1 << upsample_above)) & 0x3F) >> 1;
        val = av1o_round2(above[base_x] * (32 - shift) + above[base_x + 1] * shift, 5);
      } else {
        x = c + 1;
        y = (r << 6) - x * dy;
        const int base_y = y >> frac_bits_y;
        const int shift = ((y * (1 << upsample_left)) & 0x3F) >> 1;
        val = av1o_round2(left[base_y] * (32 - shift) + left[base_y + 1] * shift, 5);
      }
      dst[c] = (uint16_t)val;
    }
}
static void dr_z3(uint16_t *dst, int stride, int bw, int bh, const uint16_t *left, int upsample_left, int dy) {
  const int max_base_y = (bw + bh - 1) << upsample_left;
  const int frac_bits = 6 - upsample_left, base_inc = 1 << upsample_left;
  int y = dy;
  for (int c = 0; c < bw; ++c, y += dy) {
    int base = y >> frac_bits;
    const int shift = ((y << upsample_left) & 0x3F) >> 1;
    for (int r = 0; r < bh; ++r, base += base_inc) {
      if (base < max_base_y) dst[r * stride + c] = (uint16_t)av1o_round2(left[base] * (32 - shift) + left[base + 1] * shift, 5);
      else { for (; r < bh; ++r) dst[r * stride + c] = left[max_base_y]; break; }
    }
  }
}

/*
 * Predict one bw x bh transform block into pred[] (uint16, stride bw).
 * ref points at the block's top-left sample inside the reconstructed plane (uint8 if bd==8, else uint16),
 * ref_stride in samples.  n_*_px are the counts of AVAILABLE neighbour samples exactly as libaom's
 * build_intra_predictors() takes them (n_top_px in {0..bw}, n_topright_px in {0..bw},
 * n_left_px in {0..bh}, n_bottomleft_px in {0..bh}).  mode: 0..12, angle_delta: -3..3 (directional only).
 * filter_type: 1 when a neighbouring block is smooth-predicted (spec get_filter_type).
 */
int av1o_intra_predict(const void *ref, int ref_stride, int bd, int bw, int bh, int mode, int angle_delta,
                       int disable_edge_filter, int filter_type, int n_top_px, int n_topright_px, int n_left_px,
                       int n_bottomleft_px, uint16_t *pred) {
  uint16_t above_data[16 + 160], left_data[16 + 160];
  uint16_t *const above_row = above_data + 16, *const left_col = left_data + 16;
  const int base = 128 << (bd - 8);
  if (mode < 0 || mode >= INTRA_MODES || angle_delta < -3 || angle_delta > 3) return -1;
#define REF(yy, xx) (bd == 8 ? (int)((const uint8_t *)ref)[(ptrdiff_t)(yy) * ref_stride + (xx)] \
                             : (int)((const uint16_t *)ref)[(ptrdiff_t)(yy) * ref_stride + (xx)])
  const int is_dr = mode >= V_PRED && mode <= D67_PRED;
  const int p_angle = is_dr ? mode_to_angle[mode] + angle_delta * 3 : 0;
  int need_left, need_above, need_above_left;
  if (is_dr) {
    if (p_angle <= 90) { need_above = 1; need_left = 0; need_above_left = 1; }
    else if (p_angle < 180) { need_above = 1; need_left = 1; need_above_left = 1; }
    else { need_above = 0; need_left = 1; need_above_left = 1; }
  } else {
    need_above = need_left = 1;            /* DC, SMOOTH*, PAETH */
    need_above_left = mode == PAETH_PRED;
  }
  memset(above_data, 0, sizeof(above_data));
  memset(left_data, 0, sizeof(left_data));
  if ((!need_above && n_left_px == 0) || (!need_left && n_top_px == 0)) {
    int val;
    if (need_left) val = n_top_px > 0 ? REF(-1, 0) : base + 1;
    else val = n_left_px > 0 ? REF(0, -1) : base - 1;
    for (int i = 0; i < bw * bh; i++) pred[i] = (uint16_t)val;
    return 0;
  }
  if (need_left) {
    const int need_bottom = is_dr ? p_angle > 180 : 0;
    const int needed = bh + (need_bottom ? bw : 0);
    int i = 0;
    if (n_left_px > 0) {
      for (; i < n_left_px; i++) left_col[i] = (uint16_t)REF(i, -1);
      if (need_bottom && n_bottomleft_px > 0)
        for (; i < bh + n_bottomleft_px; i++) left_col[i] = (uint16_t)REF(i, -1);
      for (; i < needed; i++) left_col[i] = left_col[i - 1];
    } else if (n_top_px > 0) {
      for (; i < needed; i++) left_col[i] = (uint16_t)REF(-1, 0);
    } else {
      for (; i < needed; i++) left_col[i] = (uint16_t)(base + 1);
    }
  }
  if (need_above) {
    const int need_right = is_dr ? p_angle < 90 : 0;
    const int needed = bw + (need_right ? bh : 0);
    int i = 0;
    if (n_top_px > 0) {
      for (; i < n_top_px; i++) above_row[i] = (uint16_t)REF(-1, i);
      if (need_right && n_topright_px > 0)
        for (; i < bw + n_topright_px; i++) above_row[i] = (uint16_t)REF(-1, i);
      for (; i < needed; i++) above_row[i] = above_row[i - 1];
    } else if (n_left_px > 0) {
      for (; i < needed; i++) above_row[i] = (uint16_t)REF(0, -1);
    } else {
      for (; i < needed; i++) above_row[i] = (uint16_t)(base - 1);
    }
  }
  if (need_above_left) {
    if (n_top_px > 0 && n_left_px > 0) above_row[-1] = (uint16_t)REF(-1, -1);
    else if (n_top_px > 0) above_row[-1] = (uint16_t)REF(-1, 0);
    else if (n_left_px > 0) above_row[-1] = (uint16_t)REF(0, -1);
    else above_row[-1] = (uint16_t)base;
    left_col[-1] = above_row[-1];
  }
#undef REF
  if (is_dr) {
    int upsample_above = 0, upsample_left = 0;
    if (!disable_edge_filter) {
      const int need_right = p_angle < 90, need_bottom = p_angle > 180;
      if (p_angle != 90 && p_angle != 180) {
        const int ab_le = need_above_left ? 1 : 0;
        if (need_above && need_left && (bw + bh >= 24)) {
          const int s = (left_col[0] * 5 + above_row[-1] * 6 + above_row[0] * 5 + 8) >> 4;
          above_row[-1] = (uint16_t)s; left_col[-1] = (uint16_t)s;
        }
        if (need_above && n_top_px > 0)
          av1o_filter_intra_edge(above_row - ab_le, n_top_px + ab_le + (need_right ? bh : 0),
                                 av1o_intra_edge_filter_strength(bw, bh, p_angle - 90, filter_type));
        if (need_left && n_left_px > 0)
          av1o_filter_intra_edge(left_col - ab_le, n_left_px + ab_le + (need_bottom ? bw : 0),
                                 av1o_intra_edge_filter_strength(bh, bw, p_angle - 180, filter_type));
      }
      upsample_above = av1o_use_intra_edge_upsample(bw, bh, p_angle - 90, filter_type);
      if (need_above && upsample_above) av1o_upsample_intra_edge(above_row, bw + (need_right ? bh : 0), bd);
      upsample_left = av1o_use_intra_edge_upsample(bh, bw, p_angle - 180, filter_type);
      if (need_left && upsample_left) av1o_upsample_intra_edge(left_col, bh + (need_bottom ? bw : 0), bd);
    }
    const int dx = get_dx(p_angle), dy = get_dy(p_angle);
    if (p_angle > 0 && p_angle < 90) dr_z1(pred, bw, bw, bh, above_row, upsample_above, dx);
    else if (p_angle > 90 && p_angle < 180) dr_z2(pred, bw, bw, bh, above_row, left_col, upsample_above, upsample_left, dx, dy);
    else if (p_angle > 180 && p_angle < 270) dr_z3(pred, bw, bw, bh, left_col, upsample_left, dy);
    else if (p_angle == 90) { for (int r = 0; r < bh; r++) for (int c = 0; c < bw; c++) pred[r * bw + c] = above_row[c]; }
    else { for (int r = 0; r < bh; r++) for (int c = 0; c < bw; c++) pred[r * bw + c] = left_col[r]; }
    return 0;
  }
  if (mode == DC_PRED) {
    int sum = 0, cnt = 0;
    if (n_top_px > 0) { for (int i = 0; i < bw; i++) sum += above_row[i]; cnt += bw; }
    if (n_left_px > 0) { for (int i = 0; i < bh; i++) sum += left_col[i]; cnt += bh; }
    const int v = cnt ? (sum + (cnt >> 1)) / cnt : base;
    for (int i = 0; i < bw * bh; i++) pred[i] = (uint16_t)v;
  } else if (mode == PAETH_PRED) {
    const int tl = above_row[-1];
    for (int r = 0; r < bh; r++)
      for (int c = 0; c < bw; c++) {
        const int l = left_col[r], t = above_row[c], b = t + l - tl;
        const int pl = abs(b - l), pt = abs(b - t), ptl = abs(b - tl);
        pred[r * bw + c] = (uint16_t)((pl <= pt && pl <= ptl) ? l : (pt <= ptl) ? t : tl);
      }
  } else {
    const uint8_t *ww = sm_weights(bw), *wh = sm_weights(bh);
    const int below = left_col[bh - 1], right = above_row[bw - 1];
    for (int r = 0; r < bh; r++)
      for (int c = 0; c < bw; c++) {
        int v;
        if (mode == SMOOTH_PRED)
          v = (wh[r] * above_row[c] + (256 - wh[r]) * below + ww[c] * left_col[r] + (256 - ww[c]) * right + 256) >> 9;
        else if (mode == SMOOTH_V_PRED) v = (wh[r] * above_row[c] + (256 - wh[r]) * below + 128) >> 8;
        else v = (ww[c] * left_col[r] + (256 - ww[c]) * right + 128) >> 8;
        pred[r * bw + c] = (uint16_t)v;
      }
  }
  return 0;
}


/*
 * Chroma-from-luma prediction, 4:2:0: AV1 spec 7.11.5 (predict chroma from luma process) == libaom cfl_predict_block
 * (cfl_luma_subsampling_420_*, subtract average, cfl_predict_*).  `luma` is the reconstructed luma plane, `dst` the chroma
 * plane whose block at (x, y), bw x bh (each 4..32), already holds the DC prediction; it is overwritten with
 *   Clip1(dc + Round2Signed(alpha_q3 * (L[i][j] - Round2(sum L, log2 bw + log2 bh)), 6)),
 *   L[i][j] = (sum of the 2x2 luma samples at (2(y+i), 2(x+j))) << 1,
 * luma coordinates limited to max_luma_w - 2 / max_luma_h - 2 (the spec's MaxLumaW / MaxLumaH: the part of the luma
 * block that has been reconstructed; beyond it the last available 2x2 group is repeated).
 */
int av1o_cfl_predict(const void *luma, int luma_stride, void *dst, int dst_stride, int bd, int x, int y, int bw, int bh,
                     int alpha_q3, int max_luma_w, int max_luma_h) {
  int32_t L[32 * 32];
  int64_t sum = 0;
  int i, j, lg = 0;
  const int maxpix = (1 << bd) - 1;
  if (bw < 4 || bh < 4 || bw > 32 || bh > 32 || (bw & (bw - 1)) || (bh & (bh - 1)) || alpha_q3 < -16 || alpha_q3 > 16) return -1;
  if (max_luma_w < 2 || max_luma_h < 2) return -1;
  for (i = bw * bh; i > 1; i >>= 1) lg++;
  for (i = 0; i < bh; i++)
    for (j = 0; j < bw; j++) {
      int ly = 2 * (y + i), lx = 2 * (x + j), t = 0, dy, dx;
      if (ly > max_luma_h - 2) ly = max_luma_h - 2;
      if (lx > max_luma_w - 2) lx = max_luma_w - 2;
      for (dy = 0; dy < 2; dy++)
        for (dx = 0; dx < 2; dx++)
          t += bd == 8 ? ((const uint8_t *)luma)[(size_t)(ly + dy) * luma_stride + lx + dx]
                       : ((const uint16_t *)luma)[(size_t)(ly + dy) * luma_stride + lx + dx];
      L[i * bw + j] = t << 1;
      sum += L[i * bw + j];
    }
  {
    const int avg = (int)((sum + ((int64_t)1 << (lg - 1))) >> lg);
    for (i = 0; i < bh; i++)
      for (j = 0; j < bw; j++) {
        const int v = alpha_q3 * (L[i * bw + j] - avg);
        const int sl = v >= 0 ? (v + 32) >> 6 : -((-v + 32) >> 6);
        if (bd == 8) { uint8_t *p = (uint8_t *)dst + (size_t)(y + i) * dst_stride + x + j; *p = (uint8_t)av1o_clampi(*p + sl, 0, maxpix); }
        else { uint16_t *p = (uint16_t *)dst + (size_t)(y + i) * dst_stride + x + j; *p = (uint16_t)av1o_clampi(*p + sl, 0, maxpix); }
      }
  }
  return 0;
}
