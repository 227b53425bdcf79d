/*
 * av1o_mc.c — CPU oracle for SURVEY.md §8 row K4: sub-pel motion compensation of one block (single reference,
 * no scaling, no compound): separable 8-tap FIR, 1/16-sample phases, two-stage rounding.
 *
 * TEST INFRASTRUCTURE ONLY; pinned to dav1d, not to the reference (see av1o_common.h).  Restates, from knowledge:
 *   av1o_mc_block         AV1 spec §7.11.3.4 "block inter prediction process" (InterRound0 = 3, InterRound1 = 11
 *                         for a single 8/10-bit prediction) == libaom av1_highbd_convolve_2d_sr_c (its bias
 *                         terms cancel exactly), plus the reference-edge clamping of §7.11.3.3/4
 *   av1o_subpel_filters   spec Subpel_Filters[6][16][8] == libaom av1_sub_pel_filters_8 / _8sharp / _8smooth /
 *                         _4 / _4smooth (filter_idx 0 regular, 1 smooth, 2 sharp, 3 bilinear, 4 regular 4-tap,
 *                         5 smooth 4-tap).  The reference tree has no counterpart (transcode.go:120).
 */
#include "av1o_common.h"

const int16_t av1o_subpel_filters[6][16][8] = {
  { /* 0: EIGHTTAP (regular) */
    { 0, 0, 0, 128, 0, 0, 0, 0 }, { 0, 2, -6, 126, 8, -2, 0, 0 }, { 0, 2, -10, 122, 18, -4, 0, 0 },
    { 0, 2, -12, 116, 28, -8, 2, 0 }, { 0, 2, -14, 110, 38, -10, 2, 0 }, { 0, 2, -14, 102, 48, -12, 2, 0 },
    { 0, 2, -16, 94, 58, -12, 2, 0 }, { 0, 2, -14, 84, 66, -12, 2, 0 }, { 0, 2, -14, 76, 76, -14, 2, 0 },
    { 0, 2, -12, 66, 84, -14, 2, 0 }, { 0, 2, -12, 58, 94, -16, 2, 0 }, { 0, 2, -12, 48, 102, -14, 2, 0 },
    { 0, 2, -10, 38, 110, -14, 2, 0 }, { 0, 2, -8, 28, 116, -12, 2, 0 }, { 0, 0, -4, 18, 122, -10, 2, 0 },
    { 0, 0, -2, 8, 126, -6, 2, 0 } },
  { /* 1: EIGHTTAP_SMOOTH */
    { 0, 0, 0, 128, 0, 0, 0, 0 }, { 0, 2, 28, 62, 34, 2, 0, 0 }, { 0, 0, 26, 62, 36, 4, 0, 0 },
    { 0, 0, 22, 62, 40, 4, 0, 0 }, { 0, 0, 20, 60, 42, 6, 0, 0 }, { 0, 0, 18, 58, 44, 8, 0, 0 },
    { 0, 0, 16, 56, 46, 10, 0, 0 }, { 0, -2, 16, 54, 48, 12, 0, 0 }, { 0, -2, 14, 52, 52, 14, -2, 0 },
    { 0, 0, 12, 48, 54, 16, -2, 0 }, { 0, 0, 10, 46, 56, 16, 0, 0 }, { 0, 0, 8, 44, 58, 18, 0, 0 },
    { 0, 0, 6, 42, 60, 20, 0, 0 }, { 0, 0, 4, 40, 62, 22, 0, 0 }, { 0, 0, 4, 36, 62, 26, 0, 0 },
    { 0, 0, 2, 34, 62, 28, 2, 0 } },
  { /* 2: EIGHTTAP_SHARP */
    { 0, 0, 0, 128, 0, 0, 0, 0 }, { -2, 2, -6, 126, 8, -2, 2, 0 }, { -2, 6, -12, 124, 16, -6, 4, -2 },
    { -2, 8, -18, 120, 26, -10, 6, -2 }, { -4, 10, -22, 116, 38, -14, 6, -2 }, { -4, 10, -22, 108, 48, -18, 8, -2 },
    { -4, 10, -24, 100, 60, -20, 8, -2 }, { -4, 10, -24, 90, 70, -22, 10, -2 }, { -4, 12, -24, 80, 80, -24, 12, -4 },
    { -2, 10, -22, 70, 90, -24, 10, -4 }, { -2, 8, -20, 60, 100, -24, 10, -4 }, { -2, 8, -18, 48, 108, -22, 10, -4 },
    { -2, 6, -14, 38, 116, -22, 10, -4 }, { -2, 6, -10, 26, 120, -18, 8, -2 }, { -2, 4, -6, 16, 124, -12, 6, -2 },
    { 0, 2, -2, 8, 126, -6, 2, -2 } },
  { /* 3: BILINEAR */
    { 0, 0, 0, 128, 0, 0, 0, 0 }, { 0, 0, 0, 120, 8, 0, 0, 0 }, { 0, 0, 0, 112, 16, 0, 0, 0 },
    { 0, 0, 0, 104, 24, 0, 0, 0 }, { 0, 0, 0, 96, 32, 0, 0, 0 }, { 0, 0, 0, 88, 40, 0, 0, 0 },
    { 0, 0, 0, 80, 48, 0, 0, 0 }, { 0, 0, 0, 72, 56, 0, 0, 0 }, { 0, 0, 0, 64, 64, 0, 0, 0 },
    { 0, 0, 0, 56, 72, 0, 0, 0 }, { 0, 0, 0, 48, 80, 0, 0, 0 }, { 0, 0, 0, 40, 88, 0, 0, 0 },
    { 0, 0, 0, 32, 96, 0, 0, 0 }, { 0, 0, 0, 24, 104, 0, 0, 0 }, { 0, 0, 0, 16, 112, 0, 0, 0 },
    { 0, 0, 0, 8, 120, 0, 0, 0 } },
  { /* 4: regular, 4 taps (block dimension <= 4) */
    { 0, 0, 0, 128, 0, 0, 0, 0 }, { 0, 0, -4, 126, 8, -2, 0, 0 }, { 0, 0, -8, 122, 18, -4, 0, 0 },
    { 0, 0, -10, 116, 28, -6, 0, 0 }, { 0, 0, -12, 110, 38, -8, 0, 0 }, { 0, 0, -12, 102, 48, -10, 0, 0 },
    { 0, 0, -14, 94, 58, -10, 0, 0 }, { 0, 0, -12, 84, 66, -10, 0, 0 }, { 0, 0, -12, 76, 76, -12, 0, 0 },
    { 0, 0, -10, 66, 84, -12, 0, 0 }, { 0, 0, -10, 58, 94, -14, 0, 0 }, { 0, 0, -10, 48, 102, -12, 0, 0 },
    { 0, 0, -8, 38, 110, -12, 0, 0 }, { 0, 0, -6, 28, 116, -10, 0, 0 }, { 0, 0, -4, 18, 122, -8, 0, 0 },
    { 0, 0, -2, 8, 126, -4, 0, 0 } },
  { /* 5: smooth, 4 taps */
    { 0, 0, 0, 128, 0, 0, 0, 0 }, { 0, 0, 30, 62, 34, 2, 0, 0 }, { 0, 0, 26, 62, 36, 4, 0, 0 },
    { 0, 0, 22, 62, 40, 4, 0, 0 }, { 0, 0, 20, 60, 42, 6, 0, 0 }, { 0, 0, 18, 58, 44, 8, 0, 0 },
    { 0, 0, 16, 56, 46, 10, 0, 0 }, { 0, 0, 14, 54, 48, 12, 0, 0 }, { 0, 0, 12, 52, 52, 12, 0, 0 },
    { 0, 0, 12, 48, 54, 14, 0, 0 }, { 0, 0, 10, 46, 56, 16, 0, 0 }, { 0, 0, 8, 44, 58, 18, 0, 0 },
    { 0, 0, 6, 42, 60, 20, 0, 0 }, { 0, 0, 4, 40, 62, 22, 0, 0 }, { 0, 0, 4, 36, 62, 26, 0, 0 },
    { 0, 0, 2, 34, 62, 30, 0, 0 } },
};

/* spec §7.11.3.4: interp filter type (0 regular, 1 smooth, 2 sharp, 3 bilinear) -> Subpel_Filters row for a dimension */
int av1o_mc_filter_index(int type, int dim) {
  if (dim <= 4) { if (type == 0 || type == 2) return 4; if (type == 1) return 5; }
  return type;
}

/* tests switch the whole-sample shortcut off to check it against the filter path */
static int mc_fast_path = 1;
void av1o_mc_set_fast_path(int on) { mc_fast_path = on; }

/*
 * Predict a w x h block at (x, y) of a plane from the reference plane `ref` (plane_w x plane_h, stride in samples),
 * displaced by (mvx, mvy) in 1/16-sample units of THIS plane.  filt_x / filt_y: interpolation filter types.
 * pred: uint16, stride w.
 */
int av1o_mc_block(const void *ref, int stride, int plane_w, int plane_h, int bd, int x, int y, int w, int h, int mvx,
                  int mvy, int filt_x, int filt_y, uint16_t *pred) {
  if (w > 128 || h > 128 || w < 2 || h < 2 || filt_x < 0 || filt_x > 3 || filt_y < 0 || filt_y > 3) return -1;
  static int32_t inter[(128 + 7) * 128];
  int32_t *im = inter;
  int32_t local[(64 + 7) * 64];
  if (w <= 64 && h <= 64) im = local;
  const int posx = x * 16 + mvx, posy = y * 16 + mvy;
  const int x0 = posx >> 4, y0 = posy >> 4, px = posx & 15, py = posy & 15;
  const int16_t *fx = av1o_subpel_filters[av1o_mc_filter_index(filt_x, w)][px];
  const int16_t *fy = av1o_subpel_filters[av1o_mc_filter_index(filt_y, h)][py];
  const int round0 = 3, round1 = 11;
  if (px == 0 && py == 0 && mc_fast_path) {
    /* whole-sample position: phase 0 of every filter family is {0,0,0,128,0,0,0,0}, so both stages are exact
     * (Round2(128 v, 3) = 16 v, Round2(128 * 16 v, 11) = v): the prediction is the edge-clamped copy.  Same result as the
     * general path below (tests/test_oracle_mc.py checks it); it only keeps the integer search of the encoder loop and
     * the CPU baseline from paying 2 x 8 taps per sample for a copy. */
    for (int r = 0; r < h; r++) {
      const int ry = av1o_clampi(y0 + r, 0, plane_h - 1);
      for (int c = 0; c < w; c++) {
        const int rx = av1o_clampi(x0 + c, 0, plane_w - 1);
        pred[r * w + c] = bd == 8 ? ((const uint8_t *)ref)[(size_t)ry * stride + rx] : ((const uint16_t *)ref)[(size_t)ry * stride + rx];
      }
    }
    return 0;
  }
  for (int r = 0; r < h + 7; r++) {
    const int ry = av1o_clampi(y0 + r - 3, 0, plane_h - 1);
    for (int c = 0; c < w; c++) {
      int s = 0;
      for (int t = 0; t < 8; t++) {
        const int rx = av1o_clampi(x0 + c + t - 3, 0, plane_w - 1);
        const int v = bd == 8 ? ((const uint8_t *)ref)[(size_t)ry * stride + rx] : ((const uint16_t *)ref)[(size_t)ry * stride + rx];
        s += fx[t] * v;
      }
      im[r * w + c] = av1o_round2(s, round0);
    }
  }
  for (int r = 0; r < h; r++)
    for (int c = 0; c < w; c++) {
      int s = 0;
      for (int t = 0; t < 8; t++) s += fy[t] * im[(r + t) * w + c];
      pred[r * w + c] = (uint16_t)av1o_clampi(av1o_round2(s, round1), 0, (1 << bd) - 1);
    }
  return 0;
}
