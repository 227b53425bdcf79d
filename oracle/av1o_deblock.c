/*
 * av1o_deblock.c — CPU oracle for SURVEY.md §8 row K5: the deblocking loop filter of one plane.
 *
 * TEST INFRASTRUCTURE ONLY; pinned to dav1d, not to the reference (see av1o_common.h).  Restates, from knowledge:
 *   av1o_deblock_plane        AV1 spec §7.14.2 edge loop (pass 0: all vertical edges of the plane, then
 *                             pass 1: all horizontal edges), §7.14.3 filter size, §7.14.4 adaptive strength
 *   lf_limits                 spec §7.14.4 == libaom update_sharpness()/av1_loop_filter_init
 *   filter4/6/8/14 + masks    spec §7.14.6 (narrow / wide sample filters) == libaom aom_dsp/loopfilter.c
 *                             highbd_filter4, highbd_filter6, highbd_filter8, highbd_filter14,
 *                             highbd_filter_mask*, highbd_flat_mask*, highbd_hev_mask
 * Per 4x4 unit of the plane the caller supplies what the spec derives from the mode info:
 *   byte 0: log2(tx width) | log2(tx height) << 4   (LoopfilterTxSizes)
 *   byte 1: filter level for vertical edges (pass 0) of the block owning the unit, 0..63
 *   byte 2: filter level for horizontal edges (pass 1)
 *   byte 3: flags: bit0 = skip && is_inter (inner transform edges not filtered),
 *                  bit1 = the unit's LEFT edge is a prediction-block edge, bit2 = its TOP edge is one
 */
#include "av1o_common.h"
#include <stdlib.h>
#include <string.h>

typedef struct { int lim, mblim, hev; } lf_thr;

static lf_thr lf_limits(int lvl, int sharp) {
  lf_thr t;
  const int shift = sharp > 4 ? 2 : (sharp > 0 ? 1 : 0);
  int inside = lvl >> shift;
  if (sharp > 0 && inside > 9 - sharp) inside = 9 - sharp;
  if (inside < 1) inside = 1;
  t.lim = inside; t.mblim = 2 * (lvl + 2) + inside; t.hev = lvl >> 4;
  return t;
}
static int sclamp(int t, int bd) { return av1o_clampi(t, -(128 << (bd - 8)), (128 << (bd - 8)) - 1); }

static int mask2(int limit, int blimit, int p1, int p0, int q0, int q1, int bd) {
  const int l = limit << (bd - 8), bl = blimit << (bd - 8);
  int m = 0;
  m |= abs(p1 - p0) > l; m |= abs(q1 - q0) > l;
  m |= abs(p0 - q0) * 2 + abs(p1 - q1) / 2 > bl;
  return !m;
}
static int mask3(int limit, int blimit, int p2, int p1, int p0, int q0, int q1, int q2, int bd) {
  const int l = limit << (bd - 8);
  return mask2(limit, blimit, p1, p0, q0, q1, bd) && !(abs(p2 - p1) > l) && !(abs(q2 - q1) > l);
}
static int mask4(int limit, int blimit, int p3, int p2, int p1, int p0, int q0, int q1, int q2, int q3, int bd) {
  const int l = limit << (bd - 8);
  return mask3(limit, blimit, p2, p1, p0, q0, q1, q2, bd) && !(abs(p3 - p2) > l) && !(abs(q3 - q2) > l);
}
static int flat3(int p2, int p1, int p0, int q0, int q1, int q2, int bd) {
  const int t = 1 << (bd - 8);
  return !(abs(p1 - p0) > t || abs(q1 - q0) > t || abs(p2 - p0) > t || abs(q2 - q0) > t);
}
static int flat4(int p3, int p2, int p1, int p0, int q0, int q1, int q2, int q3, int bd) {
  const int t = 1 << (bd - 8);
  return flat3(p2, p1, p0, q0, q1, q2, bd) && !(abs(p3 - p0) > t || abs(q3 - q0) > t);
}
static void filter4(int mask, int thresh, int *op1, int *op0, int *oq0, int *oq1, int bd) {
  const int t80 = 0x80 << (bd - 8), th = thresh << (bd - 8);
  const int ps1 = *op1 - t80, ps0 = *op0 - t80, qs0 = *oq0 - t80, qs1 = *oq1 - t80;
  const int hev = abs(*op1 - *op0) > th || abs(*oq1 - *oq0) > th;
  int f = hev ? sclamp(ps1 - qs1, bd) : 0;
  f = mask ? sclamp(f + 3 * (qs0 - ps0), bd) : 0;
  const int f1 = sclamp(f + 4, bd) >> 3, f2 = sclamp(f + 3, bd) >> 3;
  *oq0 = sclamp(qs0 - f1, bd) + t80;
  *op0 = sclamp(ps0 + f2, bd) + t80;
  f = hev ? 0 : (f1 + 1) >> 1;
  *oq1 = sclamp(qs1 - f, bd) + t80;
  *op1 = sclamp(ps1 + f, bd) + t80;
}
#define RP2(v, n) (((v) + (1 << ((n) - 1))) >> (n))

/* px[0..15] = p7..p0 q0..q7 (p0 = px[7], q0 = px[8]); len in {4,6,8,14} */
static void filter_edge_samples(int *px, int len, lf_thr t, int bd) {
  int *p = px + 7, *q = px + 8; /* p[-i] = p_i, q[i] = q_i */
#define P(i) p[-(i)]
#define Q(i) q[(i)]
  if (len == 4) {
    const int m = mask2(t.lim, t.mblim, P(1), P(0), Q(0), Q(1), bd);
    filter4(m, t.hev, &P(1), &P(0), &Q(0), &Q(1), bd);
  } else if (len == 6) {
    const int m = mask3(t.lim, t.mblim, P(2), P(1), P(0), Q(0), Q(1), Q(2), bd);
    const int fl = flat3(P(2), P(1), P(0), Q(0), Q(1), Q(2), bd);
    if (fl && m) {
      const int p2 = P(2), p1 = P(1), p0 = P(0), q0 = Q(0), q1 = Q(1), q2 = Q(2);
      P(1) = RP2(p2 * 3 + p1 * 2 + p0 * 2 + q0, 3);
      P(0) = RP2(p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1, 3);
      Q(0) = RP2(p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2, 3);
      Q(1) = RP2(p0 + q0 * 2 + q1 * 2 + q2 * 3, 3);
    } else filter4(m, t.hev, &P(1), &P(0), &Q(0), &Q(1), bd);
  } else {
    const int m = mask4(t.lim, t.mblim, P(3), P(2), P(1), P(0), Q(0), Q(1), Q(2), Q(3), bd);
    const int fl = flat4(P(3), P(2), P(1), P(0), Q(0), Q(1), Q(2), Q(3), bd);
    const int fl2 = len == 14 && flat4(P(6), P(5), P(4), P(0), Q(0), Q(4), Q(5), Q(6), bd);
    const int p6 = P(6), p5 = P(5), p4 = P(4), p3 = P(3), p2 = P(2), p1 = P(1), p0 = P(0);
    const int q0 = Q(0), q1 = Q(1), q2 = Q(2), q3 = Q(3), q4 = Q(4), q5 = Q(5), q6 = Q(6);
    if (fl2 && fl && m) {
      P(5) = RP2(p6 * 7 + p5 * 2 + p4 * 2 + p3 + p2 + p1 + p0 + q0, 4);
      P(4) = RP2(p6 * 5 + p5 * 2 + p4 * 2 + p3 * 2 + p2 + p1 + p0 + q0 + q1, 4);
      P(3) = RP2(p6 * 4 + p5 + p4 * 2 + p3 * 2 + p2 * 2 + p1 + p0 + q0 + q1 + q2, 4);
      P(2) = RP2(p6 * 3 + p5 + p4 + p3 * 2 + p2 * 2 + p1 * 2 + p0 + q0 + q1 + q2 + q3, 4);
      P(1) = RP2(p6 * 2 + p5 + p4 + p3 + p2 * 2 + p1 * 2 + p0 * 2 + q0 + q1 + q2 + q3 + q4, 4);
      P(0) = RP2(p6 + p5 + p4 + p3 + p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1 + q2 + q3 + q4 + q5, 4);
      Q(0) = RP2(p5 + p4 + p3 + p2 + p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2 + q3 + q4 + q5 + q6, 4);
      Q(1) = RP2(p4 + p3 + p2 + p1 + p0 + q0 * 2 + q1 * 2 + q2 * 2 + q3 + q4 + q5 + q6 * 2, 4);
      Q(2) = RP2(p3 + p2 + p1 + p0 + q0 + q1 * 2 + q2 * 2 + q3 * 2 + q4 + q5 + q6 * 3, 4);
      Q(3) = RP2(p2 + p1 + p0 + q0 + q1 + q2 * 2 + q3 * 2 + q4 * 2 + q5 + q6 * 4, 4);
      Q(4) = RP2(p1 + p0 + q0 + q1 + q2 + q3 * 2 + q4 * 2 + q5 * 2 + q6 * 5, 4);
      Q(5) = RP2(p0 + q0 + q1 + q2 + q3 + q4 * 2 + q5 * 2 + q6 * 7, 4);
    } else if (fl && m) {
      P(2) = RP2(p3 + p3 + p3 + 2 * p2 + p1 + p0 + q0, 3);
      P(1) = RP2(p3 + p3 + p2 + 2 * p1 + p0 + q0 + q1, 3);
      P(0) = RP2(p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2, 3);
      Q(0) = RP2(p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3, 3);
      Q(1) = RP2(p1 + p0 + q0 + 2 * q1 + q2 + q3 + q3, 3);
      Q(2) = RP2(p0 + q0 + q1 + 2 * q2 + q3 + q3 + q3, 3);
    } else filter4(m, t.hev, &P(1), &P(0), &Q(0), &Q(1), bd);
  }
#undef P
#undef Q
}

static int getpx(const void *pl, int bd, size_t i) { return bd == 8 ? ((const uint8_t *)pl)[i] : ((const uint16_t *)pl)[i]; }
static void setpx(void *pl, int bd, size_t i, int v) { if (bd == 8) ((uint8_t *)pl)[i] = (uint8_t)v; else ((uint16_t *)pl)[i] = (uint16_t)v; }

/*
 * Deblock one plane in place.  w,h: plane size in samples (multiples of 4); mi: (h/4) x (w/4) units, 4 bytes each
 * (see file header), mi_stride in units.  is_chroma selects the chroma filter-length rule.  pass_mask: bit0 =
 * run pass 0 (vertical edges), bit1 = run pass 1 (horizontal edges).
 */
int av1o_deblock_plane(void *plane, int stride, int w, int h, int bd, int is_chroma, const uint8_t *mi, int mi_stride,
                       int sharpness, int pass_mask) {
  if ((w & 3) || (h & 3) || (bd != 8 && bd != 10)) return -1;
  const int cols = w / 4, rows = h / 4;
  for (int pass = 0; pass < 2; pass++) {
    if (!(pass_mask & (1 << pass))) continue;
    for (int r = 0; r < rows; r++)
      for (int c = 0; c < cols; c++) {
        const uint8_t *cur = mi + ((size_t)r * mi_stride + c) * 4;
        const int x = c * 4, y = r * 4;
        if (pass == 0 ? x == 0 : y == 0) continue;                      /* picture edge */
        const int txw = 1 << (cur[0] & 15), txh = 1 << (cur[0] >> 4);
        const int is_tx_edge = pass == 0 ? (x % txw) == 0 : (y % txh) == 0;
        if (!is_tx_edge) continue;
        const int is_blk_edge = pass == 0 ? (cur[3] >> 1) & 1 : (cur[3] >> 2) & 1;
        const int skip_inter = cur[3] & 1;
        if (!(is_blk_edge || !skip_inter)) continue;
        const uint8_t *prev = pass == 0 ? cur - 4 : cur - (size_t)mi_stride * 4;
        const int ptxw = 1 << (prev[0] & 15), ptxh = 1 << (prev[0] >> 4);
        const int base = pass == 0 ? (txw < ptxw ? txw : ptxw) : (txh < ptxh ? txh : ptxh);
        const int fsize = is_chroma ? (base < 8 ? base : 8) : (base < 16 ? base : 16);
        int lvl = cur[1 + pass];
        if (lvl == 0) lvl = prev[1 + pass];
        if (lvl == 0) continue;
        const int len = is_chroma ? (fsize == 4 ? 4 : 6) : (fsize == 4 ? 4 : fsize == 8 ? 8 : 14);
        const lf_thr t = lf_limits(lvl, sharpness);
        for (int i = 0; i < 4; i++) {
          int px[16];
          const int half = len == 14 ? 7 : len == 8 ? 4 : len == 6 ? 3 : 2;
          for (int k = -half; k < half; k++) {
            const size_t idx = pass == 0 ? (size_t)(y + i) * stride + (x + k) : (size_t)(y + k) * stride + (x + i);
            px[8 + k] = getpx(plane, bd, idx);
          }
          filter_edge_samples(px, len, t, bd);
          for (int k = -half; k < half; k++) {
            const size_t idx = pass == 0 ? (size_t)(y + i) * stride + (x + k) : (size_t)(y + k) * stride + (x + i);
            setpx(plane, bd, idx, px[8 + k]);
          }
        }
      }
  }
  return 0;
}
