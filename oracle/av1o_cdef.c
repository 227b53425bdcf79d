/*
 * av1o_cdef.c — CPU oracle for SURVEY.md §8 row K6: the constrained directional enhancement filter, 4:2:0.
 *
 * TEST INFRASTRUCTURE ONLY; pinned to dav1d, not to the reference (see av1o_common.h).  Restates, from knowledge:
 *   av1o_cdef_find_dir    AV1 spec §7.15.2 "CDEF direction process" == libaom cdef_find_dir_c (av1/common/cdef_block.c)
 *   cdef_filter_block     spec §7.15.3 "CDEF filter process" == libaom cdef_filter_block_c / constrain()
 *   av1o_cdef_frame       spec §7.15 / §7.15.1: per 64x64, per 8x8 (all-skip blocks untouched), luma primary strength
 *                         adjusted by the directional variance, chroma reuses the luma direction, damping - 1 for chroma
 * Taps: Cdef_Pri_Taps {{4,2},{3,3}}, Cdef_Sec_Taps {2,1}; Cdef_Directions as in the spec.  Input is the deblocked
 * frame, output a separate frame (taps never see filtered samples).  Reference tree: no counterpart (transcode.go:120).
 */
#include "av1o_common.h"
#include <stdlib.h>
#include <string.h>

static const int cdef_dirs[8][2][2] = { /* [dir][k] = {dy, dx} */
  { { -1, 1 }, { -2, 2 } }, { { 0, 1 }, { -1, 2 } }, { { 0, 1 }, { 0, 2 } }, { { 0, 1 }, { 1, 2 } },
  { { 1, 1 }, { 2, 2 } },   { { 1, 0 }, { 2, 1 } },  { { 1, 0 }, { 2, 0 } }, { { 1, 0 }, { 2, -1 } } };
static const int cdef_pri_taps[2][2] = { { 4, 2 }, { 3, 3 } };
static const int cdef_sec_taps[2] = { 2, 1 };

static int gp(const void *p, int bd, size_t i) { return bd == 8 ? ((const uint8_t *)p)[i] : ((const uint16_t *)p)[i]; }
static void sp(void *p, int bd, size_t i, int v) { if (bd == 8) ((uint8_t *)p)[i] = (uint8_t)v; else ((uint16_t *)p)[i] = (uint16_t)v; }

/* img: top-left of an 8x8 luma block; returns direction, *var = directional contrast */
int av1o_cdef_find_dir(const void *img, int stride, int bd, int *var) {
  static const int div_table[9] = { 0, 840, 420, 280, 210, 168, 140, 120, 105 };
  int cost[8] = { 0 }, partial[8][15];
  memset(partial, 0, sizeof(partial));
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) {
      const int x = (gp(img, bd, (size_t)i * stride + j) >> (bd - 8)) - 128;
      partial[0][i + j] += x;
      partial[1][i + j / 2] += x;
      partial[2][i] += x;
      partial[3][3 + i - j / 2] += x;
      partial[4][7 + i - j] += x;
      partial[5][3 - i / 2 + j] += x;
      partial[6][j] += x;
      partial[7][i / 2 + j] += x;
    }
  for (int i = 0; i < 8; i++) { cost[2] += partial[2][i] * partial[2][i]; cost[6] += partial[6][i] * partial[6][i]; }
  cost[2] *= div_table[8]; cost[6] *= div_table[8];
  for (int i = 0; i < 7; i++) {
    cost[0] += (partial[0][i] * partial[0][i] + partial[0][14 - i] * partial[0][14 - i]) * div_table[i + 1];
    cost[4] += (partial[4][i] * partial[4][i] + partial[4][14 - i] * partial[4][14 - i]) * div_table[i + 1];
  }
  cost[0] += partial[0][7] * partial[0][7] * div_table[8];
  cost[4] += partial[4][7] * partial[4][7] * div_table[8];
  for (int i = 1; i < 8; i += 2) {
    for (int j = 0; j < 5; j++) cost[i] += partial[i][3 + j] * partial[i][3 + j];
    cost[i] *= div_table[8];
    for (int j = 0; j < 3; j++)
      cost[i] += (partial[i][j] * partial[i][j] + partial[i][10 - j] * partial[i][10 - j]) * div_table[2 * j + 2];
  }
  int best = 0, best_cost = 0;
  for (int i = 0; i < 8; i++) if (cost[i] > best_cost) { best_cost = cost[i]; best = i; }
  *var = (best_cost - cost[(best + 4) & 7]) >> 10;
  return best;
}

static int constrain(int diff, int threshold, int damping) {
  if (!threshold) return 0;
  const int shift = damping - av1o_msb((unsigned)threshold) > 0 ? damping - av1o_msb((unsigned)threshold) : 0;
  const int mag = abs(diff);
  int lim = threshold - (mag >> shift);
  if (lim < 0) lim = 0;
  const int v = mag < lim ? mag : lim;
  return diff < 0 ? -v : v;
}

/* one bw x bh block of a plane at (x0, y0); taps outside [0,pw) x [0,ph) are unavailable */
static void cdef_filter_block(const void *src, void *dst, int stride, int pw, int ph, int bd, int x0, int y0, int bw, int bh,
                              int pri, int sec, int damping, int dir) {
  const int cs = bd - 8;
  const int *pt = cdef_pri_taps[(pri >> cs) & 1];
  for (int i = 0; i < bh; i++)
    for (int j = 0; j < bw; j++) {
      const int x = gp(src, bd, (size_t)(y0 + i) * stride + x0 + j);
      int sum = 0, mx = x, mn = x;
      for (int k = 0; k < 2; k++)
        for (int sgn = -1; sgn <= 1; sgn += 2) {
          int yy = y0 + i + sgn * cdef_dirs[dir][k][0], xx = x0 + j + sgn * cdef_dirs[dir][k][1];
          if (yy >= 0 && yy < ph && xx >= 0 && xx < pw) {
            const int p = gp(src, bd, (size_t)yy * stride + xx);
            sum += pt[k] * constrain(p - x, pri, damping);
            if (p > mx) mx = p;
            if (p < mn) mn = p;
          }
          for (int off = -2; off <= 2; off += 4) {
            const int d2 = (dir + off) & 7;
            yy = y0 + i + sgn * cdef_dirs[d2][k][0]; xx = x0 + j + sgn * cdef_dirs[d2][k][1];
            if (yy >= 0 && yy < ph && xx >= 0 && xx < pw) {
              const int s = gp(src, bd, (size_t)yy * stride + xx);
              sum += cdef_sec_taps[k] * constrain(s - x, sec, damping);
              if (s > mx) mx = s;
              if (s < mn) mn = s;
            }
          }
        }
      sp(dst, bd, (size_t)(y0 + i) * stride + x0 + j, av1o_clampi(x + ((8 + sum - (sum < 0)) >> 4), mn, mx));
    }
}

/*
 * CDEF of one 4:2:0 frame.  w,h: luma size (multiples of 8).  sb_strength: one entry of 4 bytes per 64x64 luma block in
 * raster order: {y_pri (0..15), y_sec (0..3), uv_pri, uv_sec}; y_pri == 255 switches CDEF off for that block.
 * skip8: one byte per 8x8 luma block (raster, w/8 per row): 1 = every mode-info unit of the block is skipped.
 * damping: cdef_damping_minus_3 + 3 (3..6).  src_* deblocked input, dst_* output (distinct buffers).
 */
int av1o_cdef_frame(const void *src_y, const void *src_u, const void *src_v, void *dst_y, void *dst_u, void *dst_v, int w, int h,
                    int stride_y, int stride_uv, int bd, int damping, const uint8_t *sb_strength, const uint8_t *skip8) {
  if ((w & 7) || (h & 7) || (bd != 8 && bd != 10) || damping < 3 || damping > 6) return -1;
  const int cs = bd - 8, bps = bd == 8 ? 1 : 2;
  for (int r = 0; r < h; r++) memcpy((char *)dst_y + (size_t)r * stride_y * bps, (const char *)src_y + (size_t)r * stride_y * bps, (size_t)w * bps);
  for (int r = 0; r < h / 2; r++) {
    memcpy((char *)dst_u + (size_t)r * stride_uv * bps, (const char *)src_u + (size_t)r * stride_uv * bps, (size_t)(w / 2) * bps);
    memcpy((char *)dst_v + (size_t)r * stride_uv * bps, (const char *)src_v + (size_t)r * stride_uv * bps, (size_t)(w / 2) * bps);
  }
  const int sbw = (w + 63) / 64;
  for (int by = 0; by < h / 8; by++)
    for (int bx = 0; bx < w / 8; bx++) {
      const uint8_t *st = sb_strength + ((size_t)(by / 8) * sbw + bx / 8) * 4;
      if (st[0] == 255 || skip8[(size_t)by * (w / 8) + bx]) continue;
      int var = 0;
      const int ydir = av1o_cdef_find_dir((const char *)src_y + ((size_t)by * 8 * stride_y + bx * 8) * bps, stride_y, bd, &var);
      int pri = st[0] << cs;
      const int sec = (st[1] == 3 ? 4 : st[1]) << cs;
      const int dir = pri == 0 ? 0 : ydir;
      const int vs = (var >> 6) ? (av1o_msb((unsigned)(var >> 6)) < 12 ? av1o_msb((unsigned)(var >> 6)) : 12) : 0;
      pri = var ? (pri * (4 + vs) + 8) >> 4 : 0;
      cdef_filter_block(src_y, dst_y, stride_y, w, h, bd, bx * 8, by * 8, 8, 8, pri, sec, damping + cs, dir);
      const int upri = st[2] << cs, usec = (st[3] == 3 ? 4 : st[3]) << cs;
      const int udir = upri == 0 ? 0 : ydir;
      cdef_filter_block(src_u, dst_u, stride_uv, w / 2, h / 2, bd, bx * 4, by * 4, 4, 4, upri, usec, damping + cs - 1, udir);
      cdef_filter_block(src_v, dst_v, stride_uv, w / 2, h / 2, bd, bx * 4, by * 4, 4, 4, upri, usec, damping + cs - 1, udir);
    }
  return 0;
}
