/*
 * av1o_lr.c — CPU oracle for SURVEY.md §8 row K7: loop restoration (Wiener and self-guided) of one plane.
 *
 * TEST INFRASTRUCTURE ONLY; pinned to dav1d, not to the reference (see av1o_common.h).  Restates, from knowledge:
 *   src_sample            AV1 spec §7.17.6 "get source sample process": picture-edge clamp, then 64-row stripes offset
 *                         by 8 luma rows; up to 2 rows beyond the stripe come from the DEBLOCKED (pre-CDEF) frame,
 *                         further rows replicate them; inside the stripe the CDEF output is used
 *   wiener_sample         spec §7.17.4 "Wiener filter process" == libaom av1_highbd_wiener_convolve_add_src_c
 *                         (7-tap symmetric, taps {c0,c1,c2,128-2(c0+c1+c2),c2,c1,c0}, rounds 3 / 11, clip of the
 *                         horizontal stage to [-2^(bd+3), 2^(bd+5)-1-2^(bd+3)])
 *   sgr_ab / sgr_sample   spec §7.17.3 "self guided filter process" / box filter == libaom
 *                         av1_selfguided_restoration_c (r=2 "fast" pass on odd rows, r=1 pass, Sgr_Params[16][4])
 *   unit / stripe geometry spec §7.17 / §7.17.2: units offset by 8 luma rows, last unit absorbs the remainder
 * Evaluated per output sample straight from the definitions (slow, obviously correct).  Reference tree: nothing.
 */
#include "av1o_common.h"
#include <stdlib.h>
#include <string.h>

/* Sgr_Params (spec 7.17.3): { r0, eps0, r1, eps1 }.  (Round 1 held libaom's derived `s` values here and derived s from them a
 * second time; found by decoding with dav1d, tests/test_av1_conformance.py.) */
static const int sgr_params[16][4] = {
  { 2, 12, 1, 4 }, { 2, 15, 1, 6 }, { 2, 18, 1, 8 }, { 2, 21, 1, 9 }, { 2, 24, 1, 10 }, { 2, 29, 1, 11 },
  { 2, 36, 1, 12 }, { 2, 45, 1, 13 }, { 2, 56, 1, 14 }, { 2, 68, 1, 15 }, { 0, 0, 1, 5 }, { 0, 0, 1, 8 },
  { 0, 0, 1, 11 }, { 0, 0, 1, 14 }, { 2, 30, 0, 0 }, { 2, 75, 0, 0 } };

typedef struct {
  const void *cdef, *dbl;   /* CDEF output, deblocked (pre-CDEF) frame */
  int stride, w, h, bd, ss; /* plane size; ss = vertical subsampling (0 luma, 1 chroma 4:2:0) */
  int stripe_start, stripe_end;
} lr_ctx;

static int gp(const void *p, int bd, size_t i) { return bd == 8 ? ((const uint8_t *)p)[i] : ((const uint16_t *)p)[i]; }

static int src_sample(const lr_ctx *c, int x, int y) {
  x = av1o_clampi(x, 0, c->w - 1);
  y = av1o_clampi(y, 0, c->h - 1);
  if (y < c->stripe_start) { y = y > c->stripe_start - 2 ? y : c->stripe_start - 2; return gp(c->dbl, c->bd, (size_t)y * c->stride + x); }
  if (y > c->stripe_end) { y = y < c->stripe_end + 2 ? y : c->stripe_end + 2; return gp(c->dbl, c->bd, (size_t)y * c->stride + x); }
  return gp(c->cdef, c->bd, (size_t)y * c->stride + x);
}

static int wiener_sample(const lr_ctx *c, int x, int y, const int8_t *coef /* v0 v1 v2 h0 h1 h2 */) {
  int vf[7], hf[7];
  for (int i = 0; i < 3; i++) { vf[i] = vf[6 - i] = coef[i]; hf[i] = hf[6 - i] = coef[3 + i]; }
  vf[3] = 128 - 2 * (coef[0] + coef[1] + coef[2]);
  hf[3] = 128 - 2 * (coef[3] + coef[4] + coef[5]);
  const int round0 = 3, round1 = 11, bd = c->bd;
  const int offset = 1 << (bd + 7 - round0 - 1), limit = (1 << (bd + 1 + 7 - round0)) - 1;
  int s2 = 0;
  for (int r = 0; r < 7; r++) {
    int s = 0;
    for (int t = 0; t < 7; t++) s += hf[t] * src_sample(c, x + t - 3, y + r - 3);
    s2 += vf[r] * av1o_clampi(av1o_round2(s, round0), -offset, limit - offset);
  }
  return av1o_clampi(av1o_round2(s2, round1), 0, (1 << bd) - 1);
}

/* A and B of the box filter at (x, y) for radius r, strength eps */
static void sgr_ab(const lr_ctx *c, int x, int y, int r, int eps, int *A, int *B) {
  const int bd = c->bd, n = (2 * r + 1) * (2 * r + 1);
  const uint32_t n2e = (uint32_t)n * n * eps;
  const uint32_t s = ((1u << 20) + n2e / 2) / n2e;
  const uint32_t one_by_n = ((1u << 12) + n / 2) / n;
  uint32_t a = 0, b = 0;
  for (int dy = -r; dy <= r; dy++)
    for (int dx = -r; dx <= r; dx++) { const uint32_t v = (uint32_t)src_sample(c, x + dx, y + dy); a += v * v; b += v; }
  const uint32_t as = (uint32_t)av1o_round2_64(a, 2 * (bd - 8)), d = (uint32_t)av1o_round2_64(b, bd - 8);
  const uint32_t p = as * n > d * d ? as * n - d * d : 0;
  const uint32_t z = (uint32_t)(((uint64_t)p * s + (1u << 19)) >> 20);
  const uint32_t a2 = z >= 255 ? 256 : z == 0 ? 1 : ((z << 8) + z / 2) / (z + 1);
  const uint32_t b2 = (256 - a2) * b * one_by_n;
  *A = (int)a2;
  *B = (int)((b2 + (1u << 11)) >> 12);
}
static int sgr_flt(const lr_ctx *c, int x, int y, int pass, int r, int eps) {
  int a = 0, b = 0;
  const int shift = (pass == 0 && (y & 1)) ? 4 : 5;
  for (int dy = -1; dy <= 1; dy++)
    for (int dx = -1; dx <= 1; dx++) {
      int wgt;
      if (pass == 0) wgt = ((y + dy) & 1) ? (dx == 0 ? 6 : 5) : 0;
      else wgt = (dx == 0 || dy == 0) ? 4 : 3;
      if (!wgt) continue;
      int A, B;
      sgr_ab(c, x + dx, y + dy, r, eps, &A, &B);
      a += wgt * A; b += wgt * B;
    }
  const int v = a * gp(c->cdef, c->bd, (size_t)y * c->stride + x) + b;
  return av1o_round2(v, 8 + shift - 4);
}
static int sgr_sample(const lr_ctx *c, int x, int y, int set, int xq0, int xq1) {
  const int r0 = sgr_params[set][0], r1 = sgr_params[set][2];
  const int w0 = xq0, w1 = xq1, w2 = 128 - w0 - w1;
  const int u = gp(c->cdef, c->bd, (size_t)y * c->stride + x) << 4;
  int v = w1 * u;
  v += w0 * (r0 ? sgr_flt(c, x, y, 0, r0, sgr_params[set][1]) : u);
  v += w2 * (r1 ? sgr_flt(c, x, y, 1, r1, sgr_params[set][3]) : u);
  return av1o_clampi(av1o_round2(v, 11), 0, (1 << c->bd) - 1);
}

/* spec count_units_in_frame() */
int av1o_lr_units(int unit_size, int plane_size) {
  const int n = (plane_size + (unit_size >> 1)) / unit_size;
  return n > 1 ? n : 1;
}

/*
 * Restore one plane.  units: rows x cols entries of 8 bytes {type (0 none, 1 wiener, 2 sgrproj), v0, v1, v2, h0, h1, h2 (int8
 * Wiener taps) | for sgrproj: set (byte 1), xqd0, xqd1 (int8, bytes 2 and 3)}, rows = av1o_lr_units(unit_size, h), cols likewise.
 * ss: 1 for the chroma planes of 4:2:0 (stripes of 32 rows offset by 4), 0 for luma.
 */
int av1o_lr_plane(const void *cdef, const void *dbl, void *out, int stride, int w, int h, int bd, int ss, int unit_size,
                  const int8_t *units) {
  if (bd != 8 && bd != 10) return -1;
  const int urows = av1o_lr_units(unit_size, h), ucols = av1o_lr_units(unit_size, w);
  lr_ctx c = { cdef, dbl, stride, w, h, bd, ss, 0, 0 };
  for (int y = 0; y < h; y++) {
    const int stripe = ((y << ss) + 8) / 64;
    c.stripe_start = (-8 + stripe * 64) >> ss;
    c.stripe_end = c.stripe_start + (64 >> ss) - 1;
    int ur = (y + (8 >> ss)) / unit_size;
    if (ur > urows - 1) ur = urows - 1;
    for (int x = 0; x < w; x++) {
      int uc = x / unit_size;
      if (uc > ucols - 1) uc = ucols - 1;
      const int8_t *u = units + ((size_t)ur * ucols + uc) * 8;
      int v;
      if (u[0] == 1) v = wiener_sample(&c, x, y, u + 1);
      else if (u[0] == 2) v = sgr_sample(&c, x, y, u[1], u[2], u[3]);
      else v = gp(cdef, bd, (size_t)y * stride + x);
      if (bd == 8) ((uint8_t *)out)[(size_t)y * stride + x] = (uint8_t)v; else ((uint16_t *)out)[(size_t)y * stride + x] = (uint16_t)v;
    }
  }
  return 0;
}

/*
 * Encoder policy (non-normative): restoration stays ON for a plane of a frame exactly when it lowers the squared error against
 * the source: returns 1 when sum (lr - src)^2 < sum (cdef - src)^2 over the sampled tiles (the plane the next frame predicts from is then `lr`, else
 * `cdef`, and the frame header signals lr_type NONE for the plane).  The kernels accumulate the same two integer sums (k_lr) and
 * compare them (k_lr_decide); fixed default taps lower PSNR on most inter frames of the synthetic clips and raise it on key frames.
 */
int av1o_lr_keep(const void *src, const void *cdef, const void *lr, int stride, int w, int h, int bd, int ss) {
  /* the sums run over an eighth of the plane: the tiles (64 columns x one restoration stripe) with (column + stripe) % 8 == 0,
   * every tile when the plane has fewer than 32 (k_lr reads the source for those tiles only) */
  const int sh = 64 >> ss, off = 8 >> ss;
  const int ntx = (w + 63) >> 6, nst = (h + off + sh - 1) / sh, dense = ntx * nst < 32;
  unsigned long long e_lr = 0, e_cdef = 0;
  for (int y = 0; y < h; y++) {
    const int stripe = (y + off) / sh;
    for (int x = 0; x < w; x++) {
      if (!dense && (((x >> 6) + stripe) & 7)) continue;
      const long s = gp(src, bd, (size_t)y * stride + x);
      const long a = gp(lr, bd, (size_t)y * stride + x) - s, b = gp(cdef, bd, (size_t)y * stride + x) - s;
      e_lr += (unsigned long long)(a * a); e_cdef += (unsigned long long)(b * b);
    }
  }
  return e_lr < e_cdef;
}
