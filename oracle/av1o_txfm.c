/*
 * av1o_txfm.c — CPU oracle for SURVEY.md §8 rows K1 (forward 2-D transform),
 * K2 (inverse 2-D transform + reconstruct) and K8 (quantise / dequantise).
 *
 * TEST INFRASTRUCTURE ONLY (see av1o_common.h).  The inverse transforms are pinned to dav1d (all sizes and types); restated from
 * the AV1 specification and libaom from knowledge; the reference tree holds no
 * arithmetic for this path (reference internal/ffmpeg/transcode.go:120 only
 * names the external encoder).
 *
 * What each function restates:
 *   av1o_idct / idct_core / idct_odd   AV1 spec §7.13.2.3 "inverse DCT process"
 *                                      == libaom av1_idct4/8/16/32/64 (av1_inv_txfm1d.c),
 *                                      incl. libaom's clamp_value() on add/sub stages
 *   av1o_iadst4/8/16                   spec §7.13.2.6-7.13.2.8 == libaom av1_iadst4/8/16
 *   av1o_iidentity                     spec §7.13.2.15 == libaom av1_iidentity4/8/16/32_c
 *   av1o_inv_txfm2d_add                spec §7.13.3 "2D inverse transform process" +
 *                                      §7.12.3 reconstruction == libaom inv_txfm2d_add_c
 *   av1o_fwd_txfm2d                    libaom fwd_txfm2d_c (encoder side, NON-normative)
 *   av1o_quantize / av1o_dequantize    libaom av1_quantize_fp (non-normative) /
 *                                      spec §7.12.3 dequantisation (normative)
 */
#include "av1o_common.h"
#include <stdlib.h>
#include <string.h>

const int av1o_tx_w[TX_SIZES_ALL] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
const int av1o_tx_h[TX_SIZES_ALL] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };

/* round(cos(i*pi/128) * 2^bit), i = 0..64; libaom av1_cospi_arr_data. */
static const int32_t cospi_tab[4][65] = {
  /* bit 10 */
  { 1024, 1024, 1023, 1021, 1019, 1016, 1013, 1009, 1004, 999, 993, 987, 980, 972, 964, 955,
    946, 936, 926, 915, 903, 891, 878, 865, 851, 837, 822, 807, 792, 775, 759, 742,
    724, 706, 688, 669, 650, 630, 610, 590, 569, 548, 526, 505, 483, 460, 438, 415,
    392, 369, 345, 321, 297, 273, 249, 224, 200, 175, 150, 125, 100, 75, 50, 25,
    0 },
  /* bit 11 */
  { 2048, 2047, 2046, 2042, 2038, 2033, 2026, 2018, 2009, 1998, 1987, 1974, 1960, 1945, 1928, 1911,
    1892, 1872, 1851, 1829, 1806, 1782, 1757, 1730, 1703, 1674, 1645, 1615, 1583, 1551, 1517, 1483,
    1448, 1412, 1375, 1338, 1299, 1260, 1220, 1179, 1138, 1096, 1053, 1009, 965, 921, 876, 830,
    784, 737, 690, 642, 595, 546, 498, 449, 400, 350, 301, 251, 201, 151, 100, 50,
    0 },
  /* bit 12 */
  { 4096, 4095, 4091, 4085, 4076, 4065, 4052, 4036, 4017, 3996, 3973, 3948, 3920, 3889, 3857, 3822,
    3784, 3745, 3703, 3659, 3612, 3564, 3513, 3461, 3406, 3349, 3290, 3229, 3166, 3102, 3035, 2967,
    2896, 2824, 2751, 2675, 2598, 2520, 2440, 2359, 2276, 2191, 2106, 2019, 1931, 1842, 1751, 1660,
    1567, 1474, 1380, 1285, 1189, 1092, 995, 897, 799, 700, 601, 501, 401, 301, 201, 101,
    0 },
  /* bit 13 */
  { 8192, 8190, 8182, 8170, 8153, 8130, 8103, 8071, 8035, 7993, 7946, 7895, 7839, 7779, 7713, 7643,
    7568, 7489, 7405, 7317, 7225, 7128, 7027, 6921, 6811, 6698, 6580, 6458, 6333, 6203, 6070, 5933,
    5793, 5649, 5501, 5351, 5197, 5040, 4880, 4717, 4551, 4383, 4212, 4038, 3862, 3683, 3503, 3320,
    3135, 2948, 2760, 2570, 2378, 2185, 1990, 1795, 1598, 1401, 1202, 1003, 803, 603, 402, 201,
    0 },
};
/* round(sin(k*pi/9) * sqrt(2) * 2/3 * 2^bit); libaom av1_sinpi_arr_data. */
static const int32_t sinpi_tab[4][5] = {
  { 0, 330, 621, 836, 951 }, { 0, 660, 1241, 1672, 1902 },
  { 0, 1321, 2482, 3344, 3803 }, { 0, 2642, 4965, 6689, 7606 },
};
#define NEW_SQRT2 5793      /* libaom NewSqrt2, 2^12 * sqrt(2) */
#define NEW_INV_SQRT2 2896  /* libaom NewInvSqrt2 */

static const int32_t *cospi_arr(int bit) { return cospi_tab[bit - 10]; }
static const int32_t *sinpi_arr(int bit) { return sinpi_tab[bit - 10]; }

/* libaom half_btf(): round_shift(w0*in0 + w1*in1, bit). */
static int32_t half_btf(int32_t w0, int32_t in0, int32_t w1, int32_t in1, int bit) {
  int64_t r = (int64_t)w0 * in0 + (int64_t)w1 * in1;
  return (int32_t)((r + ((int64_t)1 << (bit - 1))) >> bit);
}
/* libaom clamp_value(): clamp to a signed `bit`-bit range; bit <= 0 means no clamp. */
static int32_t clamp_value(int64_t v, int bit) {
  if (bit <= 0) return (int32_t)v;
  const int64_t hi = ((int64_t)1 << (bit - 1)) - 1, lo = -((int64_t)1 << (bit - 1));
  return (int32_t)(v < lo ? lo : (v > hi ? hi : v));
}
static int brev(int nbits, int x) {
  int r = 0;
  for (int i = 0; i < nbits; i++) r |= ((x >> i) & 1) << (nbits - 1 - i);
  return r;
}
static int ilog2(int n) { int r = 0; while ((1 << r) < n) r++; return r; }

/* spec §7.13.2.1: cos128()/sin128() over the 65-entry quarter table. */
static int32_t cos128(const int32_t *cospi, int angle) {
  int a = angle & 255;
  if (a <= 64) return cospi[a];
  if (a <= 128) return -cospi[128 - a];
  if (a <= 192) return -cospi[a - 128];
  return cospi[256 - a];
}
static int32_t sin128(const int32_t *cospi, int angle) { return cos128(cospi, angle - 64); }

/* spec §7.13.2.2 butterfly rotation B(a,b,angle,flip).  inverse direction. */
static void rot_inv(int32_t *T, int a, int b, int angle, int flip, const int32_t *cospi, int bit) {
  const int32_t c = cos128(cospi, angle), s = sin128(cospi, angle);
  const int32_t x = half_btf(c, T[a], -s, T[b], bit);
  const int32_t y = half_btf(s, T[a], c, T[b], bit);
  if (flip) { T[a] = y; T[b] = x; } else { T[a] = x; T[b] = y; }
}
/* transpose of rot_inv's 2x2 matrix (forward direction). */
static void rot_fwd(int32_t *T, int a, int b, int angle, int flip, const int32_t *cospi, int bit) {
  const int32_t c = cos128(cospi, angle), s = sin128(cospi, angle);
  if (flip) { /* [[s,c],[c,-s]] is symmetric */
    const int32_t x = half_btf(s, T[a], c, T[b], bit);
    const int32_t y = half_btf(c, T[a], -s, T[b], bit);
    T[a] = x; T[b] = y;
  } else {
    const int32_t x = half_btf(c, T[a], s, T[b], bit);
    const int32_t y = half_btf(-s, T[a], c, T[b], bit);
    T[a] = x; T[b] = y;
  }
}
/* spec §7.13.2.2 Hadamard rotation H(a,b,flip) with libaom's stage clamp. */
static void had(int32_t *T, int a, int b, int flip, int range) {
  const int64_t x = T[a], y = T[b];
  if (flip) { T[a] = clamp_value(-x + y, range); T[b] = clamp_value(x + y, range); }
  else      { T[a] = clamp_value(x + y, range);  T[b] = clamp_value(x - y, range); }
}
/* first-rotation angle of the odd block [M,2M): frequency k = 1 + 2*brev(log2 M, i). */
static int r0_angle(int M, int i) {
  const int k = 1 + 2 * brev(ilog2(M), i);
  return 64 - k * 32 / M;
}

/* odd half [M,2M) of an inverse DCT of size 2M (spec §7.13.2.3 steps 2-30 grouped by block). */
static void idct_odd(int32_t *T, int M, const int32_t *cospi, int bit, int range) {
  for (int i = 0; i < M / 2; i++) rot_inv(T, M + i, 2 * M - 1 - i, r0_angle(M, i), 0, cospi, bit);
  for (int s = 2; s <= M / 2; s *= 2) {
    for (int g = 0; g < M / s; g++)
      for (int j = 0; j < s / 2; j++) had(T, M + g * s + j, M + g * s + s - 1 - j, g & 1, range);
    if (4 * s <= M) {
      for (int g = 0; g < M / (4 * s); g++) {
        const int th = r0_angle(M / (2 * s), g);
        for (int q = 0; q < s / 2; q++) {
          const int p = g * 2 * s + s / 2 + q, p2 = g * 2 * s + s + q;
          rot_inv(T, 2 * M - 1 - p, M + p, th, 1, cospi, bit);
          rot_inv(T, 2 * M - 1 - p2, M + p2, th + 64, 1, cospi, bit);
        }
      }
    } else {
      for (int q = 0; q < s / 2; q++) {
        const int p = s / 2 + q;
        rot_inv(T, 2 * M - 1 - p, M + p, 32, 1, cospi, bit);
      }
    }
  }
}
static void idct_core(int32_t *T, int N, const int32_t *cospi, int bit, int range) {
  if (N == 2) {
    const int32_t a = half_btf(cospi[32], T[0], cospi[32], T[1], bit);
    const int32_t b = half_btf(cospi[32], T[0], -cospi[32], T[1], bit);
    T[0] = a; T[1] = b;
    return;
  }
  idct_core(T, N / 2, cospi, bit, range);
  idct_odd(T, N / 2, cospi, bit, range);
  for (int i = 0; i < N / 2; i++) had(T, i, N - 1 - i, 0, range);
}
/* 1-D inverse DCT, N in {4,8,16,32,64}. range = libaom stage_range (0 = spec behaviour, no clamp). */
void av1o_idct(const int32_t *in, int32_t *out, int N, int bit, int range) {
  int32_t T[64];
  const int n = ilog2(N);
  for (int i = 0; i < N; i++) T[i] = in[brev(n, i)];
  idct_core(T, N, cospi_arr(bit), bit, range);
  memcpy(out, T, sizeof(int32_t) * N);
}

/* forward: transposed flow graph, reversed (libaom av1_fdct4..64; non-normative). */
static void had_fwd(int32_t *T, int a, int b, int flip) {
  const int32_t x = T[a], y = T[b];
  if (flip) { T[a] = -x + y; T[b] = x + y; } else { T[a] = x + y; T[b] = x - y; }
}
static void fdct_odd(int32_t *T, int M, const int32_t *cospi, int bit) {
  for (int s = M / 2; s >= 2; s /= 2) {
    if (4 * s <= M) {
      for (int g = 0; g < M / (4 * s); g++) {
        const int th = r0_angle(M / (2 * s), g);
        for (int q = 0; q < s / 2; q++) {
          const int p = g * 2 * s + s / 2 + q, p2 = g * 2 * s + s + q;
          rot_fwd(T, 2 * M - 1 - p, M + p, th, 1, cospi, bit);
          rot_fwd(T, 2 * M - 1 - p2, M + p2, th + 64, 1, cospi, bit);
        }
      }
    } else {
      for (int q = 0; q < s / 2; q++) {
        const int p = s / 2 + q;
        rot_fwd(T, 2 * M - 1 - p, M + p, 32, 1, cospi, bit);
      }
    }
    for (int g = 0; g < M / s; g++)
      for (int j = 0; j < s / 2; j++) had_fwd(T, M + g * s + j, M + g * s + s - 1 - j, g & 1);
  }
  for (int i = 0; i < M / 2; i++) rot_fwd(T, M + i, 2 * M - 1 - i, r0_angle(M, i), 0, cospi, bit);
}
static void fdct_core(int32_t *T, int N, const int32_t *cospi, int bit) {
  if (N == 2) {
    const int32_t a = half_btf(cospi[32], T[0], cospi[32], T[1], bit);
    const int32_t b = half_btf(-cospi[32], T[1], cospi[32], T[0], bit);
    T[0] = a; T[1] = b;
    return;
  }
  for (int i = 0; i < N / 2; i++) had_fwd(T, i, N - 1 - i, 0);
  fdct_odd(T, N / 2, cospi, bit);
  fdct_core(T, N / 2, cospi, bit);
}
void av1o_fdct(const int32_t *in, int32_t *out, int N, int bit) {
  int32_t T[64];
  const int n = ilog2(N);
  memcpy(T, in, sizeof(int32_t) * N);
  fdct_core(T, N, cospi_arr(bit), bit);
  for (int i = 0; i < N; i++) out[brev(n, i)] = T[i];
}

/* spec §7.13.2.6 inverse ADST4 == libaom av1_iadst4 (no stage clamps). */
void av1o_iadst4(const int32_t *in, int32_t *out, int bit) {
  const int32_t *sinpi = sinpi_arr(bit);
  int32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
  if (!(x0 | x1 | x2 | x3)) { out[0] = out[1] = out[2] = out[3] = 0; return; }
  int32_t s0 = sinpi[1] * x0, s1 = sinpi[2] * x0, s2 = sinpi[3] * x1, s3 = sinpi[4] * x2;
  int32_t s4 = sinpi[1] * x2, s5 = sinpi[2] * x3, s6 = sinpi[4] * x3, s7 = (x0 - x2) + x3;
  s0 = s0 + s3; s1 = s1 - s4; s3 = s2; s2 = sinpi[3] * s7;
  s0 = s0 + s5; s1 = s1 - s6;
  x0 = s0 + s3; x1 = s1 + s3; x2 = s2; x3 = s0 + s1; x3 = x3 - s3;
  out[0] = av1o_round2(x0, bit); out[1] = av1o_round2(x1, bit);
  out[2] = av1o_round2(x2, bit); out[3] = av1o_round2(x3, bit);
}
/* libaom av1_fadst4 (non-normative). */
void av1o_fadst4(const int32_t *in, int32_t *out, int bit) {
  const int32_t *sinpi = sinpi_arr(bit);
  int32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
  if (!(x0 | x1 | x2 | x3)) { out[0] = out[1] = out[2] = out[3] = 0; return; }
  int32_t s0 = sinpi[1] * x0, s1 = sinpi[4] * x0, s2 = sinpi[2] * x1, s3 = sinpi[1] * x1;
  int32_t s4 = sinpi[3] * x2, s5 = sinpi[4] * x3, s6 = sinpi[2] * x3, s7 = x0 + x1 - x3;
  x0 = s0 + s2 + s5; x1 = sinpi[3] * s7; x2 = s1 - s3 + s6; x3 = s4;
  s0 = x0 + x3; s1 = x1; s2 = x2 - x3; s3 = x2 - x0 + x3;
  out[0] = av1o_round2(s0, bit); out[1] = av1o_round2(s1, bit);
  out[2] = av1o_round2(s2, bit); out[3] = av1o_round2(s3, bit);
}

/* shared pieces of ADST8/16 (spec §7.13.2.7/8 == libaom av1_iadst8/av1_iadst16). */
static void adst_addsub(int32_t *b, int N, int h, int range) { /* groups of 2h: (i, i+h) */
  for (int g = 0; g < N; g += 2 * h)
    for (int i = 0; i < h; i++) {
      const int64_t x = b[g + i], y = b[g + i + h];
      b[g + i] = clamp_value(x + y, range);
      b[g + i + h] = clamp_value(x - y, range);
    }
}
/* pair (p,p+1): [c0 c1; c1 -c0] then (p+2,p+3): [-c1 c0; c0 c1]  (both symmetric) */
static void adst_rot_pp(int32_t *b, int p, int32_t c0, int32_t c1, int bit) {
  const int32_t x = half_btf(c0, b[p], c1, b[p + 1], bit), y = half_btf(c1, b[p], -c0, b[p + 1], bit);
  b[p] = x; b[p + 1] = y;
}
static void adst_rot_np(int32_t *b, int p, int32_t c0, int32_t c1, int bit) {
  const int32_t x = half_btf(-c1, b[p], c0, b[p + 1], bit), y = half_btf(c0, b[p], c1, b[p + 1], bit);
  b[p] = x; b[p + 1] = y;
}
void av1o_iadst8(const int32_t *in, int32_t *out, int bit, int range) {
  const int32_t *c = cospi_arr(bit);
  int32_t b[8];
  for (int i = 0; i < 4; i++) { b[2 * i] = in[7 - 2 * i]; b[2 * i + 1] = in[2 * i]; }
  for (int i = 0; i < 4; i++) adst_rot_pp(b, 2 * i, c[4 + 16 * i], c[60 - 16 * i], bit);   /* stage 2 */
  adst_addsub(b, 8, 4, range);                                                           /* stage 3 */
  adst_rot_pp(b, 4, c[16], c[48], bit); adst_rot_np(b, 6, c[16], c[48], bit);            /* stage 4 */
  adst_addsub(b, 8, 2, range);                                                           /* stage 5 */
  adst_rot_pp(b, 2, c[32], c[32], bit); adst_rot_pp(b, 6, c[32], c[32], bit);            /* stage 6 */
  out[0] = b[0]; out[1] = -b[4]; out[2] = b[6]; out[3] = -b[2];                          /* stage 7 */
  out[4] = b[3]; out[5] = -b[7]; out[6] = b[5]; out[7] = -b[1];
}
void av1o_iadst16(const int32_t *in, int32_t *out, int bit, int range) {
  const int32_t *c = cospi_arr(bit);
  int32_t b[16];
  for (int i = 0; i < 8; i++) { b[2 * i] = in[15 - 2 * i]; b[2 * i + 1] = in[2 * i]; }
  for (int i = 0; i < 8; i++) adst_rot_pp(b, 2 * i, c[2 + 8 * i], c[62 - 8 * i], bit);    /* stage 2 */
  adst_addsub(b, 16, 8, range);                                                          /* stage 3 */
  adst_rot_pp(b, 8, c[8], c[56], bit);  adst_rot_pp(b, 10, c[40], c[24], bit);           /* stage 4 */
  adst_rot_np(b, 12, c[8], c[56], bit); adst_rot_np(b, 14, c[40], c[24], bit);
  adst_addsub(b, 16, 4, range);                                                          /* stage 5 */
  adst_rot_pp(b, 4, c[16], c[48], bit);  adst_rot_np(b, 6, c[16], c[48], bit);           /* stage 6 */
  adst_rot_pp(b, 12, c[16], c[48], bit); adst_rot_np(b, 14, c[16], c[48], bit);
  adst_addsub(b, 16, 2, range);                                                          /* stage 7 */
  for (int p = 2; p < 16; p += 4) adst_rot_pp(b, p, c[32], c[32], bit);                  /* stage 8 */
  out[0] = b[0];  out[1] = -b[8];   out[2] = b[12];  out[3] = -b[4];                     /* stage 9 */
  out[4] = b[6];  out[5] = -b[14];  out[6] = b[10];  out[7] = -b[2];
  out[8] = b[3];  out[9] = -b[11];  out[10] = b[15]; out[11] = -b[7];
  out[12] = b[5]; out[13] = -b[13]; out[14] = b[9];  out[15] = -b[1];
}
/* forward ADST8/16: the same symmetric stages in reverse (libaom av1_fadst8/16; non-normative). */
void av1o_fadst8(const int32_t *in, int32_t *out, int bit) {
  const int32_t *c = cospi_arr(bit);
  int32_t b[8];
  b[0] = in[0]; b[4] = -in[1]; b[6] = in[2]; b[2] = -in[3];
  b[3] = in[4]; b[7] = -in[5]; b[5] = in[6]; b[1] = -in[7];
  adst_rot_pp(b, 2, c[32], c[32], bit); adst_rot_pp(b, 6, c[32], c[32], bit);
  adst_addsub(b, 8, 2, 0);
  adst_rot_pp(b, 4, c[16], c[48], bit); adst_rot_np(b, 6, c[16], c[48], bit);
  adst_addsub(b, 8, 4, 0);
  for (int i = 0; i < 4; i++) adst_rot_pp(b, 2 * i, c[4 + 16 * i], c[60 - 16 * i], bit);
  for (int i = 0; i < 4; i++) { out[7 - 2 * i] = b[2 * i]; out[2 * i] = b[2 * i + 1]; }
}
void av1o_fadst16(const int32_t *in, int32_t *out, int bit) {
  const int32_t *c = cospi_arr(bit);
  int32_t b[16];
  b[0] = in[0];  b[8] = -in[1];   b[12] = in[2];  b[4] = -in[3];
  b[6] = in[4];  b[14] = -in[5];  b[10] = in[6];  b[2] = -in[7];
  b[3] = in[8];  b[11] = -in[9];  b[15] = in[10]; b[7] = -in[11];
  b[5] = in[12]; b[13] = -in[13]; b[9] = in[14];  b[1] = -in[15];
  for (int p = 2; p < 16; p += 4) adst_rot_pp(b, p, c[32], c[32], bit);
  adst_addsub(b, 16, 2, 0);
  adst_rot_pp(b, 4, c[16], c[48], bit);  adst_rot_np(b, 6, c[16], c[48], bit);
  adst_rot_pp(b, 12, c[16], c[48], bit); adst_rot_np(b, 14, c[16], c[48], bit);
  adst_addsub(b, 16, 4, 0);
  adst_rot_pp(b, 8, c[8], c[56], bit);  adst_rot_pp(b, 10, c[40], c[24], bit);
  adst_rot_np(b, 12, c[8], c[56], bit); adst_rot_np(b, 14, c[40], c[24], bit);
  adst_addsub(b, 16, 8, 0);
  for (int i = 0; i < 8; i++) adst_rot_pp(b, 2 * i, c[2 + 8 * i], c[62 - 8 * i], bit);
  for (int i = 0; i < 8; i++) { out[15 - 2 * i] = b[2 * i]; out[2 * i] = b[2 * i + 1]; }
}

/* spec §7.13.2.15 identity transforms == libaom av1_iidentity{4,8,16,32}_c (same maps forward). */
void av1o_identity(const int32_t *in, int32_t *out, int N) {
  for (int i = 0; i < N; i++) {
    switch (N) {
      case 4:  out[i] = (int32_t)av1o_round2_64((int64_t)in[i] * NEW_SQRT2, 12); break;
      case 8:  out[i] = in[i] * 2; break;
      case 16: out[i] = (int32_t)av1o_round2_64((int64_t)in[i] * 2 * NEW_SQRT2, 12); break;
      default: out[i] = in[i] * 4; break;
    }
  }
}

/* libaom vtx_tab / htx_tab: TX_TYPE -> 1-D column (vertical) / row (horizontal) kernels. */
static int col_1d(int tx_type) {
  static const int t[TX_TYPES] = { T1D_DCT, T1D_ADST, T1D_DCT, T1D_ADST, T1D_FLIPADST, T1D_DCT, T1D_FLIPADST,
    T1D_ADST, T1D_FLIPADST, T1D_IDTX, T1D_DCT, T1D_IDTX, T1D_ADST, T1D_IDTX, T1D_FLIPADST, T1D_IDTX };
  return t[tx_type];
}
static int row_1d(int tx_type) {
  static const int t[TX_TYPES] = { T1D_DCT, T1D_DCT, T1D_ADST, T1D_ADST, T1D_DCT, T1D_FLIPADST, T1D_FLIPADST,
    T1D_FLIPADST, T1D_ADST, T1D_IDTX, T1D_IDTX, T1D_DCT, T1D_IDTX, T1D_ADST, T1D_IDTX, T1D_FLIPADST };
  return t[tx_type];
}
/* 1: the (size,type) pair is arithmetically defined: ADST needs length 4/8/16, IDTX length <= 32. */
int av1o_txfm_valid(int tx_size, int tx_type) {
  if (tx_type == AV1O_WHT_WHT) return tx_size == TX_4X4;   /* lossless blocks: Walsh-Hadamard, 4x4 only (spec 7.13.3) */
  if (tx_size < 0 || tx_size >= TX_SIZES_ALL || tx_type < 0 || tx_type >= TX_TYPES) return 0;
  const int w = av1o_tx_w[tx_size], h = av1o_tx_h[tx_size];
  const int r = row_1d(tx_type), c = col_1d(tx_type);
  if ((r == T1D_ADST || r == T1D_FLIPADST) && w > 16) return 0;
  if ((c == T1D_ADST || c == T1D_FLIPADST) && h > 16) return 0;
  if (r == T1D_IDTX && w > 32) return 0;
  if (c == T1D_IDTX && h > 32) return 0;
  return 1;
}
static void inv_1d(int kind, const int32_t *in, int32_t *out, int N, int range) {
  if (kind == T1D_DCT) av1o_idct(in, out, N, 12, range);
  else if (kind == T1D_IDTX) av1o_identity(in, out, N);
  else if (N == 4) av1o_iadst4(in, out, 12);
  else if (N == 8) av1o_iadst8(in, out, 12, range);
  else av1o_iadst16(in, out, 12, range);
}
static void fwd_1d(int kind, const int32_t *in, int32_t *out, int N, int bit) {
  if (kind == T1D_DCT) av1o_fdct(in, out, N, bit);
  else if (kind == T1D_IDTX) av1o_identity(in, out, N);
  else if (N == 4) av1o_fadst4(in, out, bit);
  else if (N == 8) av1o_fadst8(in, out, bit);
  else av1o_fadst16(in, out, bit);
}

/*
 * Lossless 4x4: spec 7.13.2.10 (inverse Walsh-Hadamard transform process) applied as 7.13.3 prescribes for Lossless —
 * rows with shift 2, columns with shift 0, no intermediate rounding, the result is the residual itself (no final shift)
 * == libaom av1_iwht4x4_16_add_c / av1_highbd_iwht4x4_16_add_c.  Forward: libaom av1_fwht4x4_c (UNIT_QUANT_FACTOR 4).
 */
static void iwht4(int32_t *a, int32_t *b, int32_t *c, int32_t *d, int shift) {   /* in: T[0], T[3], T[1], T[2] order of the spec */
  int32_t e;
  *a >>= shift; *c >>= shift; *d >>= shift; *b >>= shift;
  *a += *c; *d -= *b; e = (*a - *d) >> 1; *b = e - *b; *c = e - *c; *a -= *b; *d += *c;
}
static int inv_wht4x4_add(const int32_t *coef, void *dst, int stride, int bd) {
  int32_t t[16];
  const int maxpix = (1 << bd) - 1;
  for (int r = 0; r < 4; r++) {
    int32_t a = coef[r * 4], c = coef[r * 4 + 1], d = coef[r * 4 + 2], b = coef[r * 4 + 3];
    iwht4(&a, &b, &c, &d, 2);
    t[r * 4] = a; t[r * 4 + 1] = b; t[r * 4 + 2] = c; t[r * 4 + 3] = d;
  }
  for (int col = 0; col < 4; col++) {
    int32_t a = t[col], c = t[4 + col], d = t[8 + col], b = t[12 + col];
    iwht4(&a, &b, &c, &d, 0);
    const int32_t res[4] = { a, b, c, d };
    for (int r = 0; r < 4; r++) {
      if (bd == 8) { uint8_t *p = (uint8_t *)dst + (size_t)r * stride + col; *p = (uint8_t)av1o_clampi(*p + res[r], 0, maxpix); }
      else { uint16_t *p = (uint16_t *)dst + (size_t)r * stride + col; *p = (uint16_t)av1o_clampi(*p + res[r], 0, maxpix); }
    }
  }
  return 0;
}
static void fwht4(int32_t *a, int32_t *b, int32_t *c, int32_t *d) {
  int32_t e;
  *a += *b; *d -= *c; e = (*a - *d) >> 1; *b = e - *b; *c = e - *c; *a -= *c; *d += *b;
}
static int fwd_wht4x4(const int16_t *resid, int stride, int32_t *coef) {
  int32_t t[16];
  for (int col = 0; col < 4; col++) {
    int32_t a = resid[col], b = resid[stride + col], c = resid[2 * stride + col], d = resid[3 * stride + col];
    fwht4(&a, &b, &c, &d);
    t[col] = a; t[4 + col] = c; t[8 + col] = d; t[12 + col] = b;
  }
  for (int r = 0; r < 4; r++) {
    int32_t a = t[r * 4], b = t[r * 4 + 1], c = t[r * 4 + 2], d = t[r * 4 + 3];
    fwht4(&a, &b, &c, &d);
    coef[r * 4] = a * 4; coef[r * 4 + 1] = c * 4; coef[r * 4 + 2] = d * 4; coef[r * 4 + 3] = b * 4;
  }
  return 0;
}

/* spec Transform_Row_Shift[] == -libaom av1_inv_txfm_shift_ls[][0]. */
static const int inv_row_shift[TX_SIZES_ALL] = { 0, 1, 2, 2, 2, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2 };

static int rect_log_ratio(int w, int h) { return ilog2(w) - ilog2(h); }

/*
 * K2.  spec §7.13.3 (2D inverse transform) + reconstruction == libaom inv_txfm2d_add_c().
 * coef: row-major, row stride min(w,32), min(h,32) rows (64-length dims carry only the
 * low 32 frequencies, libaom av1_inv_txfm2d_add_64x64_c).  dst holds the prediction on
 * entry, the reconstruction on return; uint8 when bd == 8, uint16 otherwise.
 * libaom_clamps != 0 applies libaom's per-stage clamp_value() (bd+8 rows, max(bd+6,16) cols).
 */
int av1o_inv_txfm2d_add(const int32_t *coef, void *dst, int stride, int tx_size, int tx_type, int bd,
                        int libaom_clamps) {
  if (!av1o_txfm_valid(tx_size, tx_type)) return -1;
  if (tx_type == AV1O_WHT_WHT) return inv_wht4x4_add(coef, dst, stride, bd);
  const int w = av1o_tx_w[tx_size], h = av1o_tx_h[tx_size];
  const int cw = w > 32 ? 32 : w, ch = h > 32 ? 32 : h;
  const int rkind = row_1d(tx_type), ckind = col_1d(tx_type);
  const int row_range = bd + 8, col_range = (bd + 6 > 16) ? bd + 6 : 16;
  const int rect = abs(rect_log_ratio(w, h)) == 1;
  const int rshift = inv_row_shift[tx_size];
  int32_t *buf = (int32_t *)calloc((size_t)w * h, sizeof(int32_t));
  int32_t tin[64], tout[64];
  if (!buf) return -2;
  /* rows (rows >= 32 of a 64-high block are all-zero input -> all-zero output) */
  for (int r = 0; r < ch; r++) {
    for (int c = 0; c < w; c++) {
      int32_t v = c < cw ? coef[r * cw + c] : 0;
      if (rect) v = (int32_t)av1o_round2_64((int64_t)v * NEW_INV_SQRT2, 12);
      tin[c] = clamp_value(v, row_range);
    }
    inv_1d(rkind == T1D_FLIPADST ? T1D_ADST : rkind, tin, tout, w, libaom_clamps ? row_range : 0);
    for (int c = 0; c < w; c++) buf[r * w + c] = av1o_round2(tout[c], rshift);
  }
  /* columns */
  const int maxpix = (1 << bd) - 1;
  for (int c = 0; c < w; c++) {
    const int sc = (rkind == T1D_FLIPADST) ? (w - 1 - c) : c; /* lr_flip */
    for (int r = 0; r < h; r++) tin[r] = clamp_value(buf[r * w + sc], col_range);
    inv_1d(ckind == T1D_FLIPADST ? T1D_ADST : ckind, tin, tout, h, libaom_clamps ? col_range : 0);
    for (int r = 0; r < h; r++) {
      const int sr = (ckind == T1D_FLIPADST) ? (h - 1 - r) : r; /* ud_flip */
      const int res = av1o_round2(tout[sr], 4);
      if (bd == 8) {
        uint8_t *p = (uint8_t *)dst + (size_t)r * stride + c;
        *p = (uint8_t)av1o_clampi(*p + res, 0, maxpix);
      } else {
        uint16_t *p = (uint16_t *)dst + (size_t)r * stride + c;
        *p = (uint16_t)av1o_clampi(*p + res, 0, maxpix);
      }
    }
  }
  free(buf);
  return 0;
}

/* libaom av1_fwd_txfm_shift_ls / av1_fwd_cos_bit_col / _row (non-normative encoder tables). */
static const int8_t fwd_shift[TX_SIZES_ALL][3] = {
  { 2, 0, 0 }, { 2, -1, 0 }, { 2, -2, 0 }, { 2, -4, 0 }, { 0, -2, -2 },
  { 2, -1, 0 }, { 2, -1, 0 }, { 2, -2, 0 }, { 2, -2, 0 }, { 2, -4, 0 }, { 2, -4, 0 }, { 0, -2, -2 }, { 2, -4, -2 },
  { 2, -1, 0 }, { 2, -1, 0 }, { 2, -2, 0 }, { 2, -2, 0 }, { 0, -2, 0 }, { 2, -4, 0 },
};
static const int8_t fwd_cos_bit_col[5][5] = { { 13, 13, 13, 0, 0 }, { 13, 13, 13, 12, 0 }, { 13, 13, 13, 12, 13 },
                                              { 0, 13, 13, 12, 13 }, { 0, 0, 13, 12, 13 } };
static const int8_t fwd_cos_bit_row[5][5] = { { 13, 13, 12, 0, 0 }, { 13, 13, 13, 12, 0 }, { 13, 13, 12, 13, 12 },
                                              { 0, 12, 13, 12, 11 }, { 0, 0, 12, 11, 10 } };
static int32_t shift_val(int32_t v, int sh) { /* libaom av1_round_shift_array: sh>0 => left shift */
  return sh >= 0 ? (int32_t)((uint32_t)v << sh) : av1o_round2(v, -sh);
}

/*
 * K1.  libaom fwd_txfm2d_c(): columns, then rows; 64-length dims keep the low 32
 * frequencies, packed to min(w,32) x min(h,32) row-major.  resid: int16 row-major, `stride`.
 */
int av1o_fwd_txfm2d(const int16_t *resid, int stride, int32_t *coef, int tx_size, int tx_type, int bd) {
  (void)bd;
  if (!av1o_txfm_valid(tx_size, tx_type)) return -1;
  if (tx_type == AV1O_WHT_WHT) return fwd_wht4x4(resid, stride, coef);
  const int w = av1o_tx_w[tx_size], h = av1o_tx_h[tx_size];
  const int cw = w > 32 ? 32 : w, ch = h > 32 ? 32 : h;
  const int rkind = row_1d(tx_type), ckind = col_1d(tx_type);
  const int8_t *sh = fwd_shift[tx_size];
  const int bit_col = fwd_cos_bit_col[ilog2(w) - 2][ilog2(h) - 2];
  const int bit_row = fwd_cos_bit_row[ilog2(w) - 2][ilog2(h) - 2];
  const int rect = abs(rect_log_ratio(w, h)) == 1;
  int32_t *buf = (int32_t *)calloc((size_t)w * h, sizeof(int32_t));
  int32_t tin[64], tout[64];
  if (!buf) return -2;
  for (int c = 0; c < w; c++) {
    for (int r = 0; r < h; r++) {
      const int sr = (ckind == T1D_FLIPADST) ? (h - 1 - r) : r; /* ud_flip */
      tin[r] = shift_val(resid[sr * stride + c], sh[0]);
    }
    fwd_1d(ckind == T1D_FLIPADST ? T1D_ADST : ckind, tin, tout, h, bit_col);
    const int dc = (rkind == T1D_FLIPADST) ? (w - 1 - c) : c; /* lr_flip */
    for (int r = 0; r < h; r++) buf[r * w + dc] = shift_val(tout[r], sh[1]);
  }
  for (int r = 0; r < ch; r++) {
    fwd_1d(rkind == T1D_FLIPADST ? T1D_ADST : rkind, buf + r * w, tout, w, bit_row);
    for (int c = 0; c < cw; c++) {
      int32_t v = shift_val(tout[c], sh[2]);
      if (rect) v = (int32_t)av1o_round2_64((int64_t)v * NEW_SQRT2, 12); /* x sqrt(2): undoes the inverse's 1/sqrt(2) */
      coef[r * cw + c] = v;
    }
  }
  free(buf);
  return 0;
}

/* ------------------------------------------------------------------ K8: quantiser ------------- */
/* spec §7.12.2 Dc_Qlookup / Ac_Qlookup == libaom dc_qlookup_QTX / ac_qlookup_QTX (8- and 10-bit). */
static const int16_t dc_q8[256] = {
  4, 8, 8, 9, 10, 11, 12, 12, 13, 14, 15, 16, 17, 18, 19, 19, 20, 21, 22, 23, 24, 25, 26, 26, 27, 28, 29, 30, 31, 32,
  32, 33, 34, 35, 36, 37, 38, 38, 39, 40, 41, 42, 43, 43, 44, 45, 46, 47, 48, 48, 49, 50, 51, 52, 53, 53, 54, 55, 56,
  57, 57, 58, 59, 60, 61, 62, 62, 63, 64, 65, 66, 66, 67, 68, 69, 70, 70, 71, 72, 73, 74, 74, 75, 76, 77, 78, 78, 79,
  80, 81, 81, 82, 83, 84, 85, 85, 87, 88, 90, 92, 93, 95, 96, 98, 99, 101, 102, 104, 105, 107, 108, 110, 111, 113, 114,
  116, 117, 118, 120, 121, 123, 125, 127, 129, 131, 134, 136, 138, 140, 142, 144, 146, 148, 150, 152, 154, 156, 158,
  161, 164, 166, 169, 172, 174, 177, 180, 182, 185, 187, 190, 192, 195, 199, 202, 205, 208, 211, 214, 217, 220, 223,
  226, 230, 233, 237, 240, 243, 247, 250, 253, 257, 261, 265, 269, 272, 276, 280, 284, 288, 292, 296, 300, 304, 309,
  313, 317, 322, 326, 330, 335, 340, 344, 349, 354, 359, 364, 369, 374, 379, 384, 389, 395, 400, 406, 411, 417, 423,
  429, 435, 441, 447, 454, 461, 467, 475, 482, 489, 497, 505, 513, 522, 530, 539, 549, 559, 569, 579, 590, 602, 614,
  626, 640, 654, 668, 684, 700, 717, 736, 755, 775, 796, 819, 843, 869, 896, 925, 955, 988, 1022, 1058, 1098, 1139,
  1184, 1232, 1282, 1336 };
static const int16_t ac_q8[256] = {
  4, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36,
  37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65,
  66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94,
  95, 96, 97, 98, 99, 100, 101, 102, 104, 106, 108, 110, 112, 114, 116, 118, 120, 122, 124, 126, 128, 130, 132, 134,
  136, 138, 140, 142, 144, 146, 148, 150, 152, 155, 158, 161, 164, 167, 170, 173, 176, 179, 182, 185, 188, 191, 194,
  197, 200, 203, 207, 211, 215, 219, 223, 227, 231, 235, 239, 243, 247, 251, 255, 260, 265, 270, 275, 280, 285, 290,
  295, 300, 305, 311, 317, 323, 329, 335, 341, 347, 353, 359, 366, 373, 380, 387, 394, 401, 408, 416, 424, 432, 440,
  448, 456, 465, 474, 483, 492, 501, 510, 520, 530, 540, 550, 560, 571, 582, 593, 604, 615, 627, 639, 651, 663, 676,
  689, 702, 715, 729, 743, 757, 771, 786, 801, 816, 832, 848, 864, 881, 898, 915, 933, 951, 969, 988, 1007, 1026, 1046,
  1066, 1087, 1108, 1129, 1151, 1173, 1196, 1219, 1243, 1267, 1292, 1317, 1343, 1369, 1396, 1423, 1451, 1479, 1508,
  1537, 1567, 1597, 1628, 1660, 1692, 1725, 1759, 1793, 1828 };
static const int16_t dc_q10[256] = {
  4, 9, 10, 13, 15, 17, 20, 22, 25, 28, 31, 34, 37, 40, 43, 47, 50, 53, 57, 60, 64, 68, 71, 75, 78, 82, 86, 90, 93,
  97, 101, 105, 109, 113, 116, 120, 124, 128, 132, 136, 140, 143, 147, 151, 155, 159, 163, 166, 170, 174, 178, 182,
  185, 189, 193, 197, 200, 204, 208, 212, 215, 219, 223, 226, 230, 233, 237, 241, 244, 248, 251, 255, 259, 262, 266,
  269, 273, 276, 280, 283, 287, 290, 293, 297, 300, 304, 307, 310, 314, 317, 321, 324, 327, 331, 334, 337, 343, 350,
  356, 362, 369, 375, 381, 387, 394, 400, 406, 412, 418, 424, 430, 436, 442, 448, 454, 460, 466, 472, 478, 484, 490,
  499, 507, 516, 525, 533, 542, 550, 559, 567, 576, 584, 592, 601, 609, 617, 625, 634, 644, 655, 666, 676, 687, 698,
  708, 718, 729, 739, 749, 759, 770, 782, 795, 807, 819, 831, 844, 856, 868, 880, 891, 906, 920, 933, 947, 961, 975,
  988, 1001, 1015, 1030, 1045, 1061, 1076, 1090, 1105, 1120, 1137, 1153, 1170, 1186, 1202, 1218, 1236, 1253, 1271,
  1288, 1306, 1323, 1342, 1361, 1379, 1398, 1416, 1436, 1456, 1476, 1496, 1516, 1537, 1559, 1580, 1601, 1624, 1647,
  1670, 1692, 1717, 1741, 1766, 1791, 1817, 1844, 1871, 1900, 1929, 1958, 1990, 2021, 2054, 2088, 2123, 2159, 2197,
  2236, 2276, 2319, 2363, 2410, 2458, 2508, 2561, 2616, 2675, 2737, 2802, 2871, 2944, 3020, 3102, 3188, 3280, 3375,
  3478, 3586, 3702, 3823, 3953, 4089, 4236, 4394, 4559, 4737, 4929, 5130, 5347 };
static const int16_t ac_q10[256] = {
  4, 9, 11, 13, 16, 18, 21, 24, 27, 30, 33, 37, 40, 44, 48, 51, 55, 59, 63, 67, 71, 75, 79, 83, 88, 92, 96, 100, 105,
  109, 114, 118, 122, 127, 131, 136, 140, 145, 149, 154, 158, 163, 168, 172, 177, 181, 186, 190, 195, 199, 204, 208,
  213, 217, 222, 226, 231, 235, 240, 244, 249, 253, 258, 262, 267, 271, 275, 280, 284, 289, 293, 297, 302, 306, 311,
  315, 319, 324, 328, 332, 337, 341, 345, 349, 354, 358, 362, 367, 371, 375, 379, 384, 388, 392, 396, 401, 409, 417,
  425, 433, 441, 449, 458, 466, 474, 482, 490, 498, 506, 514, 523, 531, 539, 547, 555, 563, 571, 579, 588, 596, 604,
  616, 628, 640, 652, 664, 676, 688, 700, 713, 725, 737, 749, 761, 773, 785, 797, 809, 825, 841, 857, 873, 889, 905,
  922, 938, 954, 970, 986, 1002, 1018, 1038, 1058, 1078, 1098, 1118, 1138, 1158, 1178, 1198, 1218, 1242, 1266, 1290,
  1314, 1338, 1362, 1386, 1411, 1435, 1463, 1491, 1519, 1547, 1575, 1603, 1631, 1663, 1695, 1727, 1759, 1791, 1823,
  1859, 1895, 1931, 1967, 2003, 2039, 2079, 2119, 2159, 2199, 2239, 2283, 2327, 2371, 2415, 2459, 2507, 2555, 2603,
  2651, 2703, 2755, 2807, 2859, 2915, 2971, 3027, 3083, 3143, 3203, 3263, 3327, 3391, 3455, 3523, 3591, 3659, 3731,
  3803, 3876, 3952, 4028, 4104, 4184, 4264, 4348, 4432, 4516, 4604, 4692, 4784, 4876, 4972, 5068, 5168, 5268, 5372,
  5476, 5584, 5692, 5804, 5916, 6032, 6148, 6268, 6388, 6512, 6640, 6768, 6900, 7036, 7172, 7312 };

int av1o_dc_q(int qindex, int delta, int bd) {
  const int q = av1o_clampi(qindex + delta, 0, 255);
  return bd == 8 ? dc_q8[q] : dc_q10[q];
}
int av1o_ac_q(int qindex, int delta, int bd) {
  const int q = av1o_clampi(qindex + delta, 0, 255);
  return bd == 8 ? ac_q8[q] : ac_q10[q];
}
/* libaom av1_get_tx_scale(): (pels > 256) + (pels > 1024). */
int av1o_tx_scale(int tx_size) {
  const int pels = av1o_tx_w[tx_size] * av1o_tx_h[tx_size];
  return (pels > 256) + (pels > 1024);
}

/*
 * Quantise n coefficients of one transform block (libaom quantize_fp_helper_c, no qmatrix,
 * scan-free: every position is visited so the result does not depend on scan order).
 * dc_q/ac_q are the dequant steps.  levels: int16.  dqcoef (may be NULL): int32 dequantised.
 * returns the number of non-zero levels.
 */
/* ac_round: the rounding offset of AC coefficients in 1/128 of the step (64 = one half = libaom's quantize_fp; smaller = a dead
 * zone, an encoder policy: see av1o_pipeline.c AV1O_AC_ROUND_INTER) */
int av1o_quantize_r(const int32_t *coef, int n, int dc_q, int ac_q, int log_scale, int ac_round, int16_t *levels, int32_t *dqcoef) {
  int nz = 0;
  for (int i = 0; i < n; i++) {
    const int q = i ? ac_q : dc_q;
    const int quant = (1 << 16) / q;                 /* libaom quant_fp */
    const int round = av1o_round2(((i ? ac_round : 64) * q) >> 7, log_scale); /* libaom round_fp, ROUND_POWER_OF_TWO(.., log_scale) */
    const int32_t c = coef[i];
    const int sign = c < 0;
    int64_t a = sign ? -(int64_t)c : c;
    int32_t lvl = 0;
    if ((a << (1 + log_scale)) >= q) {
      a += round;
      if (a > 32767) a = 32767;
      lvl = (int32_t)((a * quant) >> (16 - log_scale));
    }
    if (lvl > 32767) lvl = 32767;
    levels[i] = (int16_t)(sign ? -lvl : lvl);
    if (dqcoef) {
      const int32_t dq = (int32_t)(((int64_t)lvl * q) >> log_scale);
      dqcoef[i] = sign ? -dq : dq;
    }
    nz += lvl != 0;
  }
  return nz;
}
int av1o_quantize(const int32_t *coef, int n, int dc_q, int ac_q, int log_scale, int16_t *levels, int32_t *dqcoef) {
  return av1o_quantize_r(coef, n, dc_q, ac_q, log_scale, 64, levels, dqcoef);
}
/* spec §7.12.3 dequantisation (normative) == libaom read_coeffs_txb tail: (level*q & 0xFFFFFF) >> shift, clamp. */
void av1o_dequantize(const int16_t *levels, int n, int dc_q, int ac_q, int log_scale, int bd, int32_t *dqcoef) {
  const int32_t maxv = (1 << (7 + bd)) - 1, minv = -(1 << (7 + bd));
  for (int i = 0; i < n; i++) {
    const int q = i ? ac_q : dc_q;
    const int l = levels[i];
    const int sign = l < 0;
    int64_t dq = ((int64_t)(sign ? -l : l) * q) & 0xFFFFFF;
    dq >>= log_scale;
    if (sign) dq = -dq;
    dqcoef[i] = (int32_t)(dq < minv ? minv : (dq > maxv ? maxv : dq));
  }
}
