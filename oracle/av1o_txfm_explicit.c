/*
 * av1o_txfm_explicit.c — stage-by-stage restatements of libaom av1_idct4 / av1_idct8 /
 * av1_idct16 / av1_idct32 (av1/common/av1_inv_txfm1d.c) written out longhand from knowledge.
 * They exist only to pin the generic generator in av1o_txfm.c (idct_core/idct_odd): the
 * tests require both to agree bit-for-bit on random and extreme inputs.
 * TEST INFRASTRUCTURE ONLY; pinned to dav1d, not to the reference (see av1o_common.h).
 */
#include "av1o_common.h"

extern void av1o_idct(const int32_t *in, int32_t *out, int N, int bit, int range);

static const int32_t C[64] = {
  4096, 4095, 4091, 4085, 4076, 4065, 4052, 4036, 4017, 3996, 3973, 3948, 3920, 3889, 3857, 3822,
  3784, 3745, 3703, 3659, 3612, 3564, 3513, 3461, 3406, 3349, 3290, 3229, 3166, 3102, 3035, 2967,
  2896, 2824, 2751, 2675, 2598, 2520, 2440, 2359, 2276, 2191, 2106, 2019, 1931, 1842, 1751, 1660,
  1567, 1474, 1380, 1285, 1189, 1092, 995, 897, 799, 700, 601, 501, 401, 301, 201, 101 };

static int32_t hb(int32_t w0, int32_t a, int32_t w1, int32_t b) {
  return (int32_t)(((int64_t)w0 * a + (int64_t)w1 * b + 2048) >> 12);
}
static int32_t cl(int64_t v, int bit) {
  if (bit <= 0) return (int32_t)v;
  const int64_t hi = ((int64_t)1 << (bit - 1)) - 1, lo = -((int64_t)1 << (bit - 1));
  return (int32_t)(v < lo ? lo : (v > hi ? hi : v));
}
#define ADD(a, b) cl((int64_t)(a) + (b), R)
#define SUB(a, b) cl((int64_t)(a) - (b), R)

void av1o_idct4_explicit(const int32_t *in, int32_t *out, int R) {
  int32_t s[4], t[4];
  s[0] = in[0]; s[1] = in[2]; s[2] = in[1]; s[3] = in[3];
  t[0] = hb(C[32], s[0], C[32], s[1]);
  t[1] = hb(C[32], s[0], -C[32], s[1]);
  t[2] = hb(C[48], s[2], -C[16], s[3]);
  t[3] = hb(C[16], s[2], C[48], s[3]);
  out[0] = ADD(t[0], t[3]); out[1] = ADD(t[1], t[2]); out[2] = SUB(t[1], t[2]); out[3] = SUB(t[0], t[3]);
}

void av1o_idct8_explicit(const int32_t *in, int32_t *out, int R) {
  int32_t a[8], b[8];
  /* stage 1 */
  a[0] = in[0]; a[1] = in[4]; a[2] = in[2]; a[3] = in[6]; a[4] = in[1]; a[5] = in[5]; a[6] = in[3]; a[7] = in[7];
  /* stage 2 */
  b[0] = a[0]; b[1] = a[1]; b[2] = a[2]; b[3] = a[3];
  b[4] = hb(C[56], a[4], -C[8], a[7]);
  b[5] = hb(C[24], a[5], -C[40], a[6]);
  b[6] = hb(C[40], a[5], C[24], a[6]);
  b[7] = hb(C[8], a[4], C[56], a[7]);
  /* stage 3 */
  a[0] = hb(C[32], b[0], C[32], b[1]);
  a[1] = hb(C[32], b[0], -C[32], b[1]);
  a[2] = hb(C[48], b[2], -C[16], b[3]);
  a[3] = hb(C[16], b[2], C[48], b[3]);
  a[4] = ADD(b[4], b[5]); a[5] = SUB(b[4], b[5]); a[6] = SUB(b[7], b[6]); a[7] = ADD(b[6], b[7]);
  /* stage 4 */
  b[0] = ADD(a[0], a[3]); b[1] = ADD(a[1], a[2]); b[2] = SUB(a[1], a[2]); b[3] = SUB(a[0], a[3]);
  b[4] = a[4];
  b[5] = hb(-C[32], a[5], C[32], a[6]);
  b[6] = hb(C[32], a[5], C[32], a[6]);
  b[7] = a[7];
  /* stage 5 */
  for (int i = 0; i < 4; i++) { out[i] = ADD(b[i], b[7 - i]); out[7 - i] = SUB(b[i], b[7 - i]); }
}

void av1o_idct16_explicit(const int32_t *in, int32_t *out, int R) {
  int32_t a[16], b[16];
  static const int perm[16] = { 0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15 };
  for (int i = 0; i < 16; i++) a[i] = in[perm[i]];
  /* stage 2 */
  for (int i = 0; i < 8; i++) b[i] = a[i];
  b[8] = hb(C[60], a[8], -C[4], a[15]);
  b[9] = hb(C[28], a[9], -C[36], a[14]);
  b[10] = hb(C[44], a[10], -C[20], a[13]);
  b[11] = hb(C[12], a[11], -C[52], a[12]);
  b[12] = hb(C[52], a[11], C[12], a[12]);
  b[13] = hb(C[20], a[10], C[44], a[13]);
  b[14] = hb(C[36], a[9], C[28], a[14]);
  b[15] = hb(C[4], a[8], C[60], a[15]);
  /* stage 3 */
  for (int i = 0; i < 4; i++) a[i] = b[i];
  a[4] = hb(C[56], b[4], -C[8], b[7]);
  a[5] = hb(C[24], b[5], -C[40], b[6]);
  a[6] = hb(C[40], b[5], C[24], b[6]);
  a[7] = hb(C[8], b[4], C[56], b[7]);
  a[8] = ADD(b[8], b[9]); a[9] = SUB(b[8], b[9]); a[10] = SUB(b[11], b[10]); a[11] = ADD(b[10], b[11]);
  a[12] = ADD(b[12], b[13]); a[13] = SUB(b[12], b[13]); a[14] = SUB(b[15], b[14]); a[15] = ADD(b[14], b[15]);
  /* stage 4 */
  b[0] = hb(C[32], a[0], C[32], a[1]);
  b[1] = hb(C[32], a[0], -C[32], a[1]);
  b[2] = hb(C[48], a[2], -C[16], a[3]);
  b[3] = hb(C[16], a[2], C[48], a[3]);
  b[4] = ADD(a[4], a[5]); b[5] = SUB(a[4], a[5]); b[6] = SUB(a[7], a[6]); b[7] = ADD(a[6], a[7]);
  b[8] = a[8];
  b[9] = hb(-C[16], a[9], C[48], a[14]);
  b[10] = hb(-C[48], a[10], -C[16], a[13]);
  b[11] = a[11]; b[12] = a[12];
  b[13] = hb(-C[16], a[10], C[48], a[13]);
  b[14] = hb(C[48], a[9], C[16], a[14]);
  b[15] = a[15];
  /* stage 5 */
  a[0] = ADD(b[0], b[3]); a[1] = ADD(b[1], b[2]); a[2] = SUB(b[1], b[2]); a[3] = SUB(b[0], b[3]);
  a[4] = b[4];
  a[5] = hb(-C[32], b[5], C[32], b[6]);
  a[6] = hb(C[32], b[5], C[32], b[6]);
  a[7] = b[7];
  a[8] = ADD(b[8], b[11]); a[9] = ADD(b[9], b[10]); a[10] = SUB(b[9], b[10]); a[11] = SUB(b[8], b[11]);
  a[12] = SUB(b[15], b[12]); a[13] = SUB(b[14], b[13]); a[14] = ADD(b[13], b[14]); a[15] = ADD(b[12], b[15]);
  /* stage 6 */
  for (int i = 0; i < 4; i++) { b[i] = ADD(a[i], a[7 - i]); b[7 - i] = SUB(a[i], a[7 - i]); }
  b[8] = a[8]; b[9] = a[9];
  b[10] = hb(-C[32], a[10], C[32], a[13]);
  b[11] = hb(-C[32], a[11], C[32], a[12]);
  b[12] = hb(C[32], a[11], C[32], a[12]);
  b[13] = hb(C[32], a[10], C[32], a[13]);
  b[14] = a[14]; b[15] = a[15];
  /* stage 7 */
  for (int i = 0; i < 8; i++) { out[i] = ADD(b[i], b[15 - i]); out[15 - i] = SUB(b[i], b[15 - i]); }
}

/* av1_idct32: stages 2..9 for the odd half [16,32) longhand; the even half is av1_idct16. */
void av1o_idct32_explicit(const int32_t *in, int32_t *out, int R) {
  int32_t ev_in[16], ev[16], a[32], b[32];
  static const int perm_odd[16] = { 1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31 };
  for (int i = 0; i < 16; i++) ev_in[i] = in[2 * i];
  av1o_idct16_explicit(ev_in, ev, R);
  for (int i = 0; i < 16; i++) a[16 + i] = in[perm_odd[i]];
  /* stage 2 */
  b[16] = hb(C[62], a[16], -C[2], a[31]);
  b[17] = hb(C[30], a[17], -C[34], a[30]);
  b[18] = hb(C[46], a[18], -C[18], a[29]);
  b[19] = hb(C[14], a[19], -C[50], a[28]);
  b[20] = hb(C[54], a[20], -C[10], a[27]);
  b[21] = hb(C[22], a[21], -C[42], a[26]);
  b[22] = hb(C[38], a[22], -C[26], a[25]);
  b[23] = hb(C[6], a[23], -C[58], a[24]);
  b[24] = hb(C[58], a[23], C[6], a[24]);
  b[25] = hb(C[26], a[22], C[38], a[25]);
  b[26] = hb(C[42], a[21], C[22], a[26]);
  b[27] = hb(C[10], a[20], C[54], a[27]);
  b[28] = hb(C[50], a[19], C[14], a[28]);
  b[29] = hb(C[18], a[18], C[46], a[29]);
  b[30] = hb(C[34], a[17], C[30], a[30]);
  b[31] = hb(C[2], a[16], C[62], a[31]);
  /* stage 3 */
  for (int g = 16; g < 32; g += 4) {
    a[g] = ADD(b[g], b[g + 1]); a[g + 1] = SUB(b[g], b[g + 1]);
    a[g + 2] = SUB(b[g + 3], b[g + 2]); a[g + 3] = ADD(b[g + 2], b[g + 3]);
  }
  /* stage 4 */
  for (int i = 16; i < 32; i++) b[i] = a[i];
  b[17] = hb(-C[8], a[17], C[56], a[30]);
  b[18] = hb(-C[56], a[18], -C[8], a[29]);
  b[21] = hb(-C[40], a[21], C[24], a[26]);
  b[22] = hb(-C[24], a[22], -C[40], a[25]);
  b[25] = hb(-C[40], a[22], C[24], a[25]);
  b[26] = hb(C[24], a[21], C[40], a[26]);
  b[29] = hb(-C[8], a[18], C[56], a[29]);
  b[30] = hb(C[56], a[17], C[8], a[30]);
  /* stage 5 */
  a[16] = ADD(b[16], b[19]); a[17] = ADD(b[17], b[18]); a[18] = SUB(b[17], b[18]); a[19] = SUB(b[16], b[19]);
  a[20] = SUB(b[23], b[20]); a[21] = SUB(b[22], b[21]); a[22] = ADD(b[21], b[22]); a[23] = ADD(b[20], b[23]);
  a[24] = ADD(b[24], b[27]); a[25] = ADD(b[25], b[26]); a[26] = SUB(b[25], b[26]); a[27] = SUB(b[24], b[27]);
  a[28] = SUB(b[31], b[28]); a[29] = SUB(b[30], b[29]); a[30] = ADD(b[29], b[30]); a[31] = ADD(b[28], b[31]);
  /* stage 6 */
  for (int i = 16; i < 32; i++) b[i] = a[i];
  b[18] = hb(-C[16], a[18], C[48], a[29]);
  b[19] = hb(-C[16], a[19], C[48], a[28]);
  b[20] = hb(-C[48], a[20], -C[16], a[27]);
  b[21] = hb(-C[48], a[21], -C[16], a[26]);
  b[26] = hb(-C[16], a[21], C[48], a[26]);
  b[27] = hb(-C[16], a[20], C[48], a[27]);
  b[28] = hb(C[48], a[19], C[16], a[28]);
  b[29] = hb(C[48], a[18], C[16], a[29]);
  /* stage 7 */
  for (int i = 0; i < 4; i++) {
    a[16 + i] = ADD(b[16 + i], b[23 - i]); a[23 - i] = SUB(b[16 + i], b[23 - i]);
    a[24 + i] = SUB(b[31 - i], b[24 + i]); a[31 - i] = ADD(b[24 + i], b[31 - i]);
  }
  /* stage 8 */
  for (int i = 16; i < 32; i++) b[i] = a[i];
  b[20] = hb(-C[32], a[20], C[32], a[27]);
  b[21] = hb(-C[32], a[21], C[32], a[26]);
  b[22] = hb(-C[32], a[22], C[32], a[25]);
  b[23] = hb(-C[32], a[23], C[32], a[24]);
  b[24] = hb(C[32], a[23], C[32], a[24]);
  b[25] = hb(C[32], a[22], C[32], a[25]);
  b[26] = hb(C[32], a[21], C[32], a[26]);
  b[27] = hb(C[32], a[20], C[32], a[27]);
  /* stage 9 */
  for (int i = 0; i < 16; i++) { out[i] = ADD(ev[i], b[31 - i]); out[31 - i] = SUB(ev[i], b[31 - i]); }
}
