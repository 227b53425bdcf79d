// txfm1d.hpp — 1-D AV1 transforms as fully-unrolled register butterflies for gfx950.
//
// One lane owns one row (or column) of a transform block: T[N] lives in VGPRs, every index below
// is a compile-time constant after unrolling, every cosine is an SGPR/literal operand and every
// product is a full-rate 24-bit multiply (v_mul_i32_i24 / v_mad_i32_i24): operands stay inside
// +-2^23 because rows are clamped to bd+8 bits, columns to max(bd+6,16) bits and add/sub stages to
// the same range (libaom clamp_value), bd <= 10.  No MFMA: these are small fixed integer
// butterflies whose rounding points are normative.
//
// Restates AV1 spec §7.13.2 (inverse DCT / ADST4 / ADST8 / ADST16 / identity) and libaom
// av1_inv_txfm1d.c / av1_fwd_txfm1d.c; nothing to cite under /root/reference, which holds no codec
// arithmetic (rows K1/K2 of §8a).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace av1mi {

#define AV1MI_DI __device__ __forceinline__
#define AV1MI_HD __host__ __device__ constexpr

// round(cos(i*pi/128) * 2^bit), bit = 10..13, i = 0..64
__device__ constexpr int32_t kCospi[4][65] = {
  { 1024, 1024, 1023, 1021, 1019, 1016, 1013, 1009, 1004, 999, 993, 987, 980, 972, 964, 955,
    946, 936, 926, 915, 903, 891, 878, 865, 851, 837, 822, 807, 792, 775, 759, 742,
    724, 706, 688, 669, 650, 630, 610, 590, 569, 548, 526, 505, 483, 460, 438, 415,
    392, 369, 345, 321, 297, 273, 249, 224, 200, 175, 150, 125, 100, 75, 50, 25, 0 },
  { 2048, 2047, 2046, 2042, 2038, 2033, 2026, 2018, 2009, 1998, 1987, 1974, 1960, 1945, 1928, 1911,
    1892, 1872, 1851, 1829, 1806, 1782, 1757, 1730, 1703, 1674, 1645, 1615, 1583, 1551, 1517, 1483,
    1448, 1412, 1375, 1338, 1299, 1260, 1220, 1179, 1138, 1096, 1053, 1009, 965, 921, 876, 830,
    784, 737, 690, 642, 595, 546, 498, 449, 400, 350, 301, 251, 201, 151, 100, 50, 0 },
  { 4096, 4095, 4091, 4085, 4076, 4065, 4052, 4036, 4017, 3996, 3973, 3948, 3920, 3889, 3857, 3822,
    3784, 3745, 3703, 3659, 3612, 3564, 3513, 3461, 3406, 3349, 3290, 3229, 3166, 3102, 3035, 2967,
    2896, 2824, 2751, 2675, 2598, 2520, 2440, 2359, 2276, 2191, 2106, 2019, 1931, 1842, 1751, 1660,
    1567, 1474, 1380, 1285, 1189, 1092, 995, 897, 799, 700, 601, 501, 401, 301, 201, 101, 0 },
  { 8192, 8190, 8182, 8170, 8153, 8130, 8103, 8071, 8035, 7993, 7946, 7895, 7839, 7779, 7713, 7643,
    7568, 7489, 7405, 7317, 7225, 7128, 7027, 6921, 6811, 6698, 6580, 6458, 6333, 6203, 6070, 5933,
    5793, 5649, 5501, 5351, 5197, 5040, 4880, 4717, 4551, 4383, 4212, 4038, 3862, 3683, 3503, 3320,
    3135, 2948, 2760, 2570, 2378, 2185, 1990, 1795, 1598, 1401, 1202, 1003, 803, 603, 402, 201, 0 },
};
// round(sin(k*pi/9) * sqrt(2) * 2/3 * 2^bit)
__device__ constexpr int32_t kSinpi[4][5] = {
  { 0, 330, 621, 836, 951 }, { 0, 660, 1241, 1672, 1902 },
  { 0, 1321, 2482, 3344, 3803 }, { 0, 2642, 4965, 6689, 7606 } };
constexpr int32_t kNewSqrt2 = 5793, kNewInvSqrt2 = 2896;

AV1MI_HD int ilog2c(int n) { int r = 0; while ((1 << r) < n) r++; return r; }
AV1MI_HD int brevc(int nbits, int x) {
  int r = 0;
  for (int i = 0; i < nbits; i++) r |= ((x >> i) & 1) << (nbits - 1 - i);
  return r;
}
AV1MI_HD int32_t cos128c(int bit, int angle) {
  const int a = angle & 255;
  return a <= 64 ? kCospi[bit - 10][a] : a <= 128 ? -kCospi[bit - 10][128 - a]
       : a <= 192 ? -kCospi[bit - 10][a - 128] : kCospi[bit - 10][256 - a];
}
AV1MI_HD int32_t sin128c(int bit, int angle) { return cos128c(bit, angle - 64); }
// first-rotation angle of the odd block [M,2M): frequency k = 1 + 2*brev(log2 M, i)
AV1MI_HD int r0_angle(int M, int i) { return 64 - (1 + 2 * brevc(ilog2c(M), i)) * 32 / M; }

// w0 * a + w1 * b rounded: two 24-bit multiply-adds (full rate).  Written with __mul24 the compiler, which knows the operands'
// ranges from the stage clamps, turned a good part of the products back into 32-bit multiplies and 64-bit multiply-adds
// (v_mul_lo_u32 / v_mad_u64_u32: four passes each on CDNA) — a third of the multiplies of the fused kernels' residual tail.
// w0, w1 are the rotation's constants (uniform); a, b fit 24 bits by the stage clamps (header comment).
template <int BIT> AV1MI_DI int32_t hbtf(int32_t w0, int32_t a, int32_t w1, int32_t b) {
  int32_t t;
  const int32_t rnd = 1 << (BIT - 1);
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t) : "v"(a), "s"(w0), "v"(rnd));
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t) : "v"(b), "s"(w1), "v"(t));
  return t >> BIT;
}
// clamp to a signed RANGE-bit value (one v_med3_i32); RANGE == 0: no clamp
template <int RANGE> AV1MI_DI int32_t clampr(int32_t v) {
  if constexpr (RANGE == 0) return v;
  else return min(max(v, -(1 << (RANGE - 1))), (1 << (RANGE - 1)) - 1);
}
AV1MI_DI int32_t round2(int32_t v, int n) { return n ? (v + (1 << (n - 1))) >> n : v; }

// ---- butterfly primitives (spec §7.13.2.2) --------------------------------------------------
template <int BIT> AV1MI_DI void rot_inv(int32_t &a, int32_t &b, int angle, bool flip) {
  const int32_t c = cos128c(BIT, angle), s = sin128c(BIT, angle);
  const int32_t x = hbtf<BIT>(c, a, -s, b), y = hbtf<BIT>(s, a, c, b);
  if (flip) { a = y; b = x; } else { a = x; b = y; }
}
template <int BIT> AV1MI_DI void rot_fwd(int32_t &a, int32_t &b, int angle, bool flip) {
  const int32_t c = cos128c(BIT, angle), s = sin128c(BIT, angle);
  if (flip) { const int32_t x = hbtf<BIT>(s, a, c, b), y = hbtf<BIT>(c, a, -s, b); a = x; b = y; }
  else      { const int32_t x = hbtf<BIT>(c, a, s, b), y = hbtf<BIT>(-s, a, c, b); a = x; b = y; }
}
template <int RANGE> AV1MI_DI void had(int32_t &a, int32_t &b, bool flip) {
  const int32_t x = a, y = b;
  if (flip) { a = clampr<RANGE>(y - x); b = clampr<RANGE>(x + y); }
  else      { a = clampr<RANGE>(x + y); b = clampr<RANGE>(x - y); }
}

// ---- inverse DCT (spec §7.13.2.3 == libaom av1_idct4..64) -------------------------------------
template <int M, int BIT, int RANGE> AV1MI_DI void idct_odd(int32_t *T) {
#pragma unroll
  for (int i = 0; i < M / 2; i++) rot_inv<BIT>(T[M + i], T[2 * M - 1 - i], r0_angle(M, i), false);
#pragma unroll
  for (int s = 2; s <= M / 2; s *= 2) {
#pragma unroll
    for (int g = 0; g < M / s; g++)
#pragma unroll
      for (int j = 0; j < s / 2; j++) had<RANGE>(T[M + g * s + j], T[M + g * s + s - 1 - j], g & 1);
    if (4 * s <= M) {
#pragma unroll
      for (int g = 0; g < M / (4 * s); g++) {
        const int th = r0_angle(M / (2 * s), g);
#pragma unroll
        for (int q = 0; q < s / 2; q++) {
          const int p = g * 2 * s + s / 2 + q, p2 = g * 2 * s + s + q;
          rot_inv<BIT>(T[2 * M - 1 - p], T[M + p], th, true);
          rot_inv<BIT>(T[2 * M - 1 - p2], T[M + p2], th + 64, true);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < s / 2; q++) rot_inv<BIT>(T[2 * M - 1 - (s / 2 + q)], T[M + s / 2 + q], 32, true);
    }
  }
}
template <int N, int BIT, int RANGE> AV1MI_DI void idct_core(int32_t *T) {
  if constexpr (N == 2) {
    const int32_t c = kCospi[BIT - 10][32];
    const int32_t a = hbtf<BIT>(c, T[0], c, T[1]), b = hbtf<BIT>(c, T[0], -c, T[1]);
    T[0] = a; T[1] = b;
  } else {
    idct_core<N / 2, BIT, RANGE>(T);
    idct_odd<N / 2, BIT, RANGE>(T);
#pragma unroll
    for (int i = 0; i < N / 2; i++) had<RANGE>(T[i], T[N - 1 - i], false);
  }
}
// x[N] natural order in -> natural order out, in place
template <int N, int RANGE> AV1MI_DI void idct(int32_t *x) {
  int32_t T[N];
#pragma unroll
  for (int i = 0; i < N; i++) T[i] = x[brevc(ilog2c(N), i)];
  idct_core<N, 12, RANGE>(T);
#pragma unroll
  for (int i = 0; i < N; i++) x[i] = T[i];
}

// ---- forward DCT: the transposed flow graph run backwards (libaom av1_fdct4..64) -------------
template <int M, int BIT> AV1MI_DI void fdct_odd(int32_t *T) {
#pragma unroll
  for (int s = M / 2; s >= 2; s /= 2) {
    if (4 * s <= M) {
#pragma unroll
      for (int g = 0; g < M / (4 * s); g++) {
        const int th = r0_angle(M / (2 * s), g);
#pragma unroll
        for (int q = 0; q < s / 2; q++) {
          const int p = g * 2 * s + s / 2 + q, p2 = g * 2 * s + s + q;
          rot_fwd<BIT>(T[2 * M - 1 - p], T[M + p], th, true);
          rot_fwd<BIT>(T[2 * M - 1 - p2], T[M + p2], th + 64, true);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < s / 2; q++) rot_fwd<BIT>(T[2 * M - 1 - (s / 2 + q)], T[M + s / 2 + q], 32, true);
    }
#pragma unroll
    for (int g = 0; g < M / s; g++)
#pragma unroll
      for (int j = 0; j < s / 2; j++) had<0>(T[M + g * s + j], T[M + g * s + s - 1 - j], g & 1);
  }
#pragma unroll
  for (int i = 0; i < M / 2; i++) rot_fwd<BIT>(T[M + i], T[2 * M - 1 - i], r0_angle(M, i), false);
}
template <int N, int BIT> AV1MI_DI void fdct_core(int32_t *T) {
  if constexpr (N == 2) {
    const int32_t c = kCospi[BIT - 10][32];
    const int32_t a = hbtf<BIT>(c, T[0], c, T[1]), b = hbtf<BIT>(-c, T[1], c, T[0]);
    T[0] = a; T[1] = b;
  } else {
#pragma unroll
    for (int i = 0; i < N / 2; i++) had<0>(T[i], T[N - 1 - i], false);
    fdct_odd<N / 2, BIT>(T);
    fdct_core<N / 2, BIT>(T);
  }
}
template <int N, int BIT> AV1MI_DI void fdct(int32_t *x) {
  int32_t T[N];
#pragma unroll
  for (int i = 0; i < N; i++) T[i] = x[i];
  fdct_core<N, BIT>(T);
#pragma unroll
  for (int i = 0; i < N; i++) x[brevc(ilog2c(N), i)] = T[i];
}

// ---- ADST (spec §7.13.2.6-8 == libaom av1_iadst4/8/16, av1_fadst4/8/16) ------------------------
template <int BIT> AV1MI_DI void iadst4(int32_t *x) {
  const int32_t s1 = kSinpi[BIT - 10][1], s2 = kSinpi[BIT - 10][2], s3 = kSinpi[BIT - 10][3], s4 = kSinpi[BIT - 10][4];
  const int32_t x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
  const int32_t a0 = __mul24(s1, x0) + __mul24(s4, x2) + __mul24(s2, x3);
  const int32_t a1 = __mul24(s2, x0) - __mul24(s1, x2) - __mul24(s4, x3);
  const int32_t a3 = __mul24(s3, x1);
  const int32_t a2 = __mul24(s3, (x0 - x2) + x3);
  x[0] = round2(a0 + a3, BIT); x[1] = round2(a1 + a3, BIT);
  x[2] = round2(a2, BIT);      x[3] = round2(a0 + a1 - a3, BIT);
}
template <int BIT> AV1MI_DI void fadst4(int32_t *x) {
  const int32_t s1 = kSinpi[BIT - 10][1], s2 = kSinpi[BIT - 10][2], s3 = kSinpi[BIT - 10][3], s4 = kSinpi[BIT - 10][4];
  const int32_t x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
  const int32_t a0 = __mul24(s1, x0) + __mul24(s2, x1) + __mul24(s4, x3);
  const int32_t a1 = __mul24(s3, x0 + x1 - x3);
  const int32_t a2 = __mul24(s4, x0) - __mul24(s1, x1) + __mul24(s2, x3);
  const int32_t a3 = __mul24(s3, x2);
  x[0] = round2(a0 + a3, BIT); x[1] = round2(a1, BIT);
  x[2] = round2(a2 - a3, BIT); x[3] = round2(a2 - a0 + a3, BIT);
}
template <int N, int H, int RANGE> AV1MI_DI void adst_addsub(int32_t *b) {
#pragma unroll
  for (int g = 0; g < N; g += 2 * H)
#pragma unroll
    for (int i = 0; i < H; i++) {
      const int32_t x = b[g + i], y = b[g + i + H];
      b[g + i] = clampr<RANGE>(x + y); b[g + i + H] = clampr<RANGE>(x - y);
    }
}
template <int BIT> AV1MI_DI void adst_pp(int32_t *b, int p, int i0, int i1) {   // [c0 c1; c1 -c0]
  const int32_t c0 = kCospi[BIT - 10][i0], c1 = kCospi[BIT - 10][i1];
  const int32_t x = hbtf<BIT>(c0, b[p], c1, b[p + 1]), y = hbtf<BIT>(c1, b[p], -c0, b[p + 1]);
  b[p] = x; b[p + 1] = y;
}
template <int BIT> AV1MI_DI void adst_np(int32_t *b, int p, int i0, int i1) {   // [-c1 c0; c0 c1]
  const int32_t c0 = kCospi[BIT - 10][i0], c1 = kCospi[BIT - 10][i1];
  const int32_t x = hbtf<BIT>(-c1, b[p], c0, b[p + 1]), y = hbtf<BIT>(c0, b[p], c1, b[p + 1]);
  b[p] = x; b[p + 1] = y;
}
template <int BIT, int RANGE> AV1MI_DI void iadst8(int32_t *x) {
  int32_t b[8];
#pragma unroll
  for (int i = 0; i < 4; i++) { b[2 * i] = x[7 - 2 * i]; b[2 * i + 1] = x[2 * i]; }
#pragma unroll
  for (int i = 0; i < 4; i++) adst_pp<BIT>(b, 2 * i, 4 + 16 * i, 60 - 16 * i);
  adst_addsub<8, 4, RANGE>(b);
  adst_pp<BIT>(b, 4, 16, 48); adst_np<BIT>(b, 6, 16, 48);
  adst_addsub<8, 2, RANGE>(b);
  adst_pp<BIT>(b, 2, 32, 32); adst_pp<BIT>(b, 6, 32, 32);
  x[0] = b[0]; x[1] = -b[4]; x[2] = b[6]; x[3] = -b[2]; x[4] = b[3]; x[5] = -b[7]; x[6] = b[5]; x[7] = -b[1];
}
template <int BIT, int RANGE> AV1MI_DI void iadst16(int32_t *x) {
  int32_t b[16];
#pragma unroll
  for (int i = 0; i < 8; i++) { b[2 * i] = x[15 - 2 * i]; b[2 * i + 1] = x[2 * i]; }
#pragma unroll
  for (int i = 0; i < 8; i++) adst_pp<BIT>(b, 2 * i, 2 + 8 * i, 62 - 8 * i);
  adst_addsub<16, 8, RANGE>(b);
  adst_pp<BIT>(b, 8, 8, 56);  adst_pp<BIT>(b, 10, 40, 24);
  adst_np<BIT>(b, 12, 8, 56); adst_np<BIT>(b, 14, 40, 24);
  adst_addsub<16, 4, RANGE>(b);
  adst_pp<BIT>(b, 4, 16, 48);  adst_np<BIT>(b, 6, 16, 48);
  adst_pp<BIT>(b, 12, 16, 48); adst_np<BIT>(b, 14, 16, 48);
  adst_addsub<16, 2, RANGE>(b);
#pragma unroll
  for (int p = 2; p < 16; p += 4) adst_pp<BIT>(b, p, 32, 32);
  x[0] = b[0];  x[1] = -b[8];   x[2] = b[12];  x[3] = -b[4];  x[4] = b[6];  x[5] = -b[14];  x[6] = b[10];  x[7] = -b[2];
  x[8] = b[3];  x[9] = -b[11];  x[10] = b[15]; x[11] = -b[7]; x[12] = b[5]; x[13] = -b[13]; x[14] = b[9];  x[15] = -b[1];
}
template <int BIT> AV1MI_DI void fadst8(int32_t *x) {
  int32_t b[8];
  b[0] = x[0]; b[4] = -x[1]; b[6] = x[2]; b[2] = -x[3]; b[3] = x[4]; b[7] = -x[5]; b[5] = x[6]; b[1] = -x[7];
  adst_pp<BIT>(b, 2, 32, 32); adst_pp<BIT>(b, 6, 32, 32);
  adst_addsub<8, 2, 0>(b);
  adst_pp<BIT>(b, 4, 16, 48); adst_np<BIT>(b, 6, 16, 48);
  adst_addsub<8, 4, 0>(b);
#pragma unroll
  for (int i = 0; i < 4; i++) adst_pp<BIT>(b, 2 * i, 4 + 16 * i, 60 - 16 * i);
#pragma unroll
  for (int i = 0; i < 4; i++) { x[7 - 2 * i] = b[2 * i]; x[2 * i] = b[2 * i + 1]; }
}
template <int BIT> AV1MI_DI void fadst16(int32_t *x) {
  int32_t b[16];
  b[0] = x[0];  b[8] = -x[1];   b[12] = x[2];  b[4] = -x[3];  b[6] = x[4];  b[14] = -x[5];  b[10] = x[6];  b[2] = -x[7];
  b[3] = x[8];  b[11] = -x[9];  b[15] = x[10]; b[7] = -x[11]; b[5] = x[12]; b[13] = -x[13]; b[9] = x[14];  b[1] = -x[15];
#pragma unroll
  for (int p = 2; p < 16; p += 4) adst_pp<BIT>(b, p, 32, 32);
  adst_addsub<16, 2, 0>(b);
  adst_pp<BIT>(b, 4, 16, 48);  adst_np<BIT>(b, 6, 16, 48);
  adst_pp<BIT>(b, 12, 16, 48); adst_np<BIT>(b, 14, 16, 48);
  adst_addsub<16, 4, 0>(b);
  adst_pp<BIT>(b, 8, 8, 56);  adst_pp<BIT>(b, 10, 40, 24);
  adst_np<BIT>(b, 12, 8, 56); adst_np<BIT>(b, 14, 40, 24);
  adst_addsub<16, 8, 0>(b);
#pragma unroll
  for (int i = 0; i < 8; i++) adst_pp<BIT>(b, 2 * i, 2 + 8 * i, 62 - 8 * i);
#pragma unroll
  for (int i = 0; i < 8; i++) { x[15 - 2 * i] = b[2 * i]; x[2 * i] = b[2 * i + 1]; }
}

// ---- identity (spec §7.13.2.15) ---------------------------------------------------------------
template <int N> AV1MI_DI void identity(int32_t *x) {
#pragma unroll
  for (int i = 0; i < N; i++) {
    if constexpr (N == 4) x[i] = (__mul24(x[i], kNewSqrt2) + 2048) >> 12;
    else if constexpr (N == 8) x[i] = x[i] * 2;
    else if constexpr (N == 16) x[i] = (__mul24(x[i], 2 * kNewSqrt2) + 2048) >> 12;
    else x[i] = x[i] * 4;
  }
}

enum { T1D_DCT = 0, T1D_ADST = 1, T1D_FLIPADST = 2, T1D_IDTX = 3 };
constexpr int kWhtType = 16;   // AV1MI_WHT_WHT: lossless 4x4 Walsh-Hadamard, handled apart from the kind tables
// spec 7.13.2.10 on (T[0], T[1], T[2], T[3]) = x[0..3]
AV1MI_DI void iwht4(int32_t *x, int shift) {
  int32_t a = x[0] >> shift, c = x[1] >> shift, d = x[2] >> shift, b = x[3] >> shift;
  a += c; d -= b;
  const int32_t e = (a - d) >> 1;
  b = e - b; c = e - c; a -= b; d += c;
  x[0] = a; x[1] = b; x[2] = c; x[3] = d;
}
// libaom av1_fwht4x4_c butterfly: outputs in the order (a, c, d, b)
AV1MI_DI void fwht4(int32_t *x) {
  int32_t a = x[0], b = x[1], c = x[2], d = x[3];
  a += b; d -= c;
  const int32_t e = (a - d) >> 1;
  b = e - b; c = e - c; a -= c; d += b;
  x[0] = a; x[1] = c; x[2] = d; x[3] = b;
}
// TX_TYPE -> vertical (column) / horizontal (row) 1-D kind, packed 2 bits each (libaom vtx_tab/htx_tab)
AV1MI_HD int col_kind(int tx_type) {
  constexpr int t[16] = { 0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3 };
  return t[tx_type & 15];
}
AV1MI_HD int row_kind(int tx_type) {
  constexpr int t[16] = { 0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2 };
  return t[tx_type & 15];
}

// inverse 1-D of length N on x[], kind in {DCT, ADST (also for FLIPADST), IDTX}; invalid kinds for a
// length (ADST > 16, IDTX > 32) are rejected on the host before launch.
template <int N, int RANGE> AV1MI_DI void inv1d(int32_t *x, int kind) {
  if (kind == T1D_DCT) { idct<N, RANGE>(x); return; }
  if constexpr (N <= 16) {
    if (kind == T1D_ADST || kind == T1D_FLIPADST) {
      if constexpr (N == 4) iadst4<12>(x);
      else if constexpr (N == 8) iadst8<12, RANGE>(x);
      else iadst16<12, RANGE>(x);
      return;
    }
  }
  if constexpr (N <= 32) identity<N>(x);
}
template <int N, int BIT> AV1MI_DI void fwd1d(int32_t *x, int kind) {
  if (kind == T1D_DCT) { fdct<N, BIT>(x); return; }
  if constexpr (N <= 16) {
    if (kind == T1D_ADST || kind == T1D_FLIPADST) {
      if constexpr (N == 4) fadst4<BIT>(x);
      else if constexpr (N == 8) fadst8<BIT>(x);
      else fadst16<BIT>(x);
      return;
    }
  }
  if constexpr (N <= 32) identity<N>(x);
}

}  // namespace av1mi
