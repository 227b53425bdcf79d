/*
 * av1o_entropy.c — CPU ORACLE for the tile entropy coder (row H1 / kernel K9).  TEST INFRASTRUCTURE ONLY (av1o_common.h).
 *
 * PARITY UNPINNED, and more so than the pixel stages: the coded syntax is this project's own, so there is no
 * external stream to compare with.  What IS restated from the published text:
 *   - the multi-symbol interval partition and its minimum-probability term: AV1 spec §8.2.6 (libaom od_ec_encode_q15 /
 *     od_ec_decode_cdf_q15, EC_MIN_PROB = 4, EC_PROB_SHIFT = 6);
 *   - the CDF adaptation rule and its counter: AV1 spec §8.2.6 (libaom update_cdf, prob.h);
 *   - renormalisation to a 16-bit range after every symbol (libaom od_ec_enc_normalize).
 * Written independently of av1-go_amd/host/entropy.cpp: 32-bit base, carries resolved by holding back one byte plus a run
 * count of 0xFF bytes (no back-patching), explicit context arithmetic.  Both must produce identical bytes, and the decoder
 * in entropy.cpp must return the input symbols from this encoder's output.
 *
 * Every adaptive symbol is 4-ary; a CDF is (c0, c1, c2, counter), 61 of them (ids below = enum in host/entropy.hpp).
 * Frame record: [log2(tile)] [varint size of every tile, raster order] [tile payloads].
 * Tile payload: blocks in raster order inside the tile; per block
 *   key frame:  mode_y, mode_uv, each as (m >> 2, m & 3), the second symbol's CDF chosen by the first
 *   P frame:    skip; mv.x and mv.y as differences to the left block of the same tile (wrapping int16): class = bit length
 *               of |d| capped at 15 as (k >> 2, k & 3), the bits below the leading one raw (class 15: |d| - 16384 in 15
 *               bits), sign raw
 *   unless skipped, for Y 8x8 then U 4x4 then V 4x4 in zig-zag order:
 *               eob class (0,1,2,3-4,5-8,9-16,17-32,33-64) as (c >> 2, c & 3) + offset bits (raw),
 *               then the token min(|l|,3) of every coefficient below eob (context = plane type, band of position,
 *               min(previous token, 2)); then for every token 3, in the same order: k = floor(log2(|l| - 2)) as the
 *               chain min(k,3), min(k-3,3), ... stopped by the first symbol below 3 (CDF per chain position), then the k
 *               bits below the leading one (raw); then the signs of the non-zero coefficients, 8 per raw symbol
 *               (first coefficient in the most significant bit, the last group holds what is left)
 * Raw bits are coded up to 8 at a time as one symbol over 2^n equal slots of (range >> n), value v in slot 2^n-1-v from the
 * bottom, the top slot taking the remainder.
 */
#include "av1o_common.h"
#include <string.h>
#include "../av1-go_amd/host/entropy_init.hpp"

typedef struct {
  uint8_t *out; size_t cap, n;
  uint32_t low, rng; int pend;    /* low holds 16 + pend bits */
  int held, ff;                   /* held-back byte (-1 none) followed by ff bytes of 0xFF */
  int overflow;
} enc_t;

static void raw_out(enc_t *e, int b) { if (e->n < e->cap) e->out[e->n] = (uint8_t)b; else e->overflow = 1; e->n++; }
static void byte_out(enc_t *e, int b) {
  if (b == 0xFF && e->held >= 0) { e->ff++; return; }
  if (e->held >= 0) raw_out(e, e->held);
  for (; e->ff; e->ff--) raw_out(e, 0xFF);
  e->held = b;
}
static void carry(enc_t *e) {         /* +1 into the last emitted byte */
  if (e->ff) { raw_out(e, e->held + 1); for (; e->ff > 1; e->ff--) raw_out(e, 0); e->ff = 0; e->held = 0; }   /* FF run -> zeros, last one held */
  else e->held += 1;                  /* nothing behind it: keep holding the incremented byte */
}
static void renorm(enc_t *e) {
  int d = 0;
  uint64_t wide;                      /* 16 + pend + d can reach 38 bits before the bytes are taken off */
  while (!((e->rng << d) & 0x8000)) d++;
  e->rng <<= d; wide = (uint64_t)e->low << d; e->pend += d;
  while (e->pend >= 8) {
    e->pend -= 8;
    byte_out(e, (int)((wide >> (16 + e->pend)) & 0xFF));
    wide &= ((uint64_t)1 << (16 + e->pend)) - 1;
  }
  e->low = (uint32_t)wide;
}
static void add_low(enc_t *e, uint32_t v) {
  e->low += v;
  if (e->low >> (16 + e->pend)) { carry(e); e->low &= (1u << (16 + e->pend)) - 1; }
}
/* spec 8.2.6 with N = 4: symbol s owns [bound(s), bound(s-1)) with bound(-1) = rng; cdf = (c0, c1, c2, counter) */
static uint32_t bound(uint32_t rng, const uint16_t *cdf, int s) {
  const uint32_t c = s < 3 ? cdf[s] : 32768u;
  return ((rng >> 8) * ((32768u - c) >> 6) >> 1) + 4u * (uint32_t)(3 - s);
}
static void put_sym(enc_t *e, uint16_t *cdf, int s) {
  const uint32_t top = s ? bound(e->rng, cdf, s - 1) : e->rng, bot = bound(e->rng, cdf, s);
  int i, rate = 3 + (cdf[3] > 15) + (cdf[3] > 31) + 2;
  add_low(e, bot);
  e->rng = top - bot;
  renorm(e);
  for (i = 0; i < 3; i++) {
    if (i >= s) cdf[i] += (uint16_t)((32768 - cdf[i]) >> rate);
    else cdf[i] -= (uint16_t)(cdf[i] >> rate);
  }
  if (cdf[3] < 32) cdf[3]++;
}
static void put_raw(enc_t *e, uint32_t v, int nbits) {
  while (nbits > 0) {
    const int n = nbits > 8 ? 8 : nbits;
    uint32_t slots, j, r;
    nbits -= n;
    slots = 1u << n; j = slots - 1 - ((v >> nbits) & (slots - 1)); r = e->rng >> n;
    add_low(e, r * j);
    e->rng = (j == slots - 1) ? e->rng - r * j : r;
    renorm(e);
  }
}
static size_t finish(enc_t *e) {
  int bits = 16 + e->pend;
  if (e->held >= 0) raw_out(e, e->held);
  for (; e->ff; e->ff--) raw_out(e, 0xFF);
  while (bits > 0) {
    const int take = bits >= 8 ? 8 : bits;
    raw_out(e, (int)(((e->low >> (bits - take)) << (8 - take)) & 0xFF));
    bits -= take;
  }
  return e->n;
}

/* CDF ids (x4 = word offset inside kEntropyInit), same enum as av1-go_amd/host/entropy.hpp */
enum { C_TOK = 0, C_GOL = 24, C_EOB_HI = 34, C_EOB_LO = 36, C_MODE_HI = 40, C_MODE_LO = 42, C_SKIP = 50, C_MV_HI = 51, C_MV_LO = 53, C_COUNT = 61 };
#define CDF(m, id) ((m) + 4 * (id))

static int bitlen(uint32_t v) { int n = 0; while (v) { n++; v >>= 1; } return n; }

static void zigzag(int n, uint8_t *o) {
  int d, i, k = 0;
  for (d = 0; d < 2 * n - 1; d++)
    for (i = 0; i <= d; i++) {
      const int r = (d & 1) ? i : d - i, c = d - r;
      if (r < n && c < n) o[k++] = (uint8_t)(r * n + c);
    }
}
static void put_coeffs(enc_t *e, uint16_t *m, int pt, const int16_t *lv, int n, const uint8_t *scan) {
  int eob = 0, i, prev = 0, cls, nnz = 0;
  uint8_t sign[64];
  for (i = 0; i < n; i++) if (lv[scan[i]]) eob = i + 1;
  cls = eob <= 2 ? eob : 1 + bitlen((uint32_t)(eob - 1));
  put_sym(e, CDF(m, C_EOB_HI + pt), cls >> 2);
  put_sym(e, CDF(m, C_EOB_LO + pt * 2 + (cls >> 2)), cls & 3);
  if (cls >= 3) put_raw(e, (uint32_t)(eob - (1 << (cls - 2)) - 1), cls - 2);
  for (i = 0; i < eob; i++) {                                        /* 1: every token */
    const int l = lv[scan[i]], a = l < 0 ? -l : l, t = a < 3 ? a : 3, band = i == 0 ? 0 : i <= 4 ? 1 : i <= 15 ? 2 : 3;
    put_sym(e, CDF(m, C_TOK + (pt * 4 + band) * 3 + prev), t);
    if (a) sign[nnz++] = l < 0;
    prev = t < 2 ? t : 2;
  }
  for (i = 0; i < eob; i++) {                                        /* 2: remainders of the escapes */
    const int l = lv[scan[i]], a = l < 0 ? -l : l;
    if (a >= 3) {
      const uint32_t x = (uint32_t)(a - 2);
      const int k = bitlen(x) - 1;              /* a <= 32768 so k <= 14 */
      int j, rest = k;
      for (j = 0; j < 5; j++) {
        const int sy = rest < 3 ? rest : 3;
        put_sym(e, CDF(m, C_GOL + pt * 5 + j), sy);
        if (sy < 3) break;
        rest -= 3;
      }
      if (k) put_raw(e, x & ((1u << k) - 1), k);
    }
  }
  for (i = 0; i < nnz;) {                                            /* 3: signs, 8 per raw symbol, the short group last */
    const int left = nnz - i, k = left > 8 ? 8 : left;
    uint32_t v = 0;
    int j;
    for (j = 0; j < k; j++) v = (v << 1) | sign[i + j];
    put_raw(e, v, k);
    i += k;
  }
}
static void put_mvd(enc_t *e, uint16_t *m, int comp, int v) {
  const uint32_t a = (uint32_t)(v < 0 ? -v : v);
  int k = bitlen(a);
  if (k > 15) k = 15;
  put_sym(e, CDF(m, C_MV_HI + comp), k >> 2);
  put_sym(e, CDF(m, C_MV_LO + comp * 4 + (k >> 2)), k & 3);
  if (k == 15) put_raw(e, a - 16384, 15);
  else if (k > 1) put_raw(e, a & ((1u << (k - 1)) - 1), k - 1);
  if (a) put_raw(e, v < 0, 1);
}
static void put_mode(enc_t *e, uint16_t *m, int which, int mode) {
  put_sym(e, CDF(m, C_MODE_HI + which), mode >> 2);
  put_sym(e, CDF(m, C_MODE_LO + which * 4 + (mode >> 2)), mode & 3);
}

/* one tile's payload; returns its size, or (size_t)-1 when cap is too small */
size_t av1o_entropy_encode_tile(int w, int h, int key, int tile, int tx, int ty, const int16_t *lev_y, const int16_t *lev_u,
                                const int16_t *lev_v, const uint8_t *modes_y, const uint8_t *modes_uv, const int16_t *mvs,
                                const uint8_t *skip, uint8_t *out, size_t cap) {
  uint16_t m[C_COUNT * 4];
  uint8_t s8[64], s4[16];
  enc_t e;
  const int bw = w / 8, bh = h / 8, tb = tile / 8;
  const int bx0 = tx * tb, by0 = ty * tb, bx1 = bx0 + tb < bw ? bx0 + tb : bw, by1 = by0 + tb < bh ? by0 + tb : bh;
  int bx, by;
  size_t n;
  memcpy(m, kEntropyInit, sizeof(m));
  zigzag(8, s8); zigzag(4, s4);
  memset(&e, 0, sizeof(e));
  e.out = out; e.cap = cap; e.rng = 0x8000; e.held = -1;
  for (by = by0; by < by1; by++)
    for (bx = bx0; bx < bx1; bx++) {
      const int b = by * bw + bx;
      int coded = 1;
      if (key) {
        put_mode(&e, m, 0, modes_y[b] < 13 ? modes_y[b] : 0);
        put_mode(&e, m, 1, modes_uv[b] < 13 ? modes_uv[b] : 0);
      } else {
        const int px = bx > bx0 ? mvs[(b - 1) * 2] : 0, py = bx > bx0 ? mvs[(b - 1) * 2 + 1] : 0;
        put_sym(&e, CDF(m, C_SKIP), skip[b] != 0);
        put_mvd(&e, m, 0, (int16_t)(mvs[b * 2] - px));
        put_mvd(&e, m, 1, (int16_t)(mvs[b * 2 + 1] - py));
        coded = !skip[b];
      }
      if (coded) {
        put_coeffs(&e, m, 0, lev_y + (size_t)b * 64, 64, s8);
        put_coeffs(&e, m, 1, lev_u + (size_t)b * 16, 16, s4);
        put_coeffs(&e, m, 1, lev_v + (size_t)b * 16, 16, s4);
      }
    }
  n = finish(&e);
  return e.overflow ? (size_t)-1 : n;
}

/* whole frame record; returns its size or (size_t)-1 */
size_t av1o_entropy_encode_frame(int w, int h, int key, int tile, const int16_t *lev_y, const int16_t *lev_u, const int16_t *lev_v,
                                 const uint8_t *modes_y, const uint8_t *modes_uv, const int16_t *mvs, const uint8_t *skip,
                                 uint8_t *out, size_t cap) {
  const int tc = (w + tile - 1) / tile, tr = (h + tile - 1) / tile, nt = tc * tr;
  /* payloads first into the tail of out, then the size table is known: do it in two passes over a scratch half */
  size_t pos, hdr = 1, total = 0;
  int t;
  uint8_t *scratch = out + cap / 2;
  size_t scap = cap - cap / 2, spos = 0;
  size_t sizes[8192];
  if (nt > 8192) return (size_t)-1;
  for (t = 0; t < nt; t++) {
    const size_t n = av1o_entropy_encode_tile(w, h, key, tile, t % tc, t / tc, lev_y, lev_u, lev_v, modes_y, modes_uv, mvs, skip,
                                              scratch + spos, scap - spos);
    size_t v = n;
    if (n == (size_t)-1) return n;
    sizes[t] = n; spos += n; total += n;
    do { hdr++; v >>= 7; } while (v);
  }
  if (hdr + total > cap / 2) return (size_t)-1;
  out[0] = (uint8_t)(bitlen((uint32_t)tile) - 1);
  pos = 1;
  for (t = 0; t < nt; t++) {
    size_t v = sizes[t];
    while (v >= 128) { out[pos++] = (uint8_t)(v | 128); v >>= 7; }
    out[pos++] = (uint8_t)v;
  }
  memmove(out + pos, scratch, total);
  return pos + total;
}
