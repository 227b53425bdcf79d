// entropy.cpp — see entropy.hpp.  Arithmetic restated from AV1 spec §8.2 (symbol decoding / CDF adaptation) and libaom
// aom_dsp/entenc.c, entdec.c; syntax and initial CDFs are this project's own (the stream is NOT AV1).
#include "entropy.hpp"
#include "entropy_init.hpp"
#include <atomic>
#include <cstring>
#include <thread>

namespace av1mi_host {

static inline int msb32(uint32_t v) { return 31 - __builtin_clz(v); }

EntropyModels::EntropyModels() {
  static_assert(sizeof(EntropyModels) == sizeof(kEntropyInit), "entropy_init.hpp is out of step with EntropyModels");
  memcpy((void *)this, kEntropyInit, sizeof(*this));
}
void EntropyModels::set_uniform() {
  for (Cdf4 &c : cdf) { c.c[0] = 8192; c.c[1] = 16384; c.c[2] = 24576; c.count = 0; }
}
void cdf_adapt(Cdf4 &cdf, int s) {   // spec §8.2.6 with N = 4
  const int rate = 5 + (cdf.count > 15) + (cdf.count > 31);
  for (int i = 0; i < 3; i++) {
    if (i >= s) cdf.c[i] += (uint16_t)((32768 - cdf.c[i]) >> rate);
    else cdf.c[i] -= (uint16_t)(cdf.c[i] >> rate);
  }
  cdf.count += cdf.count < 32;
}
// spec §8.2.6 interval partition: lower end of symbol s in "value" space (symbol 0 sits at the top of the range)
static inline uint32_t lower_end(uint32_t rng, const Cdf4 &cdf, int s) {
  const uint32_t f = 32768u - (s < 3 ? cdf.c[s] : 32768u);
  return (((rng >> 8) * (f >> 6)) >> 1) + 4u * (uint32_t)(3 - s);
}

// ------------------------------------------------------------------------------------------------ encoder
void RangeEncoder::add(uint64_t v) {
  low += v;
  if (low >> (16 + pending)) {          // carry into the bytes already written
    for (size_t i = out.size(); i-- > 0;) if (++out[i] != 0) break;
    low &= ((uint64_t)1 << (16 + pending)) - 1;
  }
}
void RangeEncoder::normalize() {
  const int d = 15 - msb32(rng);
  rng <<= d; low <<= d; pending += d;
  while (pending >= 8) {
    pending -= 8;
    out.push_back((uint8_t)((low >> (16 + pending)) & 0xFF));
    low &= ((uint64_t)1 << (16 + pending)) - 1;
  }
}
void RangeEncoder::encode(int s, Cdf4 &cdf) {
  const uint32_t hi = s ? lower_end(rng, cdf, s - 1) : rng, lo = lower_end(rng, cdf, s);
  add(lo);
  rng = hi - lo;
  normalize();
  cdf_adapt(cdf, s);
}
void RangeEncoder::encode_bits(unsigned v, int nbits) {
  while (nbits > 0) {
    const int n = nbits > 8 ? 8 : nbits;
    nbits -= n;
    const uint32_t top = (1u << n) - 1, j = top - ((v >> nbits) & top), r = rng >> n;
    add((uint64_t)r * j);
    rng = j == top ? rng - r * j : r;
    normalize();
  }
}
void RangeEncoder::finish() {   // any value in [low, low + rng) decodes; emit `low` itself, padded with zero bits
  int bits = 16 + pending;
  while (bits > 0) {
    const int take = bits >= 8 ? 8 : bits;
    out.push_back((uint8_t)(((low >> (bits - take)) << (8 - take)) & 0xFF));
    bits -= take;
  }
}

// ------------------------------------------------------------------------------------------------ decoder
void RangeDecoder::init(const uint8_t *p, size_t n) { buf = p; len = n; pos = 0; rng = 0x8000; code = 0; avail = -16; refill(); }
void RangeDecoder::refill() {
  while (avail <= 40) { code = (code << 8) | (pos < len ? buf[pos] : 0); pos++; avail += 8; }
}
int RangeDecoder::decode(Cdf4 &cdf) {
  if (avail < 16) refill();
  const uint32_t value = (uint32_t)(code >> avail);
  uint32_t cur = rng, prev;
  int s = -1;
  do { s++; prev = cur; cur = lower_end(rng, cdf, s); } while (value < cur && s < 3);
  code -= (uint64_t)cur << avail;
  rng = prev - cur;
  const int d = 15 - msb32(rng);
  rng <<= d; avail -= d;
  cdf_adapt(cdf, s);
  return s;
}
unsigned RangeDecoder::decode_bits(int nbits) {
  unsigned v = 0;
  while (nbits > 0) {
    const int n = nbits > 8 ? 8 : nbits;
    nbits -= n;
    if (avail < 16) refill();
    const uint32_t value = (uint32_t)(code >> avail), top = (1u << n) - 1, r = rng >> n;
    uint32_t j = value / r;
    if (j > top) j = top;
    code -= (uint64_t)(r * j) << avail;
    rng = j == top ? rng - r * j : r;
    const int d = 15 - msb32(rng);
    rng <<= d; avail -= d;
    v = (v << n) | (top - j);
  }
  return v;
}

// ------------------------------------------------------------------------------------------------ syntax
namespace {

struct Scan { uint8_t s8[64], s4[16]; };
const Scan &scans() {
  static Scan sc = [] {
    Scan t;
    auto zig = [](int n, uint8_t *o) {
      int k = 0;
      for (int d = 0; d < 2 * n - 1; d++)
        for (int i = 0; i <= d; i++) {
          const int r = (d & 1) ? i : d - i, c = d - r;
          if (r < n && c < n) o[k++] = (uint8_t)(r * n + c);
        }
    };
    zig(8, t.s8); zig(4, t.s4);
    return t;
  }();
  return sc;
}
using Models = EntropyModels;
inline int eob_class(int e) { return e <= 2 ? e : 1 + (32 - __builtin_clz((unsigned)(e - 1))); }   // 3-4:3 5-8:4 9-16:5 17-32:6 33-64:7
inline int band_of(int i) { return i == 0 ? 0 : i <= 4 ? 1 : i <= 15 ? 2 : 3; }

// a value 0..15 as two 4-ary symbols, the low one conditioned on the high one
inline void put_pair(RangeEncoder &e, Models &m, int hi_id, int lo_id, int v) { e.encode(v >> 2, m.cdf[hi_id]); e.encode(v & 3, m.cdf[lo_id + (v >> 2)]); }
inline int get_pair(RangeDecoder &d, Models &m, int hi_id, int lo_id) { const int hi = d.decode(m.cdf[hi_id]); return hi * 4 + d.decode(m.cdf[lo_id + hi]); }

void put_block(RangeEncoder &e, Models &m, int pt, const int16_t *lv, int n, const uint8_t *scan) {
  int eob = 0;
  for (int i = 0; i < n; i++) if (lv[scan[i]]) eob = i + 1;
  const int c = eob_class(eob);
  e.encode(c >> 2, m.cdf[CDF_EOB_HI + pt]); e.encode(c & 3, m.cdf[CDF_EOB_LO + pt * 2 + (c >> 2)]);
  if (c >= 3) e.encode_bits((unsigned)(eob - ((1 << (c - 2)) + 1)), c - 2);
  int prev = 0, nnz = 0;
  uint64_t signs = 0;
  for (int i = 0; i < eob; i++) {
    const int l = lv[scan[i]], a = l < 0 ? -l : l, t = a < 3 ? a : 3;
    e.encode(t, m.cdf[CDF_TOK + (pt * 4 + band_of(i)) * 3 + prev]);
    if (a) { signs = (signs << 1) | (uint64_t)(l < 0); nnz++; }
    prev = t < 2 ? t : 2;
  }
  for (int i = 0; i < eob; i++) {
    const int l = lv[scan[i]], a = l < 0 ? -l : l;
    if (a < 3) continue;
    const unsigned x = (unsigned)(a - 2);
    const int k = msb32(x);                                  // 0..14
    for (int j = 0, r = k;; j++, r -= 3) { const int s = r < 3 ? r : 3; e.encode(s, m.cdf[CDF_GOL + pt * 5 + j]); if (s < 3) break; }
    if (k) e.encode_bits(x & ((1u << k) - 1), k);
  }
  while (nnz > 0) {
    const int k = nnz > 8 ? 8 : nnz;
    nnz -= k;
    e.encode_bits((unsigned)((signs >> nnz) & 0xFF), k);
  }
}
bool get_block(RangeDecoder &d, Models &m, int pt, int16_t *lv, int n, const uint8_t *scan) {
  memset(lv, 0, sizeof(int16_t) * n);
  const int chi = d.decode(m.cdf[CDF_EOB_HI + pt]), c = chi * 4 + d.decode(m.cdf[CDF_EOB_LO + pt * 2 + (chi & 1)]);
  if (chi > 1) return false;
  int eob = c;
  if (c >= 3) eob = (1 << (c - 2)) + 1 + (int)d.decode_bits(c - 2);
  if (eob > n) return false;
  int mag[64], prev = 0, nnz = 0;
  for (int i = 0; i < eob; i++) {
    const int t = d.decode(m.cdf[CDF_TOK + (pt * 4 + band_of(i)) * 3 + prev]);
    mag[i] = t;
    nnz += t != 0;
    prev = t < 2 ? t : 2;
  }
  for (int i = 0; i < eob; i++) {
    if (mag[i] != 3) continue;
    int k = 0;
    for (int j = 0;; j++) { const int s = d.decode(m.cdf[CDF_GOL + pt * 5 + j]); k += s; if (s < 3) break; if (j == 4) return false; }
    if (k > 14) return false;
    mag[i] = 2 + (int)((1u << k) | (k ? d.decode_bits(k) : 0));
    if (mag[i] > 32768) return false;
  }
  int i = 0;
  while (nnz > 0) {
    const int k = nnz > 8 ? 8 : nnz;
    nnz -= k;
    const unsigned bits = d.decode_bits(k);
    for (int j = k - 1; j >= 0; i++) {
      if (i >= eob) return false;
      if (!mag[i]) continue;
      lv[scan[i]] = (int16_t)(((bits >> j) & 1) ? -mag[i] : mag[i]);
      j--;
    }
  }
  return true;
}
void put_mv_comp(RangeEncoder &e, Models &m, int comp, int v) {
  const unsigned a = (unsigned)(v < 0 ? -v : v); int k = a ? 32 - __builtin_clz(a) : 0;
  if (k > 15) k = 15;
  put_pair(e, m, CDF_MV_HI + comp, CDF_MV_LO + comp * 4, k);
  if (k == 15) e.encode_bits(a - 16384, 15);
  else if (k > 1) e.encode_bits(a & ((1u << (k - 1)) - 1), k - 1);
  if (a) e.encode_bits(v < 0, 1);
}
int get_mv_comp(RangeDecoder &d, Models &m, int comp) {
  const int k = get_pair(d, m, CDF_MV_HI + comp, CDF_MV_LO + comp * 4);
  if (!k) return 0;
  const unsigned a = k == 15 ? 16384 + d.decode_bits(15) : (1u << (k - 1)) | (k > 1 ? d.decode_bits(k - 1) : 0);
  return d.decode_bits(1) ? -(int)a : (int)a;
}

}  // namespace

void entropy_encode_tile(const FrameSyms &f, int tx, int ty, RangeEncoder &e, EntropyModels *final_models) {
  Models m;
  const Scan &sc = scans();
  const int bw = f.width / 8, bh = f.height / 8, tb = f.tile / 8;
  const int bx0 = tx * tb, by0 = ty * tb, bx1 = bx0 + tb < bw ? bx0 + tb : bw, by1 = by0 + tb < bh ? by0 + tb : bh;
  for (int by = by0; by < by1; by++)
    for (int bx = bx0; bx < bx1; bx++) {
      const int b = by * bw + bx;
      bool coded = true;
      if (f.key) {
        put_pair(e, m, CDF_MODE_HI, CDF_MODE_LO, f.modes_y[b] < 13 ? f.modes_y[b] : 0);
        put_pair(e, m, CDF_MODE_HI + 1, CDF_MODE_LO + 4, f.modes_uv[b] < 13 ? f.modes_uv[b] : 0);
      } else {
        const int sk = f.skip[b] != 0;
        e.encode(sk, m.cdf[CDF_SKIP]);
        const int px = bx > bx0 ? f.mvs[(b - 1) * 2] : 0, py = bx > bx0 ? f.mvs[(b - 1) * 2 + 1] : 0;   // left neighbour inside the tile predicts
        put_mv_comp(e, m, 0, (int16_t)(f.mvs[b * 2] - px));         // differences wrap modulo 2^16
        put_mv_comp(e, m, 1, (int16_t)(f.mvs[b * 2 + 1] - py));
        coded = !sk;
      }
      if (coded) {
        put_block(e, m, 0, f.lev_y + (size_t)b * 64, 64, sc.s8);
        put_block(e, m, 1, f.lev_u + (size_t)b * 16, 16, sc.s4);
        put_block(e, m, 1, f.lev_v + (size_t)b * 16, 16, sc.s4);
      }
    }
  e.finish();
  if (final_models) *final_models = m;
}

static void put_varint(std::vector<uint8_t> &o, size_t v) { while (v >= 128) { o.push_back((uint8_t)(v | 128)); v >>= 7; } o.push_back((uint8_t)v); }

std::vector<uint8_t> entropy_assemble_frame(int tile, const std::vector<const uint8_t *> &tiles, const std::vector<size_t> &sizes) {
  std::vector<uint8_t> o;
  size_t tot = 0;
  for (size_t n : sizes) tot += n;
  o.reserve(tot + sizes.size() * 3 + 1);
  o.push_back((uint8_t)msb32((uint32_t)tile));
  for (size_t n : sizes) put_varint(o, n);
  for (size_t i = 0; i < sizes.size(); i++) o.insert(o.end(), tiles[i], tiles[i] + sizes[i]);
  return o;
}

std::vector<uint8_t> entropy_encode_frame(const FrameSyms &f) {
  const int tc = (f.width + f.tile - 1) / f.tile, tr = (f.height + f.tile - 1) / f.tile;
  std::vector<std::vector<uint8_t>> parts((size_t)tc * tr);
  std::vector<const uint8_t *> ptrs; std::vector<size_t> sizes;
  for (int ty = 0; ty < tr; ty++)
    for (int tx = 0; tx < tc; tx++) {
      RangeEncoder e;
      e.out.reserve((size_t)f.tile * f.tile / 2);
      entropy_encode_tile(f, tx, ty, e);
      parts[(size_t)ty * tc + tx] = std::move(e.out);
    }
  for (auto &p : parts) { ptrs.push_back(p.data()); sizes.push_back(p.size()); }
  return entropy_assemble_frame(f.tile, ptrs, sizes);
}

bool entropy_decode_frame(const uint8_t *data, size_t n, int width, int height, int key, int16_t *lev_y, int16_t *lev_u, int16_t *lev_v,
                          uint8_t *modes_y, uint8_t *modes_uv, int16_t *mvs, uint8_t *skip) {
  if (n < 1 || data[0] < 5 || data[0] > 12) return false;   // tiles of 32..4096 luma samples
  const int tile = 1 << data[0], tc = (width + tile - 1) / tile, tr = (height + tile - 1) / tile;
  size_t pos = 1;
  std::vector<size_t> sizes((size_t)tc * tr);
  for (size_t &sz : sizes) {
    size_t v = 0; int sh = 0;
    for (;;) { if (pos >= n || sh > 56) return false; const uint8_t c = data[pos++]; v |= (size_t)(c & 127) << sh; sh += 7; if (c < 128) break; }
    sz = v;
  }
  const Scan &sc = scans();
  const int bw = width / 8, bh = height / 8, tb = tile / 8;
  for (int ty = 0; ty < tr; ty++)
    for (int tx = 0; tx < tc; tx++) {
      const size_t sz = sizes[(size_t)ty * tc + tx];
      if (pos + sz > n) return false;
      RangeDecoder d;
      d.init(data + pos, sz);
      pos += sz;
      Models m;
      const int bx0 = tx * tb, by0 = ty * tb, bx1 = bx0 + tb < bw ? bx0 + tb : bw, by1 = by0 + tb < bh ? by0 + tb : bh;
      for (int by = by0; by < by1; by++)
        for (int bx = bx0; bx < bx1; bx++) {
          const int b = by * bw + bx;
          bool coded = true;
          if (key) {
            modes_y[b] = (uint8_t)get_pair(d, m, CDF_MODE_HI, CDF_MODE_LO);
            modes_uv[b] = (uint8_t)get_pair(d, m, CDF_MODE_HI + 1, CDF_MODE_LO + 4);
          } else {
            skip[b] = (uint8_t)(d.decode(m.cdf[CDF_SKIP]) != 0);
            const int px = bx > bx0 ? mvs[(b - 1) * 2] : 0, py = bx > bx0 ? mvs[(b - 1) * 2 + 1] : 0;
            mvs[b * 2] = (int16_t)(px + get_mv_comp(d, m, 0));
            mvs[b * 2 + 1] = (int16_t)(py + get_mv_comp(d, m, 1));
            coded = !skip[b];
          }
          if (coded) {
            if (!get_block(d, m, 0, lev_y + (size_t)b * 64, 64, sc.s8)) return false;
            if (!get_block(d, m, 1, lev_u + (size_t)b * 16, 16, sc.s4)) return false;
            if (!get_block(d, m, 1, lev_v + (size_t)b * 16, 16, sc.s4)) return false;
          } else {
            memset(lev_y + (size_t)b * 64, 0, 128); memset(lev_u + (size_t)b * 16, 0, 32); memset(lev_v + (size_t)b * 16, 0, 32);
          }
        }
      if (d.pos > sz + 8) return false;
    }
  return pos == n;
}

void entropy_encode_frames(const std::vector<FrameSyms> &frames, int threads, std::vector<std::vector<uint8_t>> *out) {
  out->assign(frames.size(), {});
  if (threads < 1) threads = 1;
  if ((size_t)threads > frames.size()) threads = (int)frames.size();
  std::atomic<size_t> next{0};
  auto work = [&] { for (size_t i; (i = next.fetch_add(1)) < frames.size();) (*out)[i] = entropy_encode_frame(frames[i]); };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; t++) pool.emplace_back(work);
  work();
  for (auto &t : pool) t.join();
}

}  // namespace av1mi_host
