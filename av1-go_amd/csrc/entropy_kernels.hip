// entropy_kernels.hip — K9: the tile entropy coder on the GPU (the stage SURVEY.md §8a row H1 leaves on the host cores and
// §7 / §8e name as the scaling risk: "host entropy coding throughput … needs on-GPU compaction").
//
// A range coder is serial in its state, so the parallel axis is the TILE: every entropy tile (32/64/128 luma samples
// square) has its own coder and its own adaptive CDFs, and one LANE codes one tile — a wave codes 64 tiles in lockstep,
// a 48-frame 1080p batch is 24 480 tiles (32: 97 920).  What makes lockstep cheap is the syntax (host/entropy.hpp,
// oracle/av1o_entropy.c): symbols are grouped by kind inside a transform block (eob, all tokens, the escapes, packed
// signs), so all lanes of a wave are at the SAME kind of symbol with a compile-time alphabet size: the CDF update is a
// fully unrolled LDS read-modify-write per lane with no divergence on N, and the rare kinds run in short loops.
//
//   k_ent_code    lane = tile.  Models: 61 CDFs x 8 bytes per lane in LDS, every adaptive symbol is 4-ary: one b64 load + store.
//                 The current block's 96 levels + its escape list sit in a lane-private LDS row (odd dword stride).
//                 Bits leave 32 at a time as big-endian words (a late carry ripples back through the lane's own words)
//                 into the tile's private slot of 6 bytes per coefficient (a bound: <= 43 bits per coefficient + block headers).
//   k_ent_layout  per frame: scan of the tile sizes -> header (varint) and payload offsets inside the frame record.
//   k_ent_frames  scan over frames -> record offsets, capacity check.
//   k_ent_pack    per tile: varint + payload copied to its place: the host receives one contiguous stream.
//
// Arithmetic: AV1 spec §8.2.6 interval partition + CDF adaptation (restated in oracle/av1o_entropy.c, byte-exact parity);
// initial CDFs: host/entropy_init.hpp (own constants).  HBM traffic: 2 B per coefficient in, <1 B out — the kernel is
// bound by the serial dependency chain of the coder (issue latency of one wave per SIMD), not by bandwidth.
#include "av1mi_internal.hpp"
#include "../host/entropy_init.hpp"
#include <utility>

namespace av1mi {

namespace {

// CDF ids: same enum as host/entropy.hpp; one CDF = (c0, c1, c2, counter) = 8 bytes
enum { C_TOK = 0, C_GOL = 24, C_EOB_HI = 34, C_EOB_LO = 36, C_MODE_HI = 40, C_MODE_LO = 42, C_SKIP = 50, C_MV_HI = 51, C_MV_LO = 53, C_COUNT = 61 };
static_assert(sizeof(kEntropyInit) == C_COUNT * 8, "entropy_init.hpp layout");
constexpr int ROW = 162;   // int16 per lane row: 96 levels (Y 64, U 16, V 16) + 64 escapes + 2 pad = 81 dwords (odd)

__constant__ uint2 kInit[C_COUNT];   // filled once per context from kEntropyInit
// zig-zag scan position of raster index r in an n x n block (compile time): the block is staged into the lane's LDS row
// already in scan order, so the coding loops index it with the (scalar) loop counter and never look a table up
constexpr int zz_pos(int n, int r) {
  int k = 0;
  for (int d = 0; d < 2 * n - 1; d++)
    for (int i = 0; i <= d; i++) {
      const int rr = (d & 1) ? i : d - i, c = d - rr;
      if (rr < n && c < n) { if (rr * n + c == r) return k; k++; }
    }
  return -1;
}
template <int N, int R> struct ZZ { static constexpr int pos = zz_pos(N, R); };
static_assert(ZZ<8, 8>::pos == 2 && ZZ<8, 16>::pos == 3 && ZZ<8, 63>::pos == 63 && ZZ<4, 4>::pos == 2 && ZZ<4, 13>::pos == 10, "zig-zag");
// dwords I... of a block (two raster-adjacent int16 each) scattered to their scan positions
template <int N, int... I> __device__ inline void stage_scan(int16_t *dst, const uint32_t *dw, std::integer_sequence<int, I...>) {
  ((dst[ZZ<N, 2 * I>::pos] = (int16_t)(dw[I] & 0xFFFF), dst[ZZ<N, 2 * I + 1>::pos] = (int16_t)(dw[I] >> 16)), ...);
}

// Coder state of one lane.  `low` carries 16 + pend bits (pend < 32 between symbols) plus, on top, at most one unresolved
// carry bit: a carry out of an addition is NOT resolved when it happens (the interval never leaves [low, low + rng) of the
// previous state, so a second one cannot pile up) but when the next 32 finished bits leave as a big-endian word: bit 32 of
// that word is the carry, and it ripples into the words already in the slot (lane-private memory, rare).  In a wave of 64
// coders SOME lane flushes at almost every step, so the flush itself must be a few predicated instructions, not a branchy
// hold-back state machine.  The byte stream is the same as a byte-wise coder's (oracle/av1o_entropy.c).
struct Enc {
  uint64_t low; uint32_t rng; int pend;
  uint32_t *out; uint32_t n, cap;     // words
};
__device__ inline void ripple(Enc &e) {    // +1 into the big-endian number already written
  for (uint32_t i = e.n < e.cap ? e.n : e.cap; i-- > 0;) {
    const uint32_t w = __builtin_bswap32(e.out[i]) + 1;
    e.out[i] = __builtin_bswap32(w);
    if (w) break;
  }
}
__device__ inline void flush_word(Enc &e) {
  e.pend -= 32;
  const uint64_t w = e.low >> (16 + e.pend);
  e.low &= ((uint64_t)1 << (16 + e.pend)) - 1;
  if (w >> 32) ripple(e);
  if (e.n < e.cap) e.out[e.n] = __builtin_bswap32((uint32_t)w);
  e.n++;
}
// interval [low + add, low + add + nrng) becomes the state; renormalise to a 16-bit range
__device__ inline void commit(Enc &e, uint32_t add, uint32_t nrng) {
  const int d = __clz((int)nrng) - 16;
  e.low = (e.low + add) << d;      // <= 17 + 31 + 15 bits
  e.rng = nrng << d;
  e.pend += d;
  if (e.pend >= 32) flush_word(e);
}
// adaptive 4-ary symbol with CDF `id` of this lane's model row: one 64-bit LDS load, one store
__device__ inline void enc4(Enc &e, uint2 *mrow, int id, int s) {
  const uint2 v = mrow[id];
  const uint32_t c0 = v.x & 0xFFFF, c1 = v.x >> 16, c2 = v.y & 0xFFFF, cnt = v.y >> 16;
  const uint32_t cs = s == 0 ? c0 : s == 1 ? c1 : s == 2 ? c2 : 32768u;
  const uint32_t cp = s == 1 ? c0 : s == 2 ? c1 : c2;
  const uint32_t r8 = e.rng >> 8;
  const uint32_t bot = ((r8 * ((32768u - cs) >> 6)) >> 1) + 4u * (uint32_t)(3 - s);
  const uint32_t top = s ? ((r8 * ((32768u - cp) >> 6)) >> 1) + 4u * (uint32_t)(4 - s) : e.rng;
  commit(e, bot, top - bot);
  const int rate = 5 + (cnt > 15) + (cnt > 31);
  const uint32_t n0 = s <= 0 ? c0 + ((32768u - c0) >> rate) : c0 - (c0 >> rate);
  const uint32_t n1 = s <= 1 ? c1 + ((32768u - c1) >> rate) : c1 - (c1 >> rate);
  const uint32_t n2 = s <= 2 ? c2 + ((32768u - c2) >> rate) : c2 - (c2 >> rate);
  mrow[id] = make_uint2(n0 | (n1 << 16), n2 | ((cnt + (cnt < 32)) << 16));
}
// n (1..8) equiprobable bits as one symbol over 2^n slots
__device__ inline void enc_raw(Enc &e, int n, uint32_t v) {
  const uint32_t top = (1u << n) - 1, j = top - v, r = e.rng >> n, add = r * j;
  commit(e, add, j == top ? e.rng - add : r);
}
// up to 16 raw bits, most significant chunk of <= 8 first; both steps are wave-level branches
__device__ inline void enc_raw16(Enc &e, bool on, int nbits, uint32_t v) {
  if (on && nbits > 0) { const int n = nbits > 8 ? 8 : nbits; enc_raw(e, n, (v >> (nbits - n)) & ((1u << n) - 1)); }
  if (on && nbits > 8) { const int n = nbits - 8; enc_raw(e, n, v & ((1u << n) - 1)); }
}
__device__ inline int wave_max(int v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) { const int w = __shfl_xor(v, o); v = w > v ? w : v; }
  return __builtin_amdgcn_readfirstlane(v);
}

// one transform block of n coefficients at int16 index `base` of the lane row
__device__ inline void code_block(Enc &e, uint2 *m, int16_t *row, int base, int pt, int ncoef, bool coded) {
  int eob = 0;
#pragma unroll 8
  for (int i = 0; i < ncoef; i++) eob = row[base + i] != 0 ? i + 1 : eob;
  if (!coded) eob = 0;
  if (coded) {
    const int cls = eob <= 2 ? eob : 33 - __clz(eob - 1);
    enc4(e, m, C_EOB_HI + pt, cls >> 2);
    enc4(e, m, C_EOB_LO + pt * 2 + (cls >> 2), cls & 3);
    const int xb = cls >= 3 ? cls - 2 : 0;
    enc_raw16(e, cls >= 3, xb, (uint32_t)(eob - (1 << xb) - 1));
  }
  const int emax = wave_max(eob);
  int prev = 0, nnz = 0, nesc = 0;
  uint64_t signs = 0;
  for (int i = 0; i < emax; i++) {          // uniform trip count: scan position and band are scalar
    const int band = i == 0 ? 0 : i <= 4 ? 1 : i <= 15 ? 2 : 3;
    if (i < eob) {
      const int l = row[base + i], a = l < 0 ? -l : l, t = a < 3 ? a : 3;
      enc4(e, m, C_TOK + (pt * 4 + band) * 3 + prev, t);
      if (a) { signs = (signs << 1) | (uint64_t)(l < 0); nnz++; }
      if (a >= 3) row[96 + nesc++] = (int16_t)(a - 2);           // 1..32766
      prev = t < 2 ? t : 2;
    }
  }
  const int xmax = wave_max(nesc);
  for (int j = 0; j < xmax; j++) {
    const bool on = j < nesc;
    const uint32_t x = on ? (uint32_t)row[96 + j] : 1u;
    const int k = 31 - __clz((int)x);
    int rest = on ? k : -1;                  // chain min(k,3), min(k-3,3), ... until a symbol below 3
    for (int c = 0; c < 5; c++) {
      if (__builtin_amdgcn_readfirstlane(__any(rest >= 0)) == 0) break;
      if (rest >= 0) { const int sy = rest < 3 ? rest : 3; enc4(e, m, C_GOL + pt * 5 + c, sy); rest = sy < 3 ? -1 : rest - 3; }
    }
    enc_raw16(e, on, k, x & ((1u << k) - 1));
  }
  const int smax = wave_max((nnz + 7) >> 3);
  for (int c = 0; c < smax; c++) {
    if (nnz > 0) {
      const int k = nnz > 8 ? 8 : nnz;
      nnz -= k;
      enc_raw(e, k, (uint32_t)(signs >> nnz) & ((1u << k) - 1));
    }
  }
}
__device__ inline void code_mvd(Enc &e, uint2 *m, int comp, bool on, int v) {
  const uint32_t a = (uint32_t)(v < 0 ? -v : v);
  int k = a ? 32 - __clz((int)a) : 0;
  k = k > 15 ? 15 : k;
  if (on) { enc4(e, m, C_MV_HI + comp, k >> 2); enc4(e, m, C_MV_LO + comp * 4 + (k >> 2), k & 3); }
  enc_raw16(e, on && k > 1, k == 15 ? 15 : k - 1, k == 15 ? a - 16384 : a & ((1u << (k > 0 ? k - 1 : 0)) - 1));
  if (on && a) enc_raw(e, 1, v < 0);
}

__global__ void __launch_bounds__(64) k_ent_code(EntropyLaunch L) {
  __shared__ uint2 s_models[64 * C_COUNT];      // lane rows of 61 x 8 bytes (odd multiple of 8: conflict-free for equal ids)
  __shared__ int16_t s_rows[64 * ROW];
  const int lane = threadIdx.x;
  const long long g = (long long)blockIdx.x * 64 + lane;
  const int tc = (L.w + L.tile - 1) / L.tile, tr = (L.h + L.tile - 1) / L.tile, tpf = tc * tr;
  const bool live = g < (long long)tpf * L.nframes;
  const int f = live ? (int)(g / tpf) : 0, t = live ? (int)(g % tpf) : 0;
  const int bw = L.w >> 3, bh = L.h >> 3, tb = L.tile >> 3;
  const int bx0 = (t % tc) * tb, by0 = (t / tc) * tb;
  const int bx1 = live ? (bx0 + tb < bw ? bx0 + tb : bw) : bx0, by1 = live ? (by0 + tb < bh ? by0 + tb : bh) : by0;
  const size_t nb = (size_t)bw * bh, fb = (size_t)f * nb;
  uint2 *m = s_models + lane * C_COUNT;
  int16_t *row = s_rows + lane * ROW;
  for (int i = 0; i < C_COUNT; i++) m[i] = kInit[i];
  Enc e;
  e.low = 0; e.rng = 0x8000; e.pend = 0; e.n = 0;
  e.cap = live ? L.slot_bytes / 4 : 0; e.out = (uint32_t *)(L.slots + (live ? (size_t)g * L.slot_bytes : 0));
  for (int lby = 0; lby < tb; lby++)
    for (int lbx = 0; lbx < tb; lbx++) {   // uniform over the wave; ragged tiles mask lanes off
      const int bx = bx0 + lbx, by = by0 + lby;
      const bool valid = bx < bx1 && by < by1;
      const size_t b = fb + (size_t)(valid ? by * bw + bx : 0);
      bool coded = valid;
      if (valid) {       // stage the block's levels in SCAN order
        const uint4 *sy = (const uint4 *)(L.lev[0] + b * 64), *su = (const uint4 *)(L.lev[1] + b * 16), *sv = (const uint4 *)(L.lev[2] + b * 16);
        uint32_t dw[32];
#pragma unroll
        for (int k = 0; k < 8; k++) { const uint4 v = sy[k]; dw[k * 4] = v.x; dw[k * 4 + 1] = v.y; dw[k * 4 + 2] = v.z; dw[k * 4 + 3] = v.w; }
        stage_scan<8>(row, dw, std::make_integer_sequence<int, 32>());
        uint32_t du[16];
#pragma unroll
        for (int k = 0; k < 2; k++) { const uint4 v = su[k], w = sv[k]; du[k * 4] = v.x; du[k * 4 + 1] = v.y; du[k * 4 + 2] = v.z; du[k * 4 + 3] = v.w;
                                      du[8 + k * 4] = w.x; du[9 + k * 4] = w.y; du[10 + k * 4] = w.z; du[11 + k * 4] = w.w; }
        stage_scan<4>(row + 64, du, std::make_integer_sequence<int, 8>());
        stage_scan<4>(row + 80, du + 8, std::make_integer_sequence<int, 8>());
      }
      if (L.key) {
        int my = 0, muv = 0;
        if (valid) { my = L.modes_y[b]; muv = L.modes_uv[b]; my = my < 13 ? my : 0; muv = muv < 13 ? muv : 0; }
        if (valid) { enc4(e, m, C_MODE_HI, my >> 2); enc4(e, m, C_MODE_LO + (my >> 2), my & 3); }
        if (valid) { enc4(e, m, C_MODE_HI + 1, muv >> 2); enc4(e, m, C_MODE_LO + 4 + (muv >> 2), muv & 3); }
      } else {
        int sk = 0, dx = 0, dy = 0;
        if (valid) {
          sk = L.skip[b] != 0;
          const int px = lbx ? L.mvs[(b - 1) * 2] : 0, py = lbx ? L.mvs[(b - 1) * 2 + 1] : 0;
          dx = (int16_t)(L.mvs[b * 2] - px); dy = (int16_t)(L.mvs[b * 2 + 1] - py);
          enc4(e, m, C_SKIP, sk);
        }
        code_mvd(e, m, 0, valid, dx);
        code_mvd(e, m, 1, valid, dy);
        coded = valid && !sk;
      }
#pragma unroll 1
      for (int p = 0; p < 3; p++) code_block(e, m, row, p == 0 ? 0 : 48 + p * 16, p != 0, p == 0 ? 64 : 16, coded);
    }
  if (live) {
    if (e.low >> (16 + e.pend)) { ripple(e); e.low &= ((uint64_t)1 << (16 + e.pend)) - 1; }   // a carry still riding on top
    uint32_t nbytes = e.n * 4;
    uint8_t *tail = (uint8_t *)e.out;
    int bits = 16 + e.pend;                    // <= 47
    while (bits > 0) {
      const int take = bits >= 8 ? 8 : bits;
      if (nbytes < L.slot_bytes) tail[nbytes] = (uint8_t)(((e.low >> (bits - take)) << (8 - take)) & 0xFF);
      nbytes++;
      bits -= take;
    }
    L.sizes[g] = nbytes;     // > slot_bytes would mean the (proven) bound failed: k_ent_layout turns it into an error
  }
}

__device__ inline uint32_t varint_len(uint32_t v) { return v < 128 ? 1 : v < 16384 ? 2 : v < 2097152 ? 3 : v < 268435456 ? 4 : 5; }

// one workgroup per frame: exclusive scans of varint lengths and payload sizes over the frame's tiles
__global__ void __launch_bounds__(256) k_ent_layout(EntropyLaunch L, int tpf) {
  __shared__ uint32_t s_h[256], s_p[256];
  const int f = blockIdx.x, tid = threadIdx.x, per = (tpf + 255) / 256;
  const long long g0 = (long long)f * tpf;
  const int lo = tid * per, hi = lo + per < tpf ? lo + per : tpf;
  uint32_t h = 0, p = 0;
  bool bad = false;
  for (int i = lo; i < hi; i++) { const uint32_t n = L.sizes[g0 + i]; h += varint_len(n); p += n; bad |= n > L.slot_bytes; }
  if (bad) atomicOr(L.status, 2u);   // a slot ran over its (proven) bound: reported, nothing is packed
  s_h[tid] = h; s_p[tid] = p;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const uint32_t ah = tid >= o ? s_h[tid - o] : 0, ap = tid >= o ? s_p[tid - o] : 0;
    __syncthreads();
    s_h[tid] += ah; s_p[tid] += ap;
    __syncthreads();
  }
  uint32_t oh = s_h[tid] - h, op = s_p[tid] - p;   // exclusive
  for (int i = lo; i < hi; i++) {
    const uint32_t n = L.sizes[g0 + i];
    L.hdr_off[g0 + i] = oh; L.pay_off[g0 + i] = op;
    oh += varint_len(n); op += n;
  }
  if (tid == 255) { L.frame_hdr[f] = 1 + s_h[255]; L.frame_size[f] = 1 + (uint64_t)s_h[255] + s_p[255]; }
}
__global__ void k_ent_frames(EntropyLaunch L) {
  if (threadIdx.x || blockIdx.x) return;
  uint64_t acc = 0;
  for (int f = 0; f < L.nframes; f++) { L.frame_off[f] = acc; acc += L.frame_size[f]; }
  L.frame_off[L.nframes] = acc;
  if (acc > L.out_cap) atomicOr(L.status, 1u);
}
__global__ void __launch_bounds__(64) k_ent_pack(EntropyLaunch L, int tpf, int tile_log2) {
  if (L.status[0]) return;
  const long long g = blockIdx.x;
  const int f = (int)(g / tpf), t = (int)(g % tpf), lane = threadIdx.x;
  uint8_t *base = L.out + L.frame_off[f];
  const uint32_t n = L.sizes[g];
  if (lane == 0) {
    if (t == 0) base[0] = (uint8_t)tile_log2;
    uint8_t *h = base + 1 + L.hdr_off[g];
    uint32_t v = n;
    while (v >= 128) { *h++ = (uint8_t)(v | 128); v >>= 7; }
    *h = (uint8_t)v;
  }
  const uint8_t *src = L.slots + (size_t)g * L.slot_bytes;
  uint8_t *dst = base + L.frame_hdr[f] + L.pay_off[g];
  for (uint32_t i = lane; i < n; i += 64) dst[i] = src[i];
}

}  // namespace

hipError_t entropy_init_tables() { return hipMemcpyToSymbol(HIP_SYMBOL(kInit), kEntropyInit, sizeof(kEntropyInit)); }

hipError_t launch_entropy_code(const EntropyLaunch &L, hipStream_t s) {
  const int tc = (L.w + L.tile - 1) / L.tile, tr = (L.h + L.tile - 1) / L.tile;
  const long long tiles = (long long)tc * tr * L.nframes;
  if (tiles <= 0) return hipSuccess;
  k_ent_code<<<dim3((unsigned)((tiles + 63) / 64)), dim3(64), 0, s>>>(L);
  return hipGetLastError();
}
hipError_t launch_entropy_pack(const EntropyLaunch &L, hipStream_t s) {
  const int tc = (L.w + L.tile - 1) / L.tile, tr = (L.h + L.tile - 1) / L.tile, tpf = tc * tr;
  if (L.nframes <= 0) return hipSuccess;
  int lg = 0;
  while ((1 << lg) < L.tile) lg++;
  k_ent_layout<<<dim3((unsigned)L.nframes), dim3(256), 0, s>>>(L, tpf);
  k_ent_frames<<<dim3(1), dim3(64), 0, s>>>(L);
  k_ent_pack<<<dim3((unsigned)((long long)tpf * L.nframes)), dim3(64), 0, s>>>(L, tpf, lg);
  return hipGetLastError();
}

}  // namespace av1mi
