// entropy_kernels.hip — K9: the tile entropy coder on the GPU (the stage SURVEY.md §8a row H1 leaves on the host cores and
// §7 / §8e name as the scaling risk: "host entropy coding throughput … needs on-GPU compaction").
//
// A range coder is serial in its state, so the parallel axis is the TILE: every entropy tile (32/64/128 luma samples
// square) has its own coder and its own adaptive CDFs (syntax: host/entropy.hpp, oracle/av1o_entropy.c).  The work is
// split where its nature changes:
//
//   k_ent_tokens  PARALLEL part, one thread per 8x8 block: reads the block's levels (+ modes / vectors), walks it in scan
//                 order and writes the block's symbols as 16-bit OPS into the tile's op list (count pass, workgroup scan
//                 for the block's offset, emit pass).  An op is either "adaptive 4-ary symbol s with CDF id" or "n raw
//                 bits of value v" — all syntax knowledge (contexts, chains, escapes, sign packing) ends here.
//   k_ent_code    SERIAL part, one LANE per tile: a wave runs 64 independent range coders in lockstep over their op
//                 lists — one uniform loop, no divergence on the kind of symbol: both interval formulas are evaluated and
//                 selected, the CDF (three 15-bit values + counter = 8 bytes, 61 per lane in LDS) is updated with one
//                 64-bit load and store.  Trip count = the longest op list of the wave (tiles of one frame are within a
//                 few % of each other), not the sum of per-block maxima a block-synchronous design pays.
//                 Bits leave 32 at a time as big-endian words into the tile's private slot (6 bytes per coefficient: a
//                 bound); a carry is resolved when the word leaves and ripples back through the lane's own words.
//   k_ent_layout  per frame: scan of the tile sizes -> header (varint) and payload offsets inside the frame record.
//   k_ent_frames  scan over frames -> record offsets, capacity check.
//   k_ent_pack    per tile: varint + payload copied to its place: the host receives one contiguous stream.
//
// Arithmetic: AV1 spec §8.2.6 interval partition + CDF adaptation (restated in oracle/av1o_entropy.c, byte-exact parity);
// initial CDFs: host/entropy_init.hpp (own constants).  HBM traffic: 2 B per coefficient in, 2 B per op out and in again,
// <1 B per coefficient out — k_ent_code is bound by the serial dependency chain of the coder (one wave per SIMD issues an
// instruction every ~5 cycles), not by bandwidth: what counts is instructions per op and ops per tile.
#include "av1mi_internal.hpp"
#include "../host/entropy_init.hpp"
#include <utility>

namespace av1mi {

namespace {

// CDF ids: same enum as host/entropy.hpp; one CDF = (c0, c1, c2, counter) = 8 bytes
enum { C_TOK = 0, C_GOL = 24, C_EOB_HI = 34, C_EOB_LO = 36, C_MODE_HI = 40, C_MODE_LO = 42, C_SKIP = 50, C_MV_HI = 51, C_MV_LO = 53, C_COUNT = 61 };
static_assert(sizeof(kEntropyInit) == C_COUNT * 8, "entropy_init.hpp layout");

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

// ------------------------------------------------------------------------------------------------ ops
// adaptive: id << 8 | sym (bit 15 clear, id <= 60)
// raw:      0x8000 | nbits << 8 | j, j = 2^nbits - 1 - value = the slot from the bottom (nbits 1..8), 0x4000 set for the top slot
constexpr uint32_t OP_RAW = 0x8000, OP_TOPSLOT = 0x4000;
constexpr int OPS_PER_BLOCK = 800;   // bound: Y 3 + 64 + 64 * 7 + 8, U/V 2 * (3 + 16 + 16 * 7 + 2), header <= 11

struct Emit { uint16_t *p; uint32_t n; };
template <bool WRITE> __device__ inline void put(Emit &E, uint32_t op) { if (WRITE) E.p[E.n] = (uint16_t)op; E.n++; }
template <bool WRITE> __device__ inline void put_sym(Emit &E, int id, int s) { put<WRITE>(E, (uint32_t)(id << 8 | s)); }
// up to 16 raw bits, the most significant chunk of <= 8 first (entropy.hpp "raw bits")
template <bool WRITE> __device__ inline void put_chunk(Emit &E, int n, uint32_t v) {   // n = 1..8 bits
  const uint32_t slots = (1u << n) - 1, j = slots - (v & slots);
  put<WRITE>(E, OP_RAW | (j == slots ? OP_TOPSLOT : 0u) | (uint32_t)n << 8 | j);
}
template <bool WRITE> __device__ inline void put_raw(Emit &E, int nbits, uint32_t v) {
  if (nbits > 8) { put_chunk<WRITE>(E, 8, v >> (nbits - 8)); nbits -= 8; }
  if (nbits > 0) put_chunk<WRITE>(E, nbits, v);
}
template <bool WRITE> __device__ inline void put_pair(Emit &E, int hi_id, int lo_id, int v) { put_sym<WRITE>(E, hi_id, v >> 2); put_sym<WRITE>(E, lo_id + (v >> 2), v & 3); }

// one transform block, levels in scan order at row[0..n)
template <bool WRITE> __device__ inline void block_ops(Emit &E, const int16_t *row, int n, int pt) {
  int eob = 0;
  for (int i = 0; i < n; i++) eob = row[i] != 0 ? i + 1 : eob;
  const int cls = eob <= 2 ? eob : 33 - __clz(eob - 1);
  put_pair<WRITE>(E, C_EOB_HI + pt, C_EOB_LO + pt * 2, cls);
  if (cls >= 3) put_raw<WRITE>(E, cls - 2, (uint32_t)(eob - (1 << (cls - 2)) - 1));
  int prev = 0, nnz = 0;
  uint64_t signs = 0;
  for (int i = 0; i < eob; i++) {
    const int l = row[i], a = l < 0 ? -l : l, t = a < 3 ? a : 3, band = i == 0 ? 0 : i <= 4 ? 1 : i <= 15 ? 2 : 3;
    put_sym<WRITE>(E, C_TOK + (pt * 4 + band) * 3 + prev, t);
    if (a) { signs = (signs << 1) | (uint64_t)(l < 0); nnz++; }
    prev = t < 2 ? t : 2;
  }
  for (int i = 0; i < eob; i++) {
    const int l = row[i], a = l < 0 ? -l : l;
    if (a < 3) continue;
    const uint32_t x = (uint32_t)(a - 2);
    const int k = 31 - __clz((int)x);
    for (int j = 0, rest = k;; j++, rest -= 3) { const int sy = rest < 3 ? rest : 3; put_sym<WRITE>(E, C_GOL + pt * 5 + j, sy); if (sy < 3) break; }
    put_raw<WRITE>(E, k, x & ((1u << k) - 1));
  }
  while (nnz > 0) {
    const int k = nnz > 8 ? 8 : nnz;
    nnz -= k;
    put_chunk<WRITE>(E, k, (uint32_t)(signs >> nnz));
  }
}
template <bool WRITE> __device__ inline void mvd_ops(Emit &E, int comp, int v) {
  const uint32_t a = (uint32_t)(v < 0 ? -v : v);
  int k = a ? 32 - __clz((int)a) : 0;
  k = k > 15 ? 15 : k;
  put_pair<WRITE>(E, C_MV_HI + comp, C_MV_LO + comp * 4, k);
  if (k == 15) put_raw<WRITE>(E, 15, a - 16384);
  else if (k > 1) put_raw<WRITE>(E, k - 1, a & ((1u << (k - 1)) - 1));
  if (a) put_chunk<WRITE>(E, 1, (uint32_t)(v < 0));
}
struct BlockHdr { int key, my, muv, sk, dx, dy; };
template <bool WRITE> __device__ inline void all_ops(Emit &E, const BlockHdr &H, const int16_t *row) {
  bool coded = true;
  if (H.key) {
    put_pair<WRITE>(E, C_MODE_HI, C_MODE_LO, H.my);
    put_pair<WRITE>(E, C_MODE_HI + 1, C_MODE_LO + 4, H.muv);
  } else {
    put_sym<WRITE>(E, C_SKIP, H.sk);
    mvd_ops<WRITE>(E, 0, H.dx);
    mvd_ops<WRITE>(E, 1, H.dy);
    coded = !H.sk;
  }
  if (coded) {
    block_ops<WRITE>(E, row, 64, 0);
    block_ops<WRITE>(E, row + 64, 16, 1);
    block_ops<WRITE>(E, row + 80, 16, 1);
  }
}

// thread = 8x8 block; a workgroup holds TB2 = (tile/8)^2 threads per tile and blockDim / TB2 tiles
__global__ void __launch_bounds__(256) k_ent_tokens(EntropyLaunch L) {
  extern __shared__ uint32_t s_dyn[];        // per thread: 96 levels in scan order (49-dword stride), then the scan array
  int16_t *s_rows = (int16_t *)s_dyn;
  uint32_t *s_cnt = s_dyn + blockDim.x * 49;
  const int tid = threadIdx.x, tb = L.tile >> 3, tb2 = tb * tb, tpg = blockDim.x / tb2;
  const int tc = (L.w + L.tile - 1) / L.tile, tr = (L.h + L.tile - 1) / L.tile, tpf = tc * tr;
  const long long g = (long long)blockIdx.x * tpg + tid / tb2;       // global tile
  const int lb = tid % tb2;
  const bool live = g < (long long)tpf * L.nframes;
  const int f = live ? (int)(g / tpf) : 0, t = live ? (int)(g % tpf) : 0;
  const int bw = L.w >> 3, bh = L.h >> 3;
  const int bx = (t % tc) * tb + lb % tb, by = (t / tc) * tb + lb / tb;
  const bool valid = live && bx < bw && by < bh;
  const size_t b = (size_t)f * bw * bh + (size_t)(valid ? by * bw + bx : 0);
  int16_t *row = s_rows + tid * 98;
  BlockHdr H = { L.key, 0, 0, 0, 0, 0 };
  if (valid) {
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
    if (L.key) {
      H.my = L.modes_y[b]; H.muv = L.modes_uv[b];
      H.my = H.my < 13 ? H.my : 0; H.muv = H.muv < 13 ? H.muv : 0;
    } else {
      H.sk = L.skip[b] != 0;
      const bool left = lb % tb != 0;                // the left neighbour inside the tile predicts the vector
      const int px = left ? L.mvs[(b - 1) * 2] : 0, py = left ? L.mvs[(b - 1) * 2 + 1] : 0;
      H.dx = (int16_t)(L.mvs[b * 2] - px); H.dy = (int16_t)(L.mvs[b * 2 + 1] - py);
    }
  }
  Emit E = { nullptr, 0 };
  if (valid) all_ops<false>(E, H, row);
  const uint32_t mine = E.n;
  s_cnt[tid] = mine;
  __syncthreads();
  for (int o = 1; o < (int)blockDim.x; o <<= 1) {         // inclusive scan over the workgroup
    const uint32_t add = tid >= o ? s_cnt[tid - o] : 0;
    __syncthreads();
    s_cnt[tid] += add;
    __syncthreads();
  }
  const int seg0 = tid - lb;                               // first thread of this tile
  const uint32_t before = seg0 ? s_cnt[seg0 - 1] : 0, off = s_cnt[tid] - mine - before, total = s_cnt[seg0 + tb2 - 1] - before;
  if (live && lb == 0) L.nops[g] = total;
  if (valid) {
    E.p = L.ops + (size_t)g * L.ops_per_tile + off; E.n = 0;
    all_ops<true>(E, H, row);
  }
}

// ------------------------------------------------------------------------------------------------ coder
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
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pk(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ inline u16x2 unpk(uint32_t v) { return __builtin_bit_cast(u16x2, v); }

// one op: both interval formulas are evaluated and selected (no divergence on the kind of op).
// The lane's CDFs live in LDS in INVERSE form, x = i0 | i1 << 16, y = i2 | count << 16 with i_k = 32768 - c_k (what the
// partition formula consumes): the pair (f of symbol s - 1, f of symbol s) is one 64-bit shift of (0, i0, i1, i2, 0) by
// 16 s bits, the zeros supplying "32768 - 32768" for the last symbol, and the update is packed 16-bit arithmetic.
__device__ inline uint2 step(Enc &e, const uint2 v, uint32_t op) {   // returns the adapted CDF (meaningful for adaptive ops)
  const bool raw = (op & OP_RAW) != 0;
  const uint32_t s = op & 3;
  const uint64_t w64 = ((uint64_t)((v.x >> 16) | (v.y << 16)) << 32 | (uint64_t)(v.x << 16)) >> (s * 16);
  const uint32_t t = (uint32_t)w64, fp = t & 0xFFFF, fs = t >> 16;        // f(s - 1) (0 for s = 0), f(s) (0 for s = 3)
  const uint32_t r8 = e.rng >> 8;
  const uint32_t bot = (__umul24(r8, fs >> 6) >> 1) + 4u * (3 - s);
  const uint32_t topc = (__umul24(r8, fp >> 6) >> 1) + 4u * (4 - s);
  const uint32_t top = s == 0 ? e.rng : topc;
  // raw: nb bits, j = slot counted from the bottom (precomputed by the tokenizer), bit 14 = top slot
  const uint32_t nb = (op >> 8) & 15, r = e.rng >> nb, radd = __umul24(r, op & 0xFF);
  const uint32_t rrng = (op & OP_TOPSLOT) ? e.rng - radd : r;
  const uint32_t add = raw ? radd : bot, nrng = raw ? rrng : top - bot;
  // interval [low + add, low + add + nrng) becomes the state; renormalise to a 16-bit range
  const int d = __clz((int)nrng) - 16;
  e.low = (e.low + add) << d;      // <= 17 + 31 + 15 bits
  e.rng = nrng << d;
  e.pend += d;
  // flush of 32 finished bits as straight-line code: selects instead of a branch (in a wave of 64 coders some lane flushes at
  // almost every step, so the branch would be taken anyway), only the store itself and the rare ripple are predicated
  const bool fl = e.pend >= 32;
  e.pend = fl ? e.pend - 32 : e.pend;
  const uint64_t w = e.low >> (16 + e.pend);
  e.low = fl ? e.low & (((uint64_t)1 << (16 + e.pend)) - 1) : e.low;
  if (fl && (w >> 32)) ripple(e);
  if (fl && e.n < e.cap) e.out[e.n] = __builtin_bswap32((uint32_t)w);
  e.n += fl;
  // adaptation in inverse form: i_k -= i_k >> rate for k >= s (c_k moves up), i_k += (32768 - i_k) >> rate for k < s
  const uint32_t cnt = v.y >> 16;
  const unsigned short rate = (unsigned short)(5 + (cnt > 15) + (cnt > 31));
  const u16x2 rr = { rate, rate }, k15 = { 32768, 32768 };
  const u16x2 x = unpk(v.x), y = unpk(v.y & 0xFFFF);
  const u16x2 xd = x - (x >> rr), xu = x + ((k15 - x) >> rr);
  const u16x2 yd = y - (y >> rr), yu = y + ((k15 - y) >> rr);   // high half of y is 0 here: the counter is put back below
  const uint32_t mx = s == 0 ? 0xFFFFFFFFu : s == 1 ? 0xFFFF0000u : 0u;   // halves with k >= s take the "down" form
  const uint32_t nx = (pk(xd) & mx) | (pk(xu) & ~mx);
  const uint32_t ny = (s == 3 ? pk(yu) : pk(yd)) & 0xFFFF;
  return make_uint2(nx, ny | ((cnt + (cnt < 32)) << 16));
}
__device__ inline int wave_max(int v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) { const int w = __shfl_xor(v, o); v = w > v ? w : v; }
  return __builtin_amdgcn_readfirstlane(v);
}

__global__ void __launch_bounds__(64) k_ent_code(EntropyLaunch L) {
  __shared__ uint2 s_models[64 * C_COUNT];      // lane rows of 61 x 8 bytes (odd multiple of 8)
  const int lane = threadIdx.x;
  const long long g = (long long)blockIdx.x * 64 + lane;
  const int tc = (L.w + L.tile - 1) / L.tile, tr = (L.h + L.tile - 1) / L.tile;
  const bool live = g < (long long)tc * tr * L.nframes;
  uint2 *m = s_models + lane * C_COUNT;
  for (int i = 0; i < C_COUNT; i++) { const uint2 c = kInit[i]; m[i] = make_uint2(0x80008000u - c.x, 0x8000u - (c.y & 0xFFFF)); }   // inverse form, counter 0
  Enc e;
  e.low = 0; e.rng = 0x8000; e.pend = 0; e.n = 0;
  e.cap = live ? L.slot_bytes / 4 : 0; e.out = (uint32_t *)(L.slots + (live ? (size_t)g * L.slot_bytes : 0));
  const int nops = live ? (int)L.nops[g] : 0;
  const uint4 *src = (const uint4 *)(L.ops + (live ? (size_t)g * L.ops_per_tile : 0));
  const int last = (int)(L.ops_per_tile >> 3) - 1;          // lists are padded: any index <= last is in bounds
  const int nmax = wave_max(nops);
  uint4 cur = src[0];
  for (int base = 0; base < nmax; base += 8) {
    const int nq = (base >> 3) + 1;
    const uint4 nxt = src[nq < last ? nq : last];            // next 8 ops, in flight while these are coded
    const uint32_t wd[4] = { cur.x, cur.y, cur.z, cur.w };
    // lanes past the end of their list code a no-op (zero raw bits leave the coder state as it is): no branch per op.
    // The CDF of op k + 1 is requested BEFORE op k's arithmetic (LDS keeps program order, so it returns the value from
    // before op k's write-back); if it is the CDF op k adapts, the adapted value is forwarded instead.
    auto op_at = [&](int k) -> uint32_t { return base + k < nops ? (wd[k >> 1] >> ((k & 1) * 16)) & 0xFFFF : OP_RAW; };
    uint32_t op = op_at(0), id = (op & OP_RAW) ? 0u : op >> 8;     // raw ops read CDF 0 and write it back unchanged
    uint2 v = m[id];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t opn = k < 7 ? op_at(k + 1) : OP_RAW, idn = (opn & OP_RAW) ? 0u : opn >> 8;
      uint2 early = v;
      if (k < 7) early = m[idn];
      const bool raw = (op & OP_RAW) != 0;
      const uint2 nv = step(e, v, op);
      const uint2 wv = make_uint2(raw ? v.x : nv.x, raw ? v.y : nv.y);
      m[id] = wv;
      v = idn == id ? wv : early;
      op = opn; id = idn;
    }
    cur = nxt;
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
  // a tile that overran its slot (status bit 1 of k_ent_code: nothing is packed then) is reported through the same word the host
  // already checks against its capacity: a total no buffer can hold
  L.frame_off[L.nframes] = (L.status[0] & 2u) ? ~0ull : acc;
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

size_t entropy_ops_per_tile(int tile) { const size_t tb2 = (size_t)(tile / 8) * (tile / 8); return (tb2 * OPS_PER_BLOCK + 16 + 7) & ~(size_t)7; }

hipError_t launch_entropy_tokens(const EntropyLaunch &L, hipStream_t s) {
  const int tc = (L.w + L.tile - 1) / L.tile, tr = (L.h + L.tile - 1) / L.tile;
  const long long tiles = (long long)tc * tr * L.nframes;
  if (tiles <= 0) return hipSuccess;
  const int tb2 = (L.tile / 8) * (L.tile / 8), threads = tb2 < 64 ? 64 : tb2, tpg = threads / tb2;
  k_ent_tokens<<<dim3((unsigned)((tiles + tpg - 1) / tpg)), dim3(threads), (size_t)threads * 50 * 4, s>>>(L);
  return hipGetLastError();
}
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
