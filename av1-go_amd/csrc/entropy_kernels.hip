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
//   k_ent_code    lane = tile.  Models: 237 u16 words per lane in LDS ([word][lane]: conflict-free for equal words).
//                 The current block's 96 levels + its escape list sit in a lane-private LDS row (odd dword stride).
//                 Bytes leave through a one-byte hold + 0xFF run counter (carries never touch written bytes) into the
//                 tile's private slot of 6 bytes per coefficient (a bound: <= 43 bits per coefficient + block headers).
//   k_ent_layout  per frame: scan of the tile sizes -> header (varint) and payload offsets inside the frame record.
//   k_ent_frames  scan over frames -> record offsets, capacity check.
//   k_ent_pack    per tile: varint + payload copied to its place: the host receives one contiguous stream.
//
// Arithmetic: AV1 spec §8.2.6 interval partition + CDF adaptation (restated in oracle/av1o_entropy.c, byte-exact parity);
// initial CDFs: host/entropy_init.hpp (own constants).  HBM traffic: 2 B per coefficient in, <1 B out — the kernel is
// bound by the serial dependency chain of the coder (issue latency of one wave per SIMD), not by bandwidth.
#include "av1mi_internal.hpp"
#include "../host/entropy_init.hpp"

namespace av1mi {

namespace {

enum { M_EOB = 0, M_TOK = 18, M_GOL = 138, M_MODE = 172, M_SKIP = 200, M_MVC = 203, M_WORDS = 237 };
static_assert(sizeof(kEntropyInit) == M_WORDS * 2, "entropy_init.hpp layout");
constexpr int ROW = 162;   // int16 per lane row: 96 levels (Y 64, U 16, V 16) + 64 escapes + 2 pad = 81 dwords (odd)

__constant__ uint16_t kInit[M_WORDS];   // filled once per context from kEntropyInit
__constant__ uint8_t kScan8[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
__constant__ uint8_t kScan4[16] = { 0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15 };

struct Enc {
  uint32_t low, rng;   // low holds 16 + pend bits
  int pend, held, ff;  // held-back byte (-1: none) followed by ff bytes of 0xFF
  uint8_t *out; uint32_t n, cap;
};
__device__ inline void raw_out(Enc &e, int b) { if (e.n < e.cap) e.out[e.n] = (uint8_t)b; e.n++; }
__device__ inline void byte_out(Enc &e, int b) {
  if (b == 0xFF && e.held >= 0) { e.ff++; return; }
  if (e.held >= 0) raw_out(e, e.held);
  for (; e.ff; e.ff--) raw_out(e, 0xFF);
  e.held = b;
}
__device__ inline void carry(Enc &e) {
  if (e.ff) { raw_out(e, e.held + 1); for (; e.ff > 1; e.ff--) raw_out(e, 0); e.ff = 0; e.held = 0; }
  else e.held += 1;
}
// interval [low + add, low + add + nrng) becomes the state; renormalise to a 16-bit range
__device__ inline void commit(Enc &e, uint32_t add, uint32_t nrng) {
  e.low += add;
  const int lim = 16 + e.pend;
  if (e.low >> lim) { carry(e); e.low &= (1u << lim) - 1; }
  const int d = __clz((int)nrng) - 16;
  e.rng = nrng << d;
  uint64_t wide = (uint64_t)e.low << d;
  e.pend += d;
  while (e.pend >= 8) {
    e.pend -= 8;
    byte_out(e, (int)((wide >> (16 + e.pend)) & 0xFF));
    wide &= ((uint64_t)1 << (16 + e.pend)) - 1;
  }
  e.low = (uint32_t)wide;
}
// adaptive symbol s of an N-ary alphabet whose CDF starts at model word `off` of this lane (m = &models[lane])
template <int N> __device__ inline void enc_sym(Enc &e, uint16_t *m, int off, int s) {
  uint32_t c[N - 1];
#pragma unroll
  for (int i = 0; i < N - 1; i++) c[i] = m[(off + i) * 64];
  const uint32_t cnt = m[(off + N) * 64];
  uint32_t cs = 32768, cp = 0;
#pragma unroll
  for (int i = 0; i < N - 1; i++) { cs = s == i ? c[i] : cs; cp = s == i + 1 ? c[i] : cp; }
  const uint32_t r8 = e.rng >> 8;
  const uint32_t bot = ((r8 * ((32768u - cs) >> 6)) >> 1) + 4u * (uint32_t)(N - 1 - s);
  const uint32_t top = s ? ((r8 * ((32768u - cp) >> 6)) >> 1) + 4u * (uint32_t)(N - s) : e.rng;
  commit(e, bot, top - bot);
  const int rate = 3 + (cnt > 15) + (cnt > 31) + (N >= 4 ? 2 : 1);
#pragma unroll
  for (int i = 0; i < N - 1; i++) {
    const uint32_t v = c[i];
    m[(off + i) * 64] = (uint16_t)(i >= s ? v + ((32768u - v) >> rate) : v - (v >> rate));
  }
  if (cnt < 32) m[(off + N) * 64] = (uint16_t)(cnt + 1);
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
template <int NCOEF> __device__ inline void code_block(Enc &e, uint16_t *m, int16_t *row, int base, int pt, bool coded) {
  const uint8_t *scan = NCOEF == 64 ? kScan8 : kScan4;
  int eob = 0;
#pragma unroll 8
  for (int i = 0; i < NCOEF; i++) eob = row[base + scan[i]] != 0 ? i + 1 : eob;
  if (!coded) eob = 0;
  if (coded) {
    const int cls = eob <= 2 ? eob : 33 - __clz(eob - 1);
    enc_sym<8>(e, m, M_EOB + pt * 9, cls);
    const int xb = cls >= 3 ? cls - 2 : 0;
    enc_raw16(e, cls >= 3, xb, (uint32_t)(eob - (1 << xb) - 1));
  }
  const int emax = wave_max(eob);
  int prev = 0, nnz = 0, nesc = 0;
  uint64_t signs = 0;
  for (int i = 0; i < emax; i++) {          // uniform trip count: scan position and band are scalar
    const int pos = scan[i], band = i == 0 ? 0 : i <= 4 ? 1 : i <= 15 ? 2 : 3;
    if (i < eob) {
      const int l = row[base + pos], a = l < 0 ? -l : l, t = a < 3 ? a : 3;
      enc_sym<4>(e, m, M_TOK + ((pt * 4 + band) * 3 + prev) * 5, t);
      if (a) { signs = (signs << 1) | (uint64_t)(l < 0); nnz++; }
      if (a >= 3) row[96 + nesc++] = (int16_t)(a - 2);           // 1..32766
      prev = t < 2 ? t : 2;
    }
  }
  const int xmax = wave_max(nesc);
  for (int j = 0; j < xmax; j++) {
    const bool on = j < nesc;
    uint32_t x = 1; int k = 0;
    if (on) {
      x = (uint32_t)row[96 + j];
      k = 31 - __clz((int)x);
      enc_sym<16>(e, m, M_GOL + pt * 17, k);
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
__device__ inline void code_mvd(Enc &e, uint16_t *m, int off, bool on, int v) {
  const uint32_t a = (uint32_t)(v < 0 ? -v : v);
  int k = a ? 32 - __clz((int)a) : 0;
  k = k > 15 ? 15 : k;
  if (on) enc_sym<16>(e, m, off, k);
  enc_raw16(e, on && k > 1, k == 15 ? 15 : k - 1, k == 15 ? a - 16384 : a & ((1u << (k > 0 ? k - 1 : 0)) - 1));
  if (on && a) enc_raw(e, 1, v < 0);
}

__global__ void __launch_bounds__(64) k_ent_code(EntropyLaunch L) {
  __shared__ uint16_t s_models[M_WORDS * 64];
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
  uint16_t *m = s_models + lane;
  int16_t *row = s_rows + lane * ROW;
  for (int w = 0; w < M_WORDS; w++) m[w * 64] = kInit[w];
  Enc e;
  e.low = 0; e.rng = 0x8000; e.pend = 0; e.held = -1; e.ff = 0; e.n = 0;
  e.cap = live ? (uint32_t)L.slot_bytes : 0; e.out = L.slots + (live ? (size_t)g * L.slot_bytes : 0);
  for (int lby = 0; lby < tb; lby++)
    for (int lbx = 0; lbx < tb; lbx++) {   // uniform over the wave; ragged tiles mask lanes off
      const int bx = bx0 + lbx, by = by0 + lby;
      const bool valid = bx < bx1 && by < by1;
      const size_t b = fb + (size_t)(valid ? by * bw + bx : 0);
      bool coded = valid;
      if (valid) {
        uint32_t *dst = (uint32_t *)row;
        const uint4 *sy = (const uint4 *)(L.lev[0] + b * 64), *su = (const uint4 *)(L.lev[1] + b * 16), *sv = (const uint4 *)(L.lev[2] + b * 16);
#pragma unroll
        for (int k = 0; k < 8; k++) { const uint4 v = sy[k]; dst[k * 4] = v.x; dst[k * 4 + 1] = v.y; dst[k * 4 + 2] = v.z; dst[k * 4 + 3] = v.w; }
#pragma unroll
        for (int k = 0; k < 2; k++) { const uint4 v = su[k]; dst[32 + k * 4] = v.x; dst[33 + k * 4] = v.y; dst[34 + k * 4] = v.z; dst[35 + k * 4] = v.w; }
#pragma unroll
        for (int k = 0; k < 2; k++) { const uint4 v = sv[k]; dst[40 + k * 4] = v.x; dst[41 + k * 4] = v.y; dst[42 + k * 4] = v.z; dst[43 + k * 4] = v.w; }
      }
      if (L.key) {
        int my = 0, muv = 0;
        if (valid) { my = L.modes_y[b]; muv = L.modes_uv[b]; my = my < 13 ? my : 0; muv = muv < 13 ? muv : 0; }
        if (valid) enc_sym<13>(e, m, M_MODE, my);
        if (valid) enc_sym<13>(e, m, M_MODE + 14, muv);
      } else {
        int sk = 0, dx = 0, dy = 0;
        if (valid) {
          sk = L.skip[b] != 0;
          const int px = lbx ? L.mvs[(b - 1) * 2] : 0, py = lbx ? L.mvs[(b - 1) * 2 + 1] : 0;
          dx = (int16_t)(L.mvs[b * 2] - px); dy = (int16_t)(L.mvs[b * 2 + 1] - py);
          enc_sym<2>(e, m, M_SKIP, sk);
        }
        code_mvd(e, m, M_MVC, valid, dx);
        code_mvd(e, m, M_MVC + 17, valid, dy);
        coded = valid && !sk;
      }
      code_block<64>(e, m, row, 0, 0, coded);
      code_block<16>(e, m, row, 64, 1, coded);
      code_block<16>(e, m, row, 80, 1, coded);
    }
  if (live) {
    if (e.held >= 0) raw_out(e, e.held);
    for (; e.ff; e.ff--) raw_out(e, 0xFF);
    int bits = 16 + e.pend;
    while (bits > 0) {
      const int take = bits >= 8 ? 8 : bits;
      raw_out(e, (int)(((e.low >> (bits - take)) << (8 - take)) & 0xFF));
      bits -= take;
    }
    L.sizes[g] = e.n;     // > slot_bytes would mean the (proven) bound failed: k_ent_frames turns it into an error
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
