// cdef_kernel.hip — SURVEY.md §8a row K6: CDEF of 4:2:0 frames, one 64x64 luma superblock (+ its 32x32 U and V
// blocks) per workgroup, frames of a segment along blockIdx.z.
//
// The workgroup stages the deblocked luma block with a 2-sample halo (and both chroma blocks likewise) in LDS as
// uint16, samples outside the picture stored as 0xFFFF = "not available" (taps skip them, as the spec's
// CdefAvailable: in the packed filter a marked tap is replaced by the centre sample).  Wave 0 then runs the direction search, one 8x8 block per lane with every bin index a
// compile-time constant; afterwards all 256 lanes filter: 16 luma + 8 chroma samples each, 12 taps per sample
// read from LDS.  Output goes to separate planes, so taps never see filtered samples and superblocks are
// independent.  HBM traffic: b*S read + b*S written (= 2b*S of SURVEY.md §8d); halo re-reads hit L2.
//
// Restates AV1 spec §7.15 and libaom cdef_find_dir_c / cdef_filter_block_c / constrain(); the reference has no
// counterpart (internal/ffmpeg/transcode.go:120).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "av1mi_internal.hpp"

namespace av1mi {

__device__ constexpr int8_t kCdefDir[8][2][2] = {
  { { -1, 1 }, { -2, 2 } }, { { 0, 1 }, { -1, 2 } }, { { 0, 1 }, { 0, 2 } }, { { 0, 1 }, { 1, 2 } },
  { { 1, 1 }, { 2, 2 } },   { { 1, 0 }, { 2, 1 } },  { { 1, 0 }, { 2, 0 } }, { { 1, 0 }, { 2, -1 } } };

__device__ __forceinline__ int msb(unsigned v) { return 31 - __clz(v); }
// Four horizontally adjacent samples at once with packed 16-bit VALU (v_pk_*_i16: two samples per lane-op).  SENT = the
// tile may hold 0xFFFF "outside the picture" marks (picture-border superblocks): two more ops per tap pair.
// p: 4-byte aligned centre pointer (first of the four samples) in an LDS tile with even row stride LS.
typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));
// constrain() on two samples: clamp(diff, -t, t) with t = max(0, thr - (|diff| >> shift)) — the same value as
// sign(diff) * min(|diff|, t) (spec 7.15.2), written so that it is a saturating subtract and a min/max pair
__device__ __forceinline__ v2s pk_constrain(v2s diff, v2u thr, v2u shift) {
  const v2u mag = __builtin_bit_cast(v2u, __builtin_elementwise_max(diff, -diff));
  const v2s t = __builtin_bit_cast(v2s, __builtin_elementwise_sub_sat(thr, mag >> shift));
  return __builtin_elementwise_max(__builtin_elementwise_min(diff, t), -t);
}
// Tap table of one direction in a tile of row stride LS, ready to use: for each of the six tap positions j (primary k = 0, 1;
// secondary dir + 2, k = 0, 1; secondary dir + 6, k = 0, 1) three int32: the BYTE offset of the aligned dword pair that holds
// the four samples at +offset, the same for the mirrored tap at -offset, and the funnel shift (16 when the offset is odd).
// A quad reads its direction's 18 entries with five LDS loads and adds them to its centre address; derived per quad from
// int16 sample offsets this was ~45 of the ~450 instructions of a quad.
constexpr int kTapEntries = 20;   // 18 used, row padded to a multiple of 4 dwords
template <int LS> __device__ __forceinline__ void cdef_fill_offsets(int32_t *tab, int i) {   // i in [0, 8 * 6)
  const int dir = i / 6, j = i - dir * 6;
  const int d = j < 2 ? dir : j < 4 ? (dir + 2) & 7 : (dir + 6) & 7, k = j & 1;
  const int o = kCdefDir[d][k][0] * LS + kCdefDir[d][k][1], odd = o & 1;
  tab[dir * kTapEntries + 3 * j] = 2 * (o - odd);
  tab[dir * kTapEntries + 3 * j + 1] = 2 * (-o - odd);
  tab[dir * kTapEntries + 3 * j + 2] = odd * 16;
}
// pshift / sshift: max(0, damping - msb(strength)) (0 for strength 0); pt0, pt1: the primary tap weights (4, 2) or (3, 3)
template <bool SENT>
__device__ __forceinline__ void cdef_quad_packed(const uint16_t *p, const int32_t *otab, int pri, int sec, int pshift, int sshift, int pt0, int pt1, v2s *out) {
  const uint32_t *c32 = reinterpret_cast<const uint32_t *>(p);
  const v2s x0 = __builtin_bit_cast(v2s, c32[0]), x1 = __builtin_bit_cast(v2s, c32[1]);
  v2s s0 = { 0, 0 }, s1 = { 0, 0 }, mx0 = x0, mx1 = x1, mn0 = x0, mn1 = x1;
  int ot[kTapEntries];
#pragma unroll
  for (int i = 0; i < kTapEntries; i += 4) {
    const int4 v = *reinterpret_cast<const int4 *>(otab + i);
    ot[i] = v.x; ot[i + 1] = v.y; ot[i + 2] = v.z; ot[i + 3] = v.w;
  }
  // one tap position and its mirror image share weight and strength: constrain both, add, one multiply-add per pair
  auto taps = [&](int j, int thr, int shift, int w) {
    const v2u th = { (unsigned short)thr, (unsigned short)thr }, sh = { (unsigned short)shift, (unsigned short)shift };
    v2s c0 = { 0, 0 }, c1 = { 0, 0 };
#pragma unroll
    for (int sg = 0; sg < 2; sg++) {
      // the four samples at p +- offset .. + 3 as two packed pairs; the offset may be odd: funnel-shift three aligned dwords
      const uint32_t *q = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(p) + ot[3 * j + sg]);
      const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
      v2s a0 = __builtin_bit_cast(v2s, __builtin_amdgcn_alignbit(d1, d0, ot[3 * j + 2]));
      v2s a1 = __builtin_bit_cast(v2s, __builtin_amdgcn_alignbit(d2, d1, ot[3 * j + 2]));
      if constexpr (SENT) {
        // picture-border superblocks: 0xFFFF marks a sample outside the picture (CdefAvailable = 0).  Valid samples are
        // < 2^15, so the sign bit is the mark; a marked tap is replaced by the centre sample: difference 0, and it cannot
        // move the min/max clamp — exactly "skip the tap"
        // (the sign splat goes through inline asm: written as a0 >> 15 the compiler recognises a per-element select and emits
        // a compare + v_cndmask per HALF, six instructions per pair instead of shift + v_bfi)
        uint32_t m0, m1;
        asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(m0) : "v"(__builtin_bit_cast(uint32_t, a0)));
        asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(m1) : "v"(__builtin_bit_cast(uint32_t, a1)));
        a0 = __builtin_bit_cast(v2s, (__builtin_bit_cast(uint32_t, x0) & m0) | (__builtin_bit_cast(uint32_t, a0) & ~m0));
        a1 = __builtin_bit_cast(v2s, (__builtin_bit_cast(uint32_t, x1) & m1) | (__builtin_bit_cast(uint32_t, a1) & ~m1));
      }
      c0 += pk_constrain(a0 - x0, th, sh); c1 += pk_constrain(a1 - x1, th, sh);
      mx0 = __builtin_elementwise_max(mx0, a0); mx1 = __builtin_elementwise_max(mx1, a1);
      mn0 = __builtin_elementwise_min(mn0, a0); mn1 = __builtin_elementwise_min(mn1, a1);
    }
    const v2s ww = { (short)w, (short)w };
    s0 += ww * c0; s1 += ww * c1;
  };
  // secondary strength 0 (uniform in a superblock; the policy's value for inter frames at mid quantisers): its eight taps
  // contribute nothing to the sum, and without them the result lies between the centre and a primary tap, so that they do not
  // take part in the min / max clamp changes nothing either (libaom's cdef_filter_8_1 drops the clamp altogether)
  if (sec) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      taps(k, pri, pshift, k ? pt1 : pt0);
      taps(2 + k, sec, sshift, k ? 1 : 2);
      taps(4 + k, sec, sshift, k ? 1 : 2);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 2; k++) taps(k, pri, pshift, k ? pt1 : pt0);
  }
  const v2s eight = { 8, 8 }, four = { 4, 4 }, fifteen = { 15, 15 };
  const v2s y0 = x0 + ((s0 + (s0 >> fifteen) + eight) >> four), y1 = x1 + ((s1 + (s1 >> fifteen) + eight) >> four);
  out[0] = __builtin_elementwise_min(__builtin_elementwise_max(y0, mn0), mx0);
  out[1] = __builtin_elementwise_min(__builtin_elementwise_max(y1, mn1), mx1);
}

template <typename Pix>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_cdef(CdefLaunch L) {
  constexpr int YS = 64 + 4 + 2, CSZ = 32 + 4 + 2;   // LDS row strides (halo 2 each side, +2 pad)
  __shared__ __attribute__((aligned(16))) uint16_t ty[(64 + 4) * YS + 8];
  __shared__ __attribute__((aligned(16))) uint16_t tc[2][(32 + 4) * CSZ + 8];
  __shared__ uint8_t bskip[64];   // skip flag of each 8x8 block of the superblock (1 also for blocks outside the picture)
  // per 8x8 block, from the direction search: luma primary strength after the variance adjustment (bits 0-7), its damping
  // shift (8-11), the luma filter direction (12-14), the primary tap parity (15), the block's direction itself (16-18: chroma)
  __shared__ uint32_t bpar[64];
  __shared__ __attribute__((aligned(16))) int32_t offy[8 * kTapEntries], offc[8 * kTapEntries];   // tap tables for the two tile strides
  const int tid = threadIdx.x;
  if (tid < 48) cdef_fill_offsets<YS>(offy, tid); else if (tid >= 64 && tid < 112) cdef_fill_offsets<CSZ>(offc, tid - 64);
  const Tile3 tl = xcd_tile((L.w + 63) / 64, (L.h + 63) / 64, L.nframes);
  const int sbx = tl.x, sby = tl.y, f = tl.z;
  constexpr int bd = sizeof(Pix) == 1 ? 8 : 10, cs = bd - 8;   // the launch picks the instantiation by L.bd (8 or 10)
  const Pix *sy = reinterpret_cast<const Pix *>(L.src[0]) + (size_t)f * L.h * L.stride_y;
  Pix *dy = reinterpret_cast<Pix *>(L.dst[0]) + (size_t)f * L.h * L.stride_y;
  const int cw = L.w / 2, chh = L.h / 2;
  // the superblock's strength set, requested before the tile loads so that its latency hides behind them (one aligned dword)
  const int sbw = (L.w + 63) / 64;
  const uint32_t st32 = *reinterpret_cast<const uint32_t *>(L.sb_strength + ((size_t)f * L.sb_frame_stride + (size_t)sby * sbw + sbx) * 4);
  const int st[4] = { (int)(st32 & 255), (int)((st32 >> 8) & 255), (int)((st32 >> 16) & 255), (int)(st32 >> 24) };
  // the 64 skip flags once, next to the tiles: the filter loops read one per quad, and as a global load each of those sat
  // in front of a branch with its whole latency exposed
  if (tid >= 192) {
    const int b = tid - 192, fy8 = sby * 8 + (b >> 3), fx8 = sbx * 8 + (b & 7);
    bskip[b] = (fy8 < (L.h >> 3) && fx8 < (L.w >> 3)) ? L.skip8[(size_t)f * L.skip_frame_stride + (size_t)fy8 * (L.w >> 3) + fx8] : (uint8_t)1;
  }
  // every tap of this superblock inside the picture?  (then the tile holds no 0xFFFF mark: vector staging, no mark handling)
  const bool interior = sbx > 0 && sby > 0 && sbx * 64 + 66 <= L.w && sby * 64 + 66 <= L.h;
  // stage luma 68x68 and chroma 36x36 x2 (local (0,0) = picture (sb*64-2, sb*64-2))
  if (interior) {
    // no bounds to check: 4-sample aligned loads starting 4 samples left of the block (18 per row luma, 10 chroma); the tile
    // keeps only the 2-sample halo, so local column = aligned column - 2.  The four samples are two uint16 pairs = two dword
    // stores: samples 0,1 go to tile columns 4g - 2, 4g - 1 and samples 2,3 to 4g, 4g + 1 (even columns, even row stride);
    // the first pair of a row and the last lie outside.  All of a lane's loads (5 luma + 3 chroma items) are issued before
    // its first LDS store, so their latencies overlap instead of adding up.
    constexpr int NY = 68 * 18, NC = 2 * 36 * 10, KY = (NY + 255) / 256, KC = (NC + 255) / 256;
    uint2 vy[KY], vc[KC];
#pragma unroll
    for (int k = 0; k < KY; k++) {
      const int i = tid + 256 * k;
      if (i < NY) {
        const int r = i / 18, g = i - r * 18;
        const Pix *q = sy + row_off(sby * 64 - 2 + r, L.stride_y) + sbx * 64 - 4 + g * 4;
        if constexpr (sizeof(Pix) == 1) vy[k].x = *reinterpret_cast<const uint32_t *>(q);
        else vy[k] = *reinterpret_cast<const uint2 *>(q);
      }
    }
#pragma unroll
    for (int k = 0; k < KC; k++) {
      const int i = tid + 256 * k;
      if (i < NC) {
        const int pl = i / 360, j = i - pl * 360, r = j / 10, g = j - r * 10;
        const Pix *q = reinterpret_cast<const Pix *>(L.src[1 + pl]) + (size_t)f * chh * L.stride_uv + row_off(sby * 32 - 2 + r, L.stride_uv) + sbx * 32 - 4 + g * 4;
        if constexpr (sizeof(Pix) == 1) vc[k].x = *reinterpret_cast<const uint32_t *>(q);
        else vc[k] = *reinterpret_cast<const uint2 *>(q);
      }
    }
    auto pairs = [](uint2 v, uint32_t &lo, uint32_t &hi) {
      if constexpr (sizeof(Pix) == 1) { lo = __builtin_amdgcn_perm(0u, v.x, 0x0c010c00u); hi = __builtin_amdgcn_perm(0u, v.x, 0x0c030c02u); }
      else { lo = v.x; hi = v.y; }
    };
#pragma unroll
    for (int k = 0; k < KY; k++) {
      const int i = tid + 256 * k;
      if (i < NY) {
        const int r = i / 18, g = i - r * 18;
        uint32_t lo, hi;
        pairs(vy[k], lo, hi);
        uint32_t *d = reinterpret_cast<uint32_t *>(ty + r * YS + g * 4 - 2);
        if (g > 0) d[0] = lo;
        if (g < 17) d[1] = hi;
      }
    }
#pragma unroll
    for (int k = 0; k < KC; k++) {
      const int i = tid + 256 * k;
      if (i < NC) {
        const int pl = i / 360, j = i - pl * 360, r = j / 10, g = j - r * 10;
        uint32_t lo, hi;
        pairs(vc[k], lo, hi);
        uint32_t *d = reinterpret_cast<uint32_t *>(tc[pl] + r * CSZ + g * 4 - 2);
        if (g > 0) d[0] = lo;
        if (g < 9) d[1] = hi;
      }
    }
  } else {
    for (int i = tid; i < 68 * 68; i += 256) {
      const int r = i / 68, c = i - r * 68;
      const int fy = sby * 64 - 2 + r, fx = sbx * 64 - 2 + c;
      ty[r * YS + c] = (fy >= 0 && fy < L.h && fx >= 0 && fx < L.w) ? (uint16_t)sy[row_off(fy, L.stride_y) + fx] : (uint16_t)0xFFFF;
    }
    for (int i = tid; i < 2 * 36 * 36; i += 256) {
      const int pl = i / (36 * 36), j = i - pl * 36 * 36, r = j / 36, c = j - r * 36;
      const int fy = sby * 32 - 2 + r, fx = sbx * 32 - 2 + c;
      const Pix *sc = reinterpret_cast<const Pix *>(L.src[1 + pl]) + (size_t)f * chh * L.stride_uv;
      tc[pl][r * CSZ + c] = (fy >= 0 && fy < chh && fx >= 0 && fx < cw) ? (uint16_t)sc[row_off(fy, L.stride_uv) + fx] : (uint16_t)0xFFFF;
    }
  }
  __syncthreads();
  const bool enabled = st[0] != 255;
  // strengths of the superblock (uniform) and what follows from them alone
  const int ypri0 = st[0] << cs, ysec = (st[1] == 3 ? 4 : st[1]) << cs;
  const int upri = st[2] << cs, usec = (st[3] == 3 ? 4 : st[3]) << cs;
  const int dampy = L.damping + cs, dampc = L.damping + cs - 1;
  // direction search: lane b of wave 0 owns 8x8 block b (raster within the superblock)
  if (tid < 64 && enabled) {
    const int by = tid >> 3, bx = tid & 7;
    if (sby * 64 + by * 8 < L.h && sbx * 64 + bx * 8 < L.w) {
      const uint16_t *p = ty + (2 + by * 8) * YS + 2 + bx * 8;
      int partial[8][15];
#pragma unroll
      for (int d = 0; d < 8; d++)
#pragma unroll
        for (int k = 0; k < 15; k++) partial[d][k] = 0;
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int x = (p[i * YS + j] >> cs) - 128;
          partial[0][i + j] += x; partial[1][i + j / 2] += x; partial[2][i] += x; partial[3][3 + i - j / 2] += x;
          partial[4][7 + i - j] += x; partial[5][3 - i / 2 + j] += x; partial[6][j] += x; partial[7][i / 2 + j] += x;
        }
      // squares of sums of at most 8 values in [-128, 127] (<= 2^20) times weights <= 840: 24-bit multiplies (a 32-bit integer
      // multiply is four passes)
      constexpr int div_table[9] = { 0, 840, 420, 280, 210, 168, 140, 120, 105 };
      // (pinned by inline asm: the compiler, which can bound the sums, turned __mul24(v, v) back into v_mul_lo_u32 / v_mad_u64_u32)
      auto sq = [](int v) { int d; asm("v_mul_i32_i24 %0, %1, %1" : "=v"(d) : "v"(v)); return d; };
      int cost[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
      for (int i = 0; i < 8; i++) { cost[2] += sq(partial[2][i]); cost[6] += sq(partial[6][i]); }
      cost[2] = (int)__umul24((unsigned)cost[2], 105u); cost[6] = (int)__umul24((unsigned)cost[6], 105u);
#pragma unroll
      for (int i = 0; i < 7; i++) {
        cost[0] += (int)__umul24((unsigned)(sq(partial[0][i]) + sq(partial[0][14 - i])), (unsigned)div_table[i + 1]);
        cost[4] += (int)__umul24((unsigned)(sq(partial[4][i]) + sq(partial[4][14 - i])), (unsigned)div_table[i + 1]);
      }
      cost[0] += (int)__umul24((unsigned)sq(partial[0][7]), 105u);
      cost[4] += (int)__umul24((unsigned)sq(partial[4][7]), 105u);
#pragma unroll
      for (int i = 1; i < 8; i += 2) {
#pragma unroll
        for (int j = 0; j < 5; j++) cost[i] += sq(partial[i][3 + j]);
        cost[i] = (int)__umul24((unsigned)cost[i], 105u);
#pragma unroll
        for (int j = 0; j < 3; j++) cost[i] += (int)__umul24((unsigned)(sq(partial[i][j]) + sq(partial[i][10 - j])), (unsigned)div_table[2 * j + 2]);
      }
      int best = 0, best_cost = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) if (cost[i] > best_cost) { best_cost = cost[i]; best = i; }
      int opp = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) if (i == ((best + 4) & 7)) opp = cost[i];
      // everything a quad of this block needs, once per block instead of once per quad (16 luma + 8 chroma quads per block)
      const int var = (best_cost - opp) >> 10;
      const int vs = (var >> 6) ? min(msb((unsigned)(var >> 6)), 12) : 0;
      const int pri = var ? (ypri0 * (4 + vs) + 8) >> 4 : 0;             // spec 7.15.2: luma primary strength adjusted by the variance
      const int pshift = pri ? max(0, dampy - msb((unsigned)pri)) : 0;
      const int ydir = ypri0 == 0 ? 0 : best;
      bpar[tid] = (uint32_t)pri | ((uint32_t)pshift << 8) | ((uint32_t)ydir << 12) | ((uint32_t)((pri >> cs) & 1) << 15) | ((uint32_t)best << 16);
    }
  }
  __syncthreads();
  // filter: luma 64x64 -> 16 samples per lane (4 rows x 4 columns), chroma 2 x 32x32 -> 2 x 4 samples per lane
  const int ysshift = ysec ? max(0, dampy - msb((unsigned)ysec)) : 0, usshift = usec ? max(0, dampc - msb((unsigned)usec)) : 0;
  const int upshift = upri ? max(0, dampc - msb((unsigned)upri)) : 0, upar = (upri >> cs) & 1;
  for (int q = tid; q < 64 * 16; q += 256) {         // q -> (row, group of 4 columns)
    const int r = q >> 4, c = (q & 15) * 4;
    const int fy = sby * 64 + r, fx = sbx * 64 + c;
    if (fy >= L.h || fx >= L.w) continue;
    const int b = (r >> 3) * 8 + (c >> 3);
    const bool skip = !enabled || bskip[b];
    const uint16_t *p = ty + (2 + r) * YS + 2 + c;
    int o[4];
    if (skip) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3]; }
    else {
      const uint32_t bp = bpar[b];
      const int pri = bp & 255, pshift = (bp >> 8) & 15, par = (bp >> 15) & 1;
      const int32_t *ot = offy + ((bp >> 12) & 7) * kTapEntries;
      v2s r[2];
      if (interior) cdef_quad_packed<false>(p, ot, pri, ysec, pshift, ysshift, par ? 3 : 4, par ? 3 : 2, r);
      else cdef_quad_packed<true>(p, ot, pri, ysec, pshift, ysshift, par ? 3 : 4, par ? 3 : 2, r);
      o[0] = r[0].x; o[1] = r[0].y; o[2] = r[1].x; o[3] = r[1].y;
    }
    Pix *d = dy + row_off(fy, L.stride_y) + fx;
    if constexpr (sizeof(Pix) == 1) *reinterpret_cast<uint32_t *>(d) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
    else { uint2 u; u.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16); u.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16); *reinterpret_cast<uint2 *>(d) = u; }
  }
  for (int q = tid; q < 2 * 32 * 8; q += 256) {
    const int pl = q >> 8, r = (q >> 3) & 31, c = (q & 7) * 4;
    const int fy = sby * 32 + r, fx = sbx * 32 + c;
    if (fy >= chh || fx >= cw) continue;
    const int b = (r >> 2) * 8 + (c >> 2);
    const bool skip = !enabled || bskip[b];
    const uint16_t *p = tc[pl] + (2 + r) * CSZ + 2 + c;
    int o[4];
    if (skip) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3]; }
    else {
      const int32_t *ot = offc + (upri == 0 ? 0 : (int)((bpar[b] >> 16) & 7)) * kTapEntries;
      v2s r[2];
      if (interior) cdef_quad_packed<false>(p, ot, upri, usec, upshift, usshift, upar ? 3 : 4, upar ? 3 : 2, r);
      else cdef_quad_packed<true>(p, ot, upri, usec, upshift, usshift, upar ? 3 : 4, upar ? 3 : 2, r);
      o[0] = r[0].x; o[1] = r[0].y; o[2] = r[1].x; o[3] = r[1].y;
    }
    Pix *d = reinterpret_cast<Pix *>(L.dst[1 + pl]) + (size_t)f * chh * L.stride_uv + row_off(fy, L.stride_uv) + fx;
    if constexpr (sizeof(Pix) == 1) *reinterpret_cast<uint32_t *>(d) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
    else { uint2 u; u.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16); u.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16); *reinterpret_cast<uint2 *>(d) = u; }
  }
}

hipError_t launch_cdef(const CdefLaunch &L, hipStream_t s) {
  const dim3 grid((unsigned)(((L.w + 63) / 64) * ((L.h + 63) / 64) * L.nframes));   // 1-D: k_cdef orders the tiles itself (xcd_tile)
  if (L.bd == 8) hipLaunchKernelGGL(k_cdef<uint8_t>, grid, dim3(256), 0, s, L);
  else hipLaunchKernelGGL(k_cdef<uint16_t>, grid, dim3(256), 0, s, L);
  return hipGetLastError();
}

}  // namespace av1mi
