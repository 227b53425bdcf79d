// deblock_kernels.hip — SURVEY.md §8a row K5: the AV1 deblocking loop filter as ONE gfx950 kernel per plane.
//
// A workgroup owns a TW x TH window of output samples.  It stages the window plus an 8-sample halo of the
// UNFILTERED source plane in LDS (coalesced row loads, uint16 per sample), runs pass 0 (vertical edges) on
// every edge that can reach the window or the rows pass 1 will read, then pass 1 (horizontal edges) on the
// window's columns, and writes the window back as whole rows.  Source and destination planes differ, so
// neighbouring workgroups never see each other's output; edges in the halo are recomputed, never exchanged.
// Within a pass edges are independent (a filter never reaches past half of the narrower transform block),
// so lanes filter in place in LDS without ordering.
// HBM traffic: b*S read + b*S written = the 2b*S of SURVEY.md §8d (halo re-reads come from L2).
//
// Restates AV1 spec §7.14 and libaom aom_dsp/loopfilter.c (highbd_filter4/6/8/14 and their masks); the
// reference has no counterpart (internal/ffmpeg/transcode.go:120 names the external encoder only).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "av1mi_internal.hpp"

namespace av1mi {

struct LfThr { int lim, mblim, hev; };

__device__ __forceinline__ LfThr lf_limits(int lvl, int sharp) {
  const int shift = sharp > 4 ? 2 : (sharp > 0 ? 1 : 0);
  int inside = lvl >> shift;
  if (sharp > 0) inside = min(inside, 9 - sharp);
  inside = max(inside, 1);
  return { inside, 2 * (lvl + 2) + inside, lvl >> 4 };
}

// px[0..15] = p7..p0 q0..q7.  len in {4, 6, 8, 14}.  Every index is a compile-time constant.
__device__ __forceinline__ void lf_filter(int (&px)[16], int len, LfThr t, int bd) {
  const int sh = bd - 8;
  const int lim = t.lim << sh, blim = t.mblim << sh, hevt = t.hev << sh, one = 1 << sh;
#define P(i) px[7 - (i)]
#define Q(i) px[8 + (i)]
  // The masks are conjunctions of |a - b| <= threshold tests: each group is ONE comparison of the largest difference, and a
  // difference of two samples is one v_sad_u32.  Written as a chain of abs() <= t && ... every term was three instructions and
  // a short-circuit branch (exec-mask save + s_cbranch) of its own.
  // (inline asm: there is no builtin for v_sad_u32, and __usad() is a library routine that compiles to min, max, subtract)
  auto ad = [](int a, int b) { int d; asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b)); return d; };
  const int d10 = max(ad(P(1), P(0)), ad(Q(1), Q(0)));
  int m = d10, fl = d10;
  if (len >= 6) { m = max(m, max(ad(P(2), P(1)), ad(Q(2), Q(1)))); fl = max(fl, max(ad(P(2), P(0)), ad(Q(2), Q(0)))); }
  if (len >= 8) { m = max(m, max(ad(P(3), P(2)), ad(Q(3), Q(2)))); fl = max(fl, max(ad(P(3), P(0)), ad(Q(3), Q(0)))); }
  const bool mask = (m <= lim) & (ad(P(0), Q(0)) * 2 + (ad(P(1), Q(1)) >> 1) <= blim);
  const bool flat = len >= 6 && fl <= one;
  bool flat2 = false;
  if (len == 14)
    flat2 = max(max(max(ad(P(4), P(0)), ad(Q(4), Q(0))), max(ad(P(5), P(0)), ad(Q(5), Q(0)))), max(ad(P(6), P(0)), ad(Q(6), Q(0)))) <= one;
  if (!mask) return;   // filter4 with mask == 0 leaves all four samples unchanged
  if (flat && flat2) {
    // 13 taps [1 1 1 1 1 2 2 2 1 1 1 1 1], positions clamped to p6 / q6
    int o[12];
#pragma unroll
    for (int i = -6; i < 6; i++) {
      int s = 8;
#pragma unroll
      for (int k = -6; k <= 6; k++) {
        const int pos = min(max(i + k, -7), 6);
        s += (k >= -1 && k <= 1 ? 2 : 1) * px[8 + pos];
      }
      o[i + 6] = s >> 4;
    }
#pragma unroll
    for (int i = 0; i < 12; i++) px[2 + i] = o[i];
  } else if (flat && len >= 8) {
    // 7 taps [1 1 1 2 1 1 1], clamped to p3 / q3
    int o[6];
#pragma unroll
    for (int i = -3; i < 3; i++) {
      int s = 4;
#pragma unroll
      for (int k = -3; k <= 3; k++) s += (k == 0 ? 2 : 1) * px[8 + min(max(i + k, -4), 3)];
      o[i + 3] = s >> 3;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) px[5 + i] = o[i];
  } else if (flat && len == 6) {
    // 5 taps [1 2 2 2 1], clamped to p2 / q2
    int o[4];
#pragma unroll
    for (int i = -2; i < 2; i++) {
      int s = 4;
#pragma unroll
      for (int k = -2; k <= 2; k++) s += (k >= -1 && k <= 1 ? 2 : 1) * px[8 + min(max(i + k, -3), 2)];
      o[i + 2] = s >> 3;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) px[6 + i] = o[i];
  } else {
    const int lo = -(128 << sh), hi = (128 << sh) - 1, t80 = 128 << sh;
    const int ps1 = P(1) - t80, ps0 = P(0) - t80, qs0 = Q(0) - t80, qs1 = Q(1) - t80;
    const bool hev = d10 > hevt;
    int f = hev ? min(max(ps1 - qs1, lo), hi) : 0;
    // f + 3 (qs0 - ps0) as one 24-bit multiply-add (the compiler made it a 64-bit one: four passes)
    { const int dq = qs0 - ps0; asm("v_mad_i32_i24 %0, %1, 3, %0" : "+v"(f) : "v"(dq)); }
    f = min(max(f, lo), hi);
    const int f1 = min(f + 4, hi) >> 3, f2 = min(f + 3, hi) >> 3;
    Q(0) = min(max(qs0 - f1, lo), hi) + t80;
    P(0) = min(max(ps0 + f2, lo), hi) + t80;
    f = hev ? 0 : (f1 + 1) >> 1;
    Q(1) = min(max(qs1 - f, lo), hi) + t80;
    P(1) = min(max(ps1 + f, lo), hi) + t80;
  }
#undef P
#undef Q
}

// edge decision for the unit `cur` against `prev` (the unit on the other side): returns filter length or 0.  Branch-free: as
// nested early returns it compiled to five levels of exec-mask save / branch per item.
__device__ __forceinline__ int lf_edge(uint32_t cur, uint32_t prev, int pass, int pos, bool is_chroma, int &lvl) {
  const int tx = pass == 0 ? (cur & 15) : ((cur >> 4) & 15), ptx = pass == 0 ? (prev & 15) : ((prev >> 4) & 15);
  const int flags = cur >> 24;
  const int outside = (cur == 0xFFFFFFFFu) | (prev == 0xFFFFFFFFu);                 // outside the plane
  const int not_edge = (pos & ((1 << tx) - 1)) != 0;                                 // not a transform edge
  const int inner_skip = (flags & 1) & ~(flags >> (1 + pass)) & 1;                   // skipped inter block, inner edge
  const int lc = (cur >> (8 + 8 * pass)) & 255, lp = (prev >> (8 + 8 * pass)) & 255;
  lvl = lc ? lc : lp;
  const int b = min(tx, ptx);                                                        // log2 of the narrower transform
  const int len = is_chroma ? (b == 2 ? 4 : 6) : (b == 2 ? 4 : b == 3 ? 8 : 14);
  return (outside | not_edge | inner_skip | (lvl == 0)) ? 0 : len;
}

// One edge line of compile-time filter length: reads only the 2 x HALF samples that length can look at (p3..q3 for 8, p6..q6
// for 14), filters, and writes back the 2 x M it can modify, unconditionally.  `step` = element stride across the edge (1 for
// vertical edges, the tile's row stride for horizontal ones).  The run-time-length form paid 16 loads and 12 predicated stores
// (each a compare + exec-mask round trip) per line whatever the length.
template <int LEN>
__device__ __forceinline__ void lf_line(uint16_t *p, int step, LfThr t, int bd) {
  constexpr int HALF = LEN == 14 ? 7 : LEN == 8 ? 4 : LEN == 6 ? 3 : 2, M = LEN == 14 ? 6 : LEN == 8 ? 3 : 2;
  int px[16];
#pragma unroll
  for (int k = 0; k < 16; k++) px[k] = (k >= 8 - HALF && k < 8 + HALF) ? (int)p[(k - 8) * step] : 0;
  lf_filter(px, LEN, t, bd);
#pragma unroll
  for (int k = 8 - M; k < 8 + M; k++) p[(k - 8) * step] = (uint16_t)px[k];
}
// lanes of a wave almost always share the length (it depends on the transform sizes on both sides of the edge): a chain of
// uniform branches, each body specialised
__device__ __forceinline__ void lf_line_any(uint16_t *p, int step, int len, LfThr t, int bd) {
  if (len == 8) lf_line<8>(p, step, t, bd);
  else if (len == 4) lf_line<4>(p, step, t, bd);
  else if (len == 6) lf_line<6>(p, step, t, bd);
  else lf_line<14>(p, step, t, bd);
}

// CHROMA: the plane's filter-length rule (a launch is one plane); SHARP0: sharpness 0, the only value the encoder loops use
// (other values take the run-time path) — both fold the per-line limit arithmetic.
template <typename Pix, int TW, int TH, bool CHROMA, bool SHARP0>
__global__ __launch_bounds__(256) void k_deblock(DeblockLaunch L) {
  // Halo of 8: an edge needs at most p6..q6 (13-tap filter) plus the flatness tests up to p6/q6; an edge whose filter can
  // reach the window lies in [0, TW] x [0, TH], so source samples in [-8, TW + 8) x [-8, TH + 8) are all that is ever read
  constexpr int HALO = 8, LW = TW + 2 * HALO, LH = TH + 2 * HALO, LS = LW + 2;   // +2: odd dword row stride
  constexpr int MW = LW / 4, MH = LH / 4;
  __shared__ __attribute__((aligned(16))) uint16_t tile[LH * LS];
  __shared__ uint32_t mis[MH * MW];
  const int tid = threadIdx.x;
  const Tile3 tl = xcd_tile((L.w + TW - 1) / TW, (L.h + TH - 1) / TH, L.nframes);
  const int X0 = tl.x * TW - HALO, Y0 = tl.y * TH - HALO;   // frame coords of LDS (0,0)
  // frame tl.z of a stack of frames: planes are stacked vertically, mode info per frame or shared
  const Pix *src = reinterpret_cast<const Pix *>(L.src) + (size_t)tl.z * L.h * L.src_stride;
  const uint32_t *mi = L.mi + (size_t)tl.z * L.mi_frame_stride;
  const int cols = L.w >> 2, rows = L.h >> 2;
  // mode-info units (0xFFFFFFFF outside the plane): requested first, stored after the sample loads have been issued too
  constexpr int NMI = (MH * MW + 255) / 256;
  uint32_t miv[NMI];
#pragma unroll
  for (int k = 0; k < NMI; k++) {
    const int i = tid + 256 * k;
    miv[k] = 0xFFFFFFFFu;
    if (i < MH * MW) {
      const int ur = (Y0 >> 2) + i / MW, uc = (X0 >> 2) + i % MW;
      if (ur >= 0 && ur < rows && uc >= 0 && uc < cols) miv[k] = mi[(size_t)ur * L.mi_stride + uc];
    }
  }
  // samples, coordinates clamped into the plane, 4 per lane per item: all of a lane's loads first, then its LDS stores (as one
  // loop every load was waited for before the next one was issued)
  {
    constexpr int NITEMS = LH * (LW / 4), NIT = (NITEMS + 255) / 256;
    uint2 v[NIT];
    unsigned done = 0;          // bit k: item k is already in uint16-pair form (border items, assembled sample by sample)
#pragma unroll
    for (int k = 0; k < NIT; k++) {
      const int i = tid + 256 * k;
      if (i < NITEMS) {
        const int ly = i / (LW / 4), lx = (i % (LW / 4)) * 4;
        const int fy = min(max(Y0 + ly, 0), L.h - 1), fx = X0 + lx;
        const Pix *row = src + (size_t)fy * L.src_stride;
        if (fx >= 0 && fx + 3 < L.w) {
          if constexpr (sizeof(Pix) == 1) v[k].x = *reinterpret_cast<const uint32_t *>(row + fx);
          else v[k] = *reinterpret_cast<const uint2 *>(row + fx);
        } else {
          int q[4];
#pragma unroll
          for (int c = 0; c < 4; c++) q[c] = row[min(max(fx + c, 0), L.w - 1)];
          v[k].x = (uint32_t)q[0] | ((uint32_t)q[1] << 16); v[k].y = (uint32_t)q[2] | ((uint32_t)q[3] << 16);
          done |= 1u << k;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NMI; k++) {
      const int i = tid + 256 * k;
      if (i < MH * MW) mis[i] = miv[k];
    }
#pragma unroll
    for (int k = 0; k < NIT; k++) {
      const int i = tid + 256 * k;
      if (i < NITEMS) {
        const int ly = i / (LW / 4), lx = (i % (LW / 4)) * 4;
        // four samples = two uint16 pairs = two dword stores (rows are an even number of samples, lx a multiple of 4)
        uint32_t *d32 = reinterpret_cast<uint32_t *>(tile + ly * LS + lx);
        uint2 w = v[k];
        if constexpr (sizeof(Pix) == 1)
          if (!((done >> k) & 1)) { const uint32_t u = v[k].x; w.x = __builtin_amdgcn_perm(0u, u, 0x0c010c00u); w.y = __builtin_amdgcn_perm(0u, u, 0x0c030c02u); }
        d32[0] = w.x; d32[1] = w.y;
      }
    }
  }
  __syncthreads();
  // pass 0: vertical edges at lx = 4*uc, uc in [2, TW/4+2] (frame x in [0, TW]), all LH rows (pass 1 reads 7 rows beyond
  // the window's first and last edge)
  {
    constexpr int NU = TW / 4 + 1, NR = LH;
    // lanes of a wave walk DOWN one unit column (rows are an odd number of dwords apart: no bank conflicts), so the edge
    // decision — transform edge or not, filter length — is the same for nearly the whole wave instead of alternating lane
    // by lane with 8x8 transforms
    for (int t = tid; t < NU * NR; t += 256) {
      const int uc = 2 + t / NR, ly = t % NR;
      const int fx = X0 + 4 * uc, fy = Y0 + ly;
      if (fx <= 0 || fy < 0 || fy >= L.h) continue;
      int lvl = 0;
      const int len = lf_edge(mis[(ly >> 2) * MW + uc], mis[(ly >> 2) * MW + uc - 1], 0, fx, CHROMA, lvl);
      if (!len) continue;
      // (lf_line writes only what its length can modify: a neighbouring edge 4 samples away owns the rest)
      lf_line_any(tile + ly * LS + 4 * uc, 1, len, lf_limits(lvl, SHARP0 ? 0 : L.sharpness), sizeof(Pix) == 1 ? 8 : 10);
    }
  }
  __syncthreads();
  // pass 1: horizontal edges at ly = 4*ur, ur in [2, TH/4+2] (frame y in [0, TH]), columns of the window
  {
    constexpr int NU = TH / 4 + 1;
    for (int t = tid; t < NU * TW; t += 256) {
      const int lx = HALO + t % TW, ur = 2 + t / TW;
      const int fx = X0 + lx, fy = Y0 + 4 * ur;
      if (fy <= 0 || fx >= L.w) continue;
      int lvl = 0;
      const int len = lf_edge(mis[ur * MW + (lx >> 2)], mis[(ur - 1) * MW + (lx >> 2)], 1, fy, CHROMA, lvl);
      if (!len) continue;
      lf_line_any(tile + (4 * ur) * LS + lx, LS, len, lf_limits(lvl, SHARP0 ? 0 : L.sharpness), sizeof(Pix) == 1 ? 8 : 10);
    }
  }
  __syncthreads();
  // write the window, 4 samples per lane per step
  Pix *dst = reinterpret_cast<Pix *>(L.dst) + (size_t)tl.z * L.h * L.dst_stride;
  for (int i = tid; i < TH * (TW / 4); i += 256) {
    const int wy = i / (TW / 4), wx = (i % (TW / 4)) * 4;
    const int fy = Y0 + HALO + wy, fx = X0 + HALO + wx;
    if (fy >= L.h || fx >= L.w) continue;   // w, h are multiples of 4
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(tile + (HALO + wy) * LS + HALO + wx);
    const uint32_t lo = s32[0], hi = s32[1];
    Pix *o = dst + (size_t)fy * L.dst_stride + fx;
    if constexpr (sizeof(Pix) == 1) *reinterpret_cast<uint32_t *>(o) = __builtin_amdgcn_perm(hi, lo, 0x06040200u);
    else *reinterpret_cast<uint2 *>(o) = make_uint2(lo, hi);
  }
}

hipError_t launch_deblock(const DeblockLaunch &L, hipStream_t s) {
  constexpr int TW = 64, TH = 64;
  const dim3 grid((unsigned)(((L.w + TW - 1) / TW) * ((L.h + TH - 1) / TH) * L.nframes));   // 1-D: the kernel orders the tiles (xcd_tile)
  const dim3 blk(256);
#define AV1MI_DBL(PIX, C, S0) hipLaunchKernelGGL((k_deblock<PIX, TW, TH, C, S0>), grid, blk, 0, s, L)
  const bool c = L.is_chroma != 0, s0 = L.sharpness == 0;
  if (L.bd == 8) { if (c) { if (s0) AV1MI_DBL(uint8_t, true, true); else AV1MI_DBL(uint8_t, true, false); }
                   else   { if (s0) AV1MI_DBL(uint8_t, false, true); else AV1MI_DBL(uint8_t, false, false); } }
  else           { if (c) { if (s0) AV1MI_DBL(uint16_t, true, true); else AV1MI_DBL(uint16_t, true, false); }
                   else   { if (s0) AV1MI_DBL(uint16_t, false, true); else AV1MI_DBL(uint16_t, false, false); } }
#undef AV1MI_DBL
  return hipGetLastError();
}

}  // namespace av1mi
