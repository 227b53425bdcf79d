// lr_kernel.hip — SURVEY.md §8a row K7: loop restoration (Wiener and self-guided) of one plane, frames of a
// segment along blockIdx.z.
//
// A workgroup owns the intersection of one 64-row stripe (32 for 4:2:0 chroma; stripes are offset 8 luma rows
// upwards, AV1 spec §7.17) with one column band no wider than a restoration unit, so every sample it writes has
// the same stripe limits AND the same unit parameters: no divergence.  It stages that block plus a 3-sample halo
// in LDS through the spec's get_source_sample rule (picture-edge clamp; up to two rows beyond the stripe come from
// the DEBLOCKED frame, further rows replicate them; inside the stripe the CDEF output), then runs
//   Wiener:      7-tap rows -> int16 intermediate in LDS -> 7-tap columns, rounds 3 / 11;
//   self-guided: box sums (r = 2 on odd rows, then r = 1) -> A (u16) / B (i32) in LDS -> 3x3 weighted a*x + b,
//                both passes kept in registers, projection with (w0, w1, 128 - w0 - w1).
// Reads two planes, writes one: algorithmic HBM traffic 2b*S (SURVEY.md §8d counts b read + b written; the
// deblocked rows are 4 of every 64).  Restates spec §7.17.3/4/6 and libaom av1_highbd_wiener_convolve_add_src_c /
// av1_selfguided_restoration_c; the reference has no counterpart (internal/ffmpeg/transcode.go:120).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "av1mi_internal.hpp"

namespace av1mi {

// Sgr_Params (spec 7.17.3): { r0, eps0, r1, eps1 }; s = ((1 << 20) + n*n*eps / 2) / (n*n*eps) is derived in the kernel
__device__ constexpr int kSgrParams[16][4] = {
  { 2, 12, 1, 4 }, { 2, 15, 1, 6 }, { 2, 18, 1, 8 }, { 2, 21, 1, 9 }, { 2, 24, 1, 10 }, { 2, 29, 1, 11 },
  { 2, 36, 1, 12 }, { 2, 45, 1, 13 }, { 2, 56, 1, 14 }, { 2, 68, 1, 15 }, { 0, 0, 1, 5 }, { 0, 0, 1, 8 },
  { 0, 0, 1, 11 }, { 0, 0, 1, 14 }, { 2, 30, 0, 0 }, { 2, 75, 0, 0 } };

// SUBY: vertical subsampling of the plane (0 luma, 1 chroma of 4:2:0) — stripe height and offset become constants
// SGR: the launch may hold self-guided units.  Without them (the encoder's policy: Wiener or none) the A / B planes of that path
// are not allocated — 9 instead of 27 KB of scratch, 20 instead of 38 KB of LDS per workgroup, 8 instead of 4 workgroups per CU —
// and a unit of type 2 would be copied unfiltered (the caller promises there is none: LrLaunch::no_sgr).
template <typename Pix, int SUBY, bool SGR = true>
__global__ __launch_bounds__(256) void k_lr(LrLaunch L) {
  constexpr int MAXH = 64, MAXW = 64, SS = MAXW + 16;      // source tile row stride (u16): tile column 0 = X0 - 4
  constexpr int AS = MAXW + 2 + 2;                         // A/B row stride
  __shared__ __attribute__((aligned(16))) uint16_t src[(MAXH + 6) * SS];
  __shared__ __attribute__((aligned(16))) unsigned char scratch[SGR ? (MAXH + 2) * AS * 2 + (MAXH + 2) * AS * 4 : (MAXH + 6) * MAXW * 2];
  int16_t *inter = reinterpret_cast<int16_t *>(scratch);                       // Wiener: (SH+6) x tw
  uint16_t *Abuf = reinterpret_cast<uint16_t *>(scratch);                      // self-guided: (SH+2) x (tw+2)
  int32_t *Bbuf = reinterpret_cast<int32_t *>(scratch + (MAXH + 2) * AS * 2);

  constexpr int bd = sizeof(Pix) == 1 ? 8 : 10;              // the launch picks the instantiation by L.bd (8 or 10)
  const int tid = threadIdx.x;
  constexpr int SH = 64 >> SUBY, off = 8 >> SUBY;
  const int tw = L.unit_size < 64 ? L.unit_size : 64;
  const Tile3 tl = xcd_tile((L.w + tw - 1) / tw, (L.h + off + SH - 1) / SH, L.nframes);
  const int X0 = tl.x * tw, stripe = tl.y, f = tl.z;
  const int sstart = stripe * SH - off, send = sstart + SH - 1;
  const int y0 = max(sstart, 0), y1 = min(send, L.h - 1);    // rows this workgroup writes
  if (y0 > y1 || X0 >= L.w) return;
  // The decision's sums run over an EIGHTH of the plane — the 64-column x stripe tiles with (column + stripe) % 8 == 0, a diagonal
  // pattern over the whole picture (every tile when the plane has fewer than 32) — so the decision costs an eighth of a source read.
  // Two-pass form: the workgroups that have nothing to do leave before they load anything (uniform per workgroup, before any barrier).
  const int ntx64 = (L.w + 63) >> 6, nst = (L.h + off + SH - 1) / SH;
  const bool sampled = ntx64 * nst < 32 || (((X0 >> 6) + stripe) & 7) == 0;
  if (L.pass == 1 && !sampled) return;
  if (L.pass == 2 && (sampled || !L.keep[(size_t)f * L.keep_stride])) return;
  const int bw = min(tw, L.w - X0), bh = y1 - y0 + 1;
  const Pix *cdef = reinterpret_cast<const Pix *>(L.cdef) + (size_t)f * L.h * L.stride;
  const Pix *dbl = reinterpret_cast<const Pix *>(L.dbl) + (size_t)f * L.h * L.stride;
  Pix *out = reinterpret_cast<Pix *>(L.out) + (size_t)f * L.h * L.stride;
  const int urows = max((L.h + (L.unit_size >> 1)) / L.unit_size, 1), ucols = max((L.w + (L.unit_size >> 1)) / L.unit_size, 1);
  const int ur = min(urows - 1, (y0 + off) / L.unit_size), uc = min(ucols - 1, X0 / L.unit_size);
  // the unit's 8 parameter bytes as one load, up front (they were re-read from memory after the staging barrier)
  const uint2 U8 = *reinterpret_cast<const uint2 *>(L.units + ((size_t)f * L.unit_frame_stride + (size_t)ur * ucols + uc) * 8);
  const int U[8] = { (int8_t)(U8.x & 255), (int8_t)((U8.x >> 8) & 255), (int8_t)((U8.x >> 16) & 255), (int8_t)(U8.x >> 24),
                     (int8_t)(U8.y & 255), (int8_t)((U8.y >> 8) & 255), (int8_t)((U8.y >> 16) & 255), (int8_t)(U8.y >> 24) };
  const int type = (!SGR && U[0] == 2) ? 0 : U[0];
  // the on/off decision's two sums (L.orig != nullptr): squared error of the restored samples and of the CDEF samples this workgroup
  // covers, against the source; per lane in 32 bits (16 samples x 2^20), per wave and stripe into 64-bit words (lr_finish)
  const Pix *orig = L.orig && sampled ? reinterpret_cast<const Pix *>(L.orig) + (size_t)f * L.h * L.stride : nullptr;
  unsigned e_lr = 0, e_cd = 0;
  auto lr_acc = [&](int o, int cd, int sv) { const int a = o - sv, b = cd - sv; e_lr += (unsigned)(a * a); e_cd += (unsigned)(b * b); };
  __shared__ unsigned long long s_sse[2][4];
  auto lr_finish = [&]() {      // every thread of the workgroup comes here exactly once (uniform control flow up to the returns)
    if (!orig) return;       // (uniform: no decision asked for, or a tile outside the sample)
    unsigned long long a = e_lr, b = e_cd;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_down(a, d, 64); b += __shfl_down(b, d, 64); }
    if ((tid & 63) == 0) { s_sse[0][tid >> 6] = a; s_sse[1][tid >> 6] = b; }
    __syncthreads();
    if (tid == 0) {
      unsigned long long *dst = L.sse + ((size_t)f * L.sse_stripes + stripe) * 2;
      atomicAdd(dst, s_sse[0][0] + s_sse[0][1] + s_sse[0][2] + s_sse[0][3]);
      atomicAdd(dst + 1, s_sse[1][0] + s_sse[1][1] + s_sse[1][2] + s_sse[1][3]);
    }
  };
  if (type == 0) {   // no restoration: copy the CDEF output
    for (int i = tid; i < bh * bw; i += 256) {
      const int r = i / bw, c = i - r * bw;
      const Pix v = cdef[row_off(y0 + r, L.stride) + X0 + c];
      out[row_off(y0 + r, L.stride) + X0 + c] = v;
      if (orig) lr_acc(v, v, orig[row_off(y0 + r, L.stride) + X0 + c]);
    }
    lr_finish();
    return;
  }
  // stage (bh + 6) rows x (bw + 8) columns of source samples: local row 0 == y0 - 3, local column 0 == X0 - 4 (4-aligned)
  const bool xin = X0 >= 4 && X0 + bw + 4 <= L.w && !(bw & 3);   // no horizontal clamping, whole 4-sample groups
  if (xin) {
    // item -> (row, group) by reciprocal multiplication (exact: items < 2^11 and (i + 0.5) / ng is never within 0.5 / ng of an
    // integer); an integer division by a run-time value is ~25 VALU instructions, and "full ? i / const : i / ng" evaluated both
    const float inv_ng = 1.0f / (float)((bw + 8) / 4);
    const int ng = (bw + 8) / 4, nmain = bh * ng, nhalo = 6 * ng;
    // All of a lane's loads first, then all of its LDS stores (as one loop every load was waited for before the next was issued,
    // and the kernel spent most of its time in those waits: waves stalled 64 % of their cycles by PMC).  The bh rows of the
    // block itself lie inside the stripe and the picture: plain CDEF rows, no clamping — at most 64 x 18 items = 5 per lane;
    // only the 3 + 3 halo rows go through get_source_sample's stripe rule — at most 108 items = 1 per lane.  The 8-byte loads
    // need no alignment check: the hardware takes any address.
    constexpr int NMAIN = (MAXH * ((MAXW + 8) / 4) + 255) / 256, NHALO = (6 * ((MAXW + 8) / 4) + 255) / 256;
    uint2 o[NMAIN + NHALO];
    auto load4 = [&](const Pix *q, uint2 &v) {
      if constexpr (sizeof(Pix) == 1) v.x = *reinterpret_cast<const uint32_t *>(q);
      else v = *reinterpret_cast<const uint2 *>(q);
    };
    const Pix *cdef0 = cdef + row_off(y0, L.stride) + X0 - 4;
#pragma unroll
    for (int k = 0; k < NMAIN; k++) {
      const int i = tid + 256 * k;
      if (i < nmain) {
        const int r = (int)(((float)i + 0.5f) * inv_ng), g = i - r * ng;
        load4(cdef0 + row_off(r, L.stride) + 4 * g, o[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < NHALO; k++) {
      const int i = tid + 256 * k;
      if (i < nhalo) {
        const int hr = (int)(((float)i + 0.5f) * inv_ng), g = i - hr * ng;
        int y = min(max(hr < 3 ? y0 - 3 + hr : y1 - 2 + hr, 0), L.h - 1);       // rows y0 - 3 .. y0 - 1 and y1 + 1 .. y1 + 3
        const Pix *p = cdef;
        if (y < sstart) { y = max(sstart - 2, y); p = dbl; }
        else if (y > send) { y = min(send + 2, y); p = dbl; }
        load4(p + row_off(y, L.stride) + X0 - 4 + 4 * g, o[NMAIN + k]);
      }
    }
    auto store4 = [&](int r, int g, uint2 w) {
      if constexpr (sizeof(Pix) == 1) { const uint32_t u = w.x; w.x = __builtin_amdgcn_perm(0u, u, 0x0c010c00u); w.y = __builtin_amdgcn_perm(0u, u, 0x0c030c02u); }
      *reinterpret_cast<uint2 *>(src + __mul24(r, SS) + 4 * g) = w;
    };
#pragma unroll
    for (int k = 0; k < NMAIN; k++) {
      const int i = tid + 256 * k;
      if (i < nmain) { const int r = (int)(((float)i + 0.5f) * inv_ng); store4(r + 3, i - r * ng, o[k]); }
    }
#pragma unroll
    for (int k = 0; k < NHALO; k++) {
      const int i = tid + 256 * k;
      if (i < nhalo) { const int hr = (int)(((float)i + 0.5f) * inv_ng); store4(hr < 3 ? hr : bh + hr, i - hr * ng, o[NMAIN + k]); }
    }
  } else {
    for (int i = tid; i < (bh + 6) * (bw + 6); i += 256) {
      const int r = i / (bw + 6), c = i - r * (bw + 6);
      const int x = min(max(X0 - 3 + c, 0), L.w - 1);
      int y = min(max(y0 - 3 + r, 0), L.h - 1);
      const Pix *p = cdef;
      if (y < sstart) { y = max(sstart - 2, y); p = dbl; }
      else if (y > send) { y = min(send + 2, y); p = dbl; }
      src[r * SS + c + 1] = p[row_off(y, L.stride) + x];
    }
  }
  __syncthreads();
  const int maxpix = (1 << bd) - 1;
  if (type == 1) {
    int vf[7], hf[7];
#pragma unroll
    for (int i = 0; i < 3; i++) { vf[i] = vf[6 - i] = U[1 + i]; hf[i] = hf[6 - i] = U[4 + i]; }
    vf[3] = 128 - 2 * (U[1] + U[2] + U[3]);
    hf[3] = 128 - 2 * (U[4] + U[5] + U[6]);
    const int offset = 1 << (bd + 3), limit = (1 << (bd + 5)) - 1;
    if (!(bw & 3)) {
      // Both passes on v_dot2_i32_i16 (two taps per instruction; samples and the clipped intermediate fit int16, the sums are
      // exact in int32).  Horizontal: four outputs per lane from six aligned dwords of the source row (window columns
      // c .. c + 11, the taps of output k start at column c + 1 + k): outputs 1 and 3 use the dwords as they are, outputs 0 and
      // 2 the same dwords funnel-shifted by one sample.  Tap pairs (h0,h1) (h2,h3) (h4,h5) (h6,0).
      typedef short s16x2 __attribute__((ext_vector_type(2)));
      auto pk = [](int a, int b) { return __builtin_bit_cast(s16x2, (uint32_t)(a & 0xffff) | ((uint32_t)b << 16)); };
      const s16x2 H0 = pk(hf[0], hf[1]), H1 = pk(hf[2], hf[3]), H2 = pk(hf[4], hf[5]), H3 = pk(hf[6], 0);
      const int q4 = bw / 4;
      const float inv_q4 = 1.0f / (float)q4;
      for (int i = tid; i < (bh + 6) * q4; i += 256) {
        const int r = (int)(((float)i + 0.5f) * inv_q4), c = (i - r * q4) * 4;
        const uint2 *p = reinterpret_cast<const uint2 *>(src + __mul24(r, SS) + c);   // (r * SS became a 64-bit multiply-add)
        const uint2 a = p[0], b = p[1], e = p[2];
        const uint32_t d[6] = { a.x, a.y, b.x, b.y, e.x, e.y };
        uint32_t A[6];
#pragma unroll
        for (int m = 1; m < 6; m++) A[m] = __builtin_amdgcn_alignbit(d[m], d[m - 1], 16);   // columns (c + 2m - 1, c + 2m)
        int o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          // output k: columns c + 1 + k .. c + 7 + k
          const uint32_t *w = (k & 1) ? d + (k + 1) / 2 : A + k / 2 + 1;
          int sum = 4;
          sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, w[0]), H0, sum, false);
          sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, w[1]), H1, sum, false);
          sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, w[2]), H2, sum, false);
          sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, w[3]), H3, sum, false);
          o[k] = min(max(sum >> 3, -offset), limit - offset);
        }
        uint2 w; w.x = (uint32_t)(o[0] & 0xffff) | ((uint32_t)o[1] << 16); w.y = (uint32_t)(o[2] & 0xffff) | ((uint32_t)o[3] << 16);
        *reinterpret_cast<uint2 *>(inter + r * MAXW + c) = w;
      }
      __syncthreads();
      // Vertical: a lane produces rows 2 rp and 2 rp + 1 of four columns from intermediate rows 2 rp .. 2 rp + 7, interleaved
      // once into row pairs per column; the even row pairs its taps (v0,v1) (v2,v3) (v4,v5) (v6,0), the odd row the same
      // row pairs with the taps moved up by one row: (0,v0) (v1,v2) (v3,v4) (v5,v6).
      const s16x2 VE[4] = { pk(vf[0], vf[1]), pk(vf[2], vf[3]), pk(vf[4], vf[5]), pk(vf[6], 0) };
      const s16x2 VO[4] = { pk(0, vf[0]), pk(vf[1], vf[2]), pk(vf[3], vf[4]), pk(vf[5], vf[6]) };
      const int hp = (bh + 1) >> 1;
      for (int i = tid; i < hp * q4; i += 256) {
        const int rp = (int)(((float)i + 0.5f) * inv_q4), c = (i - rp * q4) * 4, r = 2 * rp;
        int se[4] = { 1024, 1024, 1024, 1024 }, so[4] = { 1024, 1024, 1024, 1024 };
#pragma unroll
        for (int j = 0; j < 4; j++) {
          // rows r + 2j, r + 2j + 1 (the last pair of an odd-height block reads one row past the intermediate: inside the
          // scratch buffer, and only the odd output row, which is not stored then, depends on it)
          const uint2 x = *reinterpret_cast<const uint2 *>(inter + (r + 2 * j) * MAXW + c);
          const uint2 y = *reinterpret_cast<const uint2 *>(inter + (r + 2 * j + 1) * MAXW + c);
          const uint32_t pr[4] = { __builtin_amdgcn_perm(y.x, x.x, 0x05040100u), __builtin_amdgcn_perm(y.x, x.x, 0x07060302u),
                                   __builtin_amdgcn_perm(y.y, x.y, 0x05040100u), __builtin_amdgcn_perm(y.y, x.y, 0x07060302u) };
#pragma unroll
          for (int k = 0; k < 4; k++) {
            se[k] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, pr[k]), VE[j], se[k], false);
            so[k] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, pr[k]), VO[j], so[k], false);
          }
        }
#pragma unroll
        for (int half = 0; half < 2; half++) {
          if (r + half >= bh) break;
          int o[4];
#pragma unroll
          for (int k = 0; k < 4; k++) o[k] = min(max((half ? so[k] : se[k]) >> 11, 0), maxpix);
          Pix *d = out + row_off(y0 + r + half, L.stride) + X0 + c;
          if (orig) {
            // four source samples as one load (the plane rows and X0 + c are 4-sample aligned wherever the output store is),
            // the four CDEF samples as one LDS read
            const Pix *sp = orig + row_off(y0 + r + half, L.stride) + X0 + c;
            const uint2 cq = *reinterpret_cast<const uint2 *>(src + (r + half + 3) * SS + c + 4);
            const int cdv[4] = { (int)(cq.x & 0xffff), (int)(cq.x >> 16), (int)(cq.y & 0xffff), (int)(cq.y >> 16) };
            int sv[4];
            if (((uintptr_t)sp & (4 * sizeof(Pix) - 1)) == 0) {
              if constexpr (sizeof(Pix) == 1) { const uint32_t u = *reinterpret_cast<const uint32_t *>(sp); sv[0] = u & 255; sv[1] = (u >> 8) & 255; sv[2] = (u >> 16) & 255; sv[3] = u >> 24; }
              else { const uint2 u = *reinterpret_cast<const uint2 *>(sp); sv[0] = u.x & 0xffff; sv[1] = u.x >> 16; sv[2] = u.y & 0xffff; sv[3] = u.y >> 16; }
            } else {
#pragma unroll
              for (int k = 0; k < 4; k++) sv[k] = sp[k];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) lr_acc(o[k], cdv[k], sv[k]);
          }
          if (((uintptr_t)d & (4 * sizeof(Pix) - 1)) == 0) {
            if constexpr (sizeof(Pix) == 1) *reinterpret_cast<uint32_t *>(d) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
            else { uint2 u; u.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16); u.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16); *reinterpret_cast<uint2 *>(d) = u; }
          } else {
#pragma unroll
            for (int k = 0; k < 4; k++) d[k] = (Pix)o[k];
          }
        }
      }
      lr_finish();
      return;
    }
    for (int i = tid; i < (bh + 6) * bw; i += 256) {
      const int r = i / bw, c = i - r * bw;
      const uint16_t *p = src + r * SS + c + 1;
      int sum = 0;
#pragma unroll
      for (int t = 0; t < 7; t++) sum += hf[t] * p[t];
      inter[r * MAXW + c] = (int16_t)min(max((sum + 4) >> 3, -offset), limit - offset);
    }
    __syncthreads();
    for (int i = tid; i < bh * bw; i += 256) {
      const int r = i / bw, c = i - r * bw;
      int sum = 0;
#pragma unroll
      for (int t = 0; t < 7; t++) sum += vf[t] * inter[(r + t) * MAXW + c];
      const int o = min(max((sum + 1024) >> 11, 0), maxpix);
      out[row_off(y0 + r, L.stride) + X0 + c] = (Pix)o;
      if (orig) lr_acc(o, src[(r + 3) * SS + c + 4], orig[row_off(y0 + r, L.stride) + X0 + c]);
    }
    lr_finish();
    return;
  }
  // self-guided
  if constexpr (!SGR) return;
  const int set = U[1] & 15, w0 = U[2], w1 = U[3], w2 = 128 - w0 - w1;
  int flt[2][16];   // this lane's samples: index k <-> sample tid + 256 k of the block (row-major)
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int r = kSgrParams[set][2 * pass], eps = kSgrParams[set][2 * pass + 1];
    if (r == 0) continue;   // uniform across the workgroup
    const unsigned n = (2 * r + 1) * (2 * r + 1), n2e = n * n * (unsigned)eps;
    const unsigned sfac = ((1u << 20) + n2e / 2) / n2e, one_by_n = ((1u << 12) + n / 2) / n;
    // A/B at local (i, j) for i in -1..bh, j in -1..bw; stored at [(i+1) * AS + (j+1)]
    for (int q = tid; q < (bh + 2) * (bw + 2); q += 256) {
      const int i = q / (bw + 2) - 1, j = q % (bw + 2) - 1;
      if (pass == 0 && !((y0 + i) & 1)) continue;            // the r = 2 pass only ever reads odd rows
      const uint16_t *p = src + (i + 3) * SS + (j + 4);
      unsigned a = 0, b = 0;
      for (int dy = -r; dy <= r; dy++)
        for (int dx = -r; dx <= r; dx++) { const unsigned v = p[dy * SS + dx]; a += v * v; b += v; }
      const unsigned as = bd == 8 ? a : (a + 8) >> 4, d = bd == 8 ? b : (b + 2) >> 2;
      const unsigned pv = as * n > d * d ? as * n - d * d : 0;
      const unsigned z = (unsigned)(((unsigned long long)pv * sfac + (1u << 19)) >> 20);
      const unsigned a2 = z >= 255 ? 256 : z == 0 ? 1 : ((z << 8) + z / 2) / (z + 1);
      Abuf[(i + 1) * AS + j + 1] = (uint16_t)a2;
      Bbuf[(i + 1) * AS + j + 1] = (int32_t)(((256 - a2) * b * one_by_n + (1u << 11)) >> 12);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int q = tid + 256 * k;
      if (q >= bh * bw) continue;
      const int i = q / bw, j = q - i * bw;
      const uint16_t *pa = Abuf + (i + 1) * AS + j + 1;
      const int32_t *pb = Bbuf + (i + 1) * AS + j + 1;
      int a, b, shift = 5;
      if (pass == 0) {
        if ((y0 + i) & 1) {
          a = 6 * pa[0] + 5 * (pa[-1] + pa[1]); b = 6 * pb[0] + 5 * (pb[-1] + pb[1]); shift = 4;
        } else {
          a = 6 * (pa[-AS] + pa[AS]) + 5 * (pa[-AS - 1] + pa[-AS + 1] + pa[AS - 1] + pa[AS + 1]);
          b = 6 * (pb[-AS] + pb[AS]) + 5 * (pb[-AS - 1] + pb[-AS + 1] + pb[AS - 1] + pb[AS + 1]);
        }
      } else {
        a = 4 * (pa[0] + pa[-1] + pa[1] + pa[-AS] + pa[AS]) + 3 * (pa[-AS - 1] + pa[-AS + 1] + pa[AS - 1] + pa[AS + 1]);
        b = 4 * (pb[0] + pb[-1] + pb[1] + pb[-AS] + pb[AS]) + 3 * (pb[-AS - 1] + pb[-AS + 1] + pb[AS - 1] + pb[AS + 1]);
      }
      const int v = a * (int)src[(i + 3) * SS + j + 4] + b;
      flt[pass][k] = (v + (1 << (shift + 3))) >> (shift + 4);   // Round2(v, SGR_BITS 8 + shift - RST_BITS 4)
    }
    __syncthreads();
  }
  const int r0 = kSgrParams[set][0], r1 = kSgrParams[set][2];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int q = tid + 256 * k;
    if (q >= bh * bw) continue;
    const int i = q / bw, j = q - i * bw;
    const int u = (int)src[(i + 3) * SS + j + 4] << 4;
    int v = w1 * u;
    v += w0 * (r0 ? flt[0][k] : u);
    v += w2 * (r1 ? flt[1][k] : u);
    const int o = min(max((v + 1024) >> 11, 0), maxpix);
    out[row_off(y0 + i, L.stride) + X0 + j] = (Pix)o;
    if (orig) lr_acc(o, src[(i + 3) * SS + j + 4], orig[row_off(y0 + i, L.stride) + X0 + j]);
  }
  lr_finish();
}

// restoration stays on for frame f of the plane when it lowered the squared error: on[f * on_stride] = sum over the stripes
__global__ __launch_bounds__(64) void k_lr_decide(const unsigned long long *sse, int nframes, int stripes, uint8_t *on, int on_stride) {
  const int f = blockIdx.x * 64 + threadIdx.x;
  if (f >= nframes) return;
  unsigned long long a = 0, b = 0;
  for (int s = 0; s < stripes; s++) { a += sse[((size_t)f * stripes + s) * 2]; b += sse[((size_t)f * stripes + s) * 2 + 1]; }
  on[(size_t)f * on_stride] = a < b;
}

// A frame whose true size (vw x vh) is not a multiple of 8 is coded at the size rounded up (w x h, at most 7 more columns / rows).
// A decoder clamps motion-compensation and restoration reads at the TRUE last column / row (spec 7.11.3.4 lastX / lastY, 7.17
// PlaneEndX / PlaneEndY); the kernels clamp at the coded size.  Replicating the true edge into the padding makes both read the
// same values.  One thread per padded sample: first the columns right of vw for the rows above vh, then whole rows below vh
// (which copy row vh - 1 with ITS columns already clamped).
template <typename Pix> __global__ __launch_bounds__(256) void k_extend(Pix *plane, int stride, int w, int h, int vw, int vh, int nframes) {
  const int pw = w - vw, ph = h - vh;                 // padding columns / rows
  const long per = (long)pw * vh + (long)ph * w;      // padded samples of a frame
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= per * nframes) return;
  const int f = (int)(i / per);
  const long k = i - (long)f * per;
  Pix *p = plane + (size_t)f * h * stride;
  int x, y;
  if (k < (long)pw * vh) { y = (int)(k / pw); x = vw + (int)(k - (long)y * pw); }
  else { const long r = k - (long)pw * vh; y = vh + (int)(r / w); x = (int)(r - (long)(y - vh) * w); }
  p[(size_t)y * stride + x] = p[(size_t)min(y, vh - 1) * stride + min(x, vw - 1)];
}
hipError_t launch_extend(void *plane, int stride, int w, int h, int vw, int vh, int bd, int nframes, hipStream_t s) {
  const long n = ((long)(w - vw) * vh + (long)(h - vh) * w) * nframes;
  if (n <= 0) return hipSuccess;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (bd == 8) hipLaunchKernelGGL(k_extend<uint8_t>, grid, dim3(256), 0, s, (uint8_t *)plane, stride, w, h, vw, vh, nframes);
  else hipLaunchKernelGGL(k_extend<uint16_t>, grid, dim3(256), 0, s, (uint16_t *)plane, stride, w, h, vw, vh, nframes);
  return hipGetLastError();
}

// the three planes of 4:2:0 frames at once: thread = (frame, plane); plane p's sums start at sse + off[p] pairs
__global__ __launch_bounds__(64) void k_lr_decide3(const unsigned long long *sse, int nframes, int stripes_y, int stripes_c, uint8_t *on) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= nframes * 3) return;
  const int f = i / 3, p = i - 3 * f;
  const int stripes = p ? stripes_c : stripes_y;
  const unsigned long long *q = sse + 2 * ((size_t)(p ? nframes * stripes_y + (p - 1) * nframes * stripes_c : 0) + (size_t)f * stripes);
  unsigned long long a = 0, b = 0;
  for (int s = 0; s < stripes; s++) { a += q[2 * s]; b += q[2 * s + 1]; }
  on[i] = a < b;
}
__global__ __launch_bounds__(256) void k_zero16(uint4 *p, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = make_uint4(0, 0, 0, 0);
}
hipError_t launch_zero16(void *p, size_t bytes, hipStream_t s) {      // bytes: a multiple of 16
  const size_t n = bytes / 16;
  if (n) hipLaunchKernelGGL(k_zero16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (uint4 *)p, n);
  return hipGetLastError();
}
hipError_t launch_lr_decide3(const unsigned long long *sse, int nframes, int stripes_y, int stripes_c, uint8_t *on, hipStream_t s) {
  hipLaunchKernelGGL(k_lr_decide3, dim3((unsigned)((nframes * 3 + 63) / 64)), dim3(64), 0, s, sse, nframes, stripes_y, stripes_c, on);
  return hipGetLastError();
}

int lr_stripes(int h, int ss) { return (h + (8 >> ss) + (64 >> ss) - 1) / (64 >> ss); }
hipError_t launch_lr_decide(const unsigned long long *sse, int nframes, int stripes, uint8_t *on, int on_stride, hipStream_t s) {
  hipLaunchKernelGGL(k_lr_decide, dim3((unsigned)((nframes + 63) / 64)), dim3(64), 0, s, sse, nframes, stripes, on, on_stride);
  return hipGetLastError();
}
hipError_t launch_lr(const LrLaunch &L, hipStream_t s) {
  const int SH = 64 >> L.ss, off = 8 >> L.ss;
  const int tw = L.unit_size < 64 ? L.unit_size : 64;
  const dim3 grid((unsigned)(((L.w + tw - 1) / tw) * ((L.h + off + SH - 1) / SH) * L.nframes));   // 1-D: the kernel orders the tiles (xcd_tile)
#define AV1MI_LR(PIX, SUBY) do { if (L.no_sgr) hipLaunchKernelGGL((k_lr<PIX, SUBY, false>), grid, dim3(256), 0, s, L); \
                                  else hipLaunchKernelGGL((k_lr<PIX, SUBY, true>), grid, dim3(256), 0, s, L); } while (0)
  if (L.ss) { if (L.bd == 8) AV1MI_LR(uint8_t, 1); else AV1MI_LR(uint16_t, 1); }
  else { if (L.bd == 8) AV1MI_LR(uint8_t, 0); else AV1MI_LR(uint16_t, 0); }
#undef AV1MI_LR
  return hipGetLastError();
}

}  // namespace av1mi
