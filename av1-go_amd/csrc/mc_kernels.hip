// mc_kernels.hip — SURVEY.md §8a row K4: sub-pel motion compensation of a list of equally-sized blocks
// (single reference, unscaled): separable 8-tap FIR at 1/16-sample phases with the spec's two-stage rounding.
//
// A block is BH lanes of one wave (64/BH blocks per wave, four waves per workgroup).  The group stages the
// (BW+7) x (BH+7) reference window in LDS (vector loads inside the plane, clamped coordinates = the spec's edge extension
// at the borders), filters rows into an int16 intermediate in LDS with dot-product instructions, then filters columns
// and writes the prediction as whole 4-sample runs.  The 16 phases x 6 filters live in constant memory; a 4-tap filter
// is an 8-tap row with zero outer taps, so every block takes the same path.  Traffic: reference window read
// ~ b*S*(1+7/BW)(1+7/BH) out of L2, prediction written b*S.
//
// Restates AV1 spec §7.11.3.4 == libaom av1_highbd_convolve_2d_sr_c; nothing to cite in the
// reference tree (internal/ffmpeg/transcode.go:120 names the external encoder only).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "av1mi_internal.hpp"

namespace av1mi {

__constant__ int16_t kSubpel[6][16][8] = {
  { { 0, 0, 0, 128, 0, 0, 0, 0 }, { 0, 2, -6, 126, 8, -2, 0, 0 }, { 0, 2, -10, 122, 18, -4, 0, 0 }, { 0, 2, -12, 116, 28, -8, 2, 0 },
    { 0, 2, -14, 110, 38, -10, 2, 0 }, { 0, 2, -14, 102, 48, -12, 2, 0 }, { 0, 2, -16, 94, 58, -12, 2, 0 }, { 0, 2, -14, 84, 66, -12, 2, 0 },
    { 0, 2, -14, 76, 76, -14, 2, 0 }, { 0, 2, -12, 66, 84, -14, 2, 0 }, { 0, 2, -12, 58, 94, -16, 2, 0 }, { 0, 2, -12, 48, 102, -14, 2, 0 },
    { 0, 2, -10, 38, 110, -14, 2, 0 }, { 0, 2, -8, 28, 116, -12, 2, 0 }, { 0, 0, -4, 18, 122, -10, 2, 0 }, { 0, 0, -2, 8, 126, -6, 2, 0 } },
  { { 0, 0, 0, 128, 0, 0, 0, 0 }, { 0, 2, 28, 62, 34, 2, 0, 0 }, { 0, 0, 26, 62, 36, 4, 0, 0 }, { 0, 0, 22, 62, 40, 4, 0, 0 },
    { 0, 0, 20, 60, 42, 6, 0, 0 }, { 0, 0, 18, 58, 44, 8, 0, 0 }, { 0, 0, 16, 56, 46, 10, 0, 0 }, { 0, -2, 16, 54, 48, 12, 0, 0 },
    { 0, -2, 14, 52, 52, 14, -2, 0 }, { 0, 0, 12, 48, 54, 16, -2, 0 }, { 0, 0, 10, 46, 56, 16, 0, 0 }, { 0, 0, 8, 44, 58, 18, 0, 0 },
    { 0, 0, 6, 42, 60, 20, 0, 0 }, { 0, 0, 4, 40, 62, 22, 0, 0 }, { 0, 0, 4, 36, 62, 26, 0, 0 }, { 0, 0, 2, 34, 62, 28, 2, 0 } },
  { { 0, 0, 0, 128, 0, 0, 0, 0 }, { -2, 2, -6, 126, 8, -2, 2, 0 }, { -2, 6, -12, 124, 16, -6, 4, -2 }, { -2, 8, -18, 120, 26, -10, 6, -2 },
    { -4, 10, -22, 116, 38, -14, 6, -2 }, { -4, 10, -22, 108, 48, -18, 8, -2 }, { -4, 10, -24, 100, 60, -20, 8, -2 },
    { -4, 10, -24, 90, 70, -22, 10, -2 }, { -4, 12, -24, 80, 80, -24, 12, -4 }, { -2, 10, -22, 70, 90, -24, 10, -4 },
    { -2, 8, -20, 60, 100, -24, 10, -4 }, { -2, 8, -18, 48, 108, -22, 10, -4 }, { -2, 6, -14, 38, 116, -22, 10, -4 },
    { -2, 6, -10, 26, 120, -18, 8, -2 }, { -2, 4, -6, 16, 124, -12, 6, -2 }, { 0, 2, -2, 8, 126, -6, 2, -2 } },
  { { 0, 0, 0, 128, 0, 0, 0, 0 }, { 0, 0, 0, 120, 8, 0, 0, 0 }, { 0, 0, 0, 112, 16, 0, 0, 0 }, { 0, 0, 0, 104, 24, 0, 0, 0 },
    { 0, 0, 0, 96, 32, 0, 0, 0 }, { 0, 0, 0, 88, 40, 0, 0, 0 }, { 0, 0, 0, 80, 48, 0, 0, 0 }, { 0, 0, 0, 72, 56, 0, 0, 0 },
    { 0, 0, 0, 64, 64, 0, 0, 0 }, { 0, 0, 0, 56, 72, 0, 0, 0 }, { 0, 0, 0, 48, 80, 0, 0, 0 }, { 0, 0, 0, 40, 88, 0, 0, 0 },
    { 0, 0, 0, 32, 96, 0, 0, 0 }, { 0, 0, 0, 24, 104, 0, 0, 0 }, { 0, 0, 0, 16, 112, 0, 0, 0 }, { 0, 0, 0, 8, 120, 0, 0, 0 } },
  { { 0, 0, 0, 128, 0, 0, 0, 0 }, { 0, 0, -4, 126, 8, -2, 0, 0 }, { 0, 0, -8, 122, 18, -4, 0, 0 }, { 0, 0, -10, 116, 28, -6, 0, 0 },
    { 0, 0, -12, 110, 38, -8, 0, 0 }, { 0, 0, -12, 102, 48, -10, 0, 0 }, { 0, 0, -14, 94, 58, -10, 0, 0 }, { 0, 0, -12, 84, 66, -10, 0, 0 },
    { 0, 0, -12, 76, 76, -12, 0, 0 }, { 0, 0, -10, 66, 84, -12, 0, 0 }, { 0, 0, -10, 58, 94, -14, 0, 0 }, { 0, 0, -10, 48, 102, -12, 0, 0 },
    { 0, 0, -8, 38, 110, -12, 0, 0 }, { 0, 0, -6, 28, 116, -10, 0, 0 }, { 0, 0, -4, 18, 122, -8, 0, 0 }, { 0, 0, -2, 8, 126, -4, 0, 0 } },
  { { 0, 0, 0, 128, 0, 0, 0, 0 }, { 0, 0, 30, 62, 34, 2, 0, 0 }, { 0, 0, 26, 62, 36, 4, 0, 0 }, { 0, 0, 22, 62, 40, 4, 0, 0 },
    { 0, 0, 20, 60, 42, 6, 0, 0 }, { 0, 0, 18, 58, 44, 8, 0, 0 }, { 0, 0, 16, 56, 46, 10, 0, 0 }, { 0, 0, 14, 54, 48, 12, 0, 0 },
    { 0, 0, 12, 52, 52, 12, 0, 0 }, { 0, 0, 12, 48, 54, 14, 0, 0 }, { 0, 0, 10, 46, 56, 16, 0, 0 }, { 0, 0, 8, 44, 58, 18, 0, 0 },
    { 0, 0, 6, 42, 60, 20, 0, 0 }, { 0, 0, 4, 40, 62, 22, 0, 0 }, { 0, 0, 4, 36, 62, 26, 0, 0 }, { 0, 0, 2, 34, 62, 30, 0, 0 } },
};

// the lanes of a block live in one wave: an LDS fence + wave barrier orders their hand-offs
#define AV1MI_MC_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ int mc_filter_index(int type, int dim) {
  if (dim <= 4) { if (type == 0 || type == 2) return 4; if (type == 1) return 5; }
  return type;
}

typedef short s16x2 __attribute__((ext_vector_type(2)));

// Block = BH lanes of one wave (64 / BH blocks per wave, 4 waves per workgroup); lane r owns window rows r, r + BH, ... in
// the horizontal pass and output row r in the vertical pass.  What k_inter_pipe taught applies here:
//  - a window row inside the plane is ONE unaligned vector load and whole-dword LDS stores (rows that stick out clamp
//    sample by sample: the spec's edge extension);
//  - LDS rows are read back as aligned dwords; 8 taps are two v_dot4_i32_i8 on samples biased to signed bytes, tap 3
//    apart (it reaches 128), for 8-bit content, four v_dot2_i32_i16 for 10-bit;
//  - the int16 intermediate is written 16 bytes at a time; the vertical pass reads 8 rows x 16 bytes per 8 output
//    columns, interleaves them into row pairs and sums with v_dot2_i32_i16.
template <int BW, int BH, typename Pix>
__global__ __launch_bounds__(256) void k_mc(McLaunch L) {
  constexpr int PER = 4 / (int)sizeof(Pix);                   // samples per dword
  constexpr int RW = BW + 7, RH = BH + 7;
  constexpr int WD = (RW + PER - 1) / PER + 1;                // dwords per window row (one spare for the chunked reads)
  constexpr int CH = BW < 8 ? BW : 8;                         // output columns per step
  constexpr int NBW = 64 / BH;                                // blocks per wave
  constexpr int IMS = BW + (BW < 8 ? 4 : 0);                  // intermediate row stride in int16 (rows stay 8-byte aligned)
  __shared__ __attribute__((aligned(16))) uint32_t win[4 * NBW][RH * WD];
  __shared__ __attribute__((aligned(16))) int16_t inter[4 * NBW][RH * IMS + 8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane / BH, r = lane % BH;
  const int b = (blockIdx.x * 4 + wave) * NBW + grp;
  if (b >= L.nblocks) return;                                 // groups only synchronise with themselves (same wave)
  const av1mi_mc_blk d = L.blocks[b];
  const int posx = d.x * 16 + d.mvx, posy = d.y * 16 + d.mvy;
  const int x0 = (posx >> 4) - 3, y0 = (posy >> 4) - 3;
  const int16_t *fxp = kSubpel[mc_filter_index(d.filt_x, BW)][posx & 15];
  const int16_t *fyp = kSubpel[mc_filter_index(d.filt_y, BH)][posy & 15];
  const Pix *ref = reinterpret_cast<const Pix *>(L.ref);
  uint32_t *wn = win[wave * NBW + grp];
  int16_t *im = inter[wave * NBW + grp];
  // ---- stage the window: rows r, r + BH, ...
  const bool inside = x0 >= 0 && x0 + (WD - 1) * PER <= L.plane_w;
  for (int j = r; j < RH; j += BH) {
    const Pix *row = ref + (size_t)min(max(y0 + j, 0), L.plane_h - 1) * L.ref_stride;
    uint32_t u[WD - 1];
    if (inside) __builtin_memcpy(u, row + x0, sizeof(u));
    else {
#pragma unroll
      for (int i = 0; i < WD - 1; i++) {
        u[i] = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) u[i] |= (uint32_t)row[min(max(x0 + i * PER + k, 0), L.plane_w - 1)] << (k * 8 * (int)sizeof(Pix));
      }
    }
#pragma unroll
    for (int i = 0; i < WD - 1; i++) wn[j * WD + i] = u[i];
  }
  AV1MI_MC_SYNC();
  int fx[8], fy[8];
#pragma unroll
  for (int t = 0; t < 8; t++) { fx[t] = fxp[t]; fy[t] = fyp[t]; }
  // ---- horizontal pass: rows r, r + BH, ... -> int16 intermediate (InterRound0 = 3)
  for (int j = r; j < RH; j += BH) {
#pragma unroll
    for (int c0 = 0; c0 < BW; c0 += CH) {
      int sum[CH];
      if constexpr (sizeof(Pix) == 1) {
        const int D0 = c0 / 4;                                // c0 is a multiple of 4 for every BW (unrolled: a constant)
        uint32_t a[4], bb[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { a[i] = (D0 + i < WD) ? wn[j * WD + D0 + i] : 0u; bb[i] = a[i] ^ 0x80808080u; }
        const int F0 = (fx[0] & 255) | ((fx[1] & 255) << 8) | ((fx[2] & 255) << 16);
        const int F1 = (fx[4] & 255) | ((fx[5] & 255) << 8) | ((fx[6] & 255) << 16) | ((fx[7] & 255) << 24);
        const int acc0 = 4 + 128 * (128 - fx[3]);             // pays the bias back: sum over t != 3 of 128 f_t
        uint32_t dd[CH + 4];
#pragma unroll
        for (int c = 0; c < CH + 4; c++) dd[c] = (c & 3) ? __builtin_amdgcn_alignbyte(bb[(c >> 2) + 1 < 4 ? (c >> 2) + 1 : 3], bb[c >> 2], c & 3) : bb[c >> 2];
#pragma unroll
        for (int c = 0; c < CH; c++) {
          const int p3 = (int)((a[(c + 3) >> 2] >> (((c + 3) & 3) * 8)) & 255);
          sum[c] = __builtin_amdgcn_sdot4(F0, (int)dd[c], __builtin_amdgcn_sdot4(F1, (int)dd[c + 4], acc0 + fx[3] * p3, false), false);
        }
      } else {
        const int D0 = c0 / 2;
        uint32_t a[8];
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = (D0 + i < WD) ? wn[j * WD + D0 + i] : 0u;
        uint32_t pm[CH + 7];
#pragma unroll
        for (int m = 0; m < CH + 7; m++) pm[m] = (m & 1) ? __builtin_amdgcn_alignbit(a[((m + 1) >> 1) < 8 ? (m + 1) >> 1 : 7], a[m >> 1], 16) : a[m >> 1];
        s16x2 fp[4];
#pragma unroll
        for (int u = 0; u < 4; u++) fp[u] = __builtin_bit_cast(s16x2, (uint32_t)(fx[2 * u] & 0xffff) | ((uint32_t)fx[2 * u + 1] << 16));
#pragma unroll
        for (int c = 0; c < CH; c++) {
          int acc = 4;
#pragma unroll
          for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, pm[c + 2 * u]), fp[u], acc, false);
          sum[c] = acc;
        }
      }
      uint32_t o[CH / 2];
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const uint32_t hh = (uint32_t)(sum[c] >> 3) & 0xffff;   // |value| < 2^15 for bd <= 10
        o[c >> 1] = (c & 1) ? (o[c >> 1] | (hh << 16)) : hh;
      }
      uint32_t *q = reinterpret_cast<uint32_t *>(im + j * IMS + c0);
#pragma unroll
      for (int i = 0; i < CH / 2; i++) q[i] = o[i];
    }
  }
  AV1MI_MC_SYNC();
  // ---- vertical pass: output row r (InterRound1 = 11, Clip1)
  Pix *dst = reinterpret_cast<Pix *>(L.dst) + (size_t)(d.y + r) * L.dst_stride + d.x;
  const int maxpix = (1 << L.bd) - 1;
  s16x2 gp[4];
#pragma unroll
  for (int u = 0; u < 4; u++) gp[u] = __builtin_bit_cast(s16x2, (uint32_t)(fy[2 * u] & 0xffff) | ((uint32_t)fy[2 * u + 1] << 16));
#pragma unroll
  for (int c0 = 0; c0 < BW; c0 += CH) {
    int sum[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) sum[c] = 1024;
#pragma unroll
    for (int kp = 0; kp < 4; kp++) {
      const uint32_t *qa = reinterpret_cast<const uint32_t *>(im + (r + 2 * kp) * IMS + c0);
      const uint32_t *qb = reinterpret_cast<const uint32_t *>(im + (r + 2 * kp + 1) * IMS + c0);
#pragma unroll
      for (int i = 0; i < CH / 2; i++) {
        const uint32_t x = qa[i], y = qb[i];
        sum[2 * i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, __builtin_amdgcn_perm(y, x, 0x05040100u)), gp[kp], sum[2 * i], false);
        sum[2 * i + 1] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, __builtin_amdgcn_perm(y, x, 0x07060302u)), gp[kp], sum[2 * i + 1], false);
      }
    }
    int o[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) o[c] = min(max(sum[c] >> 11, 0), maxpix);
#pragma unroll
    for (int c = 0; c < CH; c += 4) {
      Pix *q = dst + c0 + c;
      if constexpr (sizeof(Pix) == 1)
        *reinterpret_cast<uint32_t *>(q) = (uint32_t)o[c] | ((uint32_t)o[c + 1] << 8) | ((uint32_t)o[c + 2] << 16) | ((uint32_t)o[c + 3] << 24);
      else {
        uint2 u; u.x = (uint32_t)o[c] | ((uint32_t)o[c + 1] << 16); u.y = (uint32_t)o[c + 2] | ((uint32_t)o[c + 3] << 16);
        *reinterpret_cast<uint2 *>(q) = u;
      }
    }
  }
}

template <int BW, int BH> static void launch_one(const McLaunch &L, hipStream_t s) {
  constexpr int per_wg = 4 * (64 / BH);
  const int grid = (L.nblocks + per_wg - 1) / per_wg;
  if (L.bd == 8) hipLaunchKernelGGL((k_mc<BW, BH, uint8_t>), dim3(grid), dim3(256), 0, s, L);
  else hipLaunchKernelGGL((k_mc<BW, BH, uint16_t>), dim3(grid), dim3(256), 0, s, L);
}

// block size ids follow the TX_SIZE numbering (w x h): 0 4x4, 1 8x8, 2 16x16, 3 32x32, 4 64x64, 5 4x8, 6 8x4, ...
hipError_t launch_mc(int size_id, const McLaunch &L, hipStream_t s) {
  if (L.nblocks <= 0) return hipSuccess;
  switch (size_id) {
#define X(id, w, h) case id: launch_one<w, h>(L, s); break;
    X(0, 4, 4) X(1, 8, 8) X(2, 16, 16) X(3, 32, 32) X(4, 64, 64) X(5, 4, 8) X(6, 8, 4) X(7, 8, 16) X(8, 16, 8)
    X(9, 16, 32) X(10, 32, 16) X(11, 32, 64) X(12, 64, 32) X(13, 4, 16) X(14, 16, 4) X(15, 8, 32) X(16, 32, 8)
    X(17, 16, 64) X(18, 64, 16)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace av1mi
