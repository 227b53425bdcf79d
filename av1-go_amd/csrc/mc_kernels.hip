// mc_kernels.hip — SURVEY.md §8a row K4: sub-pel motion compensation of a list of equally-sized blocks
// (single reference, unscaled): separable 8-tap FIR at 1/16-sample phases with the spec's two-stage rounding.
//
// One block per wave, four per workgroup.  The wave stages the (BW+7) x (BH+7) reference window in LDS with
// coordinates clamped to the plane (the spec's edge extension), filters rows into an int16 intermediate of
// (BH+7) x BW in LDS, then filters columns and writes the prediction as whole 4-sample runs.  The 16 phases x
// 6 filters live in constant memory; a 4-tap filter is an 8-tap row with zero outer taps, so every block takes
// the same path.  Bound: HBM (reference window read ~ b*S*(1+7/BW)(1+7/BH) out of L2, prediction written b*S).
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

__device__ __forceinline__ int mc_filter_index(int type, int dim) {
  if (dim <= 4) { if (type == 0 || type == 2) return 4; if (type == 1) return 5; }
  return type;
}

template <int BW, int BH, typename Pix>
__global__ __launch_bounds__(256) void k_mc(McLaunch L) {
  constexpr int RW = BW + 7, RH = BH + 7, RS = RW + 1;     // reference window, padded row stride
  __shared__ uint16_t win[4][RH * RS];
  __shared__ int16_t inter[4][RH * BW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= L.nblocks) return;                              // whole waves leave; no workgroup barrier below
  const av1mi_mc_blk d = L.blocks[b];
  const int posx = d.x * 16 + d.mvx, posy = d.y * 16 + d.mvy;
  const int x0 = (posx >> 4) - 3, y0 = (posy >> 4) - 3;
  const int16_t *fx = kSubpel[mc_filter_index(d.filt_x, BW)][posx & 15];
  const int16_t *fy = kSubpel[mc_filter_index(d.filt_y, BH)][posy & 15];
  const Pix *ref = reinterpret_cast<const Pix *>(L.ref);
  uint16_t *wn = win[wave];
  int16_t *im = inter[wave];
  for (int i = lane; i < RH * RW; i += 64) {
    const int r = i / RW, c = i - r * RW;
    const int ry = min(max(y0 + r, 0), L.plane_h - 1), rx = min(max(x0 + c, 0), L.plane_w - 1);
    wn[r * RS + c] = ref[(size_t)ry * L.ref_stride + rx];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  int tx[8], ty[8];
#pragma unroll
  for (int t = 0; t < 8; t++) { tx[t] = fx[t]; ty[t] = fy[t]; }
  for (int i = lane; i < RH * BW; i += 64) {
    const int r = i / BW, c = i - r * BW;
    const uint16_t *p = wn + r * RS + c;
    int s = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) s += tx[t] * p[t];
    im[i] = (int16_t)((s + 4) >> 3);                       // InterRound0 = 3; |value| < 2^15 for bd <= 10
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  Pix *dst = reinterpret_cast<Pix *>(L.dst) + (size_t)d.y * L.dst_stride + d.x;
  const int maxpix = (1 << L.bd) - 1;
  for (int i = lane; i < BH * (BW / 4); i += 64) {
    const int r = i / (BW / 4), c = (i - r * (BW / 4)) * 4;
    int o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int s = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) s += ty[t] * im[(r + t) * BW + c + k];
      o[k] = min(max((s + 1024) >> 11, 0), maxpix);      // InterRound1 = 11, Clip1
    }
    Pix *q = dst + (size_t)r * L.dst_stride + c;
    if constexpr (sizeof(Pix) == 1)
      *reinterpret_cast<uint32_t *>(q) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
    else {
      uint2 u; u.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16); u.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
      *reinterpret_cast<uint2 *>(q) = u;
    }
  }
}

template <int BW, int BH> static void launch_one(const McLaunch &L, hipStream_t s) {
  const int grid = (L.nblocks + 3) / 4;
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
