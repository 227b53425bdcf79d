// intra_kernels.hip — SURVEY.md §8a row K3 as a standalone gfx950 kernel: intra prediction of a list of
// equally-sized transform blocks from a reconstructed plane into a prediction plane.
//
// One block per BH consecutive lanes of one wave (one lane per row): the group builds both edges in LDS
// (intra.hpp), then each lane writes its row as whole 4-sample stores.  Blocks of a launch are independent
// (the caller guarantees their neighbours are already reconstructed in `ref`), so the grid is nblocks*BH/256
// workgroups.  Bound: HBM, b*S bytes written + < 4b*S/w read.
#include "intra.hpp"
#include "av1mi_internal.hpp"

namespace av1mi {

__device__ __forceinline__ void store4(uint8_t *p, const int *v) {
  *reinterpret_cast<uint32_t *>(p) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
}
__device__ __forceinline__ void store4(uint16_t *p, const int *v) {
  uint2 u; u.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16); u.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
  *reinterpret_cast<uint2 *>(p) = u;
}

template <int BW, int BH, typename Pix>
__global__ __launch_bounds__(256) void k_intra_pred(IntraLaunch L) {
  constexpr int NB = 256 / BH, EL = edge_len(BW, BH);
  __shared__ uint16_t lds[NB * 4 * EL];
  const int grp = threadIdx.x / BH, lane = threadIdx.x % BH;
  const int b = blockIdx.x * NB + grp;
  if (b >= L.nblocks) return;   // whole groups leave together; no workgroup barrier below
  const av1mi_intra_blk d = L.blocks[b];
  IntraBlk B;
  B.mode = d.mode; B.angle_delta = d.angle_delta; B.disable_edge_filter = d.flags & 1; B.filter_type = (d.flags >> 1) & 1;
  B.n_top = d.n_top; B.n_topright = d.n_topright; B.n_left = d.n_left; B.n_bottomleft = d.n_bottomleft;
  uint16_t *base = lds + grp * 4 * EL;
  uint16_t *above = base + kEdgePad, *left = base + EL + kEdgePad, *tmpa = base + 2 * EL + kEdgePad, *tmpl = base + 3 * EL + kEdgePad;
  const Pix *ref = reinterpret_cast<const Pix *>(L.ref) + (long long)d.y * L.ref_stride + d.x;
  const int rs = L.ref_stride;
  auto fetch = [&](int yy, int xx) -> int { return ref[(long long)yy * rs + xx]; };
  const IntraEdges E = intra_build_edges<BW, BH, BH>(B, L.bd, lane, above, left, tmpa, tmpl, fetch);
  int out[BW];
  intra_pred_row<BW, BH>(B, E, L.bd, lane, above, left, out);
  Pix *dst = reinterpret_cast<Pix *>(L.dst) + (long long)(d.y + lane) * L.dst_stride + d.x;
#pragma unroll
  for (int c = 0; c < BW; c += 4) store4(dst + c, out + c);
}

template <int BW, int BH> static void launch_one(const IntraLaunch &L, hipStream_t s) {
  constexpr int NB = 256 / BH;
  const int grid = (L.nblocks + NB - 1) / NB;
  if (L.bd == 8) hipLaunchKernelGGL((k_intra_pred<BW, BH, uint8_t>), dim3(grid), dim3(256), 0, s, L);
  else hipLaunchKernelGGL((k_intra_pred<BW, BH, uint16_t>), dim3(grid), dim3(256), 0, s, L);
}

hipError_t launch_intra_pred(int tx_size, const IntraLaunch &L, hipStream_t s) {
  if (L.nblocks <= 0) return hipSuccess;
  switch (tx_size) {
#define X(id, w, h) case id: launch_one<w, h>(L, s); break;
    X(0, 4, 4) X(1, 8, 8) X(2, 16, 16) X(3, 32, 32) X(4, 64, 64) X(5, 4, 8) X(6, 8, 4) X(7, 8, 16) X(8, 16, 8)
    X(9, 16, 32) X(10, 32, 16) X(11, 32, 64) X(12, 64, 32) X(13, 4, 16) X(14, 16, 4) X(15, 8, 32) X(16, 32, 8)
    X(17, 16, 64) X(18, 64, 16)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ chroma from luma
// Spec 7.11.5, 4:2:0.  Block = BH lanes of one wave, lane = row: two luma rows of 2 BW samples -> BW sums, the block average
// by cross-lane sums, then the row of the DC prediction in d_dst is corrected in place.  Restated in oracle/av1o_intra.c
// (av1o_cfl_predict); the reference has no counterpart.
template <int BW, int BH, typename Pix>
__global__ __launch_bounds__(256) void k_cfl_pred(CflLaunch L) {
  constexpr int NB = 256 / BH;
  const int grp = threadIdx.x / BH, lane = threadIdx.x % BH;
  const int b = blockIdx.x * NB + grp;
  if (b >= L.nblocks) return;
  const av1mi_cfl_blk d = L.blocks[b];
  const Pix *luma = reinterpret_cast<const Pix *>(L.luma);
  const int ly = min(2 * (d.y + lane), (int)d.max_luma_h - 2);
  const Pix *r0 = luma + (size_t)ly * L.luma_stride, *r1 = r0 + L.luma_stride;
  int Lq[BW], rowsum = 0;
  const bool inside = 2 * (d.x + BW) <= (int)d.max_luma_w;
  if (inside) {
    Pix a[2 * BW], c[2 * BW];
    __builtin_memcpy(a, r0 + 2 * d.x, sizeof(a));
    __builtin_memcpy(c, r1 + 2 * d.x, sizeof(c));
#pragma unroll
    for (int j = 0; j < BW; j++) Lq[j] = ((int)a[2 * j] + a[2 * j + 1] + c[2 * j] + c[2 * j + 1]) << 1;
  } else {
#pragma unroll
    for (int j = 0; j < BW; j++) {
      const int lx = min(2 * (d.x + j), (int)d.max_luma_w - 2);
      Lq[j] = ((int)r0[lx] + r0[lx + 1] + r1[lx] + r1[lx + 1]) << 1;
    }
  }
#pragma unroll
  for (int j = 0; j < BW; j++) rowsum += Lq[j];
  int total = group_sum<(BH > 16 ? 16 : BH)>(rowsum);
  if constexpr (BH == 32) total += __shfl_xor(total, 16, 64);      // the two 16-lane halves of the block
  constexpr int lg = (BW == 4 ? 2 : BW == 8 ? 3 : BW == 16 ? 4 : 5) + (BH == 4 ? 2 : BH == 8 ? 3 : BH == 16 ? 4 : 5);
  const int avg = (total + (1 << (lg - 1))) >> lg;
  Pix *dst = reinterpret_cast<Pix *>(L.dst) + (size_t)(d.y + lane) * L.dst_stride + d.x;
  const int maxpix = (1 << L.bd) - 1, alpha = d.alpha_q3;
  Pix dc[BW];
  __builtin_memcpy(dc, dst, sizeof(dc));
  int out[BW];
#pragma unroll
  for (int j = 0; j < BW; j++) {
    const int v = alpha * (Lq[j] - avg), m = v < 0 ? -v : v, sl = (m + 32) >> 6;
    out[j] = min(max((int)dc[j] + (v < 0 ? -sl : sl), 0), maxpix);
  }
#pragma unroll
  for (int c = 0; c < BW; c += 4) store4(dst + c, out + c);
}
template <int BW, int BH> static void launch_cfl_one(const CflLaunch &L, hipStream_t s) {
  constexpr int NB = 256 / BH;
  const int grid = (L.nblocks + NB - 1) / NB;
  if (L.bd == 8) hipLaunchKernelGGL((k_cfl_pred<BW, BH, uint8_t>), dim3(grid), dim3(256), 0, s, L);
  else hipLaunchKernelGGL((k_cfl_pred<BW, BH, uint16_t>), dim3(grid), dim3(256), 0, s, L);
}
hipError_t launch_cfl_pred(int tx_size, const CflLaunch &L, hipStream_t s) {
  if (L.nblocks <= 0) return hipSuccess;
  switch (tx_size) {   // CfL exists for chroma blocks up to 32x32
#define X(id, w, h) case id: launch_cfl_one<w, h>(L, s); break;
    X(0, 4, 4) X(1, 8, 8) X(2, 16, 16) X(3, 32, 32) X(5, 4, 8) X(6, 8, 4) X(7, 8, 16) X(8, 16, 8) X(9, 16, 32) X(10, 32, 16)
    X(13, 4, 16) X(14, 16, 4) X(15, 8, 32) X(16, 32, 8)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace av1mi
