// txfm_kernels.hip — SURVEY.md §8a rows K1 (forward 2-D transform), K2 (inverse 2-D transform +
// reconstruct) and K8 (quantise / dequantise) as hand-written gfx950 kernels.
//
// Work layout (all three kernels):
//   * a transform block is owned by LPB = max(rows, cols) consecutive lanes of ONE wave, so the
//     row pass -> transpose -> column pass hand-off needs no workgroup barrier, only LDS;
//   * coefficients stream HBM -> LDS with 16-byte-per-lane coalesced loads, the row pass runs one
//     row per lane in VGPRs (txfm1d.hpp), the LDS tile is the transpose buffer, the column pass runs
//     one column per lane, and the reconstruction is written back as whole pixel runs per lane;
//   * a 256-thread workgroup carries 256/LPB blocks; grids are >> 256 workgroups for any frame.
// Bound: HBM.  Algorithmic bytes per sample: K1 6, K2 4+2b, K8 6.
//
// Restates AV1 spec §7.13.3 / §7.12.3 and libaom inv_txfm2d_add_c / fwd_txfm2d_c /
// av1_quantize_fp; the reference holds no arithmetic for this path (transcode.go:120 names the
// external encoder only).
#include "txfm1d.hpp"
#include "txfm_cfg.hpp"
#include "av1mi_internal.hpp"

namespace av1mi {

template <int W, int H> struct TxGeom {
  static constexpr int CW = imin(W, 32), CH = imin(H, 32);   // stored coefficient extent
  static constexpr int LPB = imax(CH, W);                    // lanes per block (4..64, divides 64)
  static constexpr int NB = 256 / LPB;                       // blocks per workgroup
  static constexpr int RS = W + 4;                           // LDS row stride (dwords), padded
  static constexpr int BS = H * RS + 8;                      // LDS block stride (dwords), padded
};

template <typename Pix> struct PixVec4;
template <> struct PixVec4<uint8_t> { using T = uint32_t; };
template <> struct PixVec4<uint16_t> { using T = uint2; };

__device__ __forceinline__ void load_pix4(const uint8_t *p, int v[4]) {
  const uint32_t u = *reinterpret_cast<const uint32_t *>(p);
  v[0] = u & 255; v[1] = (u >> 8) & 255; v[2] = (u >> 16) & 255; v[3] = u >> 24;
}
__device__ __forceinline__ void load_pix4(const uint16_t *p, int v[4]) {
  const uint2 u = *reinterpret_cast<const uint2 *>(p);
  v[0] = u.x & 0xffff; v[1] = u.x >> 16; v[2] = u.y & 0xffff; v[3] = u.y >> 16;
}
__device__ __forceinline__ void store_pix4(uint8_t *p, const int v[4]) {
  *reinterpret_cast<uint32_t *>(p) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
}
__device__ __forceinline__ void store_pix4(uint16_t *p, const int v[4]) {
  uint2 u; u.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16); u.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
  *reinterpret_cast<uint2 *>(p) = u;
}

// resolve block b of the launch -> coefficient offset (int32 units), pixel offset, tx type
__device__ __forceinline__ void resolve_block(const TxLaunch &L, int b, int w, int h, int coef_per_blk,
                                              long long &coef_off, long long &pix_off, int &tx_type) {
  if (L.list) {
    const av1mi_txb d = L.list[b];
    coef_off = d.coef_off; pix_off = (long long)d.y * L.stride + d.x; tx_type = d.tx_type;
  } else {
    const int by = b / L.blocks_per_row, bx = b - by * L.blocks_per_row;
    coef_off = (long long)b * coef_per_blk;
    pix_off = (long long)by * h * L.stride + (long long)bx * w;
    tx_type = L.tx_types ? L.tx_types[b] : L.uniform_type;
  }
}

// ------------------------------------------------------------------------------------ K2
template <int W, int H, int BD>
__global__ __launch_bounds__(256) void k_inv_txfm_add(TxLaunch L) {
  using G = TxGeom<W, H>;
  using Pix = typename std::conditional<BD == 8, uint8_t, uint16_t>::type;
  constexpr int CW = G::CW, CH = G::CH, LPB = G::LPB, NB = G::NB, RS = G::RS, BS = G::BS;
  constexpr int ROW_RANGE = BD + 8, COL_RANGE = BD + 6 > 16 ? BD + 6 : 16;
  __shared__ __attribute__((aligned(16))) int32_t lds[NB * BS];

  const int tid = threadIdx.x;
  const int grp = tid / LPB, lane = tid % LPB;
  const int b = blockIdx.x * NB + grp;
  const bool live = b < L.nblocks;
  int32_t *tile = lds + grp * BS;
  long long coef_off = 0, pix_off = 0;
  int tx_type = 0;
  if (live) resolve_block(L, b, W, H, CW * CH, coef_off, pix_off, tx_type);
  const int rk = row_kind(tx_type), ck = col_kind(tx_type);
  const bool wht = W == 4 && H == 4 && tx_type == kWhtType;   // lossless: no clamps, no rounding, no final shift

  // phase 0: coalesced 16-byte loads HBM -> LDS tile (rows 0..CH-1, cols 0..CW-1)
  if (live) {
    const int4 *src = reinterpret_cast<const int4 *>(L.coef + coef_off);
#pragma unroll
    for (int q = lane; q < CW * CH / 4; q += LPB) {
      const int4 v = src[q];
      const int r = (q * 4) / CW, c = (q * 4) % CW;
      *reinterpret_cast<int4 *>(tile + r * RS + c) = v;
    }
  }
  __syncthreads();
  // phase 1: one row per lane
  if (live && lane < CH) {
    int32_t x[W];
#pragma unroll
    for (int c = 0; c < W; c += 4) {
      if (c < CW) {
        const int4 v = *reinterpret_cast<const int4 *>(tile + lane * RS + c);
        x[c] = v.x; x[c + 1] = v.y; x[c + 2] = v.z; x[c + 3] = v.w;
      } else { x[c] = x[c + 1] = x[c + 2] = x[c + 3] = 0; }
    }
#pragma unroll
    for (int c = 0; c < CW; c++) {
      int32_t v = x[c];
      if constexpr (is_rect2(W, H)) {
        // Round2(v * 2896, 12) in 32 bits: pre-clamping to +-2^ROW_RANGE cannot change the result,
        // because anything beyond it still lands outside the ROW_RANGE clamp after the 1/sqrt(2).
        v = min(max(v, -(1 << ROW_RANGE)), 1 << ROW_RANGE);
        v = (__mul24(v, kNewInvSqrt2) + 2048) >> 12;
      }
      x[c] = wht ? v : clampr<ROW_RANGE>(v);
    }
    if constexpr (W == 4 && H == 4) { if (wht) iwht4(x, 2); else inv1d<W, ROW_RANGE>(x, rk); }
    else inv1d<W, ROW_RANGE>(x, rk);
    constexpr int rsh = inv_row_shift(W, H);
#pragma unroll
    for (int c = 0; c < W; c += 4) {
      int4 v;
      v.x = round2(x[c], rsh); v.y = round2(x[c + 1], rsh); v.z = round2(x[c + 2], rsh); v.w = round2(x[c + 3], rsh);
      *reinterpret_cast<int4 *>(tile + lane * RS + c) = v;
    }
  }
  __syncthreads();
  // phase 2: one column per lane
  if (live && lane < W) {
    int32_t x[H];
    const int sc = rk == T1D_FLIPADST ? W - 1 - lane : lane;
#pragma unroll
    for (int r = 0; r < H; r++) x[r] = r < CH ? (wht ? tile[r * RS + sc] : clampr<COL_RANGE>(tile[r * RS + sc])) : 0;
    if constexpr (W == 4 && H == 4) { if (wht) iwht4(x, 0); else inv1d<H, COL_RANGE>(x, ck); }
    else inv1d<H, COL_RANGE>(x, ck);
    __builtin_amdgcn_wave_barrier();   // all lanes of the block have read their column (same wave)
    const bool ud = ck == T1D_FLIPADST;
#pragma unroll
    for (int r = 0; r < H; r++) {
      const int32_t v = wht ? x[r] : round2(x[r], 4);
      tile[(ud ? H - 1 - r : r) * RS + lane] = v;
    }
  }
  __syncthreads();
  // phase 3: reconstruction, 4 pixels per lane per step, rows of the block in order
  if (live) {
    Pix *dst = reinterpret_cast<Pix *>(L.plane) + pix_off;
    const int maxpix = (1 << BD) - 1;
#pragma unroll
    for (int q = lane; q < W * H / 4; q += LPB) {
      const int r = (q * 4) / W, c = (q * 4) % W;
      const int4 res = *reinterpret_cast<const int4 *>(tile + r * RS + c);
      int p[4];
      load_pix4(dst + (long long)r * L.stride + c, p);
      p[0] = min(max(p[0] + res.x, 0), maxpix); p[1] = min(max(p[1] + res.y, 0), maxpix);
      p[2] = min(max(p[2] + res.z, 0), maxpix); p[3] = min(max(p[3] + res.w, 0), maxpix);
      store_pix4(dst + (long long)r * L.stride + c, p);
    }
  }
}

// ------------------------------------------------------------------------------------ K1
// residual int16 (frame layout, stride in samples) -> int32 coefficients, block-contiguous,
// min(w,32) x min(h,32) row-major per block.
template <int W, int H>
__global__ __launch_bounds__(256) void k_fwd_txfm(TxLaunch L) {
  using G = TxGeom<W, H>;
  constexpr int CW = G::CW, CH = G::CH, LPB = G::LPB, NB = G::NB, RS = G::RS, BS = G::BS;
  __shared__ __attribute__((aligned(16))) int32_t lds[NB * BS];
  const int tid = threadIdx.x;
  const int grp = tid / LPB, lane = tid % LPB;
  const int b = blockIdx.x * NB + grp;
  const bool live = b < L.nblocks;
  int32_t *tile = lds + grp * BS;
  long long coef_off = 0, pix_off = 0;
  int tx_type = 0;
  if (live) resolve_block(L, b, W, H, CW * CH, coef_off, pix_off, tx_type);
  const int rk = row_kind(tx_type), ck = col_kind(tx_type);
  const bool wht = W == 4 && H == 4 && tx_type == kWhtType;
  // phase 0: residual rows -> LDS (4 samples = 8 bytes per lane per step)
  if (live) {
    const int16_t *src = reinterpret_cast<const int16_t *>(L.plane) + pix_off;
#pragma unroll
    for (int q = lane; q < W * H / 4; q += LPB) {
      const int r = (q * 4) / W, c = (q * 4) % W;
      const uint2 u = *reinterpret_cast<const uint2 *>(src + (long long)r * L.stride + c);
      int4 v;
      v.x = (int16_t)(u.x & 0xffff); v.y = (int16_t)(u.x >> 16); v.z = (int16_t)(u.y & 0xffff); v.w = (int16_t)(u.y >> 16);
      *reinterpret_cast<int4 *>(tile + r * RS + c) = v;
    }
  }
  __syncthreads();
  // phase 1: columns
  if (live && lane < W) {
    int32_t x[H];
    const bool ud = ck == T1D_FLIPADST;
#pragma unroll
    for (int r = 0; r < H; r++) x[r] = tile[(ud ? H - 1 - r : r) * RS + lane] << (wht ? 0 : fwd_shift(W, H, 0));
    if constexpr (W == 4 && H == 4) { if (wht) fwht4(x); else fwd1d<H, fwd_cos_bit_col(W, H)>(x, ck); }
    else fwd1d<H, fwd_cos_bit_col(W, H)>(x, ck);
    __builtin_amdgcn_wave_barrier();
    const int dc = rk == T1D_FLIPADST ? W - 1 - lane : lane;
    constexpr int s1 = -fwd_shift(W, H, 1);
#pragma unroll
    for (int r = 0; r < H; r++) tile[r * RS + dc] = wht ? x[r] : round2(x[r], s1);
  }
  __syncthreads();
  // phase 2: rows (only the CH stored rows)
  if (live && lane < CH) {
    int32_t x[W];
#pragma unroll
    for (int c = 0; c < W; c += 4) {
      const int4 v = *reinterpret_cast<const int4 *>(tile + lane * RS + c);
      x[c] = v.x; x[c + 1] = v.y; x[c + 2] = v.z; x[c + 3] = v.w;
    }
    if constexpr (W == 4 && H == 4) { if (wht) { fwht4(x); x[0] *= 4; x[1] *= 4; x[2] *= 4; x[3] *= 4; } else fwd1d<W, fwd_cos_bit_row(W, H)>(x, rk); }
    else fwd1d<W, fwd_cos_bit_row(W, H)>(x, rk);
    constexpr int s2 = -fwd_shift(W, H, 2);
#pragma unroll
    for (int c = 0; c < CW; c++) {
      int32_t v = round2(x[c], s2);
      if constexpr (is_rect2(W, H)) v = (__mul24(v, kNewSqrt2) + 2048) >> 12;  // |v| < 2^18: fits 32 bits
      x[c] = v;
    }
#pragma unroll
    for (int c = 0; c < CW; c += 4) {
      int4 v; v.x = x[c]; v.y = x[c + 1]; v.z = x[c + 2]; v.w = x[c + 3];
      *reinterpret_cast<int4 *>(tile + lane * RS + c) = v;
    }
  }
  __syncthreads();
  // phase 3: coalesced store of the block's coefficients
  if (live) {
    int4 *dst = reinterpret_cast<int4 *>(L.coef + coef_off);
#pragma unroll
    for (int q = lane; q < CW * CH / 4; q += LPB) {
      const int r = (q * 4) / CW, c = (q * 4) % CW;
      dst[q] = *reinterpret_cast<const int4 *>(tile + r * RS + c);
    }
  }
}

// ------------------------------------------------------------------------------------ K8
// Elementwise over block-contiguous coefficients: 4 coefficients per lane per step.
// position 0 of every block uses dc_q, the rest ac_q.  levels int16, optional dqcoef int32.
__global__ __launch_bounds__(256) void k_quantize(const int32_t *__restrict__ coef, int16_t *__restrict__ levels,
                                                  int32_t *__restrict__ dqcoef, long long n4, int coef_per_blk,
                                                  int dc_q, int ac_q, int log_scale) {
  const int dc_quant = (1 << 16) / dc_q, ac_quant = (1 << 16) / ac_q;
  const int dc_round = round2((64 * dc_q) >> 7, log_scale), ac_round = round2((64 * ac_q) >> 7, log_scale);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int4 c4 = reinterpret_cast<const int4 *>(coef)[i];
    const int c[4] = { c4.x, c4.y, c4.z, c4.w };
    const bool has_dc = ((i * 4) % coef_per_blk) == 0;
    int lv[4], dq[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool dc = has_dc && k == 0;
      const int q = dc ? dc_q : ac_q, quant = dc ? dc_quant : ac_quant, rnd = dc ? dc_round : ac_round;
      const bool sign = c[k] < 0;
      // |c| saturated at 2^20: beyond 32767 the level is already pinned by the int16 clamp below
      int a = (int)min((unsigned)(sign ? -(unsigned)c[k] : (unsigned)c[k]), 1u << 20);
      int l = 0;
      if ((a << (1 + log_scale)) >= q) {
        a = min(a + rnd, 32767);
        l = (a * quant) >> (16 - log_scale);
      }
      l = min(l, 32767);
      lv[k] = sign ? -l : l;
      const int d = (l * q) >> log_scale;
      dq[k] = sign ? -d : d;
    }
    uint2 o;
    o.x = (uint32_t)(lv[0] & 0xffff) | ((uint32_t)lv[1] << 16);
    o.y = (uint32_t)(lv[2] & 0xffff) | ((uint32_t)lv[3] << 16);
    reinterpret_cast<uint2 *>(levels)[i] = o;
    if (dqcoef) { int4 d4; d4.x = dq[0]; d4.y = dq[1]; d4.z = dq[2]; d4.w = dq[3]; reinterpret_cast<int4 *>(dqcoef)[i] = d4; }
  }
}
// spec §7.12.3: dq = ((|level| * q) & 0xFFFFFF) >> shift, sign restored, clamped to bd+8 bits signed.
__global__ __launch_bounds__(256) void k_dequantize(const int16_t *__restrict__ levels, int32_t *__restrict__ dqcoef,
                                                    long long n4, int coef_per_blk, int dc_q, int ac_q,
                                                    int log_scale, int bd) {
  const int maxv = (1 << (7 + bd)) - 1, minv = -(1 << (7 + bd));
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const uint2 u = reinterpret_cast<const uint2 *>(levels)[i];
    const int l[4] = { (int16_t)(u.x & 0xffff), (int16_t)(u.x >> 16), (int16_t)(u.y & 0xffff), (int16_t)(u.y >> 16) };
    const bool has_dc = ((i * 4) % coef_per_blk) == 0;
    int dq[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int q = (has_dc && k == 0) ? dc_q : ac_q;
      const bool sign = l[k] < 0;
      int d = (((sign ? -l[k] : l[k]) * q) & 0xFFFFFF) >> log_scale;
      if (sign) d = -d;
      dq[k] = min(max(d, minv), maxv);
    }
    int4 d4; d4.x = dq[0]; d4.y = dq[1]; d4.z = dq[2]; d4.w = dq[3];
    reinterpret_cast<int4 *>(dqcoef)[i] = d4;
  }
}

// ------------------------------------------------------------------------------------ launchers
static const int kTxW[19] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
static const int kTxH[19] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };

int tx_width(int tx_size) { return kTxW[tx_size]; }
int tx_height(int tx_size) { return kTxH[tx_size]; }

template <int W, int H> static void launch_inv(const TxLaunch &L, int bd, hipStream_t s) {
  constexpr int NB = TxGeom<W, H>::NB;
  const int grid = (L.nblocks + NB - 1) / NB;
  if (bd == 8) hipLaunchKernelGGL((k_inv_txfm_add<W, H, 8>), dim3(grid), dim3(256), 0, s, L);
  else hipLaunchKernelGGL((k_inv_txfm_add<W, H, 10>), dim3(grid), dim3(256), 0, s, L);
}
template <int W, int H> static void launch_fwd(const TxLaunch &L, hipStream_t s) {
  constexpr int NB = TxGeom<W, H>::NB;
  const int grid = (L.nblocks + NB - 1) / NB;
  hipLaunchKernelGGL((k_fwd_txfm<W, H>), dim3(grid), dim3(256), 0, s, L);
}

#define AV1MI_FOR_ALL_TX(X) \
  X(0, 4, 4) X(1, 8, 8) X(2, 16, 16) X(3, 32, 32) X(4, 64, 64) X(5, 4, 8) X(6, 8, 4) X(7, 8, 16) X(8, 16, 8) \
  X(9, 16, 32) X(10, 32, 16) X(11, 32, 64) X(12, 64, 32) X(13, 4, 16) X(14, 16, 4) X(15, 8, 32) X(16, 32, 8) \
  X(17, 16, 64) X(18, 64, 16)

hipError_t launch_inv_txfm(int tx_size, const TxLaunch &L, int bd, hipStream_t s) {
  if (L.nblocks <= 0) return hipSuccess;
  switch (tx_size) {
#define X(id, w, h) case id: launch_inv<w, h>(L, bd, s); break;
    AV1MI_FOR_ALL_TX(X)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_fwd_txfm(int tx_size, const TxLaunch &L, hipStream_t s) {
  if (L.nblocks <= 0) return hipSuccess;
  switch (tx_size) {
#define X(id, w, h) case id: launch_fwd<w, h>(L, s); break;
    AV1MI_FOR_ALL_TX(X)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_quantize(const int32_t *coef, int16_t *levels, int32_t *dqcoef, long long n, int coef_per_blk,
                           int dc_q, int ac_q, int log_scale, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const long long n4 = n / 4;
  const int grid = (int)imin(2048, (int)((n4 + 255) / 256));
  hipLaunchKernelGGL(k_quantize, dim3(grid), dim3(256), 0, s, coef, levels, dqcoef, n4, coef_per_blk, dc_q, ac_q, log_scale);
  return hipGetLastError();
}
hipError_t launch_dequantize(const int16_t *levels, int32_t *dqcoef, long long n, int coef_per_blk, int dc_q,
                             int ac_q, int log_scale, int bd, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const long long n4 = n / 4;
  const int grid = (int)imin(2048, (int)((n4 + 255) / 256));
  hipLaunchKernelGGL(k_dequantize, dim3(grid), dim3(256), 0, s, levels, dqcoef, n4, coef_per_blk, dc_q, ac_q, log_scale, bd);
  return hipGetLastError();
}

}  // namespace av1mi
