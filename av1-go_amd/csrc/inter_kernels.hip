// inter_kernels.hip — the inter (P-frame) block pipeline of BASELINE config 3 as two gfx950 kernels:
//
//   k_me_int     integer motion search.  One workgroup per 64x64 luma tile: source tile and the reference window
//                (tile + search range, coordinates clamped = the spec's edge extension) staged in LDS; one 8x8 block
//                per wave at a time, ONE CANDIDATE VECTOR PER LANE, the 64 source samples broadcast from LDS;
//                (SAD, raster rank) packed into one integer and min-reduced across the wave with xor-shuffles.
//   k_inter_pipe per block, 8 lanes of one wave (8 blocks per wave, no dependency between blocks): half-pel refinement scored
//                with the bilinear filter (= rounded averages, no filter passes), quarter-pel refinement with the real 8-tap
//                sub-pel filter (K4 arithmetic, window in LDS), final
//                luma + chroma motion compensation, then the shared residual tail (K1 + K8 + K2, block_code.hpp),
//                skip flag and final vector.
// Inter blocks only depend on the PREVIOUS frame, so a P frame is embarrassingly parallel; the serial dependency is
// frame-to-frame inside a GOP and the pipeline batches the t-th frames of many segments into one launch.
// HBM traffic per sample (k_inter_pipe): source b + reference window (L2-served after first touch) ~b + reconstruction
// b + levels 2.
//
// Arithmetic: AV1 spec §7.11.3.4 (see mc_kernels.hip) + §7.13.3 / §7.12.3; search policy is this project's own and
// mirrored by oracle/av1o_pipeline.c:av1o_inter_encode_frame.  Reference tree: nothing (transcode.go:120).
#include "block_code.hpp"
#include "av1mi_internal.hpp"

namespace av1mi {

// regular 8-tap (Subpel_Filters[0]) and regular 4-tap (Subpel_Filters[4]) rows, the only ones this encoder policy uses
__constant__ int16_t kRegular8[16][8] = {
  { 0, 0, 0, 128, 0, 0, 0, 0 }, { 0, 2, -6, 126, 8, -2, 0, 0 }, { 0, 2, -10, 122, 18, -4, 0, 0 }, { 0, 2, -12, 116, 28, -8, 2, 0 },
  { 0, 2, -14, 110, 38, -10, 2, 0 }, { 0, 2, -14, 102, 48, -12, 2, 0 }, { 0, 2, -16, 94, 58, -12, 2, 0 }, { 0, 2, -14, 84, 66, -12, 2, 0 },
  { 0, 2, -14, 76, 76, -14, 2, 0 }, { 0, 2, -12, 66, 84, -14, 2, 0 }, { 0, 2, -12, 58, 94, -16, 2, 0 }, { 0, 2, -12, 48, 102, -14, 2, 0 },
  { 0, 2, -10, 38, 110, -14, 2, 0 }, { 0, 2, -8, 28, 116, -12, 2, 0 }, { 0, 0, -4, 18, 122, -10, 2, 0 }, { 0, 0, -2, 8, 126, -6, 2, 0 } };
__constant__ int16_t kRegular4[16][8] = {
  { 0, 0, 0, 128, 0, 0, 0, 0 }, { 0, 0, -4, 126, 8, -2, 0, 0 }, { 0, 0, -8, 122, 18, -4, 0, 0 }, { 0, 0, -10, 116, 28, -6, 0, 0 },
  { 0, 0, -12, 110, 38, -8, 0, 0 }, { 0, 0, -12, 102, 48, -10, 0, 0 }, { 0, 0, -14, 94, 58, -10, 0, 0 }, { 0, 0, -12, 84, 66, -10, 0, 0 },
  { 0, 0, -12, 76, 76, -12, 0, 0 }, { 0, 0, -10, 66, 84, -12, 0, 0 }, { 0, 0, -10, 58, 94, -14, 0, 0 }, { 0, 0, -10, 48, 102, -12, 0, 0 },
  { 0, 0, -8, 38, 110, -12, 0, 0 }, { 0, 0, -6, 28, 116, -10, 0, 0 }, { 0, 0, -4, 18, 122, -8, 0, 0 }, { 0, 0, -2, 8, 126, -4, 0, 0 } };

// the plane frame f predicts from: the restored one, or (ref_sel given and 0 for this frame and plane) the CDEF output — the
// restoration on / off decision of the previous frame (lr_kernel.hip k_lr_decide) without a copy
// (the flag is read as the aligned dword that holds it: f is uniform in a workgroup, so this is a scalar load)
__device__ __forceinline__ const void *ref_plane(const InterLaunch &L, int f, int p) {
  if (!L.ref_sel) return L.ref[p];
  const int i = f * 3 + p;
  const uint32_t w = reinterpret_cast<const uint32_t *>(L.ref_sel)[i >> 2];
  return ((w >> (8 * (i & 3))) & 0xffu) ? L.ref[p] : L.ref_alt[p];
}

// ------------------------------------------------------------------------------------------ integer search
// The search runs on the 8 most significant bits of the samples (10-bit content is shifted down by 2: encoder policy,
// mirrored by the oracle), which lets one v_qsad_pk_u16_u8 score FOUR horizontally adjacent candidate vectors against
// four source samples at once.  A lane owns (dy, group of 4 dx): per block row it reads 12 reference bytes (three
// aligned dwords) and issues two QSADs with the row's two source dwords, which all lanes read from the same LDS address.
// RC: the search range as a compile-time constant (0 = take L.range at run time): with the default +-8 every quotient, item
// count and validity bound below is a constant.
template <typename Pix, int RC>
__global__ __launch_bounds__(256) void k_me_int(InterLaunch L) {
  // window row stride in bytes: 41 dwords.  The window reads are ds_read2_b32 / ds_read_b32 (32 banks per 32-lane group); a lane
  // reads dword (by 8 + 3 dp + r) 41 + 2 bx + g (+ 0, 1, 2): 3 x 41 = 27 mod 32, so the six dy triples of a block start 5 banks
  // apart (0, 27, 22, 17, 12, 7) and their five dx groups fill the gaps: 30 distinct banks.  That holds per BLOCK, so a block's
  // 30 items are padded to 32 (one 32-lane group = one block; 16 x 32 items are the same 8 rounds as 16 x 30).  With 35-dword
  // rows and 30 items per block every group straddled two blocks and two triples shared a bank: SQ_LDS_BANK_CONFLICT was 55 % of
  // the LDS cycles.
  constexpr int MAXR = 16, WS = 164;
  static_assert(WS >= 64 + 2 * MAXR + 4 && WS % 4 == 0, "window row must hold tile + range and be whole dwords");
  __shared__ __attribute__((aligned(16))) uint8_t win[(64 + 2 * MAXR) * WS];
  __shared__ __attribute__((aligned(16))) uint8_t srct[64 * 64];
  __shared__ uint32_t s_best[64];                    // per 8x8 block: min of SAD << 16 | rank
  constexpr int sh = sizeof(Pix) == 1 ? 0 : 2;
  const int tid = threadIdx.x, R = RC ? RC : L.range, R4 = (R + 3) & ~3, NC = 2 * R + 1;
  const int WDX = 64 + 2 * R4 + 4, WDY = 64 + 2 * R + 2;    // window columns start at x - R4 (4-aligned), rows at y - R; two more rows for the unused part of the last dy triple
  const int sbw = (L.w + 63) / 64;
  const Tile3 tl = xcd_tile(sbw, (L.h + 63) / 64, L.nframes);
  const int f = tl.z, sby = tl.y, sbx = tl.x;
  const Pix *src = reinterpret_cast<const Pix *>(L.src[0]) + (size_t)f * L.h * L.stride_y;
  const Pix *ref = reinterpret_cast<const Pix *>(ref_plane(L, f, 0)) + (size_t)f * L.h * L.stride_y;
  // staging, four samples per lane per step (the window starts on a 4-sample boundary and its width is a multiple of 4);
  // groups that lie inside the plane are one vector load, the others clamp sample by sample (the spec's edge extension)
  const int wg = WDX >> 2;
  for (int i = tid; i < WDY * wg; i += 256) {
    const int r = i / wg, c = (i - r * wg) * 4;
    const int fy = min(max(sby * 64 - R + r, 0), L.h - 1), fx = sbx * 64 - R4 + c;
    const Pix *row = ref + row_off(fy, L.stride_y);
    uint32_t u;
    if (fx >= 0 && fx + 3 < L.w) {
      if constexpr (sizeof(Pix) == 1) u = *reinterpret_cast<const uint32_t *>(row + fx);
      else { const uint2 v = *reinterpret_cast<const uint2 *>(row + fx); u = ((v.x >> sh) & 0xff) | (((v.x >> (16 + sh)) & 0xff) << 8) | (((v.y >> sh) & 0xff) << 16) | (((v.y >> (16 + sh)) & 0xff) << 24); }
    } else {
      u = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) u |= (uint32_t)((row[min(max(fx + k, 0), L.w - 1)] >> sh) & 0xff) << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(win + r * WS + c) = u;
  }
  for (int i = tid; i < 64 * 16; i += 256) {
    const int r = i >> 4, c = (i & 15) * 4;
    const int fy = min(sby * 64 + r, L.h - 1), fx = sbx * 64 + c;
    const Pix *row = src + row_off(fy, L.stride_y);
    uint32_t u;
    if (fx + 3 < L.w) {
      if constexpr (sizeof(Pix) == 1) u = *reinterpret_cast<const uint32_t *>(row + fx);
      else { const uint2 v = *reinterpret_cast<const uint2 *>(row + fx); u = ((v.x >> sh) & 0xff) | (((v.x >> (16 + sh)) & 0xff) << 8) | (((v.y >> sh) & 0xff) << 16) | (((v.y >> (16 + sh)) & 0xff) << 24); }
    } else {
      u = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) u |= (uint32_t)((row[min(fx + k, L.w - 1)] >> sh) & 0xff) << (8 * k);
    }
    *reinterpret_cast<uint32_t *>(srct + r * 64 + c) = u;
  }
  if (tid < 64) s_best[tid] = 0xFFFFFFFFu;
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int bw = L.w / 8, bh = L.h / 8;
  const int NG = (2 * R4) / 4 + 1;                 // groups of four dx starting at -R4
  constexpr int HD = 3;                            // vertical displacements per item
  const int NP = (NC + HD - 1) / HD;               // triples of vertical displacements (the last one may hold fewer)
  const int items = NP * NG;                       // (dy triple, dx group) items per block
  const int per = RC == 8 ? 32 : items;            // lanes per block: padded to a 32-lane group when the count is known to be 30
  int16_t *mvs = L.mvs + (size_t)f * bw * bh * 2;
  // A lane scores THREE vertically adjacent displacements of four horizontal ones: they share six of their eight window
  // rows, so ten rows of three dwords and ONE copy of the source block's even rows are read for 24 QSADs (pairs: nine rows and a source
  // copy per 32; the QSADs are half of the kernel's cycles, the reads and the per-item arithmetic the other half, and +-8 is
  // 17 = 6 x 3 - 1 rows: as little padding as pairs).  The 16 blocks of a wave form ONE item space (16 x per): with +-8 a
  // block has 30 items, which alone would leave a 64-lane wave half empty; the per-block minimum is an LDS atomic instead of a
  // wave reduction.
  // item -> (block, dy pair, dx group) by reciprocal multiplication: u < 16 x 144 and (u + 0.5) / per is never closer than
  // 0.5 / per to an integer, far outside float rounding, so the truncation is the exact quotient (an integer division by a
  // run-time divisor is ~25 instructions, and there are two per item)
  const float inv_per = 1.0f / (float)per, inv_ng = 1.0f / (float)NG;
  // what an item (dy triple dp, dx group g) contributes whatever the block: its window offset, the ranks of its 12 displacements
  // (~0 = outside +-R) and whether it holds the zero vector.  With 32 items per block a lane has the SAME item in every round, so
  // all of it is computed once, before the loop (the compiler does not hoist it out of the rounds' conditional bodies by itself).
  struct Item { int off; unsigned rk[HD][4]; unsigned not_zero_item; };
  auto make_item = [&](int t) {
    Item it;
    const int dp = (int)(((float)t + 0.5f) * inv_ng), g = t - __mul24(dp, NG);    // dy = 3 dp - R + {0, 1, 2}, dx0 = -R4 + 4 g
    it.off = __mul24(HD * dp, WS) + 4 * g;
    // rank: (0,0) ranks first (0), the others in raster order (1 + (dy + R) NC + dx + R < 1024); dx_i = 4 g - R4 + i is in range for
    // i in [imin, imax]; row h of the triple exists when 3 dp + h < NC
    const int imin = R4 - R - 4 * g, imax = R4 + R - 4 * g;
    const unsigned rank0 = (unsigned)(__mul24(HD * dp, NC) + 4 * g - R4 + R + 1);
#pragma unroll
    for (int h = 0; h < HD; h++)
#pragma unroll
      for (int i = 0; i < 4; i++)
        it.rk[h][i] = (rank0 + h * NC + i) | (unsigned)(((i - imin) | (imax - i)) >> 31) | (unsigned)((NC - 1 - h - HD * dp) >> 31);
    it.not_zero_item = (dp == R / HD && g == (R4 >> 2)) ? 0u : 0xFFFFFFFFu;     // dy index R = row R % 3 of triple R / 3, dx index R4 = sample 0 of group R4 >> 2
    return it;
  };
  Item fixed = {};
  if constexpr (RC == 8) fixed = make_item(lane & 31);
  for (int u0 = 0; u0 < 16 * per; u0 += 64) {
    const int u = u0 + lane;
    if (u < 16 * per) {
      // (with 32 items per block the item of a lane — its dy triple, dx group, ranks and validity masks — is the same in every round:
      // written so that the compiler sees it and keeps them in registers across the unrolled rounds)
      const int bi = RC == 8 ? (u0 >> 5) + (lane >> 5) : (int)(((float)u + 0.5f) * inv_per), t = RC == 8 ? (lane & 31) : u - __mul24(bi, per), b = wave + 4 * bi;   // (plain products here became 64-bit multiply-adds)
      const int by = b >> 3, bx = b & 7;
      if (t < items && sbx * 8 + bx < bw && sby * 8 + by < bh) {
        const Item it = RC == 8 ? fixed : make_item(t);
        const uint8_t *p = win + __mul24(by * 8, WS) + bx * 8 + it.off;
        const uint8_t *s = srct + (by * 8) * 64 + bx * 8;
        // the SAD runs over the block's EVEN rows (policy; the oracle's block_sad8): half the QSADs, the same vectors on the test clips
        uint2 sr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) sr[r] = *reinterpret_cast<const uint2 *>(s + 2 * r * 64);
        unsigned long long acc[HD] = { 0, 0, 0 };
#pragma unroll
        for (int r = 0; r < 8 + HD - 1; r++) {
          const uint32_t *q = reinterpret_cast<const uint32_t *>(p + r * WS);   // row 3 dp + r <= 2 R + 11: inside the staged window
          // (q1, q2) as a 64-bit operand costs two register moves per row — an even-aligned pair — but reading the row as two overlapping
          // 8-byte pieces instead is slower: +23 % LDS instructions, 0.300 -> 0.311 ms)
          const unsigned long long w01 = (unsigned long long)q[0] | ((unsigned long long)q[1] << 32);
          const unsigned long long w12 = (unsigned long long)q[1] | ((unsigned long long)q[2] << 32);
#pragma unroll
          for (int h = 0; h < HD; h++)
            if (r - h >= 0 && r - h < 8 && ((r - h) & 1) == 0) {
              acc[h] = __builtin_amdgcn_qsad_pk_u16_u8(w01, sr[(r - h) >> 1].x, acc[h]);
              acc[h] = __builtin_amdgcn_qsad_pk_u16_u8(w12, sr[(r - h) >> 1].y, acc[h]);
            }
        }
        // key = SAD << 16 | rank; ties keep the lower rank; a displacement outside +-R has rank ~0, which turns its key into ~0 under
        // the OR.  With the SAD in the upper half a candidate's key is ONE instruction from the packed QSAD result:
        // (word << 16) | rank or (word & 0xFFFF0000) | rank.
        unsigned best = 0xFFFFFFFFu;
#pragma unroll
        for (int h = 0; h < HD; h++) {
          const uint32_t lo = (uint32_t)acc[h], hi = (uint32_t)(acc[h] >> 32);
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const uint32_t word = i < 2 ? lo : hi;
            best = min(best, ((i & 1) ? (word & 0xFFFF0000u) : (word << 16)) | it.rk[h][i]);
          }
        }
        {
          const int hz = R % HD;      // the zero vector
          const unsigned sz = (unsigned)(hz == 0 ? acc[0] : hz == 1 ? acc[1] : acc[2]) << 16;
          best = min(best, sz | it.not_zero_item);
        }
        atomicMin(&s_best[b], best);
      }
    }
  }
  __syncthreads();
  if (tid < 64) {
    const int by = tid >> 3, bx = tid & 7;
    const int fbx = sbx * 8 + bx, fby = sby * 8 + by;
    if (fbx < bw && fby < bh) {
      const int rank = s_best[tid] & 0xFFFF;
      const int dy = rank ? (rank - 1) / NC - R : 0, dx = rank ? (rank - 1) % NC - R : 0;
      mvs[((size_t)fby * bw + fbx) * 2] = (int16_t)(dx * 8);
      mvs[((size_t)fby * bw + fbx) * 2 + 1] = (int16_t)(dy * 8);
    }
  }
}

// ------------------------------------------------------------------------------------------ refinement + coding
// 2-D sub-pel prediction of row `lane` of a B x B block from a window in LDS (origin = integer position - 4, i.e. the
// window starts 4 samples left of / above the integer-vector block): posx/posy = displacement from the window's block
// origin in 1/16 samples, in [-16, 15].  im: (B+7) x B int16 scratch of the group.  NL = lanes of the block.
typedef short s16x2 __attribute__((ext_vector_type(2)));
// The first v_dot2 of an accumulation with its start value as an operand.  The compiler always picks the two-address form
// (v_dot2c: accumulator = destination) and initialises every accumulator with a v_mov first — one extra instruction per output
// sample in loops that are nothing but dot products; the three-address VOP3P form takes the start value from an SGPR.
__device__ __forceinline__ int dot2_start(uint32_t a, uint32_t b, int start_uniform) {
  int d;
  asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(start_uniform));
  return d;
}
// One window row of N samples from plane row `row`, columns x0 .. x0 + N - 1, into the dword-aligned LDS row `dst`.
// Inside the plane it is one unaligned vector load (the hardware takes any address) and N * sizeof(Pix) / 4 dword stores;
// rows that stick out clamp sample by sample (the spec's edge extension).
template <int N, typename Pix> __device__ __forceinline__ void stage_window_row(const Pix *row, int x0, int w, Pix *dst) {
  constexpr int ND = N * (int)sizeof(Pix) / 4;
  uint32_t u[ND];
  if (x0 >= 0 && x0 + N <= w) __builtin_memcpy(u, row + x0, sizeof(u));
  else {
    constexpr int PER = 4 / (int)sizeof(Pix);
#pragma unroll
    for (int d = 0; d < ND; d++) {
      u[d] = 0;
#pragma unroll
      for (int k = 0; k < PER; k++) u[d] |= (uint32_t)row[min(max(x0 + d * PER + k, 0), w - 1)] << (k * 8 * (int)sizeof(Pix));
    }
  }
  uint32_t *q = reinterpret_cast<uint32_t *>(dst);
#pragma unroll
  for (int d = 0; d < ND; d++) q[d] = u[d];
}
// N consecutive samples starting `ox` samples into a dword-aligned LDS row: aligned dword reads + funnel shift, so that the
// row costs (N * sizeof(ES)) / 4 + 1 LDS reads instead of N (k_inter_pipe is LDS-bound).  ox in [0, 4 / sizeof(ES)).
template <int N, typename ES> __device__ __forceinline__ void row_samples(const ES *row, int ox, int *v) {
  const uint32_t *q = reinterpret_cast<const uint32_t *>(row);
  constexpr int PER = 4 / (int)sizeof(ES), ND = (N + PER - 1) / PER;
  uint32_t w[ND + 1], a[ND];
#pragma unroll
  for (int i = 0; i <= ND; i++) w[i] = q[i];
#pragma unroll
  for (int i = 0; i < ND; i++) {
    if constexpr (sizeof(ES) == 1) a[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], ox);
    else a[i] = __builtin_amdgcn_alignbit(w[i + 1], w[i], ox * 16);
  }
#pragma unroll
  for (int i = 0; i < N; i++) v[i] = (a[i / PER] >> ((i % PER) * 8 * (int)sizeof(ES))) & (sizeof(ES) == 1 ? 255u : 0xffffu);
}

// T0, T1: the filter family's non-zero taps are T0 .. T1 - 1 (the 4-tap families of blocks <= 4 wide: 2 .. 5; the other taps
// are exactly 0, so leaving them out changes nothing but the work: 7 intermediate rows of 4 taps instead of 11 of 8).
template <int B, typename ES, int T0 = 0, int T1 = 8>
__device__ __forceinline__ void mc_row(const ES *win, int ws, int16_t *im, int lane, int posx, int posy, const int16_t (*filt)[8], int bd,
                                       int *out) {
  const int ox = 4 + (posx >> 4) - 3, oy = 4 + (posy >> 4) - 3;   // window column / row of tap 0 of sample (0,0)
  constexpr int NT = T1 - T0, NR = B + NT - 1, PER = 4 / (int)sizeof(ES);
  int fx[NT], fy[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) { fx[t] = filt[posx & 15][T0 + t]; fy[t] = filt[posy & 15][T0 + t]; }
#pragma unroll
  for (int it = 0; it < (NR + B - 1) / B; it++) {   // intermediate rows T0 .. T0 + NR - 1 over B lanes
    const int j = T0 + lane + it * B;
    if (j < T0 + NR) {
      int v[NR];
      const int o = ox + T0;                        // first sample of the row that a non-zero tap touches
      row_samples<NR, ES>(win + (oy + j) * ws + (o & ~(PER - 1)), o & (PER - 1), v);
#pragma unroll
      for (int c = 0; c < B; c++) {
        int s = 0;
#pragma unroll
        for (int t = 0; t < NT; t++) s += fx[t] * v[c + t];
        im[j * B + c] = (int16_t)((s + 4) >> 3);
      }
    }
  }
  AV1MI_GROUP_SYNC();
  const int maxpix = (1 << bd) - 1;
#pragma unroll
  for (int c = 0; c < B; c++) {
    int s = 0;
#pragma unroll
    for (int t = 0; t < NT; t++) s += fy[t] * (int)im[(lane + T0 + t) * B + c];
    out[c] = min(max((s + 1024) >> 11, 0), maxpix);
  }
  AV1MI_GROUP_SYNC();
}

// The same filter split in two, so that the candidates of a refinement round that share a horizontal phase share its
// (more expensive) horizontal pass: mc_h16 filters all 16 rows of the luma window for one horizontal displacement,
// mc_rows7 + mc_v7 produce row `lane` of the prediction for the vertical displacements from that intermediate.
template <typename ES>
__device__ __forceinline__ void mc_h16(const ES *win, int ws, int16_t *im, int lane, int posx, const int16_t (*filt)[8]) {
  const int ox = 4 + (posx >> 4) - 3;            // 0 or 1
  int fx[8];
  if constexpr (sizeof(ES) == 1) {
#pragma unroll
    for (int t = 0; t < 8; t++) fx[t] = filt[posx & 15][t];
  }
  constexpr int PER = 4 / (int)sizeof(ES), ND = 16 / PER;
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int j = lane + it * 8;
    // the row's 16 samples from ox on, as aligned dwords funnel-shifted by ox samples
    const uint32_t *q = reinterpret_cast<const uint32_t *>(win + j * ws);
    uint32_t w[ND + 1], a[ND];
#pragma unroll
    for (int i = 0; i <= ND; i++) w[i] = q[i];
#pragma unroll
    for (int i = 0; i < ND; i++) {
      if constexpr (sizeof(ES) == 1) a[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], ox);
      else a[i] = __builtin_amdgcn_alignbit(w[i + 1], w[i], ox * 16);
    }
    int sum[8];
    if constexpr (sizeof(ES) == 1) {
      // 8 taps = two v_dot4_i32_i8 on the samples biased to signed bytes (p - 128), except tap 3 (it reaches 128, one more
      // than a signed byte holds), which is a separate multiply-add on the unbiased sample.  The bias is paid back in the
      // accumulator: sum over t != 3 of f_t * 128 = 128 * (128 - f_3).
      uint32_t b[4];
#pragma unroll
      for (int i = 0; i < 4; i++) b[i] = a[i] ^ 0x80808080u;
      const int F0 = (fx[0] & 255) | ((fx[1] & 255) << 8) | ((fx[2] & 255) << 16);
      const int F1 = (fx[4] & 255) | ((fx[5] & 255) << 8) | ((fx[6] & 255) << 16) | ((fx[7] & 255) << 24);
      const int acc0 = 4 + 128 * (128 - fx[3]);
      uint32_t d[12];
#pragma unroll
      for (int c = 0; c < 12; c++) d[c] = (c & 3) ? __builtin_amdgcn_alignbyte(b[(c >> 2) + 1], b[c >> 2], c & 3) : b[c >> 2];
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const int p3 = (int)((a[(c + 3) >> 2] >> (((c + 3) & 3) * 8)) & 255);
        sum[c] = __builtin_amdgcn_sdot4(F0, (int)d[c], __builtin_amdgcn_sdot4(F1, (int)d[c + 4], acc0 + fx[3] * p3, false), false);
      }
    } else {
      // 10-bit: the regular filter has SIX taps (taps 0 and 7 are 0 at every phase): sample pairs (m, m + 1) for m = 1..12 (odd
      // m by funnel shift) against the tap pairs (t1, t2) (t3, t4) (t5, t6) = three v_dot2_i32_i16 per sample
      uint32_t pm[13];
#pragma unroll
      for (int m = 1; m < 13; m++) pm[m] = (m & 1) ? __builtin_amdgcn_alignbit(a[(m + 1) >> 1], a[m >> 1], 16) : a[m >> 1];
      // the tap row is 8 int16 = (t0, t1) (t2, t3) (t4, t5) (t6, t7) as they lie in memory: the pairs one tap further by funnel shift
      const uint4 fq = *reinterpret_cast<const uint4 *>(filt[posx & 15]);
      const uint32_t g1 = __builtin_amdgcn_alignbit(fq.y, fq.x, 16), g2 = __builtin_amdgcn_alignbit(fq.z, fq.y, 16), g3 = __builtin_amdgcn_alignbit(fq.w, fq.z, 16);
#pragma unroll
      for (int c = 0; c < 8; c++) {
        int acc = dot2_start(pm[c + 1], g1, 128);   // taps x 32 (see below): the rounding constant 4 x 32
        acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, pm[c + 3]), __builtin_bit_cast(s16x2, g2), acc, false);
        acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, pm[c + 5]), __builtin_bit_cast(s16x2, g3), acc, false);
        sum[c] = acc;
      }
    }
    uint32_t o[4];
    if constexpr (sizeof(ES) == 1) {
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const uint32_t h = (uint32_t)(sum[c] >> 3) & 0xffff;
        o[c >> 1] = (c & 1) ? (o[c >> 1] | (h << 16)) : h;
      }
    } else {
      // 10-bit: `filt` is the x 32 table, (32 s + 128) >> 8 == (s + 4) >> 3: the int16 result is bytes 1..2 of the accumulator, and
      // one v_perm shifts and packs two of them (|32 s| < 2^23: no overflow)
#pragma unroll
      for (int j = 0; j < 4; j++) o[j] = __builtin_amdgcn_perm((uint32_t)sum[2 * j + 1], (uint32_t)sum[2 * j], 0x06050201u);
    }
    *reinterpret_cast<uint4 *>(im + j * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
// mc_h16 for a whole-sample horizontal position (posx = 0): phase 0 is the identity, (128 p + 4) >> 3 = 16 p exactly.
template <typename ES>
__device__ __forceinline__ void mc_h16_copy(const ES *win, int ws, int16_t *im, int lane) {
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int j = lane + it * 8;
    uint32_t o[4];
    if constexpr (sizeof(ES) == 2) {
      const uint4 a = *reinterpret_cast<const uint4 *>(win + j * ws + 4);        // samples 4 .. 11 of the row, two per dword
      o[0] = a.x << 4; o[1] = a.y << 4; o[2] = a.z << 4; o[3] = a.w << 4;         // p < 2^12: no carry between the halves
    } else {
      const uint2 a = *reinterpret_cast<const uint2 *>(win + j * ws + 4);
      o[0] = __builtin_amdgcn_perm(0u, a.x, 0x0c010c00u) << 4; o[1] = __builtin_amdgcn_perm(0u, a.x, 0x0c030c02u) << 4;
      o[2] = __builtin_amdgcn_perm(0u, a.y, 0x0c010c00u) << 4; o[3] = __builtin_amdgcn_perm(0u, a.y, 0x0c030c02u) << 4;
    }
    *reinterpret_cast<uint4 *>(im + j * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
// The vertical displacements of one refinement round differ by less than a sample and the regular filter has six taps (1..6),
// so row `lane` of all of them reads intermediate rows lane + 1 .. lane + 7: those seven rows (16 bytes each) are loaded ONCE
// per horizontal position (mc_rows7) and every vertical candidate is a 7-tap sum over them — the six taps of its phase shifted
// by its integer part, a zero at the other end (mc_v7).  The first version of k_inter_pipe was LDS-bound (LDS busy 85 % of the
// kernel by PMC); this cut the reads of the vertical pass from 8 per candidate to 7 per three candidates.
// After the LDS diet the kernel is VALU-bound, so the vertical sums run on v_dot2_i32_i16: the rows are interleaved once per
// horizontal position into row PAIRS per column, pr[kp][c] = (row 2kp + 1, row 2kp + 2) of column c ("row 8" = a constant), and
// a candidate is 4 dot2 per sample instead of 7 multiply-adds + 7 extracts (exact: int16 x int16 into int32).
__device__ __forceinline__ void mc_rows7(const int16_t *im, int lane, uint32_t (*pr)[8]) {
  uint4 rw[8];
#pragma unroll
  for (int k = 0; k < 7; k++) rw[k] = *reinterpret_cast<const uint4 *>(im + (lane + 1 + k) * 8);
  rw[7] = make_uint4(0x00020002u, 0x00020002u, 0x00020002u, 0x00020002u);   // "row 8" = the constant 2: carries mc_v7's rounding term
#pragma unroll
  for (int kp = 0; kp < 4; kp++) {
    const uint32_t x[4] = { rw[2 * kp].x, rw[2 * kp].y, rw[2 * kp].z, rw[2 * kp].w };
    const uint32_t y[4] = { rw[2 * kp + 1].x, rw[2 * kp + 1].y, rw[2 * kp + 1].z, rw[2 * kp + 1].w };
#pragma unroll
    for (int d = 0; d < 4; d++) {
      pr[kp][2 * d] = __builtin_amdgcn_perm(y[d], x[d], 0x05040100u);       // (x.lo, y.lo)
      pr[kp][2 * d + 1] = __builtin_amdgcn_perm(y[d], x[d], 0x07060302u);   // (x.hi, y.hi)
    }
  }
}
// `filt32` holds the taps times 32: (32 s + 32768) >> 16 == (s + 1024) >> 11 exactly, and the result is then the high half of
// the accumulator — one v_perm takes the high halves of two sums, shift and pack in one instruction (the products stay far
// inside int32: |intermediate| < 2^15, sum of |taps| x 32 < 2^13).  Returns row `lane` of the prediction as four packed pairs.
__device__ __forceinline__ void mc_v7(const uint32_t (*pr)[8], int posy, const int16_t (*filt32)[8], int bd, uint32_t *ow) {
  const int oy = 4 + (posy >> 4) - 3;            // 0 or 1
  // the taps over rows lane + 1 .. lane + 7 as pairs.  The tap row lies in memory as (t0, t1) (t2, t3) (t4, t5) (t6, t7), t0 = t7 = 0:
  //   oy = 1 (rows 2 .. 7 carry t1 .. t6): (0, t1) (t2, t3) (t4, t5) (t6, .)   = the row as it lies;
  //   oy = 0 (rows 1 .. 6 carry t1 .. t6): (t1, t2) (t3, t4) (t5, t6) (0, .)   = one funnel shift per pair.
  // The last pair's second row is the constant 2 (mc_rows7) and its tap 16384: 2 x 16384 = 32768 is the rounding term, so the
  // accumulators start at 0 — an inline constant of the first dot2 instead of a v_mov of a literal per column.
  const uint4 fq = *reinterpret_cast<const uint4 *>(filt32[posy & 15]);
  const uint32_t g[4] = { oy ? fq.x : __builtin_amdgcn_alignbit(fq.y, fq.x, 16), oy ? fq.y : __builtin_amdgcn_alignbit(fq.z, fq.y, 16),
                          oy ? fq.z : __builtin_amdgcn_alignbit(fq.w, fq.z, 16), (oy ? (fq.w & 0xFFFFu) : 0u) | 0x40000000u };
  int s[8];
#pragma unroll
  for (int kp = 0; kp < 4; kp++) {
    const s16x2 gp = __builtin_bit_cast(s16x2, g[kp]);
#pragma unroll
    for (int c = 0; c < 8; c++)
      s[c] = kp == 0 ? dot2_start(pr[kp][c], g[kp], 0) : __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, pr[kp][c]), gp, s[c], false);
  }
  const s16x2 zero = { 0, 0 }, top = { (short)((1 << bd) - 1), (short)((1 << bd) - 1) };
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const s16x2 v = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm((uint32_t)s[2 * j + 1], (uint32_t)s[2 * j], 0x07060302u));
    ow[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_max(v, zero), top));
  }
}

__host__ __device__ constexpr int cmax3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

template <typename Pix>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_inter_pipe(InterLaunch L) {
  using ES = Pix;
  constexpr int bd = sizeof(Pix) == 1 ? 8 : 10;
  constexpr int GPW = 32;                        // groups (blocks) per workgroup
  constexpr int YW = 16, YWS = 20;               // luma window 16 x 16 (integer vector -4 .. +11); rows are whole dwords
  constexpr int CW = 12, CWS = 16;               // chroma window columns: chroma integer position -4 .. +7; rows are whole dwords
  constexpr int CWR0 = 3, CWR = 7;               // ... of which the 4-tap family reads rows 3 .. 9 only: those are stored
  // ONE LDS region per group, reused by the phases of a block (each ends with a group sync before the next one writes):
  //   luma search     window (YW x YWS samples) | intermediate (16 x 8 int16)
  //   luma residual   transpose buffer (8 x 12 int32) over the dead window / intermediate
  //   chroma          two windows (CWR x CWS samples) | intermediates (2 x (11 x 4 + 4) int16); then the two transpose buffers
  //                   (2 x 32 int32) over the dead windows
  // Separate arrays cost 66 KB per workgroup at 10 bits = 2 waves per SIMD; the shared region is 29 KB (912 bytes per block: the luma
  // phase sets it since the chroma windows hold 7 rows instead of 12) = the 5 waves per SIMD the VGPRs allow.
  constexpr int YWIN_BYTES = YW * YWS * (int)sizeof(ES), CWIN_BYTES = 2 * CWR * CWS * (int)sizeof(ES);
  constexpr int IMY_BYTES = 16 * 8 * 2, IMC_BYTES = 2 * (11 * 4 + 4) * 2, TY_BYTES = 8 * 12 * 4, TC_BYTES = 2 * 32 * 4;
  constexpr int REG_RAW = cmax3(YWIN_BYTES + IMY_BYTES, CWIN_BYTES + IMC_BYTES, TY_BYTES > TC_BYTES ? TY_BYTES : TC_BYTES);
  // per-group stride = 16 bytes past a multiple of 128: the groups of a wave run in lockstep at equal offsets and so start 4 banks apart
  constexpr int REG_BYTES = ((REG_RAW + 127) / 128) * 128 + 16;
  static_assert(YWIN_BYTES % 16 == 0 && CWIN_BYTES % 16 == 0, "intermediates and transpose buffers need 16-byte alignment");
  __shared__ __attribute__((aligned(16))) unsigned char regb[GPW * REG_BYTES];
  // the two filter families this policy uses, copied next to the windows: a lane's taps depend on its block's vector, and a
  // per-lane indexed read of __constant__ memory is a vector memory load (hundreds of cycles) in front of every candidate
  __shared__ __attribute__((aligned(16))) int16_t s_filt[3][16][8];   // regular 8-tap, regular 4-tap, regular 8-tap x 32 (mc_h16 at 10 bits, mc_v7)
  if (threadIdx.x < 128) {
    const int t = threadIdx.x;
    reinterpret_cast<uint32_t *>(s_filt)[t] = t < 64 ? reinterpret_cast<const uint32_t *>(kRegular8)[t] : reinterpret_cast<const uint32_t *>(kRegular4)[t - 64];
    s_filt[2][t >> 3][t & 7] = (int16_t)(kRegular8[t >> 3][t & 7] * 32);
  }
  __syncthreads();

  const int grp = threadIdx.x >> 3, lane = threadIdx.x & 7;
  const int bw = L.w / 8, bh = L.h / 8;
  // workgroups in XCD-aware order: vertically adjacent block rows (their windows overlap by 8 of 16 rows) share an L2
  // frame = a whole number of workgroups (launch_inter_pipe): the frame index is a 32-bit division of the workgroup id — scalar
  // code — and the block row a reciprocal multiplication with a one-step correction.  (As blk_all / (bw * bh) in 64 bits and two
  // more integer divisions per lane this line was 117 vector instructions, 5 % of the kernel.)
  const unsigned wpf = (unsigned)(bw * bh + GPW - 1) / GPW, wg = xcd_swizzle(blockIdx.x, gridDim.x);
  const int f = (int)(wg / wpf), blk = (int)(wg - (unsigned)f * wpf) * GPW + grp;
  if (blk >= bw * bh) return;
  int by = (int)((float)blk * __builtin_amdgcn_rcpf((float)bw)), bx = blk - by * bw;       // blk < 2^22: the estimate is within 1
  if (bx < 0) { by--; bx += bw; } else if (bx >= bw) { by++; bx -= bw; }
  const int x = bx * 8, y = by * 8;
  unsigned char *reg = regb + grp * REG_BYTES;
  ES *wy = reinterpret_cast<ES *>(reg);
  int16_t *im = reinterpret_cast<int16_t *>(reg + YWIN_BYTES);
  int32_t *T = reinterpret_cast<int32_t *>(reg);

  const Pix *src_y = reinterpret_cast<const Pix *>(L.src[0]) + (size_t)f * L.h * L.stride_y;
  const Pix *ref_y = reinterpret_cast<const Pix *>(ref_plane(L, f, 0)) + (size_t)f * L.h * L.stride_y;
  Pix *rec_y = reinterpret_cast<Pix *>(L.rec[0]) + (size_t)f * L.h * L.stride_y;
  int16_t *mvs = L.mvs + ((size_t)f * bw * bh + blk) * 2;
  const int imx = mvs[0] >> 3, imy = mvs[1] >> 3;   // integer vector from k_me_int (multiples of 8)

  // luma window: samples (x + imx - 4 .. +11, y + imy - 4 .. +11), clamped into the plane
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int r = lane + it * 8;
    const int fy = min(max(y + imy - 4 + r, 0), L.h - 1);
    const Pix *row = ref_y + row_off(fy, L.stride_y);
    stage_window_row<YW, Pix>(row, x + imx - 4, L.w, wy + r * YWS);
  }
  int s[8], bp[8];
  load_row<8>(src_y + row_off(y + lane, L.stride_y) + x, s);
  AV1MI_GROUP_SYNC();
  // rows are compared as packed pairs (v_sad_u16: two samples per instruction, both bit depths) and the running best
  // prediction is kept packed: four selects per candidate; it is unpacked once after the search
  uint32_t sp[4], bpp[4];
#pragma unroll
  for (int j = 0; j < 4; j++) sp[j] = (uint32_t)s[2 * j] | ((uint32_t)s[2 * j + 1] << 16);
  auto pack4 = [&](const int *o, uint32_t *w) {
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = (uint32_t)o[2 * j] | ((uint32_t)o[2 * j + 1] << 16);
  };
  auto sad_of = [&](const uint32_t *w) {
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) v = __builtin_amdgcn_sad_u16(sp[j], w[j], v);
    return group_sum<8>((int)v);
  };
  // candidate 0: the integer vector itself
  // (phase 0 of the 8-tap filter is the identity: taps 0 0 0 128 0 0 0 0, and (128 p + 4) >> 3 = 16 p, (128 * 16 p + 1024) >> 11
  // = p exactly — so the integer position is a copy of the window's centre)
  row_samples<8, ES>(wy + (lane + 4) * YWS + 4, 0, bp);
  pack4(bp, bpp);
  int best = sad_of(bpp), bfx = 0, bfy = 0;     // fractional part in 1/8 samples relative to the integer vector
  // ---- half-sample round, scored with the BILINEAR filter (encoder policy, mirrored by the oracle; libaom's USE_2_TAPS search):
  // at phase 8 its taps are (64, 64), and through the spec's two rounding stages ((s + 4) >> 3, then (s + 1024) >> 11) the
  // prediction is EXACTLY the rounded average of the 2 (one direction) or 4 (both) samples around the position.  So this round
  // needs no filter passes at all: three window rows of ten samples per lane and adds.  Which half-sample neighbourhood a
  // block lies in is decided as well this way as with the 8-tap filter (coded bytes and PSNR unchanged); the
  // quarter-sample round below scores every position, its centre included, with the real filter.
  {
    // packed 16-bit arithmetic (v_pk_add_u16 / v_pk_lshrrev_b16: two samples per lane-op; sums of four 10-bit samples fit):
    // P[j][k][i] = the sample pair at columns 3 + k + 2 i, 4 + k + 2 i of window row lane + 3 + j (block row -1 .. +1, columns
    // -1 + k ..), straight from the window's dwords — the sample-by-sample form unpacked 30 samples, averaged in 32 bits and packed
    // every candidate again for the SAD
    typedef unsigned short v2u __attribute__((ext_vector_type(2)));
    v2u P[3][3][4];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const uint32_t *r32 = reinterpret_cast<const uint32_t *>(wy + (lane + 3 + j) * YWS);
      uint32_t w[3][4];
      if constexpr (sizeof(ES) == 2) {
        uint32_t D[6];                               // columns 2 .. 13
#pragma unroll
        for (int i = 0; i < 6; i++) D[i] = r32[1 + i];
#pragma unroll
        for (int i = 0; i < 4; i++) { w[0][i] = __builtin_amdgcn_alignbit(D[i + 1], D[i], 16); w[1][i] = D[i + 1]; w[2][i] = __builtin_amdgcn_alignbit(D[i + 2], D[i + 1], 16); }
      } else {
        const uint32_t B0 = r32[0], B1 = r32[1], B2 = r32[2], B3 = r32[3];     // columns 0 .. 15, one byte each
        w[0][0] = __builtin_amdgcn_perm(B1, B0, 0x0c040c03u); w[0][1] = __builtin_amdgcn_perm(0u, B1, 0x0c020c01u);
        w[0][2] = __builtin_amdgcn_perm(B2, B1, 0x0c040c03u); w[0][3] = __builtin_amdgcn_perm(0u, B2, 0x0c020c01u);
        w[1][0] = __builtin_amdgcn_perm(0u, B1, 0x0c010c00u); w[1][1] = __builtin_amdgcn_perm(0u, B1, 0x0c030c02u);
        w[1][2] = __builtin_amdgcn_perm(0u, B2, 0x0c010c00u); w[1][3] = __builtin_amdgcn_perm(0u, B2, 0x0c030c02u);
        w[2][0] = w[0][1]; w[2][1] = w[0][2]; w[2][2] = w[0][3]; w[2][3] = __builtin_amdgcn_perm(B3, B2, 0x0c040c03u);
      }
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 4; i++) P[j][k][i] = __builtin_bit_cast(v2u, w[k][i]);
    }
    v2u H[3][2][4];                                // horizontal pair sums: columns (k, k + 1) of row j
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int k = 0; k < 2; k++)
#pragma unroll
        for (int i = 0; i < 4; i++) H[j][k][i] = P[j][k][i] + P[j][k + 1][i];
    const v2u one = { 1, 1 }, two = { 2, 2 };
    unsigned bestkey = (unsigned)best << 4;        // (SAD, rank): rank 0 = the centre (keeps ties), k + 1 = the neighbour visited k-th
#pragma unroll
    for (int iy = 0; iy < 3; iy++) {
#pragma unroll
      for (int ix = 0; ix < 3; ix++) {
        if (ix == 1 && iy == 1) continue;
        const int r0 = iy == 0 ? 0 : 1, r1 = iy == 2 ? 2 : 1, k = ix == 0 ? 0 : 1;   // the 1, 2 or 4 samples averaged
        uint32_t ow[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          v2u o;
          if (iy == 1) o = (H[1][k][i] + one) >> one;
          else if (ix == 1) o = (P[r0][1][i] + P[r1][1][i] + one) >> one;
          else o = (H[r0][k][i] + H[r1][k][i] + two) >> two;
          ow[i] = __builtin_bit_cast(uint32_t, o);
        }
        bestkey = min(bestkey, ((unsigned)sad_of(ow) << 4) | (unsigned)(iy * 3 + ix + 1));
      }
    }
    const int rk = (int)(bestkey & 15u), kk = rk - 1, wy_ = (kk * 11) >> 5, wx_ = kk - 3 * wy_;   // kk / 3 for kk in 0..8
    bfx = rk ? (wx_ - 1) * 4 : 0;
    bfy = rk ? (wy_ - 1) * 4 : 0;
  }
  // ---- quarter-sample round with the real 8-tap filter: the centre (re-scored: rank 0) and its 8 neighbours, column by column,
  // one horizontal pass per column.  The oracle visits the neighbours in raster order with strict improvement, i.e. the winner
  // is the minimum of (SAD, rank) and the centre keeps ties; every position goes through the same (key, prediction) update, so
  // the order of evaluation does not matter.
  {
    const int step = 2, cx = bfx, cy = bfy;
    unsigned bestkey = 0xFFFFFFFFu;
#pragma unroll 1
    for (int ix = 0; ix < 3; ix++) {
      const int fx = cx + (ix - 1) * step;
      mc_h16<ES>(wy, YWS, im, lane, fx * 2, sizeof(ES) == 2 ? s_filt[2] : s_filt[0]);
      AV1MI_GROUP_SYNC();
      uint32_t pr[4][8];
      mc_rows7(im, lane, pr);
#pragma unroll
      for (int iy = 0; iy < 3; iy++) {     // unrolled: the three vertical candidates of a column overlap
        const int fy = cy + (iy - 1) * step, k = iy * 3 + ix;
        uint32_t ow[4];
        mc_v7(pr, fy * 2, s_filt[2], bd, ow);
        const unsigned key = ((unsigned)sad_of(ow) << 4) | (unsigned)(k == 4 ? 0 : k + 1);
        const bool better = key < bestkey;
        bestkey = min(bestkey, key);
#pragma unroll
        for (int j = 0; j < 4; j++) bpp[j] = better ? ow[j] : bpp[j];
      }
      AV1MI_GROUP_SYNC();
    }
    const int rk = (int)(bestkey & 15u), kk = rk - 1, wy_ = (kk * 11) >> 5, wx_ = kk - 3 * wy_;
    best = (int)(bestkey >> 4);
    bfx = rk ? cx + (wx_ - 1) * step : cx;
    bfy = rk ? cy + (wy_ - 1) * step : cy;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) { bp[2 * j] = bpp[j] & 0xffff; bp[2 * j + 1] = bpp[j] >> 16; }
  const int mvx = imx * 8 + bfx, mvy = imy * 8 + bfy;   // final vector, 1/8 luma samples
  if (lane == 0) { mvs[0] = (int16_t)mvx; mvs[1] = (int16_t)mvy; }
  int rec[8];
  int nz = code_residual<8, Pix, false, kAcRoundInter>(T, lane, s, bp, L.dc_q, L.ac_q, L.dc_quant, L.ac_quant, L.lev[0] + (size_t)f * L.w * L.h + (size_t)blk * 64 + lane * 8, rec);
  store_row<8>(rec_y + row_off(y + lane, L.stride_y) + x, rec);

  // chroma: lanes 0-3 code the U block, lanes 4-7 the V block (4x4 each, 4-tap regular filter rows)
  {
    const int pl = lane >> 2, cl = lane & 3;
    const int cw = L.w / 2, chh = L.h / 2, cx0 = x / 2, cy0 = y / 2;
    const Pix *src_c = reinterpret_cast<const Pix *>(L.src[1 + pl]) + (size_t)f * chh * L.stride_uv;
    const Pix *ref_c = reinterpret_cast<const Pix *>(ref_plane(L, f, 1 + pl)) + (size_t)f * chh * L.stride_uv;
    Pix *rec_c = reinterpret_cast<Pix *>(L.rec[1 + pl]) + (size_t)f * chh * L.stride_uv;
    // the vector in 1/16 chroma samples is the luma vector in 1/8 luma samples
    const int cix = mvx >> 4, ciy = mvy >> 4;         // integer chroma displacement (floor)
    ES *wc = wy + pl * CWR * CWS - CWR0 * CWS;       // row r of the window at wc + r * CWS, rows CWR0 .. CWR0 + CWR - 1 exist
    int16_t *imc = reinterpret_cast<int16_t *>(reg + CWIN_BYTES) + pl * (11 * 4 + 4);
    AV1MI_GROUP_SYNC();                               // the luma transpose buffer has been read: the region is free
    // the 4-tap family reads window rows 3 .. 9 only (mc_row<4, ES, 2, 6>: taps 2 .. 5 of rows -1 .. +5 around the block)
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const int r = 3 + cl + it * 4;
      if (r <= 9) {
        const int fy = min(max(cy0 + ciy - 4 + r, 0), chh - 1);
        const Pix *row = ref_c + row_off(fy, L.stride_uv);
        stage_window_row<CW, Pix>(row, cx0 + cix - 4, cw, wc + r * CWS);
      }
    }
    int sc[4], pc[4], rc[4];
    load_row<4>(src_c + row_off(cy0 + cl, L.stride_uv) + cx0, sc);
    AV1MI_GROUP_SYNC();
    mc_row<4, ES, 2, 6>(wc, CWS, imc, cl, mvx & 15, mvy & 15, s_filt[1], bd, pc);
    nz |= code_residual<4, Pix, false, kAcRoundInter>(T + pl * 32, cl, sc, pc, L.dc_q, L.ac_q, L.dc_quant, L.ac_quant,
                                L.lev[1 + pl] + (size_t)f * cw * chh + (size_t)blk * 16 + cl * 4, rc);
    store_row<4>(rec_c + row_off(cy0 + cl, L.stride_uv) + cx0, rc);
  }
  nz = group_or<8>(nz);
  if (lane == 0) L.skip[(size_t)f * bw * bh + blk] = nz == 0;
}

// integer search: writes L.mvs (whole-sample vectors), which k_inter_pipe then refines
hipError_t launch_me_int(const InterLaunch &L, hipStream_t s) {
  if (L.nframes <= 0) return hipSuccess;
  const int sbs = ((L.w + 63) / 64) * ((L.h + 63) / 64);
  const dim3 g1((unsigned)(sbs * L.nframes));   // 1-D: the kernel orders the tiles (xcd_tile)
  if (L.range == 8) {
    if (L.bd == 8) hipLaunchKernelGGL((k_me_int<uint8_t, 8>), g1, dim3(256), 0, s, L);
    else hipLaunchKernelGGL((k_me_int<uint16_t, 8>), g1, dim3(256), 0, s, L);
  } else {
    if (L.bd == 8) hipLaunchKernelGGL((k_me_int<uint8_t, 0>), g1, dim3(256), 0, s, L);
    else hipLaunchKernelGGL((k_me_int<uint16_t, 0>), g1, dim3(256), 0, s, L);
  }
  return hipGetLastError();
}
hipError_t launch_inter_pipe(const InterLaunch &L, hipStream_t s) {
  if (L.nframes <= 0) return hipSuccess;
  const long long wpf = ((long long)(L.w / 8) * (L.h / 8) + 31) / 32;       // workgroups per frame (32 blocks each)
  const dim3 g2((unsigned)(wpf * L.nframes));
  if (L.bd == 8) hipLaunchKernelGGL(k_inter_pipe<uint8_t>, g2, dim3(256), 0, s, L);
  else hipLaunchKernelGGL(k_inter_pipe<uint16_t>, g2, dim3(256), 0, s, L);
  return hipGetLastError();
}

}  // namespace av1mi
