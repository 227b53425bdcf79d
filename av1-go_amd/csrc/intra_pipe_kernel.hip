// intra_pipe_kernel.hip — the closed-loop intra-only block pipeline (BASELINE config 2) as ONE gfx950 kernel:
// per block: intra prediction from reconstructed neighbours (K3) -> mode decision (SAD over 11 candidate modes)
// -> residual -> forward DCT (K1) -> quantise (K8) -> dequantise -> inverse DCT + reconstruct (K2).
//
// Parallelism.  Intra prediction chains every block to its reconstructed neighbours, so the unit of
// independent work is the TILE: one 64x64 luma superblock (+ its two 32x32 chroma blocks), never predicted
// across its border.  A tile is walked in z-order by a group of BS lanes (BS = luma block size): lane r owns
// row r of the current block for prediction / residual / reconstruction and row-or-column r for the
// transforms, with a BS x BS int32 LDS tile as the transpose buffer.  64/BS tiles share a wave and run in
// lockstep, 256/BS share a workgroup; a 1080p frame has 510 tiles and a segment of F frames 510*F, so the grid
// is F*510*BS/256 workgroups.  For chroma the group splits into two halves that code the U and the V block
// of the same position at once (one shared uv mode, as in AV1).
// Neighbour samples never come back from HBM: each tile keeps the last reconstructed sample of every column
// ("above" line), of every row ("left" line) and each block's bottom-right sample in LDS, which is exactly
// what the edge builder of intra.hpp needs in z-order.
// HBM traffic per sample: source read b + reconstruction write b + levels write 2 (+ 1 mode byte per block).
//
// The arithmetic restates AV1 spec §7.11.2 / §7.13.3 / §7.12.3 (see intra.hpp, txfm1d.hpp); the encoder policy
// (candidate list, SAD, first-minimum tie break, DCT_DCT, tile = superblock) is this project's own and is
// mirrored by oracle/av1o_pipeline.c:av1o_intra_encode_frame for checking.  The reference has no counterpart
// (internal/ffmpeg/transcode.go:120 hands the whole job to an external encoder).
#include <type_traits>
#include "intra_fast.hpp"
#include "block_code.hpp"
#include "av1mi_internal.hpp"

// Diagnostic build only (-DAV1MI_STAMPS, never shipped): per-phase s_memtime sums of lane 0 of every wave, written to a
// debug buffer no other code reads (cdna_hip_programming.md §7 "In-kernel stamps").
#ifdef AV1MI_STAMPS
#define STAMP(i)                                                                                   \
  do {                                                                                             \
    unsigned long long t_;                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    av1mi_stamp_acc[i] += t_ - av1mi_stamp_last; av1mi_stamp_last = t_;                            \
  } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

namespace av1mi {

#ifdef AV1MI_STAMPS
__device__ unsigned long long av1mi_stamp_out[8];
#endif


__device__ __forceinline__ unsigned morton2(unsigned x, unsigned y) {
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) m |= ((x >> i) & 1u) << (2 * i) | ((y >> i) & 1u) << (2 * i + 1);
  return m;
}

// per-plane, per-tile neighbour context in LDS (indices in samples / blocks of that plane); ES = sample type
template <typename ES> struct TileCtx {
  ES *above;       // [tile width]  last reconstructed sample of each column
  ES *left;        // [tile height] last reconstructed sample of each row
  ES *tl;          // [n] per block row: the top-left neighbour sample of the NEXT block of that row (see code_block)
  ES *edge;        // raw + derived edge arrays, fast_edge_len(B) entries
  int32_t *tbuf;   // B x (B+4) transpose buffer
};

// Code one B x B block with L = B lanes (`lane` in [0,B)); `gw` = lanes that share the mode decision (the whole
// group: B for luma, 2B for the U+V pair).  Returns the chosen mode (identical in all gw lanes).
#ifdef AV1MI_STAMPS
#define STAMP_ARGS , unsigned long long *av1mi_stamp_acc, unsigned long long &av1mi_stamp_last
#define STAMP_PASS , av1mi_stamp_acc, av1mi_stamp_last
#else
#define STAMP_ARGS
#define STAMP_PASS
#endif
// one of the 13 modes by its RUN-TIME number (the open-loop pipeline: the mode was decided beforehand, k_intra_modes).  The groups
// of a wave take their cases one after the other; a case reads only its own group's edge arrays.
template <int B, typename ES>
__device__ __forceinline__ void pred_row_of_mode(int mode, const ES *edge, int lane, int bd, int n_top, int n_left, int filter_type, int *out) {
  switch (mode) {
#define AV1MI_CASE(M) case M: fast_pred_row<M, B, ES>(edge, lane, bd, n_top, n_left, filter_type, out); break;
    AV1MI_CASE(DC_PRED) AV1MI_CASE(V_PRED) AV1MI_CASE(H_PRED) AV1MI_CASE(D45_PRED) AV1MI_CASE(D135_PRED) AV1MI_CASE(D113_PRED) AV1MI_CASE(D157_PRED)
    AV1MI_CASE(D203_PRED) AV1MI_CASE(D67_PRED) AV1MI_CASE(SMOOTH_PRED) AV1MI_CASE(SMOOTH_V_PRED) AV1MI_CASE(SMOOTH_H_PRED)
    default: fast_pred_row<PAETH_PRED, B, ES>(edge, lane, bd, n_top, n_left, filter_type, out); break;
#undef AV1MI_CASE
  }
}

// the 13 candidates (order == oracle/av1o_pipeline.c:intra_candidates, first minimum wins) scored by SAD against the source row s[]
// over the GW lanes that share the decision; returns the mode, and the winning prediction row in bp[] when KEEP
template <int B, int GW, typename ES, bool KEEP>
__device__ __forceinline__ int search_modes(const ES *edge, int lane, int bd, int n_top, int n_left, int filter_type, const int *s, int *bp) {
  int best = 0x7fffffff, best_mode = 0;
  auto eval = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    int out[B];
    fast_pred_row<MODE, B, ES>(edge, lane, bd, n_top, n_left, filter_type, out);
    int sad = 0;
#pragma unroll
    for (int c = 0; c < B; c++) sad += abs(s[c] - out[c]);
    sad = group_sum<GW>(sad);
    const bool better = sad < best;
    best = better ? sad : best; best_mode = better ? MODE : best_mode;
    if constexpr (KEEP) {
#pragma unroll
      for (int c = 0; c < B; c++) bp[c] = better ? out[c] : bp[c];
    }
    // keep the scheduler from hoisting the next candidates' LDS reads above this one: that costs ~100 VGPRs and a wave per SIMD
    __builtin_amdgcn_sched_barrier(0);
  };
  // candidate order == oracle/av1o_pipeline.c:intra_candidates (first minimum wins)
  eval(std::integral_constant<int, DC_PRED>{});   eval(std::integral_constant<int, V_PRED>{});
  eval(std::integral_constant<int, H_PRED>{});    eval(std::integral_constant<int, D45_PRED>{});
  eval(std::integral_constant<int, D135_PRED>{}); eval(std::integral_constant<int, D113_PRED>{});
  eval(std::integral_constant<int, D157_PRED>{}); eval(std::integral_constant<int, D203_PRED>{});
  eval(std::integral_constant<int, D67_PRED>{});  eval(std::integral_constant<int, SMOOTH_PRED>{});
  eval(std::integral_constant<int, PAETH_PRED>{}); eval(std::integral_constant<int, SMOOTH_V_PRED>{});
  eval(std::integral_constant<int, SMOOTH_H_PRED>{});
  return best_mode;
}

template <int B, int GW, typename Pix, bool OPEN = false>
__device__ __forceinline__ int code_block(const TileCtx<Pix> &C, int lane, int bx, int by, int n, int n_top, int n_topright, int n_left,
                                          int n_bottomleft, int filter_type, int dc_q, int ac_q, int dc_quant, int ac_quant, const Pix *src_row,
                                          Pix *rec_row, int16_t *lev_row, int mode_in STAMP_ARGS) {
  constexpr int bd = sizeof(Pix) == 1 ? 8 : 10;
  using ES = Pix;
  const int x = bx * B, y = by * B;
  auto fetch = [&](int yy, int xx) -> int {
    if (yy < 0) return xx < 0 ? C.tl[by] : C.above[x + xx];
    return C.left[y + yy];
  };
  int s[B], bp[B];
  load_row<B>(src_row, s);
  STAMP(0);
  // all edge variants of the block in two LDS phases, then 13 compile-time-specialised predictions, no hand-offs
  fast_build<B>(C.edge, lane, bd, n_top, n_topright, n_left, n_bottomleft, filter_type, fetch);
  STAMP(1);
  int best_mode;
  if constexpr (OPEN) {      // the mode is given (decided on the source, k_intra_modes): one prediction
    best_mode = mode_in;
    pred_row_of_mode<B, ES>(mode_in, (const ES *)C.edge, lane, bd, n_top, n_left, filter_type, bp);
  } else {
    best_mode = search_modes<B, GW, ES, true>((const ES *)C.edge, lane, bd, n_top, n_left, filter_type, s, bp);
  }
  AV1MI_GROUP_SYNC();
  STAMP(2);
  int rec[B];
  constexpr int AC_ROUND = GW == 32 ? kAcRoundKey32 : kAcRoundIntra;      // GW = the luma block size
  if constexpr (GW != B)   // the U+V pair: transform type implied by the mode (luma types are coded in the bitstream: DCT_DCT)
    code_residual<B, Pix, true, AC_ROUND>(C.tbuf, lane, s, bp, dc_q, ac_q, dc_quant, ac_quant, lev_row, rec, (kModeVertAdst >> best_mode) & 1, (kModeHorzAdst >> best_mode) & 1);
  else
    code_residual<B, Pix, false, AC_ROUND>(C.tbuf, lane, s, bp, dc_q, ac_q, dc_quant, ac_quant, lev_row, rec);
  STAMP(3);
  store_row<B>(rec_row, rec);
  // neighbour context for the blocks to come
  C.left[y + lane] = (ES)rec[B - 1];
  if (lane == B - 1) {
    // Before this block's bottom row replaces it, above[x + B - 1] still holds sample (x + B - 1, y - 1): the top-left
    // neighbour of the block to the right.  In z-order the blocks of one block row are coded left to right (the order is
    // monotone in x for fixed y), so one entry per block row is enough — no n x n map of bottom-right samples.
    C.tl[by] = C.above[x + B - 1];
#pragma unroll
    for (int c = 0; c < B; c++) C.above[x + c] = (ES)rec[c];
  }
  AV1MI_GROUP_SYNC();
  STAMP(4);
  return best_mode;
}

// OPEN-LOOP MODE DECISION (av1mi_intra_job.open_loop, off by default): the 13 candidates of every block predicted from the SOURCE
// frame's neighbours (edge filter type 0: the neighbours' modes are being decided at the same time), all blocks of all frames in
// parallel — no tile is a serial chain here.  k_intra_pipe<.., OPEN> then walks the tiles with ONE prediction per block, from the
// reconstruction, in the mode found here.  Same availability rules as the closed loop (tile = 64x64 superblock).
// Cost in the oracle (av1o_set_intra_open_loop): +0.0 .. +0.3 % bytes up to q 128, +1 % at q 160, +3.5 % at q 200, PSNR within
// 0.07 dB.  Gain, measured (round 3): 16 frames of 1080p 8-bit 0.885 -> 0.748 ms (k_intra_modes 0.37 of it), 12 frames of 4K 10-bit
// 1.61 -> ~1.6 ms: the search is VALU work that this arrangement moves but does not remove, so the closed loop stays the default.
template <int BS, typename Pix>
__global__ __launch_bounds__(256) void k_intra_modes(IntraPipeLaunch L) {
  constexpr int CS = BS / 2, N = 64 / BS, GPW = 256 / BS;
  using ES = Pix;
  constexpr int ELY = fast_edge_len(BS), ELC = fast_edge_len(CS);
  constexpr int EDGE_N = ((ELY > 2 * ELC ? ELY : 2 * ELC) + 15) / 16 * 16 + 8;      // entries per group; + 8: groups start on different banks
  __shared__ __attribute__((aligned(16))) ES edges[GPW * EDGE_N];
  constexpr int bd = sizeof(Pix) == 1 ? 8 : 10;
  const int grp = threadIdx.x / BS, lane = threadIdx.x % BS;
  const int bw = L.w / BS, bh = L.h / BS;
  const long long per = (long long)bw * bh, blk_all = (long long)blockIdx.x * GPW + grp;
  if (blk_all >= per * L.nframes) return;
  const int f = (int)(blk_all / per), blk = (int)(blk_all - f * per), fy = blk / bw, fx = blk - fy * bw;
  const unsigned bx = fx % N, by = fy % N, k = morton2(bx, by);
  const bool have_top = by > 0, have_left = bx > 0;
  const bool have_tr = have_top && (int)bx + 1 < N && fx + 1 < bw && morton2(bx + 1, by - 1) < k;
  const bool have_bl = have_left && (int)by + 1 < N && fy + 1 < bh && morton2(bx - 1, by + 1) < k;
  ES *e = edges + grp * EDGE_N;
  const int pl = lane / CS, cl = lane % CS;
  {
    const Pix *p = reinterpret_cast<const Pix *>(L.src[0]) + (size_t)f * L.frame_rows * L.stride_y + row_off(fy * BS, L.stride_y) + (size_t)(fx * BS);
    const int st = L.stride_y;
    // fast_build asks for row -1 / column -1 unconditionally and selects afterwards (harmless on the LDS lines of the closed loop):
    // here those are global addresses, outside the allocation at the frame's first row — answer 0 without a load
    auto fetch = [&](int yy, int xx) -> int { return ((yy >= 0 || have_top) && (xx >= 0 || have_left)) ? (int)p[(ptrdiff_t)yy * st + xx] : 0; };
    int s[BS], bp[1];
    load_row<BS>(p + (size_t)lane * st, s);
    fast_build<BS>(e, lane, bd, have_top ? BS : 0, have_tr ? BS : 0, have_left ? BS : 0, have_bl ? BS : 0, 0, fetch);
    const int m = search_modes<BS, BS, ES, false>(e, lane, bd, have_top ? BS : 0, have_left ? BS : 0, 0, s, bp);
    if (lane == 0) L.modes_y[(size_t)f * L.modes_stride + blk] = (uint8_t)m;
  }
  AV1MI_GROUP_SYNC();
  {
    const Pix *p = reinterpret_cast<const Pix *>(L.src[1 + pl]) + (size_t)f * (L.frame_rows / 2) * L.stride_uv + row_off(fy * CS, L.stride_uv) + (size_t)(fx * CS);
    const int st = L.stride_uv;
    auto fetch = [&](int yy, int xx) -> int { return ((yy >= 0 || have_top) && (xx >= 0 || have_left)) ? (int)p[(ptrdiff_t)yy * st + xx] : 0; };
    int s[CS], bp[1];
    load_row<CS>(p + (size_t)cl * st, s);
    fast_build<CS>(e + pl * ELC, cl, bd, have_top ? CS : 0, have_tr ? CS : 0, have_left ? CS : 0, have_bl ? CS : 0, 0, fetch);
    const int m = search_modes<CS, BS, ES, false>(e + pl * ELC, cl, bd, have_top ? CS : 0, have_left ? CS : 0, 0, s, bp);
    if (lane == 0) L.modes_uv[(size_t)f * L.modes_stride + blk] = (uint8_t)m;
  }
}

template <int BS, typename Pix, bool OPEN = false>
__global__ __launch_bounds__(256, BS == 8 ? 3 : 1) void k_intra_pipe(IntraPipeLaunch L) {
  constexpr int CS = BS / 2, N = 64 / BS, TPW = 256 / BS;
  using ES = Pix;                                      // LDS sample type: 1 byte for 8-bit content, 2 for 10-bit
  constexpr int ELY = fast_edge_len(BS), ELC = fast_edge_len(CS);
  constexpr int T32_PER_TILE = BS * (BS + 4);          // int32 transpose buffer of the residual tail
  // luma and chroma blocks alternate: one edge region — and it doubles as the transpose buffer: the edge arrays are dead
  // once the candidates have been compared (group sync), the residual tail then transposes through the same bytes
  constexpr int EDGE_RAW = (ELY > 2 * ELC ? ELY : 2 * ELC) * (int)sizeof(ES);
  constexpr int EDGE_BYTES = ((EDGE_RAW > T32_PER_TILE * 4 ? EDGE_RAW : T32_PER_TILE * 4) + 15) / 16 * 16;
  // bytes per tile: line buffers + one top-left sample per block row (Y, U, V), edge / transpose region, four mode lines
  constexpr int LINE_N = ((64 + 64 + N) + 2 * (32 + 32 + N) + 15) / 16 * 16;
  constexpr int LINE_BYTES = LINE_N * (int)sizeof(ES);
  constexpr int CTX_BYTES_RAW = LINE_BYTES + EDGE_BYTES + 4 * N;
  // per-tile strides padded so that the tiles of a wave (which run in lockstep at equal offsets) start 4 banks apart
  constexpr int CTX_BYTES = ((CTX_BYTES_RAW + 127) / 128) * 128 + 16;
  __shared__ __attribute__((aligned(16))) unsigned char ctxb[TPW * CTX_BYTES];

  const int grp = threadIdx.x / BS, lane = threadIdx.x % BS;
  const int sbw = (L.w + 63) / 64, sbh = (L.h + 63) / 64;
  // workgroups in XCD-aware order (av1mi_internal.hpp): horizontally adjacent tiles share 128-byte lines of every plane row
  const long long tile = (long long)xcd_swizzle(blockIdx.x, gridDim.x) * TPW + grp;
  if (tile >= (long long)L.nframes * sbw * sbh) return;
  const int f = (int)(tile / (sbw * sbh)), sb = (int)(tile % (sbw * sbh)), sby = sb / sbw, sbx = sb % sbw;
  const int bw = L.w / BS, bh = L.h / BS;

  ES *u = reinterpret_cast<ES *>(ctxb + grp * CTX_BYTES);
  TileCtx<ES> Y, Cp;   // Cp: this lane's chroma plane (U for the lower half of the group, V for the upper)
  Y.above = u; Y.left = u + 64; Y.tl = u + 128;
  const int pl = lane / CS, cl = lane % CS;
  ES *cu = u + 128 + N + pl * (64 + N);
  Cp.above = cu; Cp.left = cu + 32; Cp.tl = cu + 64;
  u += LINE_N;
  Y.edge = u;
  Cp.edge = u + pl * ELC;
  Y.tbuf = reinterpret_cast<int32_t *>(u);
  Cp.tbuf = Y.tbuf + pl * (CS * (CS + 4));
  // modes of the most recent block of every block column / row (luma, chroma): the top neighbour of a block is the latest
  // block of its column, the left neighbour the latest of its row (z-order is monotone along both)
  uint8_t *col_my = reinterpret_cast<uint8_t *>(u) + EDGE_BYTES, *row_my = col_my + N, *col_mc = row_my + N, *row_mc = col_mc + N;

  const Pix *src_y = reinterpret_cast<const Pix *>(L.src[0]) + (size_t)f * L.frame_rows * L.stride_y;
  Pix *rec_y = reinterpret_cast<Pix *>(L.rec[0]) + (size_t)f * L.frame_rows * L.stride_y;
  const Pix *src_c = reinterpret_cast<const Pix *>(L.src[1 + pl]) + (size_t)f * (L.frame_rows / 2) * L.stride_uv;
  Pix *rec_c = reinterpret_cast<Pix *>(L.rec[1 + pl]) + (size_t)f * (L.frame_rows / 2) * L.stride_uv;
  int16_t *lev_y = L.lev[0] + (size_t)f * L.w * L.frame_rows;
  int16_t *lev_c = L.lev[1 + pl] + (size_t)f * (L.w / 2) * (L.frame_rows / 2);
  uint8_t *modes_y = L.modes_y + (size_t)f * L.modes_stride, *modes_uv = L.modes_uv + (size_t)f * L.modes_stride;

#ifdef AV1MI_STAMPS
  unsigned long long av1mi_stamp_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, av1mi_stamp_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(av1mi_stamp_last)::"memory");
#endif
  // (Measured and dropped, round 3: loading the source rows one block ahead — 1.61 -> 1.92 ms for 12 4K 10-bit frames.  The chain
  // is bound by the VALU work of the 13-candidate search, not by the latency of these loads.)
  for (unsigned k = 0; k < (unsigned)(N * N); k++) {
    unsigned bx = 0, by = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { bx |= ((k >> (2 * i)) & 1u) << i; by |= ((k >> (2 * i + 1)) & 1u) << i; }
    const int fx = sbx * N + bx, fy = sby * N + by;
    if (fx >= bw || fy >= bh) continue;
    const bool have_top = by > 0, have_left = bx > 0;
    const bool have_tr = have_top && (int)bx + 1 < N && fx + 1 < bw && morton2(bx + 1, by - 1) < k;
    const bool have_bl = have_left && (int)by + 1 < N && fy + 1 < bh && morton2(bx - 1, by + 1) < k;
    int ft = 0, ftc = 0;
    if (have_top) { const int m = col_my[bx], mc = col_mc[bx]; ft |= m >= 9 && m <= 11; ftc |= mc >= 9 && mc <= 11; }
    if (have_left) { const int m = row_my[by], mc = row_mc[by]; ft |= m >= 9 && m <= 11; ftc |= mc >= 9 && mc <= 11; }
    const size_t blk = row_off(fy, bw) + fx;        // 24-bit multiplies: see row_off
    {
      const size_t off = row_off(fy * BS + lane, L.stride_y) + (size_t)(fx * BS);
      const int m = code_block<BS, BS, Pix, OPEN>(Y, lane, bx, by, N, have_top ? BS : 0, have_tr ? BS : 0, have_left ? BS : 0,
                                                   have_bl ? BS : 0, ft, L.dc_q, L.ac_q, L.dc_quant, L.ac_quant, src_y + off, rec_y + off,
                                                   lev_y + blk * BS * BS + lane * BS, OPEN ? (int)modes_y[blk] : 0 STAMP_PASS);
      if (lane == 0) { if (!OPEN) modes_y[blk] = (uint8_t)m; col_my[bx] = row_my[by] = (uint8_t)m; }
    }
    {
      const size_t off = row_off(fy * CS + cl, L.stride_uv) + (size_t)(fx * CS);
      const int m = code_block<CS, BS, Pix, OPEN>(Cp, cl, bx, by, N, have_top ? CS : 0, have_tr ? CS : 0, have_left ? CS : 0,
                                                   have_bl ? CS : 0, ftc, L.dc_q, L.ac_q, L.dc_quant, L.ac_quant, src_c + off, rec_c + off,
                                                   lev_c + blk * CS * CS + cl * CS, OPEN ? (int)modes_uv[blk] : 0 STAMP_PASS);
      if (lane == 0) { if (!OPEN) modes_uv[blk] = (uint8_t)m; col_mc[bx] = row_mc[by] = (uint8_t)m; }
    }
    AV1MI_GROUP_SYNC();
    STAMP(5);
  }
#ifdef AV1MI_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 7)
    for (int i = 0; i < 8; i++) av1mi_stamp_out[i] = av1mi_stamp_acc[i];
#endif
}

#ifdef AV1MI_STAMPS
extern "C" int av1mi_debug_read_stamps(unsigned long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(av1mi_stamp_out), sizeof(av1mi_stamp_out));
}
#endif

hipError_t launch_intra_pipe(const IntraPipeLaunch &L, int bs, hipStream_t s) {
  const long long tiles = (long long)L.nframes * ((L.w + 63) / 64) * ((L.h + 63) / 64);
  if (tiles <= 0) return hipSuccess;
  if (bs != 8 && bs != 16 && bs != 32) return hipErrorInvalidValue;
  const int tpw = 256 / bs;
  const dim3 grid((unsigned)((tiles + tpw - 1) / tpw));
  if (L.open_loop) {      // the decision on the source, every block on its own; then the tiles' chains with one prediction per block
    const long long blocks = (long long)L.nframes * (L.w / bs) * (L.h / bs);
    const dim3 mgrid((unsigned)((blocks + tpw - 1) / tpw));
    if (bs == 32) return hipErrorInvalidValue;      // (the open-loop decision exists for 8x8 and 16x16)
    if (bs == 8) {
      if (L.bd == 8) { hipLaunchKernelGGL((k_intra_modes<8, uint8_t>), mgrid, dim3(256), 0, s, L); hipLaunchKernelGGL((k_intra_pipe<8, uint8_t, true>), grid, dim3(256), 0, s, L); }
      else { hipLaunchKernelGGL((k_intra_modes<8, uint16_t>), mgrid, dim3(256), 0, s, L); hipLaunchKernelGGL((k_intra_pipe<8, uint16_t, true>), grid, dim3(256), 0, s, L); }
    } else {
      if (L.bd == 8) { hipLaunchKernelGGL((k_intra_modes<16, uint8_t>), mgrid, dim3(256), 0, s, L); hipLaunchKernelGGL((k_intra_pipe<16, uint8_t, true>), grid, dim3(256), 0, s, L); }
      else { hipLaunchKernelGGL((k_intra_modes<16, uint16_t>), mgrid, dim3(256), 0, s, L); hipLaunchKernelGGL((k_intra_pipe<16, uint16_t, true>), grid, dim3(256), 0, s, L); }
    }
    return hipGetLastError();
  }
  if (bs == 8) {
    if (L.bd == 8) hipLaunchKernelGGL((k_intra_pipe<8, uint8_t>), grid, dim3(256), 0, s, L);
    else hipLaunchKernelGGL((k_intra_pipe<8, uint16_t>), grid, dim3(256), 0, s, L);
  } else if (bs == 32) {
    if (L.bd == 8) hipLaunchKernelGGL((k_intra_pipe<32, uint8_t>), grid, dim3(256), 0, s, L);
    else hipLaunchKernelGGL((k_intra_pipe<32, uint16_t>), grid, dim3(256), 0, s, L);
  } else {
    if (L.bd == 8) hipLaunchKernelGGL((k_intra_pipe<16, uint8_t>), grid, dim3(256), 0, s, L);
    else hipLaunchKernelGGL((k_intra_pipe<16, uint16_t>), grid, dim3(256), 0, s, L);
  }
  return hipGetLastError();
}

}  // namespace av1mi
