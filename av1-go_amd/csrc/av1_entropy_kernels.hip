// av1_entropy_kernels.hip — K9 for the REAL syntax: the AV1 tile entropy coder on the GPU.  SURVEY.md §8a row H1 keeps entropy
// coding on the host cores and §8e names it the scaling risk; measured (profiles/): the host writer on the box's 16
// cores sustains ~100 4K frames/s while the block pipeline produces 4 900 — so the same syntax also runs here, and what crosses
// PCIe is the coded tile payloads (0.7 MB per 4K frame) instead of 25 MB of int16 levels.
//
// The syntax is av1_ops.hpp (shared source with the host's CPU twin, byte-identical to the dav1d-verified writer), in three stages:
//   k_av1_info     thread per block: level-context summary of the block + (inter) the mode that codes its vector
//   k_av1_tokens   workgroup per tile, thread per block in z-order: the block's syntax elements, counted, placed, written —
//                  literals into the tile's list, adaptive symbols as entries grouped by CDF slot
//   k_av1_chains   a slot's CDF evolves with that slot's symbols only: one lane per (tile, slot) walks the slot's entries with the
//                  CDF in registers and writes, at each element's place in the list, the tuple (icdf[s - 1], icdf[s], n - s) the
//                  range coder needs.  The serial dependency of a tile drops from all its symbols (4000-8000) to its longest
//                  chain (700-1300), and the chains of ALL tiles run side by side
//   k_av1_code     ONE LANE PER TILE: the serial range coder over the finished list.  No CDFs: its state is five registers.
//   k_av1_scan / k_av1_gather   tile payloads -> one contiguous buffer in tile order (the host adds the frame header and the
//                  tile-size fields: they depend on the largest tile, host/av1_bitstream.cpp frame_obu_from_tiles)
#include <string.h>
#include <vector>
#include "av1_ops_cdfs.hpp"
#include "av1_ops32.hpp"
#include "av1mi_internal.hpp"

namespace av1mi {
using namespace av1ops;

struct Av1EntLaunch {
  FrameView fv;                 // pointers of frame 0; frame f adds f * per-frame strides
  int nframes, sbr_n, sbc_n;
  BlockInfo *info;              // nframes * blocks
  op_t *ops; uint32_t ops_cap;  // per tile: the list (literal ops, tuples)
  uint32_t *grouped; uint32_t grouped_cap;   // per tile: the entries of the adaptive symbols, grouped by slot
  uint16_t *slot_total, *slot_base;          // per tile x S_MAX: entries of the slot, position of its first entry
  uint16_t *rec;                             // per tile x 64 blocks x kBlockRecords: the tokenizer's records
  uint32_t *nops;               // per tile
  uint8_t *slots; uint32_t slot_cap;   // per tile payload slot
  uint32_t *tile_size;          // per tile: payload bytes (0 = overflow)
  uint64_t *tile_off;           // per tile + 1: exclusive scan of the sizes
  uint32_t *status;             // bit 0: a tile's op list overflowed, bit 1: a payload slot overflowed, bit 2: out_cap too small
  uint8_t *out; uint64_t out_cap;
  const uint16_t *cdf_image; int cdf_words;   // default slot image of this frame type / q category
  SlotTable tab;
  const uint16_t *cdf_image32; SlotTable tab32;   // the same for the tiles of a key frame's 32x32 band
  const uint8_t *lr_on_frame;   // optional: [frame * 3 + plane], 0 = restoration of the plane is off in that frame
  // key frames in 32x32 blocks (av1_ops32.hpp): the first sb_rows32 superblock rows of every frame are tiles of that kind, with their
  // own slot table and default CDFs; `band` selects which tiles a chains launch walks (0 all, 1 the 32x32 band, 2 the rows below)
  int sb_rows32, band, nslots;
};

__device__ __forceinline__ FrameView frame_view(const Av1EntLaunch &L, int f) {
  FrameView v = L.fv;
  const long nb = (long)v.w8 * v.h8;
  if (v.key) { v.y_mode += nb * f; v.uv_mode += nb * f; }
  else { v.mv += nb * 2 * f; v.skip += nb * f; }
  v.lev_y += nb * 64 * f; v.lev_u += nb * 16 * f; v.lev_v += nb * 16 * f;
  v.info = L.info + nb * f;
  if (L.lr_on_frame) for (int p = 0; p < 3; p++) v.lr_on[p] = v.lr_on[p] && L.lr_on_frame[f * 3 + p];
  return v;
}

__global__ __launch_bounds__(256) void k_av1_info(Av1EntLaunch L) {
  const long nb = (long)L.fv.w8 * L.fv.h8, i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nb * L.nframes) return;
  const int f = (int)(i / nb), b = (int)(i - (long)f * nb);
  const FrameView v = frame_view(L, f);
  BlockInfo o = {};
  block_summary(v, b, &o);
  if (!v.key) inter_mode_decision(v, b / v.w8, b % v.w8, &o);
  L.info[nb * f + b] = o;
}

// the records are replayed in two passes over half of the slots each: the 16-bit positions [slots][64 blocks] of ALL slots were 25 KB
// of LDS = 6 workgroups per CU; 12.5 KB fit beside nothing larger (counts + magnitudes: 19 KB) = 8, what the registers allow
constexpr int kReplaySlots = (S_MAX + 1) / 2;
struct TokLds {
  union {
    struct { uint8_t cnt[S_MAX * kBlocksPerTile]; alignas(16) uint8_t mag[kBlocksPerTile * kMagBytes]; } p1;    // while tokenizing
    uint16_t pos[kReplaySlots * kBlocksPerTile];                                                               // while replaying: half of the slots at a time
  };
  uint16_t total[S_MAX], base[S_MAX];
  ScanTables scan;
};

// TOKENIZE once (records + counts; every read of a coefficient after the first goes to the thread's LDS copy of the block),
// PLACE (the counts of the tile's 64 blocks -> where each block's entries of each slot go), REPLAY (records -> list words and
// grouped entries).  The counts are bytes and share their LDS with the 16-bit positions that replace them: a thread keeps the
// counts of its slots in registers across the switch.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_av1_tokens(Av1EntLaunch L) {
  const int tiles = L.sbr_n * L.sbc_n, t = blockIdx.x, f = t / tiles, tt = t - f * tiles, sbr = tt / L.sbc_n, sbc = tt - sbr * L.sbc_n;
  if (sbr < L.sb_rows32) return;           // a tile of the 32x32 band: k_av1_tokens32
  const FrameView v = frame_view(L, f);
  const int zi = threadIdx.x, nslots = v.key ? S_KEY_END : S_INTER_END;
  __shared__ TokLds S;
  {
    uint32_t *m = reinterpret_cast<uint32_t *>(S.p1.cnt);
    for (int i = zi; i < S_MAX * kBlocksPerTile / 4; i += 64) m[i] = 0;
    if (zi == 0) fill_scan_tables(&S.scan);
  }
  __syncthreads();
  uint16_t *rec = L.rec + ((size_t)t * kBlocksPerTile + zi) * kBlockRecords;
  const TokScratch ts = { S.p1.mag + zi * kMagBytes, &S.scan };
  Sink k = { rec, S.p1.cnt, zi, 0, 0, false };
  tok_block(v, k, ts, sbr, sbc, zi);
  int x = k.n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d, 64);
    if (zi >= d) x += y;
  }
  const int total = __shfl(x, 63, 64), first = x - k.n;
  if ((uint32_t)total > L.ops_cap || __any(k.overflow)) {
    if (zi == 0) { atomicOr(L.status, 1u); L.nops[t] = 0; }
    for (int sl = zi; sl < S_MAX; sl += 64) L.slot_total[(size_t)t * S_MAX + sl] = 0;
    return;
  }
  if (zi == 0) L.nops[t] = (uint32_t)total;
  __syncthreads();
  // place: thread zi owns slots zi, zi + 64, ...: their counts into registers, their totals to everybody
  constexpr int kOwn = (S_MAX + 63) / 64;
  uint32_t c[kOwn][kBlocksPerTile / 4];
#pragma unroll
  for (int q = 0; q < kOwn; q++) {
    const int sl = zi + 64 * q;
    int sum = 0;
#pragma unroll
    for (int w = 0; w < kBlocksPerTile / 4; w++) {
      const uint32_t u = sl < nslots ? reinterpret_cast<const uint32_t *>(S.p1.cnt)[sl * (kBlocksPerTile / 4) + w] : 0u;
      c[q][w] = u;
      sum = (int)__builtin_amdgcn_sad_u8(u, 0u, (unsigned)sum);      // the four counts of the dword in one instruction
    }
    if (sl < S_MAX) S.total[sl] = (uint16_t)sum;
  }
  __syncthreads();         // every count is in a register now: the positions may overwrite them
  {
    int mine = 0;       // thread zi places slots [4 zi, 4 zi + 4): slots follow each other on 16-byte boundaries
    for (int q = 0; q < 4; q++) { const int sl = 4 * zi + q; if (sl < nslots) mine += (S.total[sl] + kListAlign - 1) & ~(kListAlign - 1); }
    int inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(inc, d, 64);
      if (zi >= d) inc += y;
    }
    int run = inc - mine;
    for (int q = 0; q < 4; q++) { const int sl = 4 * zi + q; if (sl < nslots) { S.base[sl] = (uint16_t)run; run += (S.total[sl] + kListAlign - 1) & ~(kListAlign - 1); } }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kOwn; q++) {
    const int sl = zi + 64 * q;
    if (sl >= S_MAX) continue;
    const bool on = sl < nslots;
    L.slot_total[(size_t)t * S_MAX + sl] = on ? S.total[sl] : (uint16_t)0;
    L.slot_base[(size_t)t * S_MAX + sl] = on ? S.base[sl] : (uint16_t)0;
  }
  // replay: the thread's own records (it wrote them itself), literals into the list, adaptive symbols into their slot's entries
  op_t *list = L.ops + (size_t)t * L.ops_cap;
  uint32_t *grouped = L.grouped + (size_t)t * L.grouped_cap;
#pragma unroll
  for (int half = 0; half < 2; half++) {
    const int lo = half * kReplaySlots, hi = lo + kReplaySlots;
    if (half) __syncthreads();                      // the first half's positions have been used up
#pragma unroll
    for (int q = 0; q < kOwn; q++) {
      const int sl = zi + 64 * q;
      if (sl < lo || sl >= hi || sl >= nslots) continue;
      int run = S.base[sl];
#pragma unroll
      for (int w = 0; w < kBlocksPerTile / 4; w++) {
        const uint32_t u = c[q][w];
        const int p0 = run, p1 = p0 + (int)(u & 0xFF), p2 = p1 + (int)((u >> 8) & 0xFF), p3 = p2 + (int)((u >> 16) & 0xFF);
        run = p3 + (int)(u >> 24);
        uint32_t *d = reinterpret_cast<uint32_t *>(S.pos + (sl - lo) * kBlocksPerTile + 4 * w);
        d[0] = (uint32_t)p0 | ((uint32_t)p1 << 16); d[1] = (uint32_t)p2 | ((uint32_t)p3 << 16);
      }
    }
    __syncthreads();
    __threadfence_block();
    replay_block(rec, k.nrec, S.pos, zi, first, list, grouped, lo, hi, half == 0);
  }
}

// The tiles of a key frame's 32x32 band: ONE LANE PER 32x32 BLOCK (16 tiles per workgroup) tokenizes serially (av1_ops32.hpp
// tok_block32: 1024 + 2 x 256 coefficients), the four lanes of a tile then place the slots and replay their records into the list
// and the grouped entries — after which the chains, the range coder and the gather treat the tile like any other.  In LDS: per tile
// the 16-bit counts / positions of its 180 slots x 4 blocks and the blocks' level summaries, per lane the magnitude map of the
// transform block at hand.  Key frames are one frame of a GOP; see DESIGN 7-1 for a finer-grained form.
constexpr int kTiles32 = 16;      // per workgroup
static_assert(kBlocks32 * kMag32Bytes >= K_END * kBlocks32 * 2 && kMag32Bytes % 8 == 0, "a tile's counters live where its four magnitude maps were");
__global__ __launch_bounds__(64) void k_av1_tokens32(Av1EntLaunch L, int ntiles_all) {
  __shared__ ScanTables32 scan;
  __shared__ __attribute__((aligned(16))) uint8_t s_mag[64 * kMag32Bytes];      // 47 KB: three workgroups per CU
  __shared__ Sum32 s_sum[kTiles32][kBlocks32];
  __shared__ int s_n[kTiles32][kBlocks32], s_nrec[kTiles32][kBlocks32], s_bad[kTiles32];
  const int tiles = L.sbr_n * L.sbc_n, tl = threadIdx.x >> 2, b = threadIdx.x & 3, t = blockIdx.x * kTiles32 + tl;
  bool live = t < ntiles_all;
  int f = 0, sbr = 0, sbc = 0;
  if (live) { f = t / tiles; const int tt = t - f * tiles; sbr = tt / L.sbc_n; sbc = tt - sbr * L.sbc_n; live = sbr < L.sb_rows32; }
  fill_scan_tables32(&scan, (int)threadIdx.x, 64);
  if (b == 0) s_bad[tl] = 0;
  const FrameView v = frame_view(L, live ? f : 0);
  const bool inside = live && !((b & 1) && sbc * 8 + 4 >= v.w8);      // width % 64 == 32: a last column of half superblocks
  if (inside) block_sums32(v, block_index32(v, sbr, sbc, b), &s_sum[tl][b]);
  __syncthreads();
  uint16_t *rec = L.rec + ((size_t)(live ? t : 0) * kBlocks32 + b) * kBlockRecords32;
  int nrec = 0;
  if (live) { s_n[tl][b] = 0; s_nrec[tl][b] = 0; }
  if (inside) {
    Sink32 k = { rec, nullptr, b, (int)kBlockRecords32, 0, 0, false, 0, 0, 0, 0 };
    const TokScratch32 ts = { s_mag + threadIdx.x * kMag32Bytes, &scan };
    tok_block32(v, k, ts, sbr, sbc, b, s_sum[tl]);
    s_n[tl][b] = k.n; s_nrec[tl][b] = nrec = k.nrec;
    if (k.overflow) atomicOr(&s_bad[tl], 1);
  }
  __syncthreads();
  // the magnitude maps are dead: the tile's 180 x 4 counters take their place
  uint16_t *cnt = reinterpret_cast<uint16_t *>(s_mag + tl * kBlocks32 * kMag32Bytes);
  for (int i = b; i < K_END * kBlocks32; i += kBlocks32) cnt[i] = 0;
  __syncthreads();
  if (live && !count_block32(rec, nrec, cnt, b)) atomicOr(&s_bad[tl], 1);
  __syncthreads();
  if (live && b == 0) {       // place: the tile's first lane (180 slots x 4 blocks: nothing beside the tokenizing)
    uint16_t *total = L.slot_total + (size_t)t * S_MAX, *base = L.slot_base + (size_t)t * S_MAX;
    const int run = place_tile32(cnt, total, base);
    for (int sl = K_END; sl < S_MAX; sl++) { total[sl] = 0; base[sl] = 0; }
    const int n = s_n[tl][0] + s_n[tl][1] + s_n[tl][2] + s_n[tl][3];
    if (s_bad[tl] || (uint32_t)n > L.ops_cap || run > 65535) {
      s_bad[tl] = 1;
      atomicOr(L.status, 1u);
      L.nops[t] = 0;
      for (int sl = 0; sl < K_END; sl++) total[sl] = 0;
    } else {
      L.nops[t] = (uint32_t)n;
    }
  }
  __syncthreads();
  if (!live || s_bad[tl]) return;
  int first = 0;
  for (int q = 0; q < b; q++) first += s_n[tl][q];
  replay_block32(rec, nrec, cnt, b, first, L.ops + (size_t)t * L.ops_cap, L.grouped + (size_t)t * L.grouped_cap);
}

// CHAINS: workgroup = one CDF slot of 64 consecutive tiles, one lane per tile.  The slot is the same for the whole wave (no
// divergence between alphabet sizes) and its chains are about equally long in neighbouring tiles, so the lanes stay busy — a
// tile's own slots differ in length by three orders of magnitude.  Small alphabets keep the CDF in registers, large ones in LDS.
// (Measured: a workgroup that walks several slots one after the other is slower — fewer, longer waves; about half of this
// kernel's time is the 4-byte scatter of the tuples into the lists.)
__global__ __launch_bounds__(64) void k_av1_chains(Av1EntLaunch L, int ntiles_all, int ngroups) {
  // workgroups are dealt to the eight XCDs in turn (id % 8), each with its own L2: all slots of a tile group go to ONE XCD, one
  // after the other, so the 64 lists the group's chains complete word by word stay in that L2 as long as possible
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nslots = L.nslots;
  const int g = (j / nslots) * 8 + xcd, sl = j - (j / nslots) * nslots, lane = threadIdx.x, t = g * 64 + lane;
  if (g >= ngroups) return;
  __shared__ __attribute__((aligned(16))) uint16_t s_big[64 * 16];
  const int n = L.tab.nsym[sl], off = L.tab.off[sl];
  int cnt = 0, base = 0;
  bool mine = t < ntiles_all;
  if (mine && L.band) { const int tiles = L.sbr_n * L.sbc_n, sbr = (t % tiles) / L.sbc_n; mine = (sbr < L.sb_rows32) == (L.band == 1); }
  if (mine) { cnt = L.slot_total[(size_t)t * S_MAX + sl]; base = L.slot_base[(size_t)t * S_MAX + sl]; }
  if (!__any(cnt > 0)) return;
  const size_t tt = t < ntiles_all ? (size_t)t : 0;
  op_t *list = L.ops + tt * L.ops_cap;
  const uint32_t *grouped = L.grouped + tt * L.grouped_cap + base;
  if (n <= 4) {
    const uint32_t *iw = reinterpret_cast<const uint32_t *>(L.cdf_image) + (off >> 1);
    uint64_t cdf = (uint64_t)iw[0] | ((uint64_t)iw[1] << 32);
    // sixteen entries (four 16-byte loads) per round, requested a whole round ahead: the chain is serial and a load takes
    // longer than four of its steps
    const uint4 *g4 = reinterpret_cast<const uint4 *>(grouped);
    const uint4 z = make_uint4(0, 0, 0, 0);
    uint4 c0 = cnt > 0 ? g4[0] : z, c1 = cnt > 4 ? g4[1] : z, c2 = cnt > 8 ? g4[2] : z, c3 = cnt > 12 ? g4[3] : z;
    auto four = [&](const uint4 &v, int e) {
      if (e < cnt) list[v.x >> 4] = small_step(cdf, (int)(v.x & 15), n);
      if (e + 1 < cnt) list[v.y >> 4] = small_step(cdf, (int)(v.y & 15), n);
      if (e + 2 < cnt) list[v.z >> 4] = small_step(cdf, (int)(v.z & 15), n);
      if (e + 3 < cnt) list[v.w >> 4] = small_step(cdf, (int)(v.w & 15), n);
    };
#pragma unroll 1
    for (int e = 0; e < cnt; e += 16) {
      const uint4 *nx = g4 + (e >> 2) + 4;
      const uint4 n0 = e + 16 < cnt ? nx[0] : z, n1 = e + 20 < cnt ? nx[1] : z, n2 = e + 24 < cnt ? nx[2] : z, n3 = e + 28 < cnt ? nx[3] : z;
      four(c0, e); four(c1, e + 4); four(c2, e + 8); four(c3, e + 12);
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
  } else {
    uint16_t *cdf = s_big + lane * 16;
    {
      const uint4 *iw = reinterpret_cast<const uint4 *>(L.cdf_image + off);
      uint4 *d = reinterpret_cast<uint4 *>(cdf);
      d[0] = iw[0];
      d[1] = n > 8 ? iw[1] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll 1
    for (int e = 0; e < cnt; e++) {
      const uint32_t v = grouped[e];
      const int s = (int)(v & 15);
      list[v >> 4] = s >= kSplitHorz ? split_tuple(cdf, s) : big_step(cdf, s, n);
    }
  }
}

// 64 lanes (tiles) per workgroup; per lane in LDS a ring of 64 list words and the coder's byte stage (32 x 16 bits).
//
// The chain is serial and the lanes of a wave are at different places of different tiles: what counts is the latency of one step.
//  * one interval update per step for every lane, whatever the word: a tuple, or one bit of a literal;
//  * NO global memory in the steady state (a wait for memory on gfx9 waits for every outstanding vector-memory operation,
//    stores included).  Words come from the LDS ring, bytes go to the LDS stage; both meet global memory in rare PHASES the whole
//    wave takes together: 32 words per lane are requested into registers in one phase and written to the ring in the next (the
//    loads have long landed), the stages are stored.  A phase is due when a lane has fewer than 24 words staged or a stage is
//    nearly full; that is looked at every eight steps.
constexpr int kRingOps = 64, kLaneWords = kRingOps * 2 + Coder::kStage + 8;     // uint16 per lane; + 8: the lanes start on different banks

__global__ __launch_bounds__(64) void k_av1_code(Av1EntLaunch L, int ntiles_all) {
  // (Measured, round 3: s_setprio(3) for these lone latency-bound waves changes nothing, with four or with eight hardware queues —
  // they do not lose issue slots to co-resident waves; what the streams compete for is residency: LDS and wave slots per CU.)
  __shared__ __attribute__((aligned(16))) uint16_t s_lane[64 * kLaneWords];
  const int lane = threadIdx.x, t = blockIdx.x * 64 + lane;
  const bool live = t < ntiles_all;
  uint32_t *ring = reinterpret_cast<uint32_t *>(s_lane + lane * kLaneWords);
  Coder c;
  c.init(L.slots + (size_t)(live ? t : 0) * L.slot_cap, (int)L.slot_cap, s_lane + lane * kLaneWords + kRingOps * 2);
  const int n = live ? (int)L.nops[t] : 0;
  const uint4 *ops4 = reinterpret_cast<const uint4 *>(L.ops + (size_t)(live ? t : 0) * L.ops_cap);
  // words [w - 64, w) are in the ring; `pend`: words [w, w + 32) are on their way into r0..r7
  uint4 r0, r1, r2, r3, r4, r5, r6, r7;
  r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = make_uint4(0, 0, 0, 0);
  int w = 0;
  bool pend = false;
  auto request = [&]() {
    const uint4 *p = ops4 + (w >> 2);
    if (w < n) r0 = p[0];
    if (w + 4 < n) r1 = p[1];
    if (w + 8 < n) r2 = p[2];
    if (w + 12 < n) r3 = p[3];
    if (w + 16 < n) r4 = p[4];
    if (w + 20 < n) r5 = p[5];
    if (w + 24 < n) r6 = p[6];
    if (w + 28 < n) r7 = p[7];
    pend = w < n;
  };
  auto take = [&]() {
    uint4 *q = reinterpret_cast<uint4 *>(ring + (w & 32));
    q[0] = r0; q[1] = r1; q[2] = r2; q[3] = r3; q[4] = r4; q[5] = r5; q[6] = r6; q[7] = r7;
    w += 32;
  };
  request();
  if (pend) { take(); request(); }
  int i = 0, lit_left = 0;
  // the word of this step and the next two in registers; the ring read of a step is issued before its arithmetic and used after
  // it (read and moved back to back, every step waited a whole LDS latency)
  uint32_t op = ring[0], op_next = ring[1], op_next2 = ring[2];
  for (;;) {
    // the checks every eight steps: a lane advances by at most eight words and eight stage entries in between
    if (!__any(lit_left > 0 || i < n)) break;
    if (__any((pend && w - i < 24) || c.ns >= Coder::kStage - 8)) {      // a phase
      if (pend && w - i <= 32) { take(); request(); }
      c.spill();
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (lit_left > 0 || i < n) {
        const uint32_t fetched = ring[(i + 3) & (kRingOps - 1)];      // requested first, used last: the word an advance shifts in
        uint32_t tup = op;
        if (lit_left == 0 && (op >> 31)) lit_left = (int)((op >> 27) & 15);
        if (lit_left > 0) {         // one equiprobable bit: the tuples of (icdf 16384 | 0, symbol 1 of 2) and (32768 | 16384, symbol 0 of 2)
          lit_left--;
          tup = (op >> lit_left) & 1 ? (1u << 19) | (256u << 9) : (2u << 19) | (512u << 9) | 256u;
        }
        c.encode_tuple(tup);
        if (lit_left == 0) { i++; op = op_next; op_next = op_next2; op_next2 = fetched; }
      }
    }
  }
  if (!live) return;
  int sz = n ? c.finish() : -1;          // n == 0: the list overflowed (k_av1_tokens), nothing to code
  if (sz < 0) { if (n) atomicOr(L.status, 2u); sz = 0; }
  L.tile_size[t] = (uint32_t)sz;
}

// exclusive scan of all tile sizes (one workgroup; a batch has at most a few 10^4 tiles)
__global__ __launch_bounds__(1024) void k_av1_scan(Av1EntLaunch L, int ntiles_all) {
  __shared__ uint64_t part[1024];
  const int tid = threadIdx.x, per = (ntiles_all + 1023) / 1024, i0 = tid * per, i1 = min(i0 + per, ntiles_all);
  uint64_t s = 0;
  for (int i = i0; i < i1; i++) s += L.tile_size[i];
  part[tid] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint64_t y = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += y;
    __syncthreads();
  }
  uint64_t run = part[tid] - s;
  for (int i = i0; i < i1; i++) { L.tile_off[i] = run; run += L.tile_size[i]; }
  if (tid == 1023) {
    L.tile_off[ntiles_all] = part[1023];
    if (part[1023] > L.out_cap) atomicOr(L.status, 4u);
  }
}
// one workgroup per tile; `out` may be pinned host memory (the GOP session hands the payloads straight to the host this way):
// bytes up to the first 16-byte boundary of the destination, then aligned 16-byte stores, then the tail
__global__ __launch_bounds__(64) void k_av1_gather(Av1EntLaunch L) {
  const int t = blockIdx.x;
  const uint32_t n = L.tile_size[t];
  const uint64_t off = L.tile_off[t];
  if (off + n > L.out_cap) return;
  const uint8_t *src = L.slots + (size_t)t * L.slot_cap;
  uint8_t *dst = L.out + off;
  const uint32_t head = min(n, (uint32_t)((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15)), body = (n - head) >> 4, tail0 = head + (body << 4);
  if (threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x];
  for (uint32_t i = threadIdx.x; i < body; i += 64) {
    uint4 v;
    __builtin_memcpy(&v, src + head + 16 * i, 16);       // the source is not aligned with the destination
    *reinterpret_cast<uint4 *>(dst + head + 16 * i) = v;
  }
  if (tail0 + threadIdx.x < n) dst[tail0 + threadIdx.x] = src[tail0 + threadIdx.x];
}

hipError_t launch_av1_front(av1mi_ctx *ctx, const Av1EntLaunch &L, hipStream_t s) {
  const long nb = (long)L.fv.w8 * L.fv.h8 * L.nframes;
  const int ntiles_all = L.sbr_n * L.sbc_n * L.nframes, ngroups = (ntiles_all + 63) / 64;
  ProfToken t = ctx_prof_begin(ctx, AV1MI_K_ENTROPY_TOKENS, s);
  hipLaunchKernelGGL(k_av1_info, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, L);
  if (L.sb_rows32 < L.sbr_n) hipLaunchKernelGGL(k_av1_tokens, dim3((unsigned)ntiles_all), dim3(64), 0, s, L);
  if (L.sb_rows32) hipLaunchKernelGGL(k_av1_tokens32, dim3((unsigned)((ntiles_all + kTiles32 - 1) / kTiles32)), dim3(64), 0, s, L, ntiles_all);
  ctx_prof_end(ctx, t, s);
  t = ctx_prof_begin(ctx, AV1MI_K_ENTROPY_CHAINS, s);
  Av1EntLaunch C = L;
  C.nslots = L.fv.key ? S_KEY_END : S_INTER_END; C.band = L.sb_rows32 ? 2 : 0;
  if (L.sb_rows32 < L.sbr_n) hipLaunchKernelGGL(k_av1_chains, dim3((unsigned)(((ngroups + 7) / 8) * 8 * C.nslots)), dim3(64), 0, s, C, ntiles_all, ngroups);
  if (L.sb_rows32) {       // the 32x32 band's tiles: their slot table and default CDFs
    C.nslots = K_END; C.band = 1; C.tab = L.tab32; C.cdf_image = L.cdf_image32;
    hipLaunchKernelGGL(k_av1_chains, dim3((unsigned)(((ngroups + 7) / 8) * 8 * C.nslots)), dim3(64), 0, s, C, ntiles_all, ngroups);
  }
  ctx_prof_end(ctx, t, s);
  return hipGetLastError();
}
hipError_t launch_av1_back(av1mi_ctx *ctx, const Av1EntLaunch &L, hipStream_t s) {
  const int ntiles_all = L.sbr_n * L.sbc_n * L.nframes;
  ProfToken t = ctx_prof_begin(ctx, AV1MI_K_ENTROPY, s);
  hipLaunchKernelGGL(k_av1_code, dim3((unsigned)((ntiles_all + 63) / 64)), dim3(64), 0, s, L, ntiles_all);
  ctx_prof_end(ctx, t, s);
  t = ctx_prof_begin(ctx, AV1MI_K_ENTROPY_PACK, s);
  hipLaunchKernelGGL(k_av1_scan, dim3(1), dim3(1024), 0, s, L, ntiles_all);
  hipLaunchKernelGGL(k_av1_gather, dim3((unsigned)ntiles_all), dim3(64), 0, s, L);
  ctx_prof_end(ctx, t, s);
  return hipGetLastError();
}

}  // namespace av1mi

// ------------------------------------------------------------------------------------------------ C ABI
struct av1mi_av1ent_state {      // per-context scratch of the coder, grown on demand (owned through av1mi_av1_entropy_release)
  void *info = nullptr, *ops[2] = { nullptr, nullptr }, *nops[2] = { nullptr, nullptr }, *grouped = nullptr, *slot_tb = nullptr, *rec = nullptr, *slots = nullptr, *tile_off = nullptr, *status = nullptr;
  size_t info_b = 0, ops_b[2] = { 0, 0 }, nops_b[2] = { 0, 0 }, grouped_b = 0, slot_tb_b = 0, rec_b = 0, slots_b = 0, off_b = 0;
  // the lists (ops, nops) exist twice: job k uses set k & 1, so the front half of job k + 1 may run beside the back half of job k
  hipEvent_t lists_ready[2] = { nullptr, nullptr }, lists_free[2] = { nullptr, nullptr };
  bool free_recorded[2] = { false, false };
  unsigned long jobs = 0;
  av1mi::Av1EntLaunch pending[2];             // a job whose back half (range coder, scan, gather) is still to be launched (av1_entropy_back)
  uint64_t *pending_total[2] = { nullptr, nullptr };
  bool pending_valid[2] = { false, false };
  size_t last_tiles = 0;          // tiles of the most recent job (av1mi_av1_entropy_last_list_words)
  uint16_t *d_image[2][4] = {};   // [key][qcat] default CDF images
  uint16_t *d_image32[4] = {};    // [qcat] the 32x32 band's
  av1ops::SlotTable tab32;
  int image_words[2] = { 0, 0 };
  av1ops::SlotTable tab[2];
};

namespace {
int grow(av1mi_ctx *ctx, void **p, size_t *have, size_t need) {
  if (*have >= need) return AV1MI_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr; *have = 0;
  if (hipMalloc(p, need) != hipSuccess) return av1mi::ctx_fail(ctx, AV1MI_E_NOMEM, "AV1 entropy scratch: hipMalloc(%zu) failed", need);
  *have = need;
  return AV1MI_OK;
}
}  // namespace

extern "C" {

// Capacities, sized on the densest content measured (tools: AV1MI_TOK_STATS in host/av1_opstream.cpp; 4K 10-bit noise-dominated key
// frames at the reference's quality 23: <= 485 records in a block before the tile's restoration / CDEF syntax, which rides on the
// tile's first block — 512 overflowed there —, <= 22 394 list words in a tile).  Not the syntax's worst case (~820 records per block,
// ~52 000 words per tile): a tile beyond them raises a status bit and the session hands the batch to the host writer.
uint32_t av1mi_av1_entropy_ops_per_tile(void) { return 32768; }
uint32_t av1mi_av1_entropy_slot_bytes(void) { return 16384; }

}  // extern "C"

int av1mi::av1_entropy_front(av1mi_ctx *ctx, const av1mi_av1_entropy_job *j, hipStream_t front, int *ticket) {
  if (!ctx || !ticket) return AV1MI_E_INVAL;
  *ticket = -1;
  if (!j) return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "null job");
  if (hipSetDevice(av1mi::ctx_device(ctx)) != hipSuccess) return av1mi::ctx_fail(ctx, AV1MI_E_DEVICE, "hipSetDevice failed");
  if (j->width <= 0 || j->height <= 0 || (j->width & 7) || (j->height & 7) || j->width > 4096 || j->height > 4096)
    return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "frame %dx%d must be a multiple of 8, at most 4096x4096", j->width, j->height);
  if (j->nframes < 0 || j->nframes > 4096 || j->base_q_idx < 1 || j->base_q_idx > 255) return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "bad nframes / base_q_idx");
  const void *need[] = { j->d_lev_y, j->d_lev_u, j->d_lev_v, j->d_out, j->d_tile_size, j->d_total, j->key ? (const void *)j->d_modes_y : (const void *)j->d_mvs,
                         j->key ? (const void *)j->d_modes_uv : (const void *)j->d_skip };
  for (const void *p : need) if (!p) return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "null device pointer");
  if (j->nframes == 0) return AV1MI_OK;
  av1mi_av1ent_state *st = av1mi::ctx_av1ent(ctx);
  if (!st) return av1mi::ctx_fail(ctx, AV1MI_E_NOMEM, "AV1 entropy state");
  const int key = j->key ? 1 : 0, qcat = j->base_q_idx <= 20 ? 0 : j->base_q_idx <= 60 ? 1 : j->base_q_idx <= 120 ? 2 : 3;
  if (!st->d_image[key][qcat]) {
    av1ops::SlotTable tab;
    const std::vector<uint16_t> img = av1ops::default_slot_image(key != 0, qcat, &tab);
    st->tab[key] = tab; st->image_words[key] = tab.words;
    if (hipMalloc((void **)&st->d_image[key][qcat], img.size() * 2) != hipSuccess ||
        hipMemcpy(st->d_image[key][qcat], img.data(), img.size() * 2, hipMemcpyHostToDevice) != hipSuccess)
      return av1mi::ctx_fail(ctx, AV1MI_E_DEVICE, "uploading the default CDF image failed");
  }
  if (j->key_rows32 && (!key || (j->key_rows32 & 63) || j->key_rows32 > j->height || (j->width & 31)))
    return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "key_rows32 %d: whole superblock rows of a key frame whose width is a multiple of 32", j->key_rows32);
  if (j->key_rows32 && !st->d_image32[qcat]) {
    const std::vector<uint16_t> img = av1ops::default_slot_image_k32(qcat, &st->tab32);
    if (hipMalloc((void **)&st->d_image32[qcat], img.size() * 2) != hipSuccess ||
        hipMemcpy(st->d_image32[qcat], img.data(), img.size() * 2, hipMemcpyHostToDevice) != hipSuccess)
      return av1mi::ctx_fail(ctx, AV1MI_E_DEVICE, "uploading the default CDF image failed");
  }
  av1mi::Av1EntLaunch L;
  memset(&L, 0, sizeof(L));
  L.sb_rows32 = j->key_rows32 / 64;
  if (j->key_rows32) { av1ops::build_slot_table_k32(&L.tab32); L.cdf_image32 = st->d_image32[qcat]; }
  L.fv.w8 = j->width / 8; L.fv.h8 = j->height / 8; L.fv.key = key;
  L.fv.y_mode = j->d_modes_y; L.fv.uv_mode = j->d_modes_uv; L.fv.mv = j->d_mvs; L.fv.skip = j->d_skip;
  L.fv.lev_y = j->d_lev_y; L.fv.lev_u = j->d_lev_u; L.fv.lev_v = j->d_lev_v;
  for (int p = 0; p < 3; p++) {
    L.fv.lr_on[p] = j->lr_on[p] != 0;
    const int vh = j->visible_height ? j->visible_height : j->height, vw = j->visible_width ? j->visible_width : j->width;   // the units tile the TRUE frame
    const int ph = p ? (vh + 1) / 2 : vh, pw = p ? (vw + 1) / 2 : vw;
    L.fv.lr_rows[p] = (ph + 32) / 64 > 1 ? (ph + 32) / 64 : 1; L.fv.lr_cols[p] = (pw + 32) / 64 > 1 ? (pw + 32) / 64 : 1;
  }
  memcpy(L.fv.lr_unit[0], j->lr_unit_y, 8); memcpy(L.fv.lr_unit[1], j->lr_unit_uv, 8);
  L.nframes = j->nframes; L.sbr_n = (L.fv.h8 + 7) / 8; L.sbc_n = (L.fv.w8 + 7) / 8;
  const size_t nb = (size_t)L.fv.w8 * L.fv.h8 * j->nframes, nt = (size_t)L.sbr_n * L.sbc_n * j->nframes;
  L.ops_cap = av1mi_av1_entropy_ops_per_tile(); L.slot_cap = av1mi_av1_entropy_slot_bytes();
  L.grouped_cap = L.ops_cap + av1ops::kListAlign * av1ops::S_MAX;      // every slot's entries start on 16 bytes
  const int par = (int)(st->jobs++ & 1);
  st->last_tiles = nt;
  int rc;
  if ((rc = grow(ctx, &st->info, &st->info_b, nb * sizeof(av1ops::BlockInfo))) ||
      (rc = grow(ctx, &st->ops[par], &st->ops_b[par], nt * L.ops_cap * sizeof(av1ops::op_t))) || (rc = grow(ctx, &st->nops[par], &st->nops_b[par], nt * 4)) ||
      (rc = grow(ctx, &st->grouped, &st->grouped_b, nt * L.grouped_cap * sizeof(uint32_t))) ||
      (rc = grow(ctx, &st->slot_tb, &st->slot_tb_b, nt * av1ops::S_MAX * 4)) ||
      (rc = grow(ctx, &st->rec, &st->rec_b, nt * av1ops::kBlocksPerTile * av1ops::kBlockRecords * sizeof(uint16_t))) ||
      (rc = grow(ctx, &st->slots, &st->slots_b, nt * L.slot_cap)) || (rc = grow(ctx, &st->tile_off, &st->off_b, (nt + 1) * 8)))
    return rc;
  L.info = (av1ops::BlockInfo *)st->info; L.ops = (av1ops::op_t *)st->ops[par]; L.nops = (uint32_t *)st->nops[par]; L.grouped = (uint32_t *)st->grouped;
  L.slot_total = (uint16_t *)st->slot_tb; L.slot_base = L.slot_total + nt * av1ops::S_MAX; L.rec = (uint16_t *)st->rec; L.slots = (uint8_t *)st->slots;
  L.tile_off = (uint64_t *)st->tile_off;
  L.tile_size = j->d_tile_size; L.out = j->d_out; L.out_cap = j->out_cap;
  L.status = (uint32_t *)(j->d_total + 1);
  L.cdf_image = st->d_image[key][qcat]; L.cdf_words = st->image_words[key]; L.tab = st->tab[key];
  L.lr_on_frame = j->d_lr_on;
#define E_HIP(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return av1mi::ctx_fail(ctx, AV1MI_E_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); } while (0)
  for (hipEvent_t *ev : { &st->lists_ready[par], &st->lists_free[par] }) if (!*ev) E_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
  if (st->free_recorded[par]) E_HIP(hipStreamWaitEvent(front, st->lists_free[par], 0));      // the coder of two jobs ago still reads this set of lists
  E_HIP(hipMemsetAsync(j->d_total, 0, 16, front));
  E_HIP(av1mi::launch_av1_front(ctx, L, front));
  E_HIP(hipEventRecord(st->lists_ready[par], front));
  st->pending[par] = L; st->pending_total[par] = j->d_total; st->pending_valid[par] = true;
  *ticket = par;
  return AV1MI_OK;
}

// the back half of the job `ticket` names (av1_entropy_front): the serial range coder, the scan of the tile sizes, the gather
int av1mi::av1_entropy_back(av1mi_ctx *ctx, int ticket, hipStream_t back) {
  if (!ctx || ticket < 0 || ticket > 1) return AV1MI_E_INVAL;
  av1mi_av1ent_state *st = av1mi::ctx_av1ent(ctx);
  if (!st || !st->pending_valid[ticket]) return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "no entropy job pending under this ticket");
  const av1mi::Av1EntLaunch &L = st->pending[ticket];
  const size_t nt = (size_t)L.sbr_n * L.sbc_n * L.nframes;
  E_HIP(hipStreamWaitEvent(back, st->lists_ready[ticket], 0));
  E_HIP(av1mi::launch_av1_back(ctx, L, back));
  E_HIP(hipMemcpyAsync(st->pending_total[ticket], L.tile_off + nt, 8, hipMemcpyDeviceToDevice, back));      // total bytes next to the status: tile_off[nt]
  E_HIP(hipEventRecord(st->lists_free[ticket], back)); st->free_recorded[ticket] = true;
  st->pending_valid[ticket] = false;
#undef E_HIP
  return AV1MI_OK;
}

int av1mi::av1_entropy_submit(av1mi_ctx *ctx, const av1mi_av1_entropy_job *j, hipStream_t front, hipStream_t back) {
  int ticket = -1;
  const int rc = av1_entropy_front(ctx, j, front, &ticket);
  if (rc != AV1MI_OK || ticket < 0) return rc;      // (an empty job has no back half)
  return av1_entropy_back(ctx, ticket, back);
}

extern "C" {

int av1mi_av1_entropy_encode_on(av1mi_ctx *ctx, const av1mi_av1_entropy_job *j, void *stream_handle) {
  if (!ctx) return AV1MI_E_INVAL;
  hipStream_t s = stream_handle ? (hipStream_t)stream_handle : av1mi::ctx_stream(ctx);
  return av1mi::av1_entropy_submit(ctx, j, s, s);
}

int av1mi_av1_entropy_encode(av1mi_ctx *ctx, const av1mi_av1_entropy_job *j) { return av1mi_av1_entropy_encode_on(ctx, j, nullptr); }

int av1mi_av1_entropy_last_list_words(av1mi_ctx *ctx, uint64_t *words) {
  if (!ctx || !words) return AV1MI_E_INVAL;
  *words = 0;
  av1mi_av1ent_state *st = av1mi::ctx_av1ent(ctx);
  if (!st || !st->jobs || !st->last_tiles) return AV1MI_OK;
  if (hipSetDevice(av1mi::ctx_device(ctx)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return av1mi::ctx_fail(ctx, AV1MI_E_DEVICE, "synchronising failed");
  std::vector<uint32_t> n(st->last_tiles);
  if (hipMemcpy(n.data(), st->nops[(st->jobs - 1) & 1], n.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return av1mi::ctx_fail(ctx, AV1MI_E_DEVICE, "reading the list sizes failed");
  for (uint32_t v : n) *words += v;
  return AV1MI_OK;
}

}  // extern "C"

namespace av1mi {
void av1ent_free(av1mi_av1ent_state *st) {
  if (!st) return;
  for (void *p : { st->info, st->ops[0], st->ops[1], st->nops[0], st->nops[1], st->grouped, st->slot_tb, st->rec, st->slots, st->tile_off }) if (p) (void)hipFree(p);
  for (hipEvent_t ev : { st->lists_ready[0], st->lists_ready[1], st->lists_free[0], st->lists_free[1] }) if (ev) (void)hipEventDestroy(ev);
  for (auto &k : st->d_image) for (uint16_t *p : k) if (p) (void)hipFree(p);
  delete st;
}
av1mi_av1ent_state *av1ent_new() { return new (std::nothrow) av1mi_av1ent_state(); }
}  // namespace av1mi
