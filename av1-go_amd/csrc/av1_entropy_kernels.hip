// av1_entropy_kernels.hip — K9 for the REAL syntax: the AV1 tile entropy coder on the GPU.  SURVEY.md §8a row H1 keeps entropy
// coding on the host cores and §8e names it the scaling risk; measured (profiles/): the host writer on the box's 16
// cores sustains ~100 4K frames/s while the block pipeline produces 4 900 — so the same syntax also runs here, and what crosses
// PCIe is the coded tile payloads (0.7 MB per 4K frame) instead of 25 MB of int16 levels.
//
// The syntax is av1_ops.hpp (shared source with the host's CPU twin, byte-identical to the dav1d-verified writer):
//   k_av1_info     thread per block: level-context summary of the block + (inter) the mode that codes its vector
//   k_av1_tokens   workgroup per tile, thread per block in z-order: the block's ops, counted, scanned, written
//   k_av1_code     ONE LANE PER TILE: the serial range coder over the tile's op list, the tile's CDFs in LDS
//   k_av1_scan / k_av1_gather   tile payloads -> one contiguous buffer in tile order (the host adds the frame header and the
//                  tile-size fields: they depend on the largest tile, host/av1_bitstream.cpp frame_obu_from_tiles)
// The coder is a latency-bound dependent chain per lane (a wave = up to 32 tiles in lockstep); it is meant to run BESIDE the
// block pipeline of the next frame (GOP session, side stream), not to be fast alone.
#include <string.h>
#include <vector>
#include "av1_ops_cdfs.hpp"
#include "av1mi_internal.hpp"

namespace av1mi {
using namespace av1ops;

struct Av1EntLaunch {
  FrameView fv;                 // pointers of frame 0; frame f adds f * per-frame strides
  int nframes, sbr_n, sbc_n;
  BlockInfo *info;              // nframes * blocks
  op_t *ops; uint32_t ops_cap;  // per tile (32-bit ops)
  uint32_t *nops;               // per tile
  uint8_t *slots; uint32_t slot_cap;   // per tile payload slot
  uint32_t *tile_size;          // per tile: payload bytes (0 = overflow)
  uint64_t *tile_off;           // per tile + 1: exclusive scan of the sizes
  uint32_t *status;             // bit 0: a tile's op list overflowed, bit 1: a payload slot overflowed, bit 2: out_cap too small
  uint8_t *out; uint64_t out_cap;
  const uint16_t *cdf_image; int cdf_words;   // default slot image of this frame type / q category
  SlotTable tab;
};

__device__ __forceinline__ FrameView frame_view(const Av1EntLaunch &L, int f) {
  FrameView v = L.fv;
  const long nb = (long)v.w8 * v.h8;
  if (v.key) { v.y_mode += nb * f; v.uv_mode += nb * f; }
  else { v.mv += nb * 2 * f; v.skip += nb * f; }
  v.lev_y += nb * 64 * f; v.lev_u += nb * 16 * f; v.lev_v += nb * 16 * f;
  v.info = L.info + nb * f;
  return v;
}

__global__ __launch_bounds__(256) void k_av1_info(Av1EntLaunch L) {
  const long nb = (long)L.fv.w8 * L.fv.h8, i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nb * L.nframes) return;
  const int f = (int)(i / nb), b = (int)(i - (long)f * nb);
  const FrameView v = frame_view(L, f);
  BlockInfo o;
  o.mode = 0; o.flags = 0;
  block_summary(v, b, &o);
  if (!v.key) inter_mode_decision(v, b / v.w8, b % v.w8, &o);
  L.info[nb * f + b] = o;
}

__global__ __launch_bounds__(64) void k_av1_tokens(Av1EntLaunch L) {
  const int tiles = L.sbr_n * L.sbc_n, t = blockIdx.x, f = t / tiles, tt = t - f * tiles, sbr = tt / L.sbc_n, sbc = tt - sbr * L.sbc_n;
  const FrameView v = frame_view(L, f);
  const int zi = threadIdx.x;
  __shared__ SlotTable s_tab;          // per-thread indexed lookups when the ops are written: from LDS, not from the kernel arguments
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&L.tab);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&s_tab);
    for (int i = zi; i < (int)(sizeof(SlotTable) / 4); i += 64) dst[i] = src[i];
  }
  __syncthreads();
  Sink cnt = { nullptr, 0, &s_tab };
  tok_block(v, cnt, sbr, sbc, zi);
  // exclusive scan of the 64 counts (one wave)
  int x = cnt.n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(x, d, 64);
    if (zi >= d) x += y;
  }
  const int total = __shfl(x, 63, 64), off = x - cnt.n;
  if (zi == 0) L.nops[t] = (uint32_t)total;
  if ((uint32_t)total > L.ops_cap) {
    if (zi == 0) { atomicOr(L.status, 1u); L.nops[t] = 0; }
    return;
  }
  Sink w = { L.ops + (size_t)t * L.ops_cap + off, 0, &s_tab };
  tok_block(v, w, sbr, sbc, zi);
}

// blockDim.x lanes (tiles) per workgroup, chosen so that their CDF copies fit 64 KB of LDS (16 for key frames, 32 for inter
// frames); each lane's CDF storage is `stride` uint16 apart in dynamic LDS
__global__ __launch_bounds__(32) void k_av1_code(Av1EntLaunch L, int ntiles_all, int stride) {
  const int LPW = (int)blockDim.x;
  extern __shared__ uint16_t s_cdf[];
  const int lane = threadIdx.x, t = blockIdx.x * LPW + lane;
  const bool live = t < ntiles_all;
  uint16_t *cdf = s_cdf + lane * stride;
  for (int i = 0; i < L.cdf_words; i++) cdf[i] = L.cdf_image[i];     // init_symbol: the frame's default CDFs
  Coder c;
  c.init(L.slots + (size_t)(live ? t : 0) * L.slot_cap, (int)L.slot_cap);
  const int n = live ? (int)L.nops[t] : 0;
  // ONE interval update per loop iteration for every lane, whatever the op: an adaptive symbol, or one bit of a literal (a
  // literal of k bits takes k iterations).  The lanes of a wave code different tiles and meet different ops at every step; with
  // "symbol" and "literal" as separate paths the wave executed both, literal loop included, at almost every step (measured:
  // 4 us per op).  Here the paths differ only in where (fl, fh) come from and whether the CDF adapts.
  // The op list is read four ops (one 16-byte load) at a time, the next chunk requested before the current one is coded.
  const uint4 *ops4 = reinterpret_cast<const uint4 *>(L.ops + (size_t)(live ? t : 0) * L.ops_cap);
  uint4 cur = n > 0 ? ops4[0] : make_uint4(0, 0, 0, 0), nxt = n > 4 ? ops4[1] : make_uint4(0, 0, 0, 0);
  int i = 0, lit_left = 0;
  uint32_t lit_val = 0;
  while (__any(lit_left > 0 || i < n)) {
    if (!(lit_left > 0 || i < n)) continue;
    uint32_t fl = 32768u, fh = 0;
    int sym = 0, ns = 2;
    uint16_t *v = nullptr;
    if (lit_left == 0) {
      const int k = i & 3;
      const uint32_t op = k == 0 ? cur.x : k == 1 ? cur.y : k == 2 ? cur.z : cur.w;
      i++;
      if ((i & 3) == 0) { cur = nxt; if (i + 4 < n) nxt = ops4[(i >> 2) + 1]; }
      if (op & 0x80000000u) {
        const int nbits = (op >> 27) & 15;
        if (nbits) { lit_left = nbits; lit_val = op & 0x7FFu; }
        else {   // split_or_horz / split_or_vert: "split" with the probability gathered from the partition CDF as it stands
          const uint16_t *p = cdf + (op & 0xFFF);
          const uint32_t p1 = p[0] - p[1], p2 = p[1] - p[2], p3 = p[2] - p[3], p4 = p[3] - p[4], p5 = p[4] - p[5], p6 = p[5] - p[6], p7 = p[6] - p[7],
                         p8 = p[7] - p[8], p9 = p[8];
          fl = ((op >> 26) & 1) == 0 ? p2 + p3 + p4 + p6 + p7 + p9 : p1 + p3 + p4 + p5 + p6 + p8;
          fh = 0; sym = 1; ns = 2;
        }
      } else {
        sym = op & 15; ns = (op >> 4) & 31;
        v = cdf + ((op >> 9) & 0xFFF);
        fl = sym ? v[sym - 1] : 32768u;
        fh = sym == ns - 1 ? 0u : v[sym];
      }
    }
    if (lit_left > 0) {
      lit_left--;
      sym = (lit_val >> lit_left) & 1;
      fl = sym ? 16384u : 32768u; fh = sym ? 0u : 16384u; ns = 2;
    }
    c.encode(fl, fh, sym, ns);
    if (v) {   // adaptation (8.2.6): alphabets of up to four symbols (95 % of all) without a loop
      const int count = v[ns - 1];
      const int rate = 3 + (count > 15) + (count > 31) + (ns >= 4 ? 2 : 1);
      if (ns <= 4) {
#pragma unroll
        for (int q = 0; q < 3; q++)
          if (q < ns - 1) { const int x = v[q]; v[q] = (uint16_t)(q < sym ? x + ((32768 - x) >> rate) : x - (x >> rate)); }
      } else {
        for (int q = 0; q < ns - 1; q++) { const int x = v[q]; v[q] = (uint16_t)(q < sym ? x + ((32768 - x) >> rate) : x - (x >> rate)); }
      }
      v[ns - 1] = (uint16_t)(count + (count < 32));
    }
  }
  if (!live) return;
  int sz = n ? c.finish() : -1;          // n == 0: the op list overflowed (k_av1_tokens), nothing to code
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
__global__ __launch_bounds__(64) void k_av1_gather(Av1EntLaunch L) {
  const int t = blockIdx.x;
  const uint32_t n = L.tile_size[t];
  const uint64_t off = L.tile_off[t];
  if (off + n > L.out_cap) return;
  const uint8_t *src = L.slots + (size_t)t * L.slot_cap;
  for (uint32_t i = threadIdx.x; i < n; i += 64) L.out[off + i] = src[i];
}

hipError_t launch_av1_entropy(const Av1EntLaunch &L, hipStream_t s) {
  if (L.nframes <= 0) return hipSuccess;
  const long nb = (long)L.fv.w8 * L.fv.h8 * L.nframes;
  const int ntiles_all = L.sbr_n * L.sbc_n * L.nframes;
  hipLaunchKernelGGL(k_av1_info, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, L);
  hipLaunchKernelGGL(k_av1_tokens, dim3((unsigned)ntiles_all), dim3(64), 0, s, L);
  const int stride = (((L.cdf_words + 1) / 2) | 1) * 2;     // an odd number of dwords: the lanes' copies of a slot fall on different banks
  int LPW = 32;
  while (LPW > 1 && (size_t)LPW * stride * sizeof(uint16_t) > 60 * 1024) LPW >>= 1;      // static + dynamic LDS stay below 64 KB per workgroup
  const size_t lds = (size_t)LPW * stride * sizeof(uint16_t);
  hipLaunchKernelGGL(k_av1_code, dim3((unsigned)((ntiles_all + LPW - 1) / LPW)), dim3(LPW), lds, s, L, ntiles_all, stride);
  hipLaunchKernelGGL(k_av1_scan, dim3(1), dim3(1024), 0, s, L, ntiles_all);
  hipLaunchKernelGGL(k_av1_gather, dim3((unsigned)ntiles_all), dim3(64), 0, s, L);
  return hipGetLastError();
}

}  // namespace av1mi

// ------------------------------------------------------------------------------------------------ C ABI
struct av1mi_av1ent_state {      // per-context scratch of the coder, grown on demand (owned through av1mi_av1_entropy_release)
  void *info = nullptr, *ops = nullptr, *nops = nullptr, *slots = nullptr, *tile_off = nullptr, *status = nullptr;
  size_t info_b = 0, ops_b = 0, nops_b = 0, slots_b = 0, off_b = 0;
  uint16_t *d_image[2][4] = {};   // [key][qcat] default CDF images
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

uint32_t av1mi_av1_entropy_ops_per_tile(void) { return 24576; }
uint32_t av1mi_av1_entropy_slot_bytes(void) { return 16384; }

int av1mi_av1_entropy_encode_on(av1mi_ctx *ctx, const av1mi_av1_entropy_job *j, void *stream_handle) {
  if (!ctx) return AV1MI_E_INVAL;
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
  hipStream_t s = stream_handle ? (hipStream_t)stream_handle : av1mi::ctx_stream(ctx);
  const int key = j->key ? 1 : 0, qcat = j->base_q_idx <= 20 ? 0 : j->base_q_idx <= 60 ? 1 : j->base_q_idx <= 120 ? 2 : 3;
  if (!st->d_image[key][qcat]) {
    av1ops::SlotTable tab;
    const std::vector<uint16_t> img = av1ops::default_slot_image(key != 0, qcat, &tab);
    st->tab[key] = tab; st->image_words[key] = tab.words;
    if (hipMalloc((void **)&st->d_image[key][qcat], img.size() * 2) != hipSuccess ||
        hipMemcpy(st->d_image[key][qcat], img.data(), img.size() * 2, hipMemcpyHostToDevice) != hipSuccess)
      return av1mi::ctx_fail(ctx, AV1MI_E_DEVICE, "uploading the default CDF image failed");
  }
  av1mi::Av1EntLaunch L;
  memset(&L, 0, sizeof(L));
  L.fv.w8 = j->width / 8; L.fv.h8 = j->height / 8; L.fv.key = key;
  L.fv.y_mode = j->d_modes_y; L.fv.uv_mode = j->d_modes_uv; L.fv.mv = j->d_mvs; L.fv.skip = j->d_skip;
  L.fv.lev_y = j->d_lev_y; L.fv.lev_u = j->d_lev_u; L.fv.lev_v = j->d_lev_v;
  for (int p = 0; p < 3; p++) {
    L.fv.lr_on[p] = j->lr_on[p] != 0;
    const int ph = p ? j->height / 2 : j->height, pw = p ? j->width / 2 : j->width;
    L.fv.lr_rows[p] = (ph + 32) / 64 > 1 ? (ph + 32) / 64 : 1; L.fv.lr_cols[p] = (pw + 32) / 64 > 1 ? (pw + 32) / 64 : 1;
  }
  memcpy(L.fv.lr_unit[0], j->lr_unit_y, 8); memcpy(L.fv.lr_unit[1], j->lr_unit_uv, 8);
  L.nframes = j->nframes; L.sbr_n = (L.fv.h8 + 7) / 8; L.sbc_n = (L.fv.w8 + 7) / 8;
  const size_t nb = (size_t)L.fv.w8 * L.fv.h8 * j->nframes, nt = (size_t)L.sbr_n * L.sbc_n * j->nframes;
  L.ops_cap = av1mi_av1_entropy_ops_per_tile(); L.slot_cap = av1mi_av1_entropy_slot_bytes();
  int rc;
  if ((rc = grow(ctx, &st->info, &st->info_b, nb * sizeof(av1ops::BlockInfo))) || (rc = grow(ctx, &st->ops, &st->ops_b, nt * L.ops_cap * sizeof(av1ops::op_t))) ||
      (rc = grow(ctx, &st->nops, &st->nops_b, nt * 4)) || (rc = grow(ctx, &st->slots, &st->slots_b, nt * L.slot_cap)) ||
      (rc = grow(ctx, &st->tile_off, &st->off_b, (nt + 1) * 8)))
    return rc;
  L.info = (av1ops::BlockInfo *)st->info; L.ops = (av1ops::op_t *)st->ops; L.nops = (uint32_t *)st->nops; L.slots = (uint8_t *)st->slots;
  L.tile_off = (uint64_t *)st->tile_off;
  L.tile_size = j->d_tile_size; L.out = j->d_out; L.out_cap = j->out_cap;
  L.status = (uint32_t *)(j->d_total + 1);
  L.cdf_image = st->d_image[key][qcat]; L.cdf_words = st->image_words[key]; L.tab = st->tab[key];
  if (hipMemsetAsync(j->d_total, 0, 16, s) != hipSuccess) return av1mi::ctx_fail(ctx, AV1MI_E_DEVICE, "hipMemsetAsync failed");
  const hipError_t e = av1mi::launch_av1_entropy(L, s);
  if (e != hipSuccess) return av1mi::ctx_fail(ctx, AV1MI_E_DEVICE, "AV1 entropy launch: %s", hipGetErrorString(e));
  // total bytes next to the status: tile_off[nt]
  if (hipMemcpyAsync(j->d_total, L.tile_off + nt, 8, hipMemcpyDeviceToDevice, s) != hipSuccess) return av1mi::ctx_fail(ctx, AV1MI_E_DEVICE, "copy of the total failed");
  return AV1MI_OK;
}

int av1mi_av1_entropy_encode(av1mi_ctx *ctx, const av1mi_av1_entropy_job *j) { return av1mi_av1_entropy_encode_on(ctx, j, nullptr); }

}  // extern "C"

namespace av1mi {
void av1ent_free(av1mi_av1ent_state *st) {
  if (!st) return;
  for (void *p : { st->info, st->ops, st->nops, st->slots, st->tile_off }) if (p) (void)hipFree(p);
  for (auto &k : st->d_image) for (uint16_t *p : k) if (p) (void)hipFree(p);
  delete st;
}
av1mi_av1ent_state *av1ent_new() { return new (std::nothrow) av1mi_av1ent_state(); }
}  // namespace av1mi
