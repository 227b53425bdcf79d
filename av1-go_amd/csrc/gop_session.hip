// gop_session.hip — the GOP session of include/av1mi.h: closed-GOP orchestration, the encoder's filter-parameter policy and
// the PCIe plumbing around the block pipeline, in ONE place (round 1 had two hand-kept copies, pipeline.py and
// host/backend.cpp, that disagreed on the P-frame deblocking level).  What stands in for the encode the reference delegates
// to its FFmpeg child (internal/ffmpeg/transcode.go:120,194); the caller it serves is internal/daemon/daemon.go:101.
//
// Streams.  The context's stream runs the kernels; the session adds an upload stream and a download stream.  Per batch:
//   up:   wait slot.filters_done (the kernels that last read this slot's source)  -> H2D of the three source planes -> uploaded
//   main: wait uploaded -> k_intra_pipe | k_me_int + k_inter_pipe -> symbols_ready -> deblock x3, CDEF, LR x3 -> reference
//   down: wait symbols_ready -> D2H of the symbols into pinned memory -> downloaded
//   side: (GPU entropy coding) wait symbols_ready -> k_av1_* -> payloads gathered into the slot's pinned buffer -> ent_done
// Source and symbol buffers exist kSlots = 3 times (slot = batch % 3): batch t + 2 uploads while batch t + 1 is in the block
// pipeline and the coder works on batch t, whose predecessor the host is still reading.  Four streams, one hardware queue each
// (a fifth would share a queue with one of these and serialise behind it).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>
#include "av1mi_internal.hpp"

namespace {

// ---- the policy (non-normative encoder choices; mirrored nowhere else) ---------------------------------------------------
int lf_level_from_q(int ac_q, int bd, bool key) {   // libaom LPF_PICK_FROM_Q: a linear fit of the level to the AC step
  long g;
  if (bd == 8) g = key ? ((long)ac_q * 17563 - 421574 + (1 << 17)) >> 18 : ((long)ac_q * 6017 + 650707 + (1 << 17)) >> 18;
  else g = ((long)ac_q * 20723 + 4060632 + (1 << 19)) >> 20;
  return (int)(g < 0 ? 0 : g > 63 ? 63 : g);
}
void frame_params(int q, int bd, int frame_type, av1mi_frame_params *p) {
  memset(p, 0, sizeof(*p));
  const int ac_q = av1mi_ac_q(q, bd), q8 = ac_q >> (bd - 8);
  p->frame_type = frame_type; p->base_q_idx = q;
  const int lvl = lf_level_from_q(ac_q, bd, frame_type == 0);
  p->lf_level[0] = p->lf_level[1] = p->lf_level[2] = p->lf_level[3] = lvl;
  p->lf_sharpness = 0;
  // CDEF strengths from the quantiser step: libaom's CDEF_PICK_FROM_Q fit (pickcdef.c av1_pick_cdef_from_qp: quadratic fits of the
  // strengths its full search picks, separate for intra-only and inter frames; secondary codes 0..3 = strengths 0, 1, 2, 4).  At
  // mid quantisers inter frames get secondary strength 0: two thirds of k_cdef's taps drop out (cdef_kernel.hip).
  p->cdef_damping = 3 + (q >> 6);
  {
    const float x = (float)q8, x2 = x * x;
    auto fit = [&](float a, float b, float c, int hi) { const int v = (int)lroundf(x2 * a + x * b + c); return v < 0 ? 0 : v > hi ? hi : v; };
    int y1, y2, c1, c2;
    if (frame_type == 0) {
      y1 = fit(0.0000033731974f, 0.008070594f, 0.0187634f, 15); y2 = fit(0.0000029167343f, 0.0027798624f, 0.0079405f, 3);
      c1 = fit(-0.0000130790995f, 0.012892405f, -0.00748388f, 15); c2 = fit(0.0000032651783f, 0.00035520183f, 0.00228092f, 3);
    } else {
      y1 = fit(-0.0000023593946f, 0.0068615186f, 0.02709886f, 15); y2 = fit(-0.00000057629734f, 0.0013993345f, 0.03831067f, 3);
      c1 = fit(-0.0000007095069f, 0.0034628846f, 0.00887099f, 15); c2 = fit(0.00000023874085f, 0.00028223585f, 0.05576307f, 3);
    }
    p->cdef_y = (uint8_t)(y1 << 2 | y2); p->cdef_uv = (uint8_t)(c1 << 2 | c2);
  }
  p->lr_unit_size = 64;
  static const int8_t wy[8] = { 1, 3, -7, 15, 3, -7, 15, 0 }, wc[8] = { 1, 0, -7, 15, 0, -7, 15, 0 };   // Wiener, libaom's mid-range taps
  memcpy(p->lr_unit_y, wy, 8); memcpy(p->lr_unit_uv, wc, 8);
}

enum { kSlots = 3, kFallbacksToHostMode = 3 };      // batches in flight: one uploading / in the block pipeline, one in the coder, one being read by the host

struct Slot {
  void *h_src[3] = { nullptr, nullptr, nullptr };       // pinned
  void *d_src[3] = { nullptr, nullptr, nullptr };
  // symbols: device + pinned host mirror
  void *d_lev[3] = { nullptr, nullptr, nullptr }, *h_lev[3] = { nullptr, nullptr, nullptr };
  void *d_modes[2] = { nullptr, nullptr }, *h_modes[2] = { nullptr, nullptr };
  void *d_mv = nullptr, *h_mv = nullptr, *d_skip = nullptr, *h_skip = nullptr;
  hipEvent_t uploaded = nullptr, kernel_done = nullptr, filters_done = nullptr, downloaded = nullptr;
  bool upload_pending = false, kernel_pending = false;
  // restoration on / off per (segment, plane), decided by the GPU against the source (k_lr + k_lr_decide): device + pinned mirror
  void *d_lr_on = nullptr, *h_lr_on = nullptr;
  int frame_type = 0;
  // GPU entropy coding (gpu_entropy != 0): coded tile payloads (pinned host memory, written by the GPU) + sizes (device + pinned mirror)
  void *h_ent_out = nullptr, *d_tile_size = nullptr, *h_tile_size = nullptr, *d_total = nullptr, *h_total = nullptr;
  hipEvent_t ent_done = nullptr;      // coder + download of sizes / total finished (side stream)
  bool ent_pending = false;
  bool symbols_down = false;          // this batch's symbols were sent to the host at submit time
  int ent_ticket = -1;                // >= 0: the range coder of this batch is still to be launched (coder_streams 3: av1_entropy_back)
};

}  // namespace

struct av1mi_gop {
  av1mi_ctx *ctx = nullptr;
  av1mi_gop_config cfg{};
  size_t ny = 0, nc = 0, nb = 0, bps = 1;      // per BATCH (segments stacked): luma samples, chroma samples, blocks
  hipStream_t up = nullptr, down = nullptr;     // with the context's main and side streams: four, one hardware queue each
  Slot slot[kSlots];
  void *d_rec[3] = {}, *d_dbl[3] = {}, *d_cdef[3] = {}, *d_ref[3] = {};
  void *d_mi[3][2] = {};                       // [key / inter / key in 32x32 blocks][luma / chroma] deblocking mode-info maps (one frame, shared by the batch)
  int key32 = 0, key_rows32 = 0;               // key frames in 32x32 blocks over the first key_rows32 luma rows (the complete superblock rows)
  int key_modes_band = 0, key_modes_stride = 0;  // mode bytes per segment: the 32x32 blocks, then (from key_modes_band) the 8x8 blocks of the last rows
  void *d_cdef_sb[2] = {}, *d_lr[2] = {}, *d_zero_skip = nullptr;
  void *d_lr_scratch = nullptr;                // the restoration decision's partial sums (three planes)
  int vw = 0, vh = 0;                          // the true frame size (== the coded size unless cfg.visible_* say otherwise)
  int last = 0;                                // slot of the most recent batch (its d_lr_on selects the next batch's references)
  av1mi_frame_params params[2];                // key, inter
  size_t ent_cap = 0; int tiles = 0;           // GPU entropy coding: payload capacity of a batch, tiles per frame
  long submitted = 0, collected = 0;           // batches
  long fallbacks = 0;                          // batches the GPU coder could not hold (handed out as symbols instead)
  bool symbols_always = false;                 // after kFallbacksToHostMode of them: the symbols go down with every batch, beside the filters
  int gop_pos = 0;
  bool acquired = false;
  int coder_streams = 0;                       // 0 = tokenizer + chains on the side stream, range coder on the back stream (default)
  int intra_open_loop = 0;                     // key frames: open-loop mode decision (k_intra_modes) instead of the closed-loop search
  std::vector<void *> dev_allocs, host_allocs;
};

namespace {

#define G_HIP(expr)                                                                                          \
  do {                                                                                                       \
    hipError_t e_ = (expr);                                                                                  \
    if (e_ != hipSuccess) return av1mi::ctx_fail(g->ctx, AV1MI_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define G_TRY(expr)                              \
  do {                                           \
    int rc_ = (expr);                            \
    if (rc_ != AV1MI_OK) return rc_;             \
  } while (0)

int dev_alloc(av1mi_gop *g, void **p, size_t bytes) {
  G_HIP(hipMalloc(p, bytes ? bytes : 8));
  g->dev_allocs.push_back(*p);
  return AV1MI_OK;
}
int host_alloc(av1mi_gop *g, void **p, size_t bytes) {
  G_HIP(hipHostMalloc(p, bytes ? bytes : 8, hipHostMallocDefault));
  g->host_allocs.push_back(*p);
  return AV1MI_OK;
}

int setup(av1mi_gop *g) {
  const av1mi_gop_config &c = g->cfg;
  const int w = c.width, h = c.height, S = c.segments;
  g->bps = c.bit_depth == 8 ? 1 : 2;
  g->ny = (size_t)w * h * S; g->nc = g->ny / 4; g->nb = g->ny / 64;
  g->vw = c.visible_width ? c.visible_width : w; g->vh = c.visible_height ? c.visible_height : h;
  G_HIP(hipSetDevice(av1mi::ctx_device(g->ctx)));
  G_HIP(hipStreamCreateWithFlags(&g->up, hipStreamNonBlocking));
  // (created in every mode, used only where symbols go to the host at submit time or a batch falls back.  HIP deals its four
  // default hardware queues to streams in creation order: main, up, down, side — and the coder's back stream, the fifth, shares
  // the main stream's queue, i.e. the range coder of batch t and the block pipeline of batch t + 1 run one after the other.
  // Measured A/B on one box: that is the FASTEST arrangement, 1 800 frames/s end to end at 4K against 1 500-1 600 with a queue
  // per stream (without this stream, or with GPU_MAX_HW_QUEUES=8) and 1 500 with the coder on the main stream itself: the
  // range coder is one long wave per CU, and beside it every kernel of the block pipeline runs at a fraction of its speed.)
  G_HIP(hipStreamCreateWithFlags(&g->down, hipStreamNonBlocking));
  for (Slot &s : g->slot) {
    for (int p = 0; p < 3; p++) {
      const size_t n = (p ? g->nc : g->ny);
      G_TRY(host_alloc(g, &s.h_src[p], n * g->bps)); G_TRY(dev_alloc(g, &s.d_src[p], n * g->bps));
      // the pinned mirror of the levels (as large as the source) is needed when the symbols go to the host; with the GPU coder
      // only a batch the coder gives back needs it, and it is allocated then (pinning memory is a good part of the start-up time)
      if (c.gpu_entropy != 1) G_TRY(host_alloc(g, &s.h_lev[p], n * 2));
      G_TRY(dev_alloc(g, &s.d_lev[p], n * 2));
    }
    for (int k = 0; k < 2; k++) { G_TRY(host_alloc(g, &s.h_modes[k], g->nb)); G_TRY(dev_alloc(g, &s.d_modes[k], g->nb)); }
    G_TRY(host_alloc(g, &s.h_mv, g->nb * 4)); G_TRY(dev_alloc(g, &s.d_mv, g->nb * 4));
    G_TRY(host_alloc(g, &s.h_skip, g->nb)); G_TRY(dev_alloc(g, &s.d_skip, g->nb));
    G_HIP(hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming));
    G_HIP(hipEventCreateWithFlags(&s.kernel_done, hipEventDisableTiming));
    G_HIP(hipEventCreateWithFlags(&s.filters_done, hipEventDisableTiming));
    G_TRY(dev_alloc(g, &s.d_lr_on, (size_t)S * 3 + 4)); G_TRY(host_alloc(g, &s.h_lr_on, (size_t)S * 3));      // (+ 4: the kernels read the flags as aligned dwords)
    G_HIP(hipEventCreateWithFlags(&s.downloaded, hipEventDisableTiming));
    if (c.gpu_entropy) {
      g->tiles = ((w + 63) / 64) * ((h + 63) / 64);
      g->ent_cap = (size_t)w * h * S;            // one byte per luma sample: several times what a frame codes to at any sane quantiser
      G_TRY(host_alloc(g, &s.h_ent_out, g->ent_cap));        // pinned and device-visible: the gather kernel writes it over PCIe
      G_TRY(dev_alloc(g, &s.d_tile_size, (size_t)g->tiles * S * 4)); G_TRY(host_alloc(g, &s.h_tile_size, (size_t)g->tiles * S * 4));
      G_TRY(dev_alloc(g, &s.d_total, 16)); G_TRY(host_alloc(g, &s.h_total, 16));
      G_HIP(hipEventCreateWithFlags(&s.ent_done, hipEventDisableTiming));
    }
  }
  for (int p = 0; p < 3; p++) {
    const size_t n = (p ? g->nc : g->ny) * g->bps;
    G_TRY(dev_alloc(g, &g->d_rec[p], n)); G_TRY(dev_alloc(g, &g->d_dbl[p], n)); G_TRY(dev_alloc(g, &g->d_cdef[p], n)); G_TRY(dev_alloc(g, &g->d_ref[p], n));
  }
  G_TRY(dev_alloc(g, &g->d_lr_scratch, av1mi_lr_yuv_decide_scratch_bytes(h, S)));
  G_TRY(dev_alloc(g, &g->d_zero_skip, g->nb));
  G_TRY(av1mi_memset(g->ctx, g->d_zero_skip, 0, g->nb));
  // constant side information: one map per frame type, shared by every frame of a batch (frame stride 0)
  const size_t fy = (size_t)w * h, fc = fy / 4;
  const int nsb = ((w + 63) / 64) * ((h + 63) / 64);
  auto units = [](int n) { const int u = (n + 32) / 64; return u > 1 ? u : 1; };
  const size_t uy = (size_t)units(h) * units(w), uc = (size_t)units(h / 2) * units(w / 2);
  if (c.key_block_size == 32) {
    g->key32 = 1; g->key_rows32 = (h / 64) * 64;
    // mode bytes of a frame: the 32x32 blocks from entry 0, the 8x8 blocks of the last rows where the 8x8 grid has them anyway
    g->key_modes_band = (g->key_rows32 / 8) * (w / 8);
    g->key_modes_stride = (h / 8) * (w / 8);
  }
  for (int t = 0; t < 2 + g->key32; t++) {
    if (t < 2) frame_params(c.base_q_idx, c.bit_depth, t, &g->params[t]);
    const av1mi_frame_params &P = g->params[t == 2 ? 0 : t];
    // deblocking mode-info words (av1mi_deblock_plane): 8x8 luma / 4x4 chroma transforms, every block edge a prediction edge
    std::vector<uint32_t> mi(fy / 16, 3u | (3u << 4) | ((uint32_t)P.lf_level[0] << 8) | ((uint32_t)P.lf_level[1] << 16) | (3u << 25));
    std::vector<uint32_t> mic(fc / 16, 2u | (2u << 4) | ((uint32_t)P.lf_level[2] << 8) | ((uint32_t)P.lf_level[2] << 16) | (3u << 25));
    // units that start at or beyond the true size (spec 7.14.2 onScreen; a chroma unit is two luma units wide) are never filtered:
    // "skipped inter block, no block edge"
    for (int r = 0; r < h / 4; r++) for (int cc = 0; cc < w / 4; cc++) if (4 * r >= g->vh || 4 * cc >= g->vw) mi[(size_t)r * (w / 4) + cc] = 3u | (3u << 4) | (1u << 24);
    for (int r = 0; r < h / 8; r++) for (int cc = 0; cc < w / 8; cc++) if (8 * r >= g->vh || 8 * cc >= g->vw) mic[(size_t)r * (w / 8) + cc] = 2u | (2u << 4) | (1u << 24);
    if (t == 2) {      // key frames in 32x32 blocks: 32x32 luma / 16x16 chroma transforms over the complete superblock rows (always on screen)
      for (int r = 0; r < g->key_rows32 / 4; r++) for (int cc = 0; cc < w / 4; cc++) mi[(size_t)r * (w / 4) + cc] = (mi[(size_t)r * (w / 4) + cc] & ~0xFFu) | 5u | (5u << 4);
      for (int r = 0; r < g->key_rows32 / 8; r++) for (int cc = 0; cc < w / 8; cc++) mic[(size_t)r * (w / 8) + cc] = (mic[(size_t)r * (w / 8) + cc] & ~0xFFu) | 4u | (4u << 4);
    }
    G_TRY(dev_alloc(g, &g->d_mi[t][0], mi.size() * 4)); G_TRY(av1mi_upload(g->ctx, g->d_mi[t][0], mi.data(), mi.size() * 4));
    G_TRY(dev_alloc(g, &g->d_mi[t][1], mic.size() * 4)); G_TRY(av1mi_upload(g->ctx, g->d_mi[t][1], mic.data(), mic.size() * 4));
  }
  for (int t = 0; t < 2; t++) {      // CDEF strengths: one set per frame type
    const av1mi_frame_params &P = g->params[t];
    std::vector<uint8_t> sb((size_t)nsb * 4);
    for (int i = 0; i < nsb; i++) { sb[4 * i] = P.cdef_y >> 2; sb[4 * i + 1] = P.cdef_y & 3; sb[4 * i + 2] = P.cdef_uv >> 2; sb[4 * i + 3] = P.cdef_uv & 3; }
    G_TRY(dev_alloc(g, &g->d_cdef_sb[t], sb.size())); G_TRY(av1mi_upload(g->ctx, g->d_cdef_sb[t], sb.data(), sb.size()));
  }
  {
    const av1mi_frame_params &P = g->params[0];     // the restoration units do not depend on the frame type
    std::vector<int8_t> lr((uy > uc ? uy : uc) * 8);
    for (size_t i = 0; i < uy; i++) memcpy(&lr[i * 8], P.lr_unit_y, 8);
    G_TRY(dev_alloc(g, &g->d_lr[0], uy * 8)); G_TRY(av1mi_upload(g->ctx, g->d_lr[0], lr.data(), uy * 8));
    for (size_t i = 0; i < uc; i++) memcpy(&lr[i * 8], P.lr_unit_uv, 8);
    G_TRY(dev_alloc(g, &g->d_lr[1], uc * 8)); G_TRY(av1mi_upload(g->ctx, g->d_lr[1], lr.data(), uc * 8));
  }
  G_TRY(av1mi_sync(g->ctx));
  return AV1MI_OK;
}

}  // namespace

extern "C" {

int av1mi_policy_frame_params(int base_q_idx, int bit_depth, int frame_type, av1mi_frame_params *out) {
  if (!out || base_q_idx < 0 || base_q_idx > 255 || (bit_depth != 8 && bit_depth != 10) || frame_type < 0 || frame_type > 1) return AV1MI_E_INVAL;
  frame_params(base_q_idx, bit_depth, frame_type, out);
  return AV1MI_OK;
}

int av1mi_gop_open(av1mi_ctx *ctx, const av1mi_gop_config *cfg, av1mi_gop **out) {
  if (!ctx || !out) return AV1MI_E_INVAL;
  *out = nullptr;
  if (!cfg) return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "null config");
  if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width & 7) || (cfg->height & 7) || cfg->width > 16384 || cfg->height > 16384)
    return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "frame %dx%d must be a multiple of 8", cfg->width, cfg->height);
  if (cfg->visible_width < 0 || cfg->visible_height < 0 || (cfg->visible_width && (cfg->visible_width > cfg->width || cfg->width - cfg->visible_width >= 8)) ||
      (cfg->visible_height && (cfg->visible_height > cfg->height || cfg->height - cfg->visible_height >= 8)))
    return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "visible size %dx%d must lie within 7 samples below the coded size %dx%d", cfg->visible_width, cfg->visible_height,
                           cfg->width, cfg->height);
  if (cfg->bit_depth != 8 && cfg->bit_depth != 10) return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "bit depth %d not supported (8 or 10)", cfg->bit_depth);
  if (cfg->base_q_idx < 1 || cfg->base_q_idx > 255 || cfg->gop_length < 1 || cfg->segments < 1 || cfg->segments > 4096 || cfg->search_range < 0 ||
      cfg->search_range > 15 || cfg->gpu_entropy < 0 || cfg->gpu_entropy > 2 || cfg->coder_streams < 0 || cfg->coder_streams > 3)
    return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "bad base_q_idx / gop_length / segments / search_range / gpu_entropy");
  if (cfg->key_block_size != 0 && cfg->key_block_size != 8 && cfg->key_block_size != 32) return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "key_block_size %d not supported (8 or 32)", cfg->key_block_size);
  if (cfg->key_block_size == 32 && (cfg->width & 31))
    return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "key_block_size 32 needs a width that is a multiple of 32");
  if (cfg->gpu_entropy && (cfg->width > 4096 || cfg->height > 4096)) return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "the AV1 tile coder takes frames up to 4096x4096");
  if ((size_t)cfg->height * cfg->segments > 65535u * 8u) return av1mi::ctx_fail(ctx, AV1MI_E_INVAL, "segments x height too large for one launch");
  av1mi_gop *g = new (std::nothrow) av1mi_gop();
  if (!g) return AV1MI_E_NOMEM;
  g->ctx = ctx; g->cfg = *cfg;
  g->coder_streams = cfg->coder_streams;
  if (const char *e = getenv("AV1MI_INTRA_OPEN_LOOP")) g->intra_open_loop = atoi(e) ? 1 : 0;
  if (const char *e = getenv("AV1MI_CODER_STREAMS")) g->coder_streams = !strcmp(e, "side") ? 1 : !strcmp(e, "main") ? 2 : !strcmp(e, "defer") ? 3 : 0;
  const int rc = setup(g);
  if (rc != AV1MI_OK) { av1mi_gop_close(g); return rc; }
  *out = g;
  return AV1MI_OK;
}

void av1mi_gop_close(av1mi_gop *g) {
  if (!g) return;
  (void)hipSetDevice(av1mi::ctx_device(g->ctx));
  (void)av1mi_sync(g->ctx);
  if (g->up) { (void)hipStreamSynchronize(g->up); (void)hipStreamDestroy(g->up); }
  if (g->down) { (void)hipStreamSynchronize(g->down); (void)hipStreamDestroy(g->down); }
  for (Slot &s : g->slot) {
    if (s.uploaded) (void)hipEventDestroy(s.uploaded);
    if (s.kernel_done) (void)hipEventDestroy(s.kernel_done);
    if (s.filters_done) (void)hipEventDestroy(s.filters_done);
    if (s.downloaded) (void)hipEventDestroy(s.downloaded);
    if (s.ent_done) (void)hipEventDestroy(s.ent_done);
  }
  for (void *p : g->dev_allocs) (void)hipFree(p);
  for (void *p : g->host_allocs) (void)hipHostFree(p);
  delete g;
}

int av1mi_gop_max_in_flight(void) { return kSlots; }

}  // extern "C"

// modes (key frames) or vectors + skip flags (inter frames) of a slot -> its pinned host buffers
static int download_modes(av1mi_gop *g, Slot &s, hipStream_t st) {
  if (s.frame_type == 0) {
    for (int k = 0; k < 2; k++) G_HIP(hipMemcpyAsync(s.h_modes[k], s.d_modes[k], g->nb, hipMemcpyDeviceToHost, st));
  } else {
    G_HIP(hipMemcpyAsync(s.h_mv, s.d_mv, g->nb * 4, hipMemcpyDeviceToHost, st));
    G_HIP(hipMemcpyAsync(s.h_skip, s.d_skip, g->nb, hipMemcpyDeviceToHost, st));
  }
  return AV1MI_OK;
}

extern "C" {

int av1mi_gop_pending(av1mi_gop *g) { return g ? (int)(g->submitted - g->collected) : 0; }
long av1mi_gop_entropy_fallbacks(av1mi_gop *g) { return g ? g->fallbacks : 0; }

int av1mi_gop_acquire_input(av1mi_gop *g, void **y, void **u, void **v) {
  if (!g || !y || !u || !v) return AV1MI_E_INVAL;
  if (g->submitted - g->collected >= kSlots) return av1mi::ctx_fail(g->ctx, AV1MI_E_INVAL, "%d batches in flight: collect before acquiring the next input", (int)kSlots);
  G_HIP(hipSetDevice(av1mi::ctx_device(g->ctx)));
  Slot &s = g->slot[g->submitted % kSlots];
  if (s.upload_pending) { G_HIP(hipEventSynchronize(s.uploaded)); s.upload_pending = false; }   // the copy engine still reads these buffers
  *y = s.h_src[0]; *u = s.h_src[1]; *v = s.h_src[2];
  g->acquired = true;
  return AV1MI_OK;
}

// the coder's results of a slot -> its pinned mirrors, then ent_done (on the stream the range coder ran on)
static int entropy_results(av1mi_gop *g, Slot &s, hipStream_t st) {
  const int S = g->cfg.segments;
  G_HIP(hipMemcpyAsync(s.h_tile_size, s.d_tile_size, (size_t)g->tiles * S * 4, hipMemcpyDeviceToHost, st));
  G_HIP(hipMemcpyAsync(s.h_total, s.d_total, 16, hipMemcpyDeviceToHost, st));
  G_HIP(hipMemcpyAsync(s.h_lr_on, s.d_lr_on, (size_t)S * 3, hipMemcpyDeviceToHost, st));
  G_HIP(hipEventRecord(s.ent_done, st));
  return AV1MI_OK;
}
// the deferred back half of a slot's entropy job (coder_streams 3)
static int finish_entropy(av1mi_gop *g, Slot &s, hipStream_t st) {
  G_TRY(av1mi::av1_entropy_back(g->ctx, s.ent_ticket, st));
  s.ent_ticket = -1;
  return entropy_results(g, s, st);
}

// one batch through the block pipeline, the filters and (gpu_entropy) the tile coder; dev_src: the source planes in device memory
// (av1mi_gop_submit_device), or null = upload the slot's pinned planes first
static int submit_batch(av1mi_gop *g, int frame_type, const void *const *dev_src) {
  if (g->submitted - g->collected >= kSlots) return av1mi::ctx_fail(g->ctx, AV1MI_E_INVAL, "%d batches in flight: collect first", (int)kSlots);
  if (frame_type < -1 || frame_type > 1) return av1mi::ctx_fail(g->ctx, AV1MI_E_INVAL, "frame_type %d", frame_type);
  if (frame_type < 0) frame_type = g->gop_pos == 0 ? 0 : 1;
  if (frame_type == 1 && g->submitted == 0) return av1mi::ctx_fail(g->ctx, AV1MI_E_INVAL, "the first frame of a session must be a key frame");
  G_HIP(hipSetDevice(av1mi::ctx_device(g->ctx)));
  const av1mi_gop_config &c = g->cfg;
  const int w = c.width, h = c.height, S = c.segments, bd = c.bit_depth;
  Slot &s = g->slot[g->submitted % kSlots];
  hipStream_t main = av1mi::ctx_stream(g->ctx);
  const void *src[3] = { s.d_src[0], s.d_src[1], s.d_src[2] };
  if (dev_src) {
    for (int p = 0; p < 3; p++) src[p] = dev_src[p];
  } else {
    // upload: not before the kernels that last read this slot's source have finished (the restoration decision is the last reader)
    if (s.kernel_pending) G_HIP(hipStreamWaitEvent(g->up, s.filters_done, 0));
    for (int p = 0; p < 3; p++) G_HIP(hipMemcpyAsync(s.d_src[p], s.h_src[p], (p ? g->nc : g->ny) * g->bps, hipMemcpyHostToDevice, g->up));
    G_HIP(hipEventRecord(s.uploaded, g->up));
    s.upload_pending = true;
    G_HIP(hipStreamWaitEvent(main, s.uploaded, 0));
  }
  if (s.ent_pending) G_HIP(hipStreamWaitEvent(main, s.ent_done, 0));      // the GPU coder of the slot's previous batch still reads its symbols
  // the block pipeline (the symbols of this slot were downloaded before the slot was collected, so they may be overwritten)
  if (frame_type == 0) {
    av1mi_intra_job j;
    memset(&j, 0, sizeof(j));
    j.width = w; j.height = h; j.bit_depth = bd; j.nframes = S; j.qindex = c.base_q_idx; j.block_size = 8; j.stride_y = w; j.stride_uv = w / 2;
    j.d_src_y = src[0]; j.d_src_u = src[1]; j.d_src_v = src[2];
    j.d_rec_y = g->d_rec[0]; j.d_rec_u = g->d_rec[1]; j.d_rec_v = g->d_rec[2];
    j.d_lev_y = (int16_t *)s.d_lev[0]; j.d_lev_u = (int16_t *)s.d_lev[1]; j.d_lev_v = (int16_t *)s.d_lev[2];
    j.d_modes_y = (uint8_t *)s.d_modes[0]; j.d_modes_uv = (uint8_t *)s.d_modes[1];
    j.open_loop = g->intra_open_loop;
    if (!g->key32) {
      G_TRY(av1mi_intra_encode(g->ctx, &j));
    } else {
      // two bands of every frame: the complete superblock rows in 32x32 blocks, a last partial row (if any) in 8x8 blocks.  Tiles are
      // single superblocks, so the bands share nothing.
      const int hA = g->key_rows32, hB = h - hA;
      j.open_loop = 0; j.frame_rows = h; j.modes_frame_stride = g->key_modes_stride;
      if (hA) { j.height = hA; j.block_size = 32; G_TRY(av1mi_intra_encode(g->ctx, &j)); }
      if (hB) {
        const size_t oy = (size_t)hA * w, oc = (size_t)(hA / 2) * (w / 2);
        j.height = hB; j.block_size = 8;
        j.d_src_y = (const char *)src[0] + oy * g->bps; j.d_src_u = (const char *)src[1] + oc * g->bps; j.d_src_v = (const char *)src[2] + oc * g->bps;
        j.d_rec_y = (char *)g->d_rec[0] + oy * g->bps; j.d_rec_u = (char *)g->d_rec[1] + oc * g->bps; j.d_rec_v = (char *)g->d_rec[2] + oc * g->bps;
        j.d_lev_y = (int16_t *)s.d_lev[0] + oy; j.d_lev_u = (int16_t *)s.d_lev[1] + oc; j.d_lev_v = (int16_t *)s.d_lev[2] + oc;
        j.d_modes_y = (uint8_t *)s.d_modes[0] + g->key_modes_band; j.d_modes_uv = (uint8_t *)s.d_modes[1] + g->key_modes_band;
        G_TRY(av1mi_intra_encode(g->ctx, &j));
      }
    }
  } else {
    av1mi_inter_job j;
    memset(&j, 0, sizeof(j));
    j.width = w; j.height = h; j.bit_depth = bd; j.nframes = S; j.qindex = c.base_q_idx; j.search_range = c.search_range; j.stride_y = w; j.stride_uv = w / 2;
    j.d_src_y = src[0]; j.d_src_u = src[1]; j.d_src_v = src[2];
    j.d_ref_y = g->d_ref[0]; j.d_ref_u = g->d_ref[1]; j.d_ref_v = g->d_ref[2];
    j.d_rec_y = g->d_rec[0]; j.d_rec_u = g->d_rec[1]; j.d_rec_v = g->d_rec[2];
    j.d_lev_y = (int16_t *)s.d_lev[0]; j.d_lev_u = (int16_t *)s.d_lev[1]; j.d_lev_v = (int16_t *)s.d_lev[2];
    j.d_mvs = (int16_t *)s.d_mv; j.d_skip = (uint8_t *)s.d_skip;
    // per segment and plane: the restored plane of the previous frame, or its CDEF output where restoration was switched off
    j.d_ref_alt_y = g->d_cdef[0]; j.d_ref_alt_u = g->d_cdef[1]; j.d_ref_alt_v = g->d_cdef[2];
    j.d_ref_sel = (const uint8_t *)g->slot[g->last].d_lr_on;
    G_TRY(av1mi_inter_encode(g->ctx, &j));
  }
  G_HIP(hipEventRecord(s.kernel_done, main));
  s.kernel_pending = true;
  s.frame_type = frame_type;
  // symbols -> pinned host memory, beside the filters.  Not when the GPU codes the tiles (gpu_entropy == 1): the host then needs
  // the payloads only (and a fifth busy stream would share a hardware queue with one of the other four)
  s.symbols_down = c.gpu_entropy != 1 || g->symbols_always;
  if (s.symbols_down) {
    G_HIP(hipStreamWaitEvent(g->down, s.kernel_done, 0));
    for (int p = 0; p < 3; p++) G_HIP(hipMemcpyAsync(s.h_lev[p], s.d_lev[p], (p ? g->nc : g->ny) * 2, hipMemcpyDeviceToHost, g->down));
    G_TRY(download_modes(g, s, g->down));
  }
  // in-loop filters: reconstruction -> what the next frame predicts from
  const av1mi_frame_params &P = g->params[frame_type];
  for (int p = 0; p < 3; p++) {
    const int pw = p ? w / 2 : w, ph = p ? h / 2 : h;
    G_TRY(av1mi_deblock_frames(g->ctx, g->d_rec[p], pw, g->d_dbl[p], pw, pw, ph, bd, p > 0, (const uint32_t *)g->d_mi[frame_type == 0 && g->key32 ? 2 : frame_type][p > 0], pw / 4, 0,
                               P.lf_sharpness, S));
  }
  av1mi_cdef_job cj;
  memset(&cj, 0, sizeof(cj));
  cj.width = w; cj.height = h; cj.bit_depth = bd; cj.nframes = S; cj.damping = P.cdef_damping; cj.stride_y = w; cj.stride_uv = w / 2;
  cj.d_src_y = g->d_dbl[0]; cj.d_src_u = g->d_dbl[1]; cj.d_src_v = g->d_dbl[2];
  cj.d_dst_y = g->d_cdef[0]; cj.d_dst_u = g->d_cdef[1]; cj.d_dst_v = g->d_cdef[2];
  cj.d_sb_strength = (const uint8_t *)g->d_cdef_sb[frame_type]; cj.sb_frame_stride = 0;
  // key frames are coded with skip = 0 everywhere (no block is exempt from CDEF); P frames: the kernel's skip flags, per frame
  // (the slot's next inter kernel is kSlots batches away and ordered behind this CDEF on the main stream, nothing else writes them)
  cj.d_skip8 = (const uint8_t *)(frame_type == 0 ? g->d_zero_skip : s.d_skip); cj.skip_frame_stride = frame_type == 0 ? 0 : (size_t)(w / 8) * (h / 8);
  G_TRY(av1mi_cdef_frames(g->ctx, &cj));
  // a true size that is not a multiple of 8: the decoder's restoration clamps at the true last column / row (CDEF above read the
  // planes as they were: it works on the coded size in a decoder too)
  const bool padded = g->vw != w || g->vh != h;
  if (padded)
    for (int p = 0; p < 3; p++) {
      const int pw = p ? w / 2 : w, ph = p ? h / 2 : h, pvw = p ? (g->vw + 1) / 2 : g->vw, pvh = p ? (g->vh + 1) / 2 : g->vh;
      G_TRY(av1mi_extend_frames(g->ctx, g->d_dbl[p], pw, pw, ph, pvw, pvh, bd, S));
      G_TRY(av1mi_extend_frames(g->ctx, g->d_cdef[p], pw, pw, ph, pvw, pvh, bd, S));
    }
  // loop restoration of every frame, and the decision per segment and plane whether it stays ON (it must lower the squared error
  // against the source): d_ref always receives the restored planes, the next batch's kernels choose between d_ref and d_cdef
  {
    av1mi_lr_decide_job lj;
    memset(&lj, 0, sizeof(lj));
    lj.width = w; lj.height = h; lj.bit_depth = bd; lj.nframes = S; lj.unit_size = P.lr_unit_size; lj.stride_y = w; lj.stride_uv = w / 2;
    lj.d_cdef_y = g->d_cdef[0]; lj.d_cdef_u = g->d_cdef[1]; lj.d_cdef_v = g->d_cdef[2];
    lj.d_dbl_y = g->d_dbl[0]; lj.d_dbl_u = g->d_dbl[1]; lj.d_dbl_v = g->d_dbl[2];
    lj.d_out_y = g->d_ref[0]; lj.d_out_u = g->d_ref[1]; lj.d_out_v = g->d_ref[2];
    lj.d_orig_y = src[0]; lj.d_orig_u = src[1]; lj.d_orig_v = src[2];
    lj.d_units_y = (const int8_t *)g->d_lr[0]; lj.d_units_uv = (const int8_t *)g->d_lr[1];
    lj.d_scratch = g->d_lr_scratch; lj.d_on = (uint8_t *)s.d_lr_on;
    lj.no_self_guided_units = P.lr_unit_y[0] != 2 && P.lr_unit_uv[0] != 2;
    G_TRY(av1mi_lr_yuv_decide(g->ctx, &lj));
  }
  if (padded)      // ... and so do its motion-compensation reads of this frame
    for (int p = 0; p < 3; p++)
      G_TRY(av1mi_extend_frames(g->ctx, g->d_ref[p], p ? w / 2 : w, p ? w / 2 : w, p ? h / 2 : h, p ? (g->vw + 1) / 2 : g->vw, p ? (g->vh + 1) / 2 : g->vh, bd, S));
  G_HIP(hipEventRecord(s.filters_done, main));
  if (s.symbols_down) {
    G_HIP(hipStreamWaitEvent(g->down, s.filters_done, 0));
    G_HIP(hipMemcpyAsync(s.h_lr_on, s.d_lr_on, (size_t)S * 3, hipMemcpyDeviceToHost, g->down));
    G_HIP(hipEventRecord(s.downloaded, g->down));
  }
  if (c.gpu_entropy) {
    // the AV1 tile entropy coder beside the next batch's block pipeline: tokenizer + chains on the context's side stream (after
    // the filters: the restoration units a tile codes depend on the decision), the serial range coder on its back stream (so the
    // next batch's tokenizer does not wait for it)
    hipStream_t side = av1mi::ctx_side_stream(g->ctx), back = av1mi::ctx_back_stream(g->ctx);
    if (!side || !back) return av1mi::ctx_fail(g->ctx, AV1MI_E_DEVICE, "no side stream");
    if (g->coder_streams == 1) back = side;                 // diagnostic arrangements (AV1MI_CODER_STREAMS): the whole coder on the side stream
    else if (g->coder_streams == 2) side = back = main;     // ... or on the main stream, serialised behind the filters
    if (side != main) G_HIP(hipStreamWaitEvent(side, s.filters_done, 0));
    av1mi_av1_entropy_job ej;
    memset(&ej, 0, sizeof(ej));
    ej.width = w; ej.height = h; ej.nframes = S; ej.key = frame_type == 0; ej.base_q_idx = c.base_q_idx;
    ej.d_lev_y = (const int16_t *)s.d_lev[0]; ej.d_lev_u = (const int16_t *)s.d_lev[1]; ej.d_lev_v = (const int16_t *)s.d_lev[2];
    ej.d_modes_y = (const uint8_t *)s.d_modes[0]; ej.d_modes_uv = (const uint8_t *)s.d_modes[1];
    ej.d_mvs = (const int16_t *)s.d_mv; ej.d_skip = (const uint8_t *)s.d_skip;
    ej.lr_on[0] = P.lr_unit_y[0] == 1; ej.lr_on[1] = ej.lr_on[2] = P.lr_unit_uv[0] == 1;
    ej.d_lr_on = (const uint8_t *)s.d_lr_on;
    ej.visible_width = g->vw; ej.visible_height = g->vh;
    ej.key_rows32 = frame_type == 0 && g->key32 ? g->key_rows32 : 0;
    memcpy(ej.lr_unit_y, P.lr_unit_y, 8); memcpy(ej.lr_unit_uv, P.lr_unit_uv, 8);
    ej.d_out = (uint8_t *)s.h_ent_out; ej.out_cap = g->ent_cap; ej.d_tile_size = (uint32_t *)s.d_tile_size; ej.d_total = (uint64_t *)s.d_total;
    if (g->coder_streams == 3) {
      // tokenizer + chains of THIS batch on the side stream; the range coder of the PREVIOUS batch on the main stream, behind this
      // batch's filters: two queues that are both busy all the time ((pipeline + filters + coder) beside (tokenizer + chains)) instead of
      // a short one and a long one
      G_TRY(av1mi::av1_entropy_front(g->ctx, &ej, side, &s.ent_ticket));
      Slot &prev = g->slot[(g->submitted + kSlots - 1) % kSlots];
      if (g->submitted > 0 && prev.ent_ticket >= 0) G_TRY(finish_entropy(g, prev, main));
    } else {
      G_TRY(av1mi::av1_entropy_submit(g->ctx, &ej, side, back));
      G_TRY(entropy_results(g, s, back));
    }
    s.ent_pending = true;
  }
  g->last = (int)(g->submitted % kSlots);
  g->submitted++;
  g->gop_pos = frame_type == 0 ? 1 % c.gop_length : (g->gop_pos + 1) % c.gop_length;
  g->acquired = false;
  return AV1MI_OK;
}

int av1mi_gop_submit(av1mi_gop *g, int frame_type) {
  if (!g) return AV1MI_E_INVAL;
  if (!g->acquired) return av1mi::ctx_fail(g->ctx, AV1MI_E_INVAL, "submit without av1mi_gop_acquire_input");
  return submit_batch(g, frame_type, nullptr);
}

int av1mi_gop_submit_device(av1mi_gop *g, const void *d_y, const void *d_u, const void *d_v, int frame_type) {
  if (!g) return AV1MI_E_INVAL;
  if (!d_y || !d_u || !d_v || (((uintptr_t)d_y | (uintptr_t)d_u | (uintptr_t)d_v) & 7)) return av1mi::ctx_fail(g->ctx, AV1MI_E_INVAL, "null or misaligned device source plane");
  const void *src[3] = { d_y, d_u, d_v };
  return submit_batch(g, frame_type, src);
}

int av1mi_gop_collect(av1mi_gop *g, av1mi_gop_frame *out) {
  if (!g || !out) return AV1MI_E_INVAL;
  if (g->submitted == g->collected) return av1mi::ctx_fail(g->ctx, AV1MI_E_INVAL, "nothing in flight");
  G_HIP(hipSetDevice(av1mi::ctx_device(g->ctx)));
  Slot &s = g->slot[g->collected % kSlots];
  if (s.symbols_down) G_HIP(hipEventSynchronize(s.downloaded));
  memset(out, 0, sizeof(*out));
  out->params = g->params[s.frame_type];
  out->lr_on = (const uint8_t *)s.h_lr_on;
  out->segments = g->cfg.segments;
  out->blocks_per_frame = g->nb / (size_t)g->cfg.segments;
  out->key_block_size = s.frame_type == 0 && g->key32 ? 32 : 8;
  out->key_modes_stride = g->key_modes_stride; out->key_modes_band = g->key_modes_band;
  auto symbols = [&](bool levels) {
    if (s.frame_type == 0) { out->y_mode = (const uint8_t *)s.h_modes[0]; out->uv_mode = (const uint8_t *)s.h_modes[1]; }
    else { out->mv = (const int16_t *)s.h_mv; out->skip = (const uint8_t *)s.h_skip; }
    if (levels) { out->lev_y = (const int16_t *)s.h_lev[0]; out->lev_u = (const int16_t *)s.h_lev[1]; out->lev_v = (const int16_t *)s.h_lev[2]; }
  };
  if (s.symbols_down) symbols(true);
  if (g->cfg.gpu_entropy) {
    if (s.ent_ticket >= 0) G_TRY(finish_entropy(g, s, av1mi::ctx_stream(g->ctx)));      // no later batch came to carry it (coder_streams 3)
    G_HIP(hipEventSynchronize(s.ent_done));
    const uint64_t total = ((const uint64_t *)s.h_total)[0], status = ((const uint64_t *)s.h_total)[1];
    // the payloads are already here: k_av1_gather wrote them into the slot's pinned buffer (a copy enqueued NOW would queue
    // behind the next batches' work, which is already submitted)
    if (status || total > g->ent_cap) {
      if (getenv("AV1MI_DEBUG"))
        fprintf(stderr, "[av1mi] batch %ld (frame type %d) falls back to the host coder: status %llx (1 list / records, 2 payload slot, 4 output), %llu bytes of %zu\n",
                (long)g->collected, s.frame_type, (unsigned long long)status, (unsigned long long)total, g->ent_cap);
      // A tile exceeded the coder's list / record / payload capacity (very fine quantisers on dense content).  The batch is not
      // lost: its symbols are still in the slot's device buffers (the next kernel that overwrites them is kSlots submits away),
      // so they are downloaded now and handed out like in host mode: tile_size stays NULL, the caller entropy-codes this batch.
      if (!s.symbols_down) {
        // the pinned mirrors of the levels exist only once a batch has needed them: all slots' at the first fallback (pinning memory
        // stalls the device), not one slot at a time in the middle of later batches
        for (Slot &o : g->slot)
          for (int p = 0; p < 3; p++)
            if (!o.h_lev[p]) G_TRY(host_alloc(g, &o.h_lev[p], (p ? g->nc : g->ny) * 2));
        for (int p = 0; p < 3; p++) G_HIP(hipMemcpyAsync(s.h_lev[p], s.d_lev[p], (p ? g->nc : g->ny) * 2, hipMemcpyDeviceToHost, g->down));
        G_TRY(download_modes(g, s, g->down));
        G_HIP(hipStreamSynchronize(g->down));
        symbols(true);
      }
      // content the GPU coder cannot hold tends to stay that way: from the third such batch on the symbols are sent down with every
      // batch, beside the filters (as in host mode), instead of after the coder has given the batch back
      if (g->fallbacks + 1 >= kFallbacksToHostMode) g->symbols_always = true;
      g->fallbacks++;
    } else {
      out->tiles_per_frame = g->tiles; out->tile_size = (const uint32_t *)s.h_tile_size; out->tile_payload = (const uint8_t *)s.h_ent_out; out->payload_bytes = total;
    }
  }
  g->collected++;
  return AV1MI_OK;
}

int av1mi_gop_download_reference(av1mi_gop *g, void *y, void *u, void *v) {
  if (!g || !y || !u || !v) return AV1MI_E_INVAL;
  G_TRY(av1mi_sync(g->ctx));
  // per segment and plane the restored plane or, where restoration was switched off, the CDEF output: what the decoder outputs
  const int S = g->cfg.segments;
  std::vector<uint8_t> on((size_t)S * 3, 1);
  if (g->submitted) G_TRY(av1mi_download(g->ctx, on.data(), g->slot[g->last].d_lr_on, on.size()));
  void *dst[3] = { y, u, v };
  for (int p = 0; p < 3; p++) {
    const size_t per = (p ? g->nc : g->ny) * g->bps / (size_t)S;
    for (int sg = 0; sg < S; sg++)
      G_TRY(av1mi_download(g->ctx, (char *)dst[p] + per * sg, (const char *)(on[(size_t)sg * 3 + p] ? g->d_ref[p] : g->d_cdef[p]) + per * sg, per));
  }
  return AV1MI_OK;
}

}  // extern "C"
