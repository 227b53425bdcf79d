/*
 * av1mi.h — C ABI of libav1mi.so, the MI355X-native AV1 block-processing backend.
 *
 * Drop-in boundary.  The reference (IONIQ6000/av1-go) has no FFI for this path: its seam is
 * the process contract of internal/ffmpeg/transcode.go:194
 *     func RunTranscode(ffmpegPath string, args []string) (int, error)
 * fed by TranscodeArgs (transcode.go:17) and called only from daemon.ProcessJob
 * (internal/daemon/daemon.go:90,101).  All pixel work happens in the FFmpeg child
 * (transcode.go:120 "-c:v:0 av1_vaapi").  This header is what a cgo replacement of
 * RunTranscode binds instead; every entry point names the part of that
 * contract, or the SURVEY.md §8a kernel row, it stands in for.
 *
 * Conventions: plain C, no C++ types, no exceptions cross the boundary.  Return 0 = OK,
 * negative = AV1MI_E_*; av1mi_last_error(ctx) gives the text the Go wrapper turns into the
 * (int, error) pair of transcode.go:194/311.  One context per GPU; a context is not
 * thread-safe, distinct contexts are.  Every call selects the context's device first, so
 * any OS thread (goroutines migrate) may call.  No callbacks.
 * Pointers named d_* are DEVICE pointers obtained from av1mi_malloc(); others are host.
 */
#ifndef AV1MI_H
#define AV1MI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AV1MI_OK 0
#define AV1MI_E_INVAL (-1)   /* bad argument (size/type/alignment) */
#define AV1MI_E_DEVICE (-2)  /* HIP runtime error; text in av1mi_last_error */
#define AV1MI_E_NOMEM (-3)
#define AV1MI_E_NODEV (-4)   /* no gfx950 device / HIP extension unusable: the product never falls back to CPU */

/* AV1 TX_SIZE / TX_TYPE numbering (AV1 spec §6.10.19). */
enum av1mi_tx_size {
  AV1MI_TX_4X4, AV1MI_TX_8X8, AV1MI_TX_16X16, AV1MI_TX_32X32, AV1MI_TX_64X64, AV1MI_TX_4X8, AV1MI_TX_8X4,
  AV1MI_TX_8X16, AV1MI_TX_16X8, AV1MI_TX_16X32, AV1MI_TX_32X16, AV1MI_TX_32X64, AV1MI_TX_64X32, AV1MI_TX_4X16,
  AV1MI_TX_16X4, AV1MI_TX_8X32, AV1MI_TX_32X8, AV1MI_TX_16X64, AV1MI_TX_64X16, AV1MI_TX_SIZES_ALL
};
enum av1mi_tx_type {
  AV1MI_DCT_DCT, AV1MI_ADST_DCT, AV1MI_DCT_ADST, AV1MI_ADST_ADST, AV1MI_FLIPADST_DCT, AV1MI_DCT_FLIPADST,
  AV1MI_FLIPADST_FLIPADST, AV1MI_ADST_FLIPADST, AV1MI_FLIPADST_ADST, AV1MI_IDTX, AV1MI_V_DCT, AV1MI_H_DCT,
  AV1MI_V_ADST, AV1MI_H_ADST, AV1MI_V_FLIPADST, AV1MI_H_FLIPADST, AV1MI_TX_TYPES,
  AV1MI_WHT_WHT = 16   /* lossless blocks: 4x4 Walsh-Hadamard (spec 7.13.2.10), valid with AV1MI_TX_4X4 only */
};

typedef struct av1mi_ctx av1mi_ctx;

/* One transform block of a block list (16 bytes, one dwordx4 load on the device). */
typedef struct av1mi_txb {
  uint32_t coef_off; /* offset of the block's coefficients in the coefficient buffer, int32 units, multiple of 4 */
  uint16_t x, y;     /* top-left sample of the block in the plane; x multiple of 4 */
  uint32_t tx_type;  /* enum av1mi_tx_type */
  uint32_t reserved;
} av1mi_txb;

/* ---- library / context: stands in for ffmpeg provisioning + process spawn (binary.go:218, transcode.go:195) */
const char *av1mi_version(void);
int av1mi_device_count(void);                       /* number of HIP devices, 0 if none */
int av1mi_open(int device, av1mi_ctx **out);        /* AV1MI_E_NODEV when no GPU: callers must fail the job */
void av1mi_close(av1mi_ctx *ctx);
const char *av1mi_last_error(av1mi_ctx *ctx);       /* <= 800 chars, the cap of transcode.go:295-297 */
const char *av1mi_device_name(av1mi_ctx *ctx);

/* ---- device memory and stream plumbing (all on the context's own HIP stream) */
int av1mi_malloc(av1mi_ctx *ctx, void **d_ptr, size_t bytes);
int av1mi_free(av1mi_ctx *ctx, void *d_ptr);
int av1mi_upload(av1mi_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int av1mi_download(av1mi_ctx *ctx, void *dst, const void *d_src, size_t bytes);
int av1mi_memset(av1mi_ctx *ctx, void *d_dst, int value, size_t bytes);
/* device-to-device copy on the context's stream (e.g. handing a reconstructed frame to the next GOP stage; bench.py uses it to
 * measure the box's copy bandwidth, the second roofline BASELINE.md §3 asks for). */
int av1mi_copy(av1mi_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
int av1mi_sync(av1mi_ctx *ctx);
/* HIP-event stopwatch on the context's stream: begin .. end brackets whatever was enqueued between. */
int av1mi_timer_begin(av1mi_ctx *ctx);
int av1mi_timer_end(av1mi_ctx *ctx, float *elapsed_ms);

/* Per-kernel HIP-event profile (bench.py's roofline leg).  While enabled, every kernel launch made through
 * this context is bracketed by its own event pair on the context's stream; av1mi_prof_get() synchronises and
 * returns, for one kernel kind, the number of launches and the summed device time since the last reset. */
enum av1mi_kernel_kind {
  AV1MI_K_FWD_TXFM, AV1MI_K_INV_TXFM, AV1MI_K_QUANT, AV1MI_K_DEQUANT, AV1MI_K_INTRA_PRED, AV1MI_K_MC,
  AV1MI_K_DEBLOCK, AV1MI_K_CDEF, AV1MI_K_LR, AV1MI_K_INTRA_PIPE, AV1MI_K_INTER_PIPE, AV1MI_K_MISC, AV1MI_K_ENTROPY,
  AV1MI_K_ENTROPY_PACK, AV1MI_K_ENTROPY_TOKENS, AV1MI_K_ME_INT, AV1MI_K_ENTROPY_CHAINS, AV1MI_K_KINDS
};
int av1mi_prof_enable(av1mi_ctx *ctx, int on);
int av1mi_prof_reset(av1mi_ctx *ctx);
int av1mi_prof_get(av1mi_ctx *ctx, int kind, int *launches, double *total_ms);
const char *av1mi_kernel_kind_name(int kind);

/* 1 when (tx_size, tx_type) is arithmetically defined: ADST needs length 4/8/16, identity <= 32. */
int av1mi_txfm_valid(int tx_size, int tx_type);
int av1mi_tx_width(int tx_size);
int av1mi_tx_height(int tx_size);

/* ---- K2: inverse 2-D transform + add to prediction + clip.  Asynchronous.
 * d_plane holds the prediction on entry and the reconstruction afterwards; uint8 samples when
 * bd == 8, uint16 when bd == 10; stride in samples, multiple of 4.
 * Coefficients: int32, per block row-major min(w,32) x min(h,32) (64-point dimensions carry only
 * the low 32 frequencies), blocks contiguous.
 * grid form: block i covers (i % blocks_per_row, i / blocks_per_row) in units of the block size, its
 * coefficients start at i * min(w,32)*min(h,32); d_tx_types (1 byte per block) may be NULL, then
 * every block uses uniform_type. */
int av1mi_inv_txfm_add_grid(av1mi_ctx *ctx, int tx_size, const int32_t *d_coef, void *d_plane, int stride, int bd,
                            int blocks_per_row, int nblocks, const uint8_t *d_tx_types, int uniform_type);
/* list form: explicit blocks, all of size tx_size. */
int av1mi_inv_txfm_add_list(av1mi_ctx *ctx, int tx_size, const int32_t *d_coef, void *d_plane, int stride, int bd,
                            const av1mi_txb *d_list, int nblocks);

/* ---- K1: forward 2-D transform.  d_resid: int16 residual plane, stride in samples (multiple of 4). */
int av1mi_fwd_txfm_grid(av1mi_ctx *ctx, int tx_size, const int16_t *d_resid, int stride, int32_t *d_coef,
                        int blocks_per_row, int nblocks, const uint8_t *d_tx_types, int uniform_type);
int av1mi_fwd_txfm_list(av1mi_ctx *ctx, int tx_size, const int16_t *d_resid, int stride, int32_t *d_coef,
                        const av1mi_txb *d_list, int nblocks);

/* ---- K8: quantise (encoder side) / dequantise (normative).  n coefficients, multiple of 4; the first
 * coefficient of every run of coef_per_blk uses dc_q, the others ac_q.  d_dqcoef may be NULL in quantize. */
int av1mi_dc_q(int qindex, int bd);
int av1mi_ac_q(int qindex, int bd);
int av1mi_quantize(av1mi_ctx *ctx, const int32_t *d_coef, int16_t *d_levels, int32_t *d_dqcoef, size_t n,
                   int coef_per_blk, int dc_q, int ac_q, int log_scale);
int av1mi_dequantize(av1mi_ctx *ctx, const int16_t *d_levels, int32_t *d_dqcoef, size_t n, int coef_per_blk,
                     int dc_q, int ac_q, int log_scale, int bd);

/* ---- K3: intra prediction of a list of equally-sized transform blocks (AV1 spec §7.11.2).
 * d_ref is the reconstructed plane the neighbours are read from, d_dst the plane the prediction is written to
 * (may be the same allocation when no listed block is a neighbour of another).  Per block the caller passes what
 * libaom's build_intra_predictors() takes: mode (0 DC, 1 V, 2 H, 3 D45, 4 D135, 5 D113, 6 D157, 7 D203, 8 D67,
 * 9 SMOOTH, 10 SMOOTH_V, 11 SMOOTH_H, 12 PAETH), angle_delta -3..3 (directional modes), and the numbers of
 * AVAILABLE neighbour samples (top <= w, top-right <= w, left <= h, bottom-left <= h). */
typedef struct av1mi_intra_blk {
  uint16_t x, y;       /* top-left sample; x multiple of 4 */
  uint8_t mode;
  int8_t angle_delta;
  uint8_t flags;       /* bit 0: disable intra edge filter; bit 1: filter type (a neighbour is smooth-predicted) */
  uint8_t n_top, n_topright, n_left, n_bottomleft;
  uint8_t reserved[5];
} av1mi_intra_blk;
int av1mi_intra_pred_list(av1mi_ctx *ctx, int tx_size, const void *d_ref, int ref_stride, void *d_dst, int dst_stride,
                          int bd, const av1mi_intra_blk *d_list, int nblocks);

/* ---- K3, chroma-from-luma (AV1 spec §7.11.5; SURVEY.md §8f rank 4 "remaining intra modes"), 4:2:0.  For every listed
 * chroma block (tx_size up to 32x32) d_dst already holds the DC prediction; the block becomes
 *   Clip1(dc + Round2Signed(alpha_q3 * (L - avg L), 6)),  L = (2x2 sum of the reconstructed luma) << 1,
 * with luma coordinates limited to max_luma_w - 2 / max_luma_h - 2 (the spec's MaxLumaW / MaxLumaH).  alpha_q3 in -16..16. */
typedef struct av1mi_cfl_blk {
  uint16_t x, y;                 /* chroma block position in the chroma plane; x multiple of 4 */
  uint16_t max_luma_w, max_luma_h;
  int8_t alpha_q3;
  uint8_t reserved[7];
} av1mi_cfl_blk;
int av1mi_cfl_pred_list(av1mi_ctx *ctx, int tx_size, const void *d_luma, int luma_stride, void *d_dst, int dst_stride, int bd,
                        const av1mi_cfl_blk *d_list, int nblocks);

/* ---- K4: sub-pel motion compensation (AV1 spec §7.11.3.4; single reference, unscaled, no compound) of a list
 * of blocks of one size.  size_id uses the TX_SIZE numbering for w x h (0 4x4 .. 4 64x64, 5 4x8 ...).  Each block
 * at (x, y) of the plane (x multiple of 4) is predicted from d_ref displaced by (mvx, mvy) in 1/16-sample units of
 * THIS plane; reference coordinates are clamped to [0, plane_w-1] x [0, plane_h-1].  filt_x / filt_y: 0 regular,
 * 1 smooth, 2 sharp, 3 bilinear (dimensions <= 4 switch to the 4-tap variants as the spec does). */
typedef struct av1mi_mc_blk {
  uint16_t x, y;
  int16_t mvx, mvy;
  uint8_t filt_x, filt_y;
  uint8_t reserved[6];
} av1mi_mc_blk;
int av1mi_mc_list(av1mi_ctx *ctx, int size_id, const void *d_ref, int ref_stride, int plane_w, int plane_h, void *d_dst,
                  int dst_stride, int bd, const av1mi_mc_blk *d_list, int nblocks);

/* ---- K5: deblocking loop filter of one plane (AV1 spec §7.14), both passes in one launch, d_src -> d_dst
 * (different allocations; w, h multiples of 4; strides in samples, multiples of 4).
 * d_mi: (h/4) x (w/4) mode-info units of the PLANE (already subsampled for chroma), one uint32 each:
 *   bits 0-3  log2(transform width), bits 4-7 log2(transform height) of the transform block covering the unit
 *   bits 8-15 filter level used for vertical edges (pass 0), bits 16-23 for horizontal edges (pass 1), 0..63
 *   bit 24    skip && is_inter (inner transform edges are not filtered)
 *   bit 25    the unit's left edge is a prediction-block edge, bit 26 its top edge is one */
int av1mi_deblock_plane(av1mi_ctx *ctx, const void *d_src, int src_stride, void *d_dst, int dst_stride, int w, int h,
                        int bd, int is_chroma, const uint32_t *d_mi, int mi_stride, int sharpness);

/* the same over nframes frames stacked vertically in d_src / d_dst (h rows each); the mode info of frame f starts
 * at d_mi + f * mi_frame_stride (units), mi_frame_stride 0 = one shared map. */
int av1mi_deblock_frames(av1mi_ctx *ctx, const void *d_src, int src_stride, void *d_dst, int dst_stride, int w, int h,
                         int bd, int is_chroma, const uint32_t *d_mi, int mi_stride, size_t mi_frame_stride, int sharpness,
                         int nframes);

/* ---- K6: CDEF (AV1 spec §7.15) of nframes 4:2:0 frames stacked vertically; deblocked planes in, separate planes out.
 * width/height: luma size, multiples of 8.  d_sb_strength: 4 bytes per 64x64 luma block in raster order
 * {y_pri 0..15, y_sec 0..3, uv_pri, uv_sec}; y_pri = 255 switches CDEF off for that block.  d_skip8: one byte per
 * 8x8 luma block, 1 = all of its mode-info units are skipped (block left untouched).  damping 3..6.
 * sb_frame_stride (entries) / skip_frame_stride (bytes) separate the per-frame maps; 0 = one map shared by all frames. */
typedef struct av1mi_cdef_job {
  int width, height, bit_depth, nframes, damping;
  int stride_y, stride_uv;
  const void *d_src_y, *d_src_u, *d_src_v;
  void *d_dst_y, *d_dst_u, *d_dst_v;
  const uint8_t *d_sb_strength; size_t sb_frame_stride;
  const uint8_t *d_skip8; size_t skip_frame_stride;
} av1mi_cdef_job;
int av1mi_cdef_frames(av1mi_ctx *ctx, const av1mi_cdef_job *job);

/* ---- K7: loop restoration (AV1 spec §7.17) of one plane of nframes frames stacked vertically.  d_cdef: the CDEF
 * output, d_deblocked: the deblocked (pre-CDEF) plane used beyond stripe boundaries, d_out: the restored plane
 * (distinct from both).  subsampled = 1 for the chroma planes of 4:2:0 (32-row stripes offset by 4), 0 for luma.
 * unit_size: 32 (chroma only), 64, 128 or 256.  d_units: rows x cols entries of 8 bytes, rows =
 * max(1, (h + unit/2) / unit), cols likewise: {type: 0 none / 1 Wiener / 2 self-guided,
 *   Wiener: v0 v1 v2 h0 h1 h2 (int8 taps; the centre tap is 128 - 2*(sum)), pad |
 *   self-guided: set 0..15, xqd0, xqd1 (int8), pad}.  unit_frame_stride: units between frames, 0 = shared. */
int av1mi_lr_frames(av1mi_ctx *ctx, const void *d_cdef, const void *d_deblocked, void *d_out, int stride, int w, int h,
                    int bd, int subsampled, int unit_size, const int8_t *d_units, size_t unit_frame_stride, int nframes);

/* The same restoration followed by the encoder's ON / OFF decision per frame of the plane (policy, non-normative; oracle:
 * av1o_lr_keep): d_on[f * on_stride] = 1 when the restored samples are closer to the source d_orig (same geometry) than the CDEF
 * samples — sum of squared differences, strictly smaller — else 0: the plane the next frame predicts from is then d_cdef's, and
 * the frame header signals lr_type NONE for it.  d_out always receives the restored samples.  d_scratch: device memory of
 * av1mi_lr_decide_scratch_bytes(h, subsampled, nframes) bytes, 8-byte aligned. */
/* The same decision for the three planes of 4:2:0 frames in one call (what the GOP session uses), in two passes: the tiles the
 * decision's sums run over are restored first (an eighth of a large plane), the decision follows, and the other tiles are restored
 * only in the frames that keep their restoration.  So d_out_* hold the restored plane where d_on[f * 3 + plane] == 1; where it is
 * 0 — the next frame predicts from d_cdef_* there — only the sampled tiles were written.  d_scratch: 16-byte aligned,
 * av1mi_lr_yuv_decide_scratch_bytes(height, nframes) bytes.  unit_size applies to the samples of each plane (luma and chroma). */
typedef struct av1mi_lr_decide_job {
  int width, height, bit_depth, nframes, unit_size;
  int stride_y, stride_uv;
  const void *d_cdef_y, *d_cdef_u, *d_cdef_v;
  const void *d_dbl_y, *d_dbl_u, *d_dbl_v;       /* the deblocked planes (rows beside the stripes) */
  void *d_out_y, *d_out_u, *d_out_v;
  const void *d_orig_y, *d_orig_u, *d_orig_v;    /* the source */
  const int8_t *d_units_y, *d_units_uv; size_t unit_frame_stride_y, unit_frame_stride_uv;
  void *d_scratch; uint8_t *d_on;
  int no_self_guided_units;                      /* != 0: the caller promises that no unit record has type 2 (the session's policy:
                                                    Wiener or none) — the kernel then runs without the self-guided path's 18 KB of LDS, at
                                                    twice the occupancy; a type-2 unit would be left unfiltered */
} av1mi_lr_decide_job;
size_t av1mi_lr_yuv_decide_scratch_bytes(int height, int nframes);
int av1mi_lr_yuv_decide(av1mi_ctx *ctx, const av1mi_lr_decide_job *job);

/* In place: the column visible_w - 1 of every row replicated into columns [visible_w, w), then row visible_h - 1 into rows
 * [visible_h, h), for nframes stacked frames of one plane (see av1mi_gop_config.visible_width for when an encoder loop needs it). */
int av1mi_extend_frames(av1mi_ctx *ctx, void *d_plane, int stride, int w, int h, int visible_w, int visible_h, int bd, int nframes);
size_t av1mi_lr_decide_scratch_bytes(int h, int subsampled, int nframes);
int av1mi_lr_frames_decide(av1mi_ctx *ctx, const void *d_cdef, const void *d_deblocked, void *d_out, int stride, int w, int h, int bd, int subsampled,
                           int unit_size, const int8_t *d_units, size_t unit_frame_stride, int nframes, const void *d_orig, void *d_scratch, uint8_t *d_on,
                           int on_stride);

/* ---- the intra-only segment pipeline (BASELINE config 2): what stands in for the encode the reference delegates
 * to `ffmpeg -c:v:0 av1_vaapi` (transcode.go:120) for key frames.  One launch codes `nframes` frames that are
 * stacked in the plane buffers (frame f starts at row f*height of the luma planes, f*height/2 of the chroma planes).
 * Every 64x64 superblock is an independent tile; blocks are block_size x block_size (8 or 16; width and height must
 * be multiples of it), transform = block size, DCT_DCT, mode chosen per block by SAD among all 13 intra modes (DC, V, H, 6
 * diagonals, SMOOTH, PAETH, SMOOTH_V, SMOOTH_H; angle delta 0).  Outputs: reconstruction planes, int16 levels (block-contiguous, raster order of blocks, per plane),
 * one mode byte per block for luma and one for the chroma pair. */
typedef struct av1mi_intra_job {
  int width, height, bit_depth, nframes, qindex, block_size;   /* block_size 8 (the session), 16 or 32 (32: closed loop only) */
  int stride_y, stride_uv;                 /* samples; multiples of 4 */
  const void *d_src_y, *d_src_u, *d_src_v; /* source frames */
  void *d_rec_y, *d_rec_u, *d_rec_v;       /* reconstruction (output) */
  int16_t *d_lev_y, *d_lev_u, *d_lev_v;    /* quantised levels (output): nframes * width*height (/4 for chroma) */
  uint8_t *d_modes_y, *d_modes_uv;         /* nframes * (width/bs)*(height/bs) each */
  int open_loop;                           /* mode decision: 0 = closed loop (13 candidates predicted from the reconstruction, inside the
                                              tile's serial chain), 1 = open loop (decided for all blocks at once from the SOURCE frame's
                                              neighbours, then one prediction per block in the chain).  The GOP session codes its key frames closed
                                              loop (AV1MI_INTRA_OPEN_LOOP=1 in the environment switches a session over, for measurements) */
  int frame_rows;                          /* 0 = height.  Otherwise the job codes a BAND of `height` rows of every frame: the frames of a plane
                                              are frame_rows luma rows apart (levels likewise), the pointers address the band's first row */
  int modes_frame_stride;                  /* 0 = the job's blocks per frame; otherwise the mode bytes of consecutive frames are this far apart */
} av1mi_intra_job;
int av1mi_intra_encode(av1mi_ctx *ctx, const av1mi_intra_job *job);

/* ---- the inter (P-frame) pipeline (BASELINE config 3): every 8x8 block predicted from ONE reference frame
 * (the previous reconstructed, loop-filtered frame): integer full search +-search_range (0..15) by SAD, half- and
 * quarter-pel refinement with the regular 8-tap filter, luma + chroma motion compensation (spec §7.11.3.4), then the
 * same DCT_DCT residual coding as the intra pipeline.  nframes independent frames are stacked like in av1mi_intra_job
 * (typically the t-th frames of many closed-GOP segments).  Outputs additionally: one vector per block (int16 x, y in
 * 1/8 luma samples) and one skip byte per block (1 = no non-zero level in Y, U and V). */
typedef struct av1mi_inter_job {
  int width, height, bit_depth, nframes, qindex, search_range;
  int stride_y, stride_uv;
  const void *d_src_y, *d_src_u, *d_src_v;
  const void *d_ref_y, *d_ref_u, *d_ref_v;
  void *d_rec_y, *d_rec_u, *d_rec_v;
  int16_t *d_lev_y, *d_lev_u, *d_lev_v;
  int16_t *d_mvs;       /* nframes * (w/8)*(h/8) * 2 */
  uint8_t *d_skip;      /* nframes * (w/8)*(h/8) */
  /* optional (NULL = not used): the reference per frame and plane without a copy.  d_ref_sel[f * 3 + p] == 0 makes frame f predict
   * plane p from d_ref_alt_* (the CDEF output of the previous frame: its restoration was switched off, av1mi_lr_frames_decide)
   * instead of d_ref_* (the restored planes).  d_ref_sel: 4-byte aligned, allocated up to a multiple of 4 bytes (read as dwords). */
  const void *d_ref_alt_y, *d_ref_alt_u, *d_ref_alt_v;
  const uint8_t *d_ref_sel;
} av1mi_inter_job;
int av1mi_inter_encode(av1mi_ctx *ctx, const av1mi_inter_job *job);

/* ---- K9: the AV1 tile entropy coder on the GPU (av1-go_amd/csrc/av1_entropy_kernels.hip).  Codes the outputs
 * of av1mi_intra_encode (key = 1) or av1mi_inter_encode (key = 0) of `nframes` stacked frames in AV1's tile syntax — the tool
 * set of the block pipeline: 8x8 blocks, one 64x64 superblock per tile, TX_MODE_LARGEST, DCT_DCT luma, cdef_bits = 0, Wiener
 * restoration on 64x64 units — byte-identical to the host writer (include/av1mi_host.h), which dav1d verifies.  Output: the tile
 * payloads of all frames back to back in tile order (frame-major, raster inside a frame) in d_out, their sizes in d_tile_size
 * (nframes * tiles entries; tiles = ceil(w / 64) * ceil(h / 64)), and d_total[0] = total bytes, d_total[1] = status (0 = OK; bit 0 /
 * 1: a tile exceeded the coder's op / payload capacity, bit 2: out_cap too small — then the batch must be coded on the host).
 * The frame header and the tile-size fields are added by the host (av1mi_obu_assemble_temporal_unit): they depend on the
 * largest tile.  lr_on / lr_unit_*: the restoration parameters the frames were filtered with (av1mi_frame_params). */
typedef struct av1mi_av1_entropy_job {
  int width, height, nframes, key, base_q_idx;
  const int16_t *d_lev_y, *d_lev_u, *d_lev_v;
  const uint8_t *d_modes_y, *d_modes_uv;       /* key = 1 */
  const int16_t *d_mvs; const uint8_t *d_skip; /* key = 0 */
  int lr_on[3];
  int8_t lr_unit_y[8], lr_unit_uv[8];
  uint8_t *d_out; size_t out_cap;
  uint32_t *d_tile_size;
  uint64_t *d_total;                           /* 2 entries, 8-byte aligned */
  const uint8_t *d_lr_on;                      /* optional: [frame * 3 + plane] 0 switches lr_on[plane] off for that frame */
  int visible_width, visible_height;           /* the true frame size when width / height are it rounded up to 8 (the restoration units
                                                  a tile codes tile the TRUE frame); 0 = width / height */
  int key_rows32;                              /* key = 1: the first key_rows32 luma rows (whole superblock rows; width % 32 == 0) are coded in
                                                  32x32 blocks: their modes one per 32x32 block from entry 0 of d_modes_*, their levels
                                                  block-contiguous over the 32x32 grid; the rows below in 8x8 blocks at their usual places */
} av1mi_av1_entropy_job;
int av1mi_av1_entropy_encode(av1mi_ctx *ctx, const av1mi_av1_entropy_job *job);
/* the same on another HIP stream of the caller's (hipStream_t passed as void *; NULL = the context's stream) */
int av1mi_av1_entropy_encode_on(av1mi_ctx *ctx, const av1mi_av1_entropy_job *job, void *stream);
uint32_t av1mi_av1_entropy_ops_per_tile(void);
/* measurement: 32-bit list words (one per syntax element) the most recent job handed from the tokenizer to the range coder, summed
 * over its tiles; synchronises the device */
int av1mi_av1_entropy_last_list_words(av1mi_ctx *ctx, uint64_t *words);
uint32_t av1mi_av1_entropy_slot_bytes(void);

/* ---- GOP session: the encoder object a cgo replacement of RunTranscode drives (reference call site
 * internal/daemon/daemon.go:101 -> internal/ffmpeg/transcode.go:194; SURVEY.md §8b "av1mi_open(config) / av1mi_encode /
 * av1mi_flush").  It owns the closed-GOP orchestration and the encoder's filter-parameter POLICY, so that no caller
 * re-implements them: `segments` independent closed GOPs are coded in lockstep (the t-th frames of all of them share every
 * launch: SURVEY.md §8e shards by closed-GOP segment), frame t = 0 of a GOP is a key frame, the others are P frames
 * predicted from the previous frame after deblocking + CDEF + loop restoration.
 *
 * Data path per frame batch: the caller fills the session's pinned host buffers with the source planes
 * (av1mi_gop_acquire_input), av1mi_gop_submit() queues the upload (own copy stream), the block pipeline + in-loop filters
 * (the context's stream) and the download of the frame's SYMBOLS (modes or vectors + skip flags + int16 levels; own copy
 * stream) into pinned host memory; av1mi_gop_collect() waits for the oldest submitted batch and hands those symbols out
 * together with the frame-header parameters the policy chose — exactly what the host bitstream writer
 * (av1-go_amd/host/av1_bitstream.hpp, entropy coding stays on the host cores) needs.  av1mi_gop_max_in_flight() = 3 batches
 * can be in flight: submit(t + 2) before collect(t) overlaps the upload of one batch, the kernels of the next, the GPU coder
 * of the third and the host's work on what was collected. */
typedef struct av1mi_gop_config {
  int width, height;     /* luma samples, multiples of 8 */
  int bit_depth;         /* 8 or 10 */
  int base_q_idx;        /* 1..255 (the reference's only quality knob is DetermineQuality, transcode.go:157-165) */
  int gop_length;        /* frames per closed GOP, >= 1 */
  int segments;          /* closed GOPs coded in lockstep, >= 1 */
  int search_range;      /* integer motion search range in samples, 0..15 */
  int gpu_entropy;       /* 0: the symbols are downloaded, the host entropy-codes them (north_star's split);
                            1: the AV1 tile entropy coder runs on the GPU (side stream), only tile payloads are downloaded;
                            2: both (tests compare the two) */
  /* Sources whose size is not a multiple of 8: width / height above are the CODED size (the true size rounded up to 8; the caller
   * replicates the source's last column / row into the padding of the input planes) and these the TRUE size that goes into the
   * sequence header (av1mi_obu_frame.visible_*); coded - visible < 8; 0 = the coded size.  The session then does what makes a
   * decoder — which works at the coded size except where the spec says FrameWidth / FrameHeight — reconstruct the same pictures:
   * deblocking units that start beyond the true size are not filtered (spec 7.14.2 onScreen), the true last column / row of the
   * deblocked, CDEF and restored planes is replicated into the padding (what the decoder's clamps at lastX / lastY, 7.11.3.4, and
   * PlaneEndX / PlaneEndY, 7.17, read), the restoration units of the tile syntax are counted on the true size. */
  int visible_width, visible_height;
  /* gpu_entropy != 0: where the tile coder runs.  0 (default) = tokenizer + chains on a side stream and the range coder on a third,
   * beside the next batches' block pipeline (fastest); 1 = the whole coder on one side stream; 2 = on the main stream, serialised
   * behind the filters — slower, but every kernel then runs ALONE on the GPU: the arrangement for per-kernel measurements (bench.py
   * `kernels_isolated`, rocprofv3 passes).  The environment variable AV1MI_CODER_STREAMS = split | side | main overrides it. */
  int coder_streams;
  int key_block_size;    /* 0 / 8: key frames in 8x8 blocks like every frame.  32: key frames in 32x32 blocks (luma 32x32 DCT, chroma 16x16,
                            transform type by mode) over every COMPLETE superblock row, 8x8 blocks in a last partial row: +3.7 dB at equal
                            size on the synthetic key frames at q 128, +1.85 dB at q 24.  Needs width % 32 == 0 */
} av1mi_gop_config;

/* Frame-header parameters chosen by the session's policy for one frame (non-normative encoder choices; the bitstream carries
 * them): deblocking level from the quantiser step (libaom's LPF_PICK_FROM_Q fit, key / inter frames differ for 8-bit), one
 * CDEF strength set and damping from the step, Wiener restoration with fixed taps on 64x64 units. */
typedef struct av1mi_frame_params {
  int frame_type;        /* 0 key frame, 1 inter frame */
  int base_q_idx;
  int lf_level[4];       /* luma vertical, luma horizontal, U, V */
  int lf_sharpness;
  int cdef_damping;
  uint8_t cdef_y, cdef_uv;      /* (primary strength << 2) | secondary code */
  int lr_unit_size;             /* luma and chroma restoration unit size in samples of the plane */
  int8_t lr_unit_y[8], lr_unit_uv[8];   /* the unit record every unit of the plane uses (layout of av1mi_lr_frames) */
} av1mi_frame_params;
/* the policy alone (tests build the oracle's chain from it) */
int av1mi_policy_frame_params(int base_q_idx, int bit_depth, int frame_type, av1mi_frame_params *out);

typedef struct av1mi_gop_frame {    /* one collected frame batch; host pointers into the session's pinned buffers, valid until
                                       the next av1mi_gop_submit */
  av1mi_frame_params params;
  int segments;                     /* batch size */
  size_t blocks_per_frame;          /* (width / 8) * (height / 8); per-block arrays hold segments * blocks_per_frame entries */
  /* the symbols: NULL with gpu_entropy = 1 (the host gets the coded payloads only), unless the GPU coder gave the batch back */
  const uint8_t *y_mode, *uv_mode;  /* key frames: intra modes per 8x8 block (0 DC .. 12 PAETH) */
  const int16_t *mv;                /* inter frames: (x, y) per block in 1/8 luma samples */
  const uint8_t *skip;              /* inter frames: 1 = no non-zero level in the block */
  const int16_t *lev_y, *lev_u, *lev_v;   /* 64 / 16 / 16 levels per block, row-major inside a block */
  /* gpu_entropy != 0: the finished tile payloads of the batch, frame-major / raster inside a frame, back to back */
  int tiles_per_frame;
  const uint32_t *tile_size;        /* segments * tiles_per_frame entries */
  const uint8_t *tile_payload;
  uint64_t payload_bytes;
  /* restoration ON (1) / OFF (0) per segment and plane, [segment * 3 + plane]: the encoder keeps params.lr_unit_* for a plane only
   * where it lowered the squared error against the source; an OFF plane is signalled with lr_type NONE in the frame header */
  const uint8_t *lr_on;
  int key_block_size;               /* of this frame: 8, or 32 (key frames of a key_block_size 32 session).  32: y_mode / uv_mode hold, per
                                       segment (stride key_modes_stride bytes), the modes of the 32x32 blocks of the complete superblock rows in
                                       raster order from entry 0, and from byte offset key_modes_band (their place in the 8x8 grid) the 8x8
                                       blocks of the last partial row; the levels
                                       are block-contiguous per region in the same planes (a region's blocks tile its rows of the plane) */
  int key_modes_stride, key_modes_band;
} av1mi_gop_frame;

typedef struct av1mi_gop av1mi_gop;
int av1mi_gop_open(av1mi_ctx *ctx, const av1mi_gop_config *cfg, av1mi_gop **out);
void av1mi_gop_close(av1mi_gop *g);
/* pinned host planes for the NEXT batch: segment s occupies rows [s * height, (s + 1) * height) of the luma plane
 * (stride = width samples, uint8 for 8-bit, uint16 otherwise) and the matching rows of the half-size chroma planes.
 * Blocks until the upload that last used these buffers has finished. */
int av1mi_gop_acquire_input(av1mi_gop *g, void **y, void **u, void **v);
/* queue the batch in the acquired buffers.  frame_type: 0 key, 1 inter, -1 = by position in the GOP (gop_length).
 * AV1MI_E_INVAL when av1mi_gop_max_in_flight() batches are already in flight (collect first). */
int av1mi_gop_submit(av1mi_gop *g, int frame_type);
/* The same for a batch whose source planes are ALREADY in device memory (same layout as the pinned planes: segments stacked, stride =
 * width): no upload is queued, the kernels read the caller's buffers, which must stay valid and unchanged until the batch has been
 * collected.  No av1mi_gop_acquire_input() before it.  (A decoder that leaves its frames in HBM feeds the session this way; bench.py
 * times this path, the bench contract's "inputs already resident in HBM".) */
int av1mi_gop_submit_device(av1mi_gop *g, const void *d_y, const void *d_u, const void *d_v, int frame_type);
/* wait for the oldest batch in flight and describe its symbols; AV1MI_E_INVAL when nothing is in flight */
int av1mi_gop_collect(av1mi_gop *g, av1mi_gop_frame *out);
/* number of batches in flight (0..av1mi_gop_max_in_flight()) */
int av1mi_gop_max_in_flight(void);
int av1mi_gop_pending(av1mi_gop *g);
/* gpu_entropy != 0: batches whose tiles exceeded the GPU coder's capacity so far.  Such a batch is handed out by
 * av1mi_gop_collect with tile_size == NULL and the symbols filled in instead (the caller entropy-codes it on the host). */
long av1mi_gop_entropy_fallbacks(av1mi_gop *g);
/* the reference frame(s) produced by the LAST submitted batch (after all in-loop filters): host buffers of the stacked-plane
 * sizes; synchronises the session.  For tests and PSNR. */
int av1mi_gop_download_reference(av1mi_gop *g, void *y, void *u, void *v);

/* ---- host-pointer single-block forms (SURVEY.md §8b "per-stage test entry points"): copy in, run the
 * same kernels, copy out, synchronous. */
int av1mi_inv_txfm2d_add(av1mi_ctx *ctx, const int32_t *coef, void *dst, int stride, int tx_size, int tx_type, int bd);
int av1mi_fwd_txfm2d(av1mi_ctx *ctx, const int16_t *resid, int stride, int32_t *coef, int tx_size, int tx_type);

#ifdef __cplusplus
}
#endif
#endif /* AV1MI_H */
