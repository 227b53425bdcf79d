/*
 * av1mi_host.h — C ABI of libav1mi_host.so: the host side of the MI355X AV1 backend (entropy coding, OBU packing and the
 * transcode-job contract stay on the host cores).  Together with include/av1mi.h this is everything a
 * cgo replacement of the reference's transcode step binds:
 *
 *   av1mi_run_transcode            stands in for  internal/ffmpeg/transcode.go:194  RunTranscode(ffmpegPath, args) (int, error)
 *   av1mi_obu_assemble_temporal_unit  frame header + tile group around tile payloads the GPU coder produced
 *   av1mi_obu_write_temporal_unit  the bitstream writer alone, for callers that drive the GOP session (av1mi.h av1mi_gop_*)
 *                                  themselves: symbols of one frame in, one AV1 temporal unit (Section-5 OBUs) out
 *
 * Plain C, no C++ types, no exceptions across the boundary, no callbacks; thread-safe (no global state).
 */
#ifndef AV1MI_HOST_H
#define AV1MI_HOST_H
#include <stddef.h>
#include <stdint.h>

#include "av1mi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Frame description for the bitstream writer.  All block arrays are raster order over the (width/8) x (height/8) grid of
 * 8x8 luma blocks; host pointers.  The tool set that can be described is listed in av1-go_amd/host/av1_bitstream.hpp. */
/* Frame description (plain C: also bound from Python through libav1mi_host.so, and by the cgo shim of INTEGRATION.md). */
/* All block arrays are raster order over the (width/8) x (height/8) grid of 8x8 luma blocks. */
typedef struct av1mi_obu_frame {
  int32_t width, height;       /* luma samples, multiples of 8 */
  int32_t bit_depth;           /* 8 or 10 */
  int32_t frame_type;          /* 0 key frame, 1 inter frame (reference LAST = the previously coded frame) */
  int32_t base_q_idx;          /* 1..255 */
  int32_t lf_level[4];         /* deblocking levels: luma vertical edges, luma horizontal edges, U, V (0..63) */
  int32_t lf_sharpness;        /* 0..7 */
  int32_t cdef_damping;        /* 3..6 */
  int32_t cdef_bits;           /* 0..3: 1 << cdef_bits strength sets */
  uint8_t cdef_y[8];           /* per set: (primary strength 0..15) << 2 | secondary code 0..3 (3 stands for strength 4) */
  uint8_t cdef_uv[8];
  const uint8_t *cdef_idx;     /* per 64x64 superblock (raster), index of its strength set; NULL = all 0 */
  int32_t lr_type[3];          /* per plane: 0 none, 1 Wiener, 2 self-guided, 3 switchable (frame restoration type) */
  int32_t lr_unit_shift;       /* luma restoration unit = 64 << shift (0..2) */
  int32_t lr_uv_shift;         /* chroma unit = luma unit >> lr_uv_shift (0 or 1) */
  const int8_t *lr_units[3];   /* per plane: unit rows x unit cols records of 8 bytes as in include/av1mi.h (av1mi_lr_frames) */
  int32_t reduced_tx_set;      /* 0 or 1 */
  int32_t disable_cdf_update;  /* 0 or 1 */
  int32_t tile_cols_log2, tile_rows_log2;  /* -1 = one superblock per tile (what the GPU pipeline's prediction assumes) */
  const uint8_t *y_mode;       /* intra blocks: 0 DC .. 12 PAETH */
  const int8_t *angle_y;       /* -3..3 for directional modes; NULL = 0 */
  const uint8_t *uv_mode;      /* 0..12, 13 = chroma from luma */
  const int8_t *angle_uv;      /* NULL = 0 */
  const int8_t *cfl_alpha;     /* 2 per block (U, V), -16..16, used where uv_mode == 13; NULL = none */
  const uint8_t *skip;         /* 1 = block coded with skip (no residual); NULL = 0 */
  const uint8_t *tx_type;      /* luma transform type per block (enum av1mi_tx_type); NULL = DCT_DCT */
  const uint8_t *is_inter;     /* inter frames: 1 = inter block (NULL = all inter) */
  const int16_t *mv;           /* inter blocks: (x, y) in 1/8 luma samples, multiples of 2 (quarter-sample precision) */
  const int16_t *lev_y;        /* 64 levels per block, row-major (row = vertical frequency) */
  const int16_t *lev_u, *lev_v;/* 16 levels per block */
  /* The size the decoder outputs, when the source is not a multiple of 8: width / height above are then the CODED size (the
   * true size rounded up to 8, the source edge replicated into the padding) and these the true one, written to the sequence
   * header; coded - visible < 8.  0 = same as width / height.  What the encoder loop must do for such a frame so that a decoder
   * reconstructs the same pictures: av1mi.h, av1mi_gop_config.visible_width. */
  int32_t visible_width, visible_height;
} av1mi_obu_frame;

/* One temporal unit: temporal delimiter [+ sequence header] + OBU_FRAME.  threads > 1 codes the tiles on that many host
 * threads.  Returns the size in bytes (the bytes are copied into out when they fit in cap), or -1 when the description
 * cannot be coded (text in err). */
long long av1mi_obu_write_temporal_unit(const av1mi_obu_frame *f, int with_sequence_header, int threads, uint8_t *out, long long cap,
                                        char *err, int errcap);

/* The same temporal unit when the tile payloads were coded by the GPU tile entropy coder (av1mi.h: av1mi_av1_entropy_encode, or
 * the GOP session with gpu_entropy != 0): f carries only the header fields (geometry, base_q_idx, filter parameters; the symbol
 * pointers are not read), payloads = the frame's ntiles finished tile payloads back to back in raster order, sizes[t] bytes each. */
long long av1mi_obu_assemble_temporal_unit(const av1mi_obu_frame *f, const uint8_t *payloads, const uint32_t *sizes, int ntiles,
                                           int with_sequence_header, uint8_t *out, long long cap, char *err, int errcap);

/* A collected GOP-session batch (av1mi.h av1mi_gop_collect) -> the temporal unit of its segment `seg`: what a caller that drives the
 * session itself does per frame (INTEGRATION.md §3 collect()).  Fills the frame description from fr->params and the restoration
 * decision fr->lr_on, then either wraps the segment's GPU-coded tile payloads (fr->tile_size != NULL) or entropy-codes its symbols
 * on `threads` host threads (gpu_entropy = 0, or a batch the GPU coder gave back).  width x height: the CODED size of the session;
 * visible_*: the true size when that is not a multiple of 8 (0 = the coded size).  Same return convention as above. */
long long av1mi_session_temporal_unit(const av1mi_gop_frame *fr, int seg, int width, int height, int bit_depth, int visible_width,
                                      int visible_height, int with_sequence_header, int threads, uint8_t *out, long long cap, char *err,
                                      int errcap);

/* ---- general block structure: every AV1 block size 4x4 .. 64x64 (incl. the 1:2 / 2:1 / 1:4 / 4:1 shapes), every partition type,
 * every transform size (TX_MODE_LARGEST or TX_MODE_SELECT) and all 16 transform types, the four interpolation filters.  It is
 * what the encoder behind transcode.go:120 (`-c:v:0 av1_vaapi`) may emit and what north_star's "4x4-64x64" names; the 8x8
 * description above is the subset the GPU block pipeline produces today.  Symbols are ARBITRARY (tests hand the writer random
 * ones and compare dav1d's decode with the oracle primitives, tests/test_av1_blocks.py).
 * Inter blocks: single reference LAST, coded as NEWMV against an EMPTY prediction list — the writer refuses an inter block that
 * has another inter block of the same tile within the reach of the MV prediction scan (5 units of 4 samples above / to the
 * left, one to the right); av1_bitstream.cpp has the full list for 8x8 blocks. */
typedef struct av1mi_obu_block {
  uint16_t mi_row, mi_col;     /* position in units of 4 luma samples */
  uint8_t bsize;               /* AV1 BLOCK_* order: 4X4 4X8 8X4 8X8 8X16 16X8 16X16 16X32 32X16 32X32 32X64 64X32 64X64 (12), then
                                  4X16 = 16, 16X4, 8X32, 32X8, 16X64, 64X16 = 21 */
  uint8_t skip;                /* no residual */
  uint8_t is_inter;            /* inter frames only */
  uint8_t y_mode, uv_mode;     /* intra blocks: 0 DC .. 12 PAETH; uv_mode 13 = chroma from luma (blocks up to 32x32) */
  int8_t angle_y, angle_uv;    /* -3..3, directional modes of blocks >= 8x8 (BLOCK_* order: also 4x16 / 16x4) */
  int8_t cfl_alpha_u, cfl_alpha_v;
  uint8_t tx_depth;            /* TX_MODE_SELECT: 0..2 halvings of the block's largest transform (intra: tx_depth, inter: txfm_split
                                  down to that depth everywhere); TX_MODE_LARGEST: must be 0 */
  uint8_t interp_filter;       /* inter blocks when the frame's filter is switchable: 0 regular, 1 smooth, 2 sharp */
  uint8_t reserved;
  int16_t mv_x, mv_y;          /* inter blocks: 1/8 luma samples, multiples of 2 */
  uint32_t tx_type_off;        /* first entry of the block in tx_type[]: one byte per LUMA transform block, coding order */
  uint32_t lev_off[3];         /* first level of the block's Y / U / V transform blocks in levels[]: the plane's transform blocks back
                                  to back in coding order, each min(h, 32) rows x min(w, 32) columns, row-major (row = vertical
                                  frequency).  Transform blocks that start outside the frame are not stored (nor coded). */
} av1mi_obu_block;

typedef struct av1mi_obu_blocks {
  av1mi_obu_frame hdr;         /* geometry, quantiser, filter parameters, tiles; its per-8x8 symbol pointers are not read */
  int32_t tx_mode_select;      /* 0 TX_MODE_LARGEST, 1 TX_MODE_SELECT */
  int32_t interp_filter;       /* inter frames: 0 regular, 1 smooth, 2 sharp, 3 bilinear for every block; 4 = switchable per block */
  int32_t high_precision_mv;   /* inter frames: allow_high_precision_mv (vectors in 1/8 luma samples: odd values allowed) */
  const uint8_t *partition;    /* one partition type (0 NONE, 1 HORZ, 2 VERT, 3 SPLIT, 4 HORZ_A, 5 HORZ_B, 6 VERT_A, 7 VERT_B, 8 HORZ_4,
                                  9 VERT_4) per decode_partition() call that starts inside the frame with a block of 8x8 or more, in
                                  decoding order (tiles in raster order, superblocks in raster order inside a tile) */
  size_t n_partition;
  const av1mi_obu_block *blocks;  /* in decoding order */
  size_t n_blocks;
  const uint8_t *tx_type;      /* enum av1mi_tx_type order (DCT_DCT 0 .. H_FLIPADST 15) */
  const int16_t *levels;
} av1mi_obu_blocks;

/* One temporal unit for a general block description; same return convention as av1mi_obu_write_temporal_unit. */
long long av1mi_obu_write_blocks_temporal_unit(const av1mi_obu_blocks *f, int with_sequence_header, uint8_t *out, long long cap,
                                               char *err, int errcap);

/* The drop-in for RunTranscode (transcode.go:194-315): argv as TranscodeArgs (transcode.go:17) builds it — the backend reads
 * "-i <input.y4m>", "-global_quality:v:0 <q>" and the output path (last argument), plus its own "-g", "-av1mi_device",
 * "-av1mi_segments", "-av1mi_gpu_entropy", "-av1mi_key_block_size", "-av1mi_tracks", "-threads"; everything else is accepted and ignored.  Returns 0 and leaves the output file in place on
 * success; -1 when the backend could not run at all (no HIP device: transcode.go:311); another non-zero code on failure.
 * err receives the reference-shaped text ("av1mi failed with exit code N: ...", at most 800 characters + "..."). */
int av1mi_run_transcode(int argc, const char *const *argv, char *err, size_t errcap);

#ifdef __cplusplus
}
#endif
#endif /* AV1MI_HOST_H */
