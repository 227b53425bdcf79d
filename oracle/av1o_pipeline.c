/*
 * av1o_pipeline.c — oracle-side frame loops: the same per-block stages the GPU pipeline runs,
 * applied block after block over a plane.  Used as the checker for whole-plane parity tests and as
 * bench.py's cpu_baseline ("port": own CPU restatement, not the reference — the reference's CPU path
 * does not exist in its tree).  TEST INFRASTRUCTURE ONLY; the encoder loops are non-normative: no external pin (see av1o_common.h).
 */
#include "av1o_common.h"
#include <string.h>

int av1o_fwd_txfm2d(const int16_t *resid, int stride, int32_t *coef, int tx_size, int tx_type, int bd);
int av1o_inv_txfm2d_add(const int32_t *coef, void *dst, int stride, int tx_size, int tx_type, int bd, int libaom_clamps);
int av1o_quantize(const int32_t *coef, int n, int dc_q, int ac_q, int log_scale, int16_t *levels, int32_t *dqcoef);
int av1o_quantize_r(const int32_t *coef, int n, int dc_q, int ac_q, int log_scale, int ac_round, int16_t *levels, int32_t *dqcoef);
/* Encoder policy (non-normative): the quantiser's AC rounding offset in inter frames, in 1/128 of the step: 0.4, a dead zone
 * (key frames: 64 = one half, libaom's quantize_fp).  -3.6 % BD-rate on the synthetic GOPs; kernels: txfm_cfg.hpp kAcRoundInter. */
#define AV1O_AC_ROUND_INTER 51
void av1o_dequantize(const int16_t *levels, int n, int dc_q, int ac_q, int log_scale, int bd, int32_t *dqcoef);
int av1o_tx_scale(int tx_size);
int av1o_dc_q(int qindex, int delta, int bd);
int av1o_ac_q(int qindex, int delta, int bd);
#include <stdlib.h>

/*
 * rows [by0,by1) of blocks of a plane split into equal tx_size blocks:
 * residual -> forward transform -> quantise -> dequantise -> inverse transform + add to `recon`
 * (which holds the prediction on entry).  levels: block-contiguous int16.  tx_types: one per block or NULL.
 */
int av1o_txq_plane(const int16_t *resid, void *recon, int stride, int blocks_per_row, int by0, int by1,
                   int tx_size, const uint8_t *tx_types, int uniform_type, int dc_q, int ac_q, int bd,
                   int16_t *levels) {
  const int w = av1o_tx_w[tx_size], h = av1o_tx_h[tx_size];
  const int cw = w > 32 ? 32 : w, ch = h > 32 ? 32 : h, n = cw * ch;
  const int ls = av1o_tx_scale(tx_size);
  const int bps = bd == 8 ? 1 : 2;
  int32_t coef[1024], dq[1024];
  for (int by = by0; by < by1; by++)
    for (int bx = 0; bx < blocks_per_row; bx++) {
      const int b = by * blocks_per_row + bx;
      const int tt = tx_types ? tx_types[b] : uniform_type;
      const size_t off = (size_t)by * h * stride + (size_t)bx * w;
      int rc = av1o_fwd_txfm2d(resid + off, stride, coef, tx_size, tt, bd);
      if (rc) return rc;
      av1o_quantize(coef, n, dc_q, ac_q, ls, levels + (size_t)b * n, NULL);
      av1o_dequantize(levels + (size_t)b * n, n, dc_q, ac_q, ls, bd, dq);
      rc = av1o_inv_txfm2d_add(dq, (char *)recon + off * bps, stride, tx_size, tt, bd, 1);
      if (rc) return rc;
    }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Intra-only frame encoder loop (BASELINE config 2), the checker of the GPU kernel k_intra_pipe.
 * The DECISIONS here (candidate list, SAD cost, first-minimum tie break, DCT_DCT only, one 64x64
 * superblock per tile, all blocks bs x bs in z-order) are this project's own encoder policy; the
 * ARITHMETIC it applies per block is the spec restatement above (intra prediction §7.11.2, forward /
 * inverse transform, quantiser, reconstruction).
 */
int av1o_intra_predict(const void *ref, int ref_stride, int bd, int bw, int bh, int mode, int angle_delta,
                       int disable_edge_filter, int filter_type, int n_top_px, int n_topright_px, int n_left_px,
                       int n_bottomleft_px, uint16_t *pred);

/* all 13 intra modes, angle delta 0: DC V H D45 D135 D113 D157 D203 D67 SMOOTH PAETH SMOOTH_V SMOOTH_H (first minimum wins) */
static const int intra_candidates[13] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 10, 11 };
/* Mode_To_Txfm (AV1 spec 5.11.47 compute_tx_type / libaom intra_mode_to_tx_type): the transform type an intra CHROMA block
 * takes from its prediction mode (it is not coded in the bitstream); luma types are coded explicitly and stay DCT_DCT here. */
const uint8_t av1o_mode_to_txfm[14] = { DCT_DCT, ADST_DCT, DCT_ADST, DCT_DCT, ADST_ADST, ADST_DCT, DCT_ADST, DCT_ADST, ADST_DCT,
                                         ADST_ADST, ADST_DCT, DCT_ADST, ADST_ADST, DCT_DCT };

static unsigned morton(unsigned x, unsigned y) {
  unsigned m = 0;
  for (int i = 0; i < 8; i++) m |= ((x >> i) & 1u) << (2 * i) | ((y >> i) & 1u) << (2 * i + 1);
  return m;
}
static int px_get(const void *p, int bd, size_t i) { return bd == 8 ? ((const uint8_t *)p)[i] : ((const uint16_t *)p)[i]; }
static void px_set(void *p, int bd, size_t i, int v) { if (bd == 8) ((uint8_t *)p)[i] = (uint8_t)v; else ((uint16_t *)p)[i] = (uint16_t)v; }

/* EXPERIMENT switch (tests / measurements only): 1 = OPEN-LOOP mode decision — the 13 candidates are predicted from the SOURCE
 * planes' neighbours (edge filter type 0), so that the decisions of all blocks are independent of each other; the chosen mode is
 * then predicted from the reconstruction as always. */
static int intra_open_loop = 0;
void av1o_set_intra_open_loop(int on) { intra_open_loop = on; }
/* The AC rounding offset of INTRA blocks in 1 / 128 of the step: 64 (one half, libaom's quantize_fp) in 8x8 and 16x16 blocks; 58 in the
 * 32x32 blocks of key frames (and their 16x16 chroma blocks) — measured on the synthetic key frames through the block writer: +0.26 dB
 * at equal size at q 128, +0.05 dB at q 24 (52: +0.33 / -0.14; 46: +0.30 / -0.10).  Kernels: txfm_cfg.hpp kAcRoundKey32.
 * av1o_set_intra_ac_round: EXPERIMENT switch (measurements only), 0 = the policy. */
#define AV1O_AC_ROUND_KEY32 58
static int intra_ac_round = 0;
void av1o_set_intra_ac_round(int r) { intra_ac_round = r == 64 ? 0 : r; }

/* encode one bs x bs block of nplanes planes sharing one mode (luma: 1 plane; chroma: U and V) */
static int encode_block(int nplanes, const void *const *src, void *const *rec, int stride, int bd, int bs, int x, int y,
                        int n_top, int n_topright, int n_left, int n_bottomleft, int filter_type, int dc_q, int ac_q,
                        int16_t *const *levels /* per plane, this block's bs*bs */, int ac_round) {
  uint16_t pred[2][64 * 64];
  const int tx_size = bs == 4 ? TX_4X4 : bs == 8 ? TX_8X8 : bs == 16 ? TX_16X16 : bs == 32 ? TX_32X32 : TX_64X64;
  const int bps = bd == 8 ? 1 : 2;
  long best = -1; int best_mode = 0;
  for (int ci = 0; ci < 13; ci++) {
    const int mode = intra_candidates[ci];
    long sad = 0;
    for (int p = 0; p < nplanes; p++) {
      av1o_intra_predict((const char *)(intra_open_loop ? src[p] : (const void *)rec[p]) + ((size_t)y * stride + x) * bps, stride, bd, bs, bs, mode, 0, 0,
                         intra_open_loop ? 0 : filter_type, n_top, n_topright, n_left, n_bottomleft, pred[0]);
      for (int r = 0; r < bs; r++)
        for (int c = 0; c < bs; c++) sad += labs((long)px_get(src[p], bd, (size_t)(y + r) * stride + x + c) - pred[0][r * bs + c]);
    }
    if (best < 0 || sad < best) { best = sad; best_mode = mode; }
  }
  /* chroma: implied by the mode (32x32 transforms have the DCT only, spec 5.11.40 / get_tx_set); luma: coded, DCT_DCT */
  const int tx_type = nplanes == 2 && bs < 32 ? av1o_mode_to_txfm[best_mode] : DCT_DCT;
  for (int p = 0; p < nplanes; p++) {
    int16_t resid[64 * 64];
    int32_t coef[1024], dq[1024];
    av1o_intra_predict((const char *)rec[p] + ((size_t)y * stride + x) * bps, stride, bd, bs, bs, best_mode, 0, 0, filter_type,
                       n_top, n_topright, n_left, n_bottomleft, pred[p]);
    for (int r = 0; r < bs; r++)
      for (int c = 0; c < bs; c++)
        resid[r * bs + c] = (int16_t)(px_get(src[p], bd, (size_t)(y + r) * stride + x + c) - pred[p][r * bs + c]);
    const int n = bs > 32 ? 1024 : bs * bs, ls = av1o_tx_scale(tx_size);
    av1o_fwd_txfm2d(resid, bs, coef, tx_size, tx_type, bd);
    av1o_quantize_r(coef, n, dc_q, ac_q, ls, intra_ac_round ? intra_ac_round : ac_round, levels[p], NULL);
    av1o_dequantize(levels[p], n, dc_q, ac_q, ls, bd, dq);
    /* reconstruct: write the prediction, then add the residual in place */
    for (int r = 0; r < bs; r++)
      for (int c = 0; c < bs; c++) px_set(rec[p], bd, (size_t)(y + r) * stride + x + c, pred[p][r * bs + c]);
    av1o_inv_txfm2d_add(dq, (char *)rec[p] + ((size_t)y * stride + x) * bps, stride, tx_size, tx_type, bd, 1);
  }
  return best_mode;
}

/*
 * One frame.  Planes are w x h (luma) and w/2 x h/2 (chroma), w and h multiples of bs (luma block size: 8 or 16 as the GPU pipeline
 * has them; 32 and 64 for rate-distortion measurements, tools/rd_blocksize.py).
 * Tiles are 64x64 luma superblocks; nothing is predicted across a tile edge.  levels_*: block-contiguous int16 in
 * raster order of blocks; modes_*: one byte per block, raster order (modes_uv shared by U and V).
 */
int av1o_intra_encode_frame(const void *src_y, const void *src_u, const void *src_v, void *rec_y, void *rec_u, void *rec_v,
                            int w, int h, int stride_y, int stride_uv, int bd, int bs, int qindex, int16_t *lev_y,
                            int16_t *lev_u, int16_t *lev_v, uint8_t *modes_y, uint8_t *modes_uv) {
  if ((bs != 8 && bs != 16 && bs != 32 && bs != 64) || (w % bs) || (h % bs) || (bd != 8 && bd != 10)) return -1;
  const int dc_q = av1o_dc_q(qindex, 0, bd), ac_q = av1o_ac_q(qindex, 0, bd);
  const int n = 64 / bs;                                   /* blocks per superblock side */
  const int bw = w / bs, bh = h / bs;                      /* frame size in blocks */
  const int cs = bs / 2;
  for (int sby = 0; sby * 64 < h; sby++)
    for (int sbx = 0; sbx * 64 < w; sbx++)
      for (unsigned k = 0; k < (unsigned)(n * n); k++) {
        /* k-th block in z-order */
        int bx = 0, by = 0;
        for (int i = 0; i < 4; i++) { bx |= ((k >> (2 * i)) & 1) << i; by |= ((k >> (2 * i + 1)) & 1) << i; }
        const int fx = sbx * n + bx, fy = sby * n + by;     /* frame block coordinates */
        if (fx >= bw || fy >= bh) continue;
        const int have_top = by > 0, have_left = bx > 0;
        const int have_tr = have_top && bx + 1 < n && fx + 1 < bw && morton(bx + 1, by - 1) < k;
        const int have_bl = have_left && by + 1 < n && fy + 1 < bh && morton(bx - 1, by + 1) < k;
        const size_t blk = (size_t)fy * bw + fx;
        /* filter type: a smooth-predicted neighbour inside the tile (spec get_filter_type) */
        int ft = 0, ftc = 0;
        if (have_top) { const int m = modes_y[blk - bw], mc = modes_uv[blk - bw]; ft |= m >= 9 && m <= 11; ftc |= mc >= 9 && mc <= 11; }
        if (have_left) { const int m = modes_y[blk - 1], mc = modes_uv[blk - 1]; ft |= m >= 9 && m <= 11; ftc |= mc >= 9 && mc <= 11; }
        {
          const void *s[1] = { src_y }; void *r[1] = { rec_y }; int16_t *l[1] = { lev_y + blk * bs * bs };
          modes_y[blk] = (uint8_t)encode_block(1, s, r, stride_y, bd, bs, fx * bs, fy * bs, have_top ? bs : 0, have_tr ? bs : 0,
                                               have_left ? bs : 0, have_bl ? bs : 0, ft, dc_q, ac_q, l, bs == 32 ? AV1O_AC_ROUND_KEY32 : 64);
        }
        {
          const void *s[2] = { src_u, src_v }; void *r[2] = { rec_u, rec_v };
          int16_t *l[2] = { lev_u + blk * cs * cs, lev_v + blk * cs * cs };
          modes_uv[blk] = (uint8_t)encode_block(2, s, r, stride_uv, bd, cs, fx * cs, fy * cs, have_top ? cs : 0, have_tr ? cs : 0,
                                                have_left ? cs : 0, have_bl ? cs : 0, ftc, dc_q, ac_q, l, bs == 32 ? AV1O_AC_ROUND_KEY32 : 64);
        }
      }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Inter (P) frame encoder loop (BASELINE config 3), the checker of k_me_int + k_inter_pipe.
 * Encoder policy (ours, non-normative): every block bs x bs (8) is inter-predicted from ONE reference frame
 * (the previous reconstructed, loop-filtered frame); integer full search +-range around the co-located block
 * by SAD of the 8 most significant bits on the block's even rows ((0,0) first, then raster order, strict improvement), then one half-pel refinement round
 * scored with the bilinear filter and one quarter-pel round scored with the regular 8-tap filter (8 neighbours each, fixed
 * order, strict improvement); the prediction itself always uses the regular 8-tap filter; chroma uses the same
 * vector; DCT_DCT residual coding as in the intra loop.  The prediction arithmetic is av1o_mc_block (spec 7.11.3.4).
 * mvs: int16 pairs (x, y) in 1/8 luma sample units, one per block, raster.  skip: 1 = all levels of the block zero.
 */
int av1o_mc_block(const void *ref, int stride, int plane_w, int plane_h, int bd, int x, int y, int w, int h, int mvx,
                  int mvy, int filt_x, int filt_y, uint16_t *pred);

/* integer-search cost: SAD on the 8 most significant bits of source and prediction, over the EVEN rows of the block (policy of
 * k_me_int; libaom's "downsampled SAD" speed feature: on the synthetic GOPs the vectors, bytes and PSNR do not change at all) */
static long block_sad8(const void *src, int stride, int bd, int x, int y, int bs, const uint16_t *pred) {
  long s = 0;
  for (int r = 0; r < bs; r += 2)
    for (int c = 0; c < bs; c++)
      s += labs((long)(px_get(src, bd, (size_t)(y + r) * stride + x + c) >> (bd - 8)) - (pred[r * bs + c] >> (bd - 8)));
  return s;
}
static long block_sad(const void *src, int stride, int bd, int x, int y, int bs, const uint16_t *pred) {
  long s = 0;
  for (int r = 0; r < bs; r++)
    for (int c = 0; c < bs; c++) s += labs((long)px_get(src, bd, (size_t)(y + r) * stride + x + c) - pred[r * bs + c]);
  return s;
}
static void code_inter_plane(const void *src, void *rec, int stride, int bd, int bs, int x, int y, const uint16_t *pred, int dc_q,
                             int ac_q, int16_t *levels) {
  int16_t resid[64 * 64];
  int32_t coef[1024], dq[1024];
  const int tx_size = bs == 4 ? TX_4X4 : bs == 8 ? TX_8X8 : TX_16X16;
  const int bps = bd == 8 ? 1 : 2, n = bs * bs;
  for (int r = 0; r < bs; r++)
    for (int c = 0; c < bs; c++) {
      resid[r * bs + c] = (int16_t)(px_get(src, bd, (size_t)(y + r) * stride + x + c) - pred[r * bs + c]);
      px_set(rec, bd, (size_t)(y + r) * stride + x + c, pred[r * bs + c]);
    }
  av1o_fwd_txfm2d(resid, bs, coef, tx_size, DCT_DCT, bd);
  av1o_quantize_r(coef, n, dc_q, ac_q, 0, AV1O_AC_ROUND_INTER, levels, NULL);
  av1o_dequantize(levels, n, dc_q, ac_q, 0, bd, dq);
  av1o_inv_txfm2d_add(dq, (char *)rec + ((size_t)y * stride + x) * bps, stride, tx_size, DCT_DCT, bd, 1);
}

int av1o_inter_encode_frame(const void *src_y, const void *src_u, const void *src_v, const void *ref_y, const void *ref_u,
                            const void *ref_v, void *rec_y, void *rec_u, void *rec_v, int w, int h, int stride_y, int stride_uv,
                            int bd, int bs, int qindex, int range, int16_t *lev_y, int16_t *lev_u, int16_t *lev_v, int16_t *mvs,
                            uint8_t *skip) {
  if (bs != 8 || (w % bs) || (h % bs) || (bd != 8 && bd != 10) || range < 0 || range > 15) return -1;
  const int dc_q = av1o_dc_q(qindex, 0, bd), ac_q = av1o_ac_q(qindex, 0, bd);
  const int bw = w / bs, bh = h / bs, cs = bs / 2;
  uint16_t pred[64], best_pred[64], pu[16], pv[16];
  for (int by = 0; by < bh; by++)
    for (int bx = 0; bx < bw; bx++) {
      const int x = bx * bs, y = by * bs;
      const size_t blk = (size_t)by * bw + bx;
      /* integer search */
      av1o_mc_block(ref_y, stride_y, w, h, bd, x, y, bs, bs, 0, 0, 0, 0, pred);
      long best = block_sad8(src_y, stride_y, bd, x, y, bs, pred);
      int bmx = 0, bmy = 0;   /* 1/8 units */
      for (int dy = -range; dy <= range; dy++)
        for (int dx = -range; dx <= range; dx++) {
          if (!dx && !dy) continue;
          av1o_mc_block(ref_y, stride_y, w, h, bd, x, y, bs, bs, dx * 16, dy * 16, 0, 0, pred);
          const long s = block_sad8(src_y, stride_y, bd, x, y, bs, pred);
          if (s < best) { best = s; bmx = dx * 8; bmy = dy * 8; }
        }
      /* Sub-sample refinement, two rounds of 8 neighbours in raster order with strict improvement (the centre keeps ties):
       *  - half-sample round with the BILINEAR filter (libaom's USE_2_TAPS sub-pel search): which half-sample neighbourhood the
       *    block lies in is decided as well by the rounded average of 2 / 4 reference samples as by the 8-tap filter;
       *  - quarter-sample round with the real (regular 8-tap) filter, the centre re-scored with it first so that all nine
       *    positions of the round are compared like with like.
       * Measured against 8-tap filters in both rounds on the synthetic clip: coded bytes +-0.05 %, PSNR-Y +-0.002 dB. */
      av1o_mc_block(ref_y, stride_y, w, h, bd, x, y, bs, bs, bmx * 2, bmy * 2, 0, 0, pred);
      best = block_sad(src_y, stride_y, bd, x, y, bs, pred);
      for (int step = 4; step >= 2; step >>= 1) {
        const int cx = bmx, cy = bmy, filt = step == 4 ? 3 : 0;      /* 3 = bilinear, 0 = regular 8-tap */
        if (step == 2) {
          av1o_mc_block(ref_y, stride_y, w, h, bd, x, y, bs, bs, cx * 2, cy * 2, 0, 0, pred);
          best = block_sad(src_y, stride_y, bd, x, y, bs, pred);
        }
        for (int k = 0; k < 9; k++) {
          if (k == 4) continue;
          const int mx = cx + (k % 3 - 1) * step, my = cy + (k / 3 - 1) * step;
          av1o_mc_block(ref_y, stride_y, w, h, bd, x, y, bs, bs, mx * 2, my * 2, filt, filt, pred);
          const long s = block_sad(src_y, stride_y, bd, x, y, bs, pred);
          if (s < best) { best = s; bmx = mx; bmy = my; }
        }
      }
      mvs[blk * 2] = (int16_t)bmx; mvs[blk * 2 + 1] = (int16_t)bmy;
      av1o_mc_block(ref_y, stride_y, w, h, bd, x, y, bs, bs, bmx * 2, bmy * 2, 0, 0, best_pred);
      av1o_mc_block(ref_u, stride_uv, w / 2, h / 2, bd, x / 2, y / 2, cs, cs, bmx, bmy, 0, 0, pu);
      av1o_mc_block(ref_v, stride_uv, w / 2, h / 2, bd, x / 2, y / 2, cs, cs, bmx, bmy, 0, 0, pv);
      code_inter_plane(src_y, rec_y, stride_y, bd, bs, x, y, best_pred, dc_q, ac_q, lev_y + blk * bs * bs);
      code_inter_plane(src_u, rec_u, stride_uv, bd, cs, x / 2, y / 2, pu, dc_q, ac_q, lev_u + blk * cs * cs);
      code_inter_plane(src_v, rec_v, stride_uv, bd, cs, x / 2, y / 2, pv, dc_q, ac_q, lev_v + blk * cs * cs);
      int nz = 0;
      for (int i = 0; i < bs * bs; i++) nz |= lev_y[blk * bs * bs + i];
      for (int i = 0; i < cs * cs; i++) nz |= lev_u[blk * cs * cs + i] | lev_v[blk * cs * cs + i];
      skip[blk] = nz == 0;
    }
  return 0;
}
