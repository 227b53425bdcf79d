/*
 * av1o_pipeline.c — oracle-side frame loops: the same per-block stages the GPU pipeline runs,
 * applied block after block over a plane.  Used as the checker for whole-plane parity tests and as
 * bench.py's cpu_baseline ("port": own CPU restatement, not the reference — the reference's CPU path
 * does not exist in its tree).  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED.
 */
#include "av1o_common.h"
#include <string.h>

int av1o_fwd_txfm2d(const int16_t *resid, int stride, int32_t *coef, int tx_size, int tx_type, int bd);
int av1o_inv_txfm2d_add(const int32_t *coef, void *dst, int stride, int tx_size, int tx_type, int bd, int libaom_clamps);
int av1o_quantize(const int32_t *coef, int n, int dc_q, int ac_q, int log_scale, int16_t *levels, int32_t *dqcoef);
void av1o_dequantize(const int16_t *levels, int n, int dc_q, int ac_q, int log_scale, int bd, int32_t *dqcoef);
int av1o_tx_scale(int tx_size);

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
