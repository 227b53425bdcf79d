// av1mi_internal.hpp — declarations shared by the kernel translation units and the C ABI (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/av1mi.h"

namespace av1mi {

// XCD-aware block order (cdna_hip_programming.md T1).  The dispatcher deals consecutive workgroup ids to the 8 XCDs in turn, and
// each XCD has its own L2: with a plain raster grid every neighbour of a tile (left, right, above, below) runs on another XCD,
// so the halo rows and columns that adjacent tiles share are fetched through several L2s (the filter kernels read 3-4x their
// algorithmic bytes by FETCH_SIZE).  The remap gives the ids that share an XCD (id % 8) one contiguous run of `nwg / 8` tiles in
// raster order, bijective for any nwg; it is an affinity hint only, nothing depends on where a block actually runs.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
// row * stride of a plane row: both operands are below 2^24 and the product below 2^32 (the C ABI admits frames up to 16384 x
// 16384), so it is ONE full-rate 24-bit multiply; written as (size_t)row * stride it is a 64-bit multiply-add, four passes.
__device__ __forceinline__ size_t row_off(int row, int stride) { return (size_t)__umul24((unsigned)row, (unsigned)stride); }

// raster tile index -> (x, y, z) of a gx x gy x gz grid
struct Tile3 { int x, y, z; };
__device__ __forceinline__ Tile3 xcd_tile(unsigned gx, unsigned gy, unsigned gz) {
  (void)gz;   // the launch's 1-D grid is exactly gx * gy * gz workgroups (every launch_* computes it from the same expressions)
  const unsigned t = xcd_swizzle(blockIdx.x, gridDim.x);
  const unsigned z = t / (gx * gy), rem = t - z * gx * gy, y = rem / gx;
  return { (int)(rem - y * gx), (int)y, (int)z };
}

// One transform launch: either a block list or an implicit grid of equal blocks.
struct TxLaunch {
  int32_t *coef;            // int32 coefficients (K2: in, K1: out)
  void *plane;              // K2: uint8/uint16 prediction->reconstruction; K1: int16 residual
  int stride;               // samples
  int nblocks;
  const av1mi_txb *list;    // non-null: list form
  const uint8_t *tx_types;  // grid form: per-block type or null
  int uniform_type;
  int blocks_per_row;
};

// K3: a list of equally-sized blocks predicted from `ref` into `dst`
struct IntraLaunch {
  const void *ref; void *dst;
  int ref_stride, dst_stride, bd, nblocks;
  const av1mi_intra_blk *blocks;
};
hipError_t launch_intra_pred(int tx_size, const IntraLaunch &L, hipStream_t s);
struct CflLaunch {
  const void *luma; void *dst;
  int luma_stride, dst_stride, bd, nblocks;
  const av1mi_cfl_blk *blocks;
};
hipError_t launch_cfl_pred(int tx_size, const CflLaunch &L, hipStream_t s);

// K5: one plane, source -> destination (different allocations)
struct DeblockLaunch {
  const void *src; void *dst;
  int src_stride, dst_stride, w, h, bd, is_chroma;
  const uint32_t *mi; int mi_stride;   // (h/4) x (w/4) units of 4 bytes, see av1mi.h
  int sharpness;
  int nframes; size_t mi_frame_stride;   // frames stacked vertically (h rows each); mi units between frames, 0 = shared
};
hipError_t launch_deblock(const DeblockLaunch &L, hipStream_t s);

// fused intra-only pipeline over a segment of stacked frames
struct IntraPipeLaunch {
  const void *src[3]; void *rec[3]; int16_t *lev[3];
  uint8_t *modes_y, *modes_uv;
  int w, h, stride_y, stride_uv, bd, nframes, dc_q, ac_q;
  int dc_quant, ac_quant;   // (1 << 16) / step (libaom quant_fp): computed once by the host, see block_code.hpp
  int open_loop;            // 1: the modes are decided on the source first (k_intra_modes), the chain predicts each block once
  int frame_rows;           // luma rows between the stacked frames of a plane (= h, or more when the job codes a band of rows of every frame)
  int modes_stride;         // mode bytes between frames (= blocks of the job, or more)
};
hipError_t launch_intra_pipe(const IntraPipeLaunch &L, int bs, hipStream_t s);

// K4: a list of equally-sized blocks predicted from `ref` into `dst`
struct McLaunch {
  const void *ref; void *dst;
  int ref_stride, dst_stride, plane_w, plane_h, bd, nblocks;
  const av1mi_mc_blk *blocks;
};
hipError_t launch_mc(int size_id, const McLaunch &L, hipStream_t s);

// K6: CDEF of 4:2:0 frames stacked vertically
struct CdefLaunch {
  const void *src[3]; void *dst[3];
  int w, h, stride_y, stride_uv, bd, damping, nframes;
  const uint8_t *sb_strength; size_t sb_frame_stride;   // 4 bytes per 64x64; entries between frames (0 = shared)
  const uint8_t *skip8; size_t skip_frame_stride;       // 1 byte per 8x8 luma block; bytes between frames (0 = shared)
};
hipError_t launch_cdef(const CdefLaunch &L, hipStream_t s);

// K7: loop restoration of one plane, frames stacked vertically
struct LrLaunch {
  const void *cdef, *dbl; void *out;
  int stride, w, h, bd, ss, unit_size, nframes;
  const int8_t *units; size_t unit_frame_stride;   // 8 bytes per unit; units between frames (0 = shared)
  // the on/off decision (orig != nullptr): the source plane, and per (frame, stripe) two 64-bit sums: squared error of the restored
  // samples, of the CDEF samples (zeroed by the caller; sse_stripes = lr_stripes(h, ss))
  const void *orig; unsigned long long *sse; int sse_stripes;
  // two-pass form of the decision (pass 0 = everything in one launch): pass 1 restores only the tiles the sums run over, pass 2 the
  // others, and only in the frames whose flag keep[f * keep_stride] (written by the decision between the two) is set
  int pass; const uint8_t *keep; int keep_stride;
  int no_sgr;       // the caller's promise that no unit is self-guided (type 2): the kernel without that path's LDS
};
hipError_t launch_lr(const LrLaunch &L, hipStream_t s);
int lr_stripes(int h, int ss);
hipError_t launch_zero16(void *p, size_t bytes, hipStream_t s);
hipError_t launch_lr_decide3(const unsigned long long *sse, int nframes, int stripes_y, int stripes_c, uint8_t *on, hipStream_t s);
hipError_t launch_extend(void *plane, int stride, int w, int h, int vw, int vh, int bd, int nframes, hipStream_t s);
hipError_t launch_lr_decide(const unsigned long long *sse, int nframes, int stripes, uint8_t *on, int on_stride, hipStream_t s);

// inter (P-frame) pipeline over the t-th frames of a batch of segments, stacked like the intra job
struct InterLaunch {
  const void *src[3]; const void *ref[3]; void *rec[3]; int16_t *lev[3];
  int16_t *mvs; uint8_t *skip;
  int w, h, stride_y, stride_uv, bd, nframes, dc_q, ac_q, range;
  int dc_quant, ac_quant;   // (1 << 16) / step (libaom quant_fp): computed once by the host, see block_code.hpp
  const void *ref_alt[3]; const uint8_t *ref_sel;   // optional: ref_sel[f * 3 + p] == 0 -> frame f predicts plane p from ref_alt[p]
};
hipError_t launch_me_int(const InterLaunch &L, hipStream_t s);
hipError_t launch_inter_pipe(const InterLaunch &L, hipStream_t s);

int tx_width(int tx_size);
int tx_height(int tx_size);
hipError_t launch_inv_txfm(int tx_size, const TxLaunch &L, int bd, hipStream_t s);
hipError_t launch_fwd_txfm(int tx_size, const TxLaunch &L, hipStream_t s);
hipError_t launch_quantize(const int32_t *coef, int16_t *levels, int32_t *dqcoef, long long n, int coef_per_blk,
                           int dc_q, int ac_q, int log_scale, hipStream_t s);
hipError_t launch_dequantize(const int16_t *levels, int32_t *dqcoef, long long n, int coef_per_blk, int dc_q,
                             int ac_q, int log_scale, int bd, hipStream_t s);

// the AV1 tile entropy coder's per-context scratch (av1_entropy_kernels.hip); created on first use, freed by av1mi_close
struct av1mi_av1ent_state_fwd;
}  // namespace av1mi
struct av1mi_av1ent_state;
namespace av1mi {
av1mi_av1ent_state *av1ent_new();
void av1ent_free(av1mi_av1ent_state *st);
av1mi_av1ent_state *ctx_av1ent(av1mi_ctx *ctx);
hipStream_t ctx_side_stream(av1mi_ctx *ctx);
hipStream_t ctx_back_stream(av1mi_ctx *ctx);
// the AV1 tile coder of include/av1mi.h's av1mi_av1_entropy_job in two halves: info + tokens + chains on `front`, the serial range
// coder + scan + gather on `back` (the same stream, or a second one: the lists are double-buffered, so the front half of the next
// job runs beside the back half of this one)
int av1_entropy_submit(av1mi_ctx *ctx, const struct av1mi_av1_entropy_job *j, hipStream_t front, hipStream_t back);
// ... and the halves on their own: the back half of a job may be launched later (the session defers the range coder of batch t behind
// the filters of batch t + 1 on the main stream); at most two jobs' back halves can be outstanding (the list sets)
int av1_entropy_front(av1mi_ctx *ctx, const struct av1mi_av1_entropy_job *j, hipStream_t front, int *ticket);
int av1_entropy_back(av1mi_ctx *ctx, int ticket, hipStream_t back);
// accessors of the opaque context for translation units other than capi.hip (gop_session.hip)
hipStream_t ctx_stream(av1mi_ctx *ctx);
int ctx_device(av1mi_ctx *ctx);
int ctx_fail(av1mi_ctx *ctx, int code, const char *fmt, ...);
// the per-kernel profile (av1mi_prof_*) for launches made outside capi.hip, on any stream of the context: an event pair around the
// launch(es) while profiling is enabled, nothing otherwise
struct ProfToken { hipEvent_t e0 = nullptr; int kind = 0; };
ProfToken ctx_prof_begin(av1mi_ctx *ctx, int kind, hipStream_t st);
void ctx_prof_end(av1mi_ctx *ctx, const ProfToken &t, hipStream_t st);

}  // namespace av1mi
