// av1_bitstream.hpp — host-side AV1 bitstream writer ("entropy coding and OBU packing stay
// on the host cores"; it produces what the reference's FFmpeg child emits for `-c:v:0 av1_vaapi`, internal/ffmpeg/
// transcode.go:120 — an AV1 elementary stream).
//
// Input: the symbols the GPU block pipeline leaves behind for one frame (per 8x8 block: prediction mode(s) or motion
// vector, skip flag, int16 quantised levels of the 8x8 luma and the two 4x4 chroma transform blocks) plus the frame's
// filter parameters.  Output: a Section-5 ("low overhead") OBU stream: temporal delimiter, sequence header, one
// OBU_FRAME per frame.  Written from the AV1 Bitstream & Decoding Process Specification (syntax sections 5.5 sequence
// header, 5.9 frame header, 5.11 tile group / block / residual syntax, 8.2 symbol coder, 9.3 CDF selection); every
// function names the syntax table it writes.  Verified by decoding with dav1d 1.5.3 (tests/test_av1_conformance.py).
//
// Tool set coded (the encoder policy of the GPU kernels, DESIGN.md "Encoder policy"): 4:2:0, 8 or 10 bit, 64x64
// superblocks, every superblock a tile of its own, partition split down to 8x8 blocks, TX_MODE_LARGEST (8x8 luma / 4x4
// chroma transforms), all 13 intra modes with angle deltas, chroma-from-luma, explicit luma transform type (2-D
// DCT/ADST/FLIPADST classes), single-reference inter blocks (LAST) with NEWMV coding against the spec's MV prediction
// list, regular 8-tap interpolation, deblocking, CDEF (up to 8 strength sets, per-superblock index), loop restoration
// (Wiener / self-guided / switchable per unit).  Not coded: other block sizes, compound / OBMC / warped motion, palette,
// intra block copy, filter-intra, segmentation, delta q / lf, quantiser matrices, super-resolution, film grain.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

extern "C" {
// Frame description (plain C: also bound from Python through libav1mi_host.so, and by the cgo shim of INTEGRATION.md).
// All block arrays are raster order over the (width/8) x (height/8) grid of 8x8 luma blocks.
typedef struct av1mi_obu_frame {
  int32_t width, height;       // luma samples, multiples of 8
  int32_t bit_depth;           // 8 or 10
  int32_t frame_type;          // 0 key frame, 1 inter frame (reference LAST = the previously coded frame)
  int32_t base_q_idx;          // 1..255
  int32_t lf_level[4];         // deblocking levels: luma vertical edges, luma horizontal edges, U, V (0..63)
  int32_t lf_sharpness;        // 0..7
  int32_t cdef_damping;        // 3..6
  int32_t cdef_bits;           // 0..3: 1 << cdef_bits strength sets
  uint8_t cdef_y[8];           // per set: (primary strength 0..15) << 2 | secondary code 0..3 (3 stands for strength 4)
  uint8_t cdef_uv[8];
  const uint8_t *cdef_idx;     // per 64x64 superblock (raster), index of its strength set; NULL = all 0
  int32_t lr_type[3];          // per plane: 0 none, 1 Wiener, 2 self-guided, 3 switchable (frame restoration type)
  int32_t lr_unit_shift;       // luma restoration unit = 64 << shift (0..2)
  int32_t lr_uv_shift;         // chroma unit = luma unit >> lr_uv_shift (0 or 1)
  const int8_t *lr_units[3];   // per plane: unit rows x unit cols records of 8 bytes as in include/av1mi.h (av1mi_lr_frames)
  int32_t reduced_tx_set;      // 0 or 1
  int32_t disable_cdf_update;  // 0 or 1
  int32_t tile_cols_log2, tile_rows_log2;  // -1 = one superblock per tile (what the GPU pipeline's prediction assumes)
  const uint8_t *y_mode;       // intra blocks: 0 DC .. 12 PAETH
  const int8_t *angle_y;       // -3..3 for directional modes; NULL = 0
  const uint8_t *uv_mode;      // 0..12, 13 = chroma from luma
  const int8_t *angle_uv;      // NULL = 0
  const int8_t *cfl_alpha;     // 2 per block (U, V), -16..16, used where uv_mode == 13; NULL = none
  const uint8_t *skip;         // 1 = block coded with skip (no residual); NULL = 0
  const uint8_t *tx_type;      // luma transform type per block (enum av1mi_tx_type); NULL = DCT_DCT
  const uint8_t *is_inter;     // inter frames: 1 = inter block (NULL = all inter)
  const int16_t *mv;           // inter blocks: (x, y) in 1/8 luma samples, multiples of 2 (quarter-sample precision)
  const int16_t *lev_y;        // 64 levels per block, row-major (row = vertical frequency)
  const int16_t *lev_u, *lev_v;// 16 levels per block
} av1mi_obu_frame;
}

namespace av1mi_host {
namespace av1 {

struct SequenceParams {
  int width = 0, height = 0, bit_depth = 8;
  bool operator==(const SequenceParams &o) const { return width == o.width && height == o.height && bit_depth == o.bit_depth; }
};

// OBU_TEMPORAL_DELIMITER (spec 5.6)
std::vector<uint8_t> temporal_delimiter_obu();
// OBU_SEQUENCE_HEADER (spec 5.5)
std::vector<uint8_t> sequence_header_obu(const SequenceParams &sp);
// OBU_FRAME (spec 5.10: frame header + tile group).  threads > 1 codes tiles on that many host threads.
// Returns false and fills *err when the description cannot be coded with the tool set above.
bool frame_obu(const av1mi_obu_frame &f, int threads, std::vector<uint8_t> *out, std::string *err);
// one temporal unit: delimiter [+ sequence header] + frame
bool temporal_unit(const av1mi_obu_frame &f, bool with_sequence_header, int threads, std::vector<uint8_t> *out, std::string *err);

}  // namespace av1
}  // namespace av1mi_host
