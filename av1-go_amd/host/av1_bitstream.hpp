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

#include "../../include/av1mi_host.h"

namespace av1mi_host {
namespace av1 {

struct SequenceParams {
  int width = 0, height = 0, bit_depth = 8;
  bool operator==(const SequenceParams &o) const { return width == o.width && height == o.height && bit_depth == o.bit_depth; }
};

// the size a decoder outputs (av1mi_obu_frame.visible_*: 0 = the coded size) and the sequence header that announces it
inline int visible_width(const av1mi_obu_frame &f) { return f.visible_width ? f.visible_width : f.width; }
inline int visible_height(const av1mi_obu_frame &f) { return f.visible_height ? f.visible_height : f.height; }
inline SequenceParams sequence_params(const av1mi_obu_frame &f) {
  SequenceParams sp; sp.width = visible_width(f); sp.height = visible_height(f); sp.bit_depth = f.bit_depth;
  return sp;
}
// OBU_TEMPORAL_DELIMITER (spec 5.6)
std::vector<uint8_t> temporal_delimiter_obu();
// OBU_SEQUENCE_HEADER (spec 5.5)
std::vector<uint8_t> sequence_header_obu(const SequenceParams &sp);
// OBU_FRAME (spec 5.10: frame header + tile group).  threads > 1 codes tiles on that many host threads.
// Returns false and fills *err when the description cannot be coded with the tool set above.
bool frame_obu(const av1mi_obu_frame &f, int threads, std::vector<uint8_t> *out, std::string *err);
// the same OBU from tile payloads coded elsewhere (the GPU tile entropy coder): `payloads` = the ntiles finished tile payloads
// back to back in raster order, sizes[t] bytes each
bool frame_obu_from_tiles(const av1mi_obu_frame &f, const uint8_t *payloads, const uint32_t *sizes, int ntiles, std::vector<uint8_t> *out,
                          std::string *err);
// the op-stream formulation of the tile syntax (csrc/av1_ops.hpp) run on the host: the GPU coder's CPU twin (av1_opstream.cpp)
// the writer's range coder over a raw list of interval updates (tests: the op-stream coder's lazy byte output against this one)
std::vector<uint8_t> range_code_raw(const uint32_t *fl, const uint32_t *fh, const uint8_t *sym, const uint8_t *nsym, size_t count);
bool opstream_supported(const av1mi_obu_frame &f, std::string *why);
bool opstream_tiles(const av1mi_obu_frame &f, std::vector<std::vector<uint8_t>> *tiles, std::string *err, int key_rows32 = 0);
// the general block-structured writer (av1_blockstream.cpp, include/av1mi_host.h av1mi_obu_blocks): one temporal unit
bool blocks_temporal_unit(const av1mi_obu_blocks &d, bool with_sequence_header, std::vector<uint8_t> *out, std::string *err, int threads = 1,
                          const size_t (*tile_start)[2] = nullptr);
// one temporal unit: delimiter [+ sequence header] + frame
bool temporal_unit(const av1mi_obu_frame &f, bool with_sequence_header, int threads, std::vector<uint8_t> *out, std::string *err);

}  // namespace av1
}  // namespace av1mi_host
