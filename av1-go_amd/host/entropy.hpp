// entropy.hpp — host-side entropy coding of the GPU pipeline's output ("stays on the host cores").
//
// What this IS: the AV1 multi-symbol arithmetic coder mechanics — 15-bit CDFs, the spec's interval partition
// ((R >> 8) * (f >> 6) >> 1 + 4 * (N - 1 - s), AV1 spec §8.2.6) and its CDF adaptation rule (rate 3 + (count > 15) +
// (count > 31) + min(log2 N, 2), §8.2.6) — driving a coefficient syntax shaped like AV1's: end-of-block class + offset bits,
// base tokens 0..3 with position/neighbour contexts, escape remainders, raw signs.  EVERY adaptive symbol is 4-ary (larger
// alphabets are coded as a chain of two 4-ary symbols, the second conditioned on the first — the same information by the
// chain rule): one CDF is three 15-bit values + a counter = 8 bytes, so a SIMD coder updates it with one 64-bit load and
// store (csrc/entropy_kernels.hip, one tile per lane).  Every tile has its own coder and CDF state: tiles are coded in
// parallel, on the host one frame per thread, on the GPU one tile per lane.
// What this is NOT: a conformant AV1 bitstream.  AV1's default CDF tables and full syntax cannot be restated from memory
// and nothing in the container holds them; the initial CDFs are this project's own constants
// (entropy_init.hpp).  The decoder below exists to prove the stream is complete and lossless w.r.t. the symbols.
//
// Tile payload, blocks in raster order inside the tile; per block
//   key frame: mode_y, mode_uv (each hi = m >> 2, lo = m & 3)
//   P frame:   skip; mv.x, mv.y as wrapping int16 differences to the left block of the same tile: class = min(bit length
//              of |d|, 15) as (hi, lo), the bits below the leading one raw (class 15: |d| - 16384 in 15 bits), sign raw
//   unless skipped, for Y 8x8, U 4x4, V 4x4 in zig-zag order:
//     eob class 0,1,2,3-4,5-8,9-16,17-32,33-64 as (hi, lo) + offset bits raw;
//     the token min(|l|, 3) of every coefficient below eob (context: plane type, band of the position, min(prev token, 2));
//     for every token 3, in order: k = floor(log2(|l| - 2)) as a chain of up to five symbols min(k - 3j, 3) that stops at
//     the first value below 3, then the k bits below the leading one raw;
//     the signs of the non-zero coefficients, 8 per raw symbol (first coefficient = most significant bit).
//   Raw bits: up to 8 at a time as one symbol over 2^n equal slots of (range >> n), value v in slot 2^n-1-v from the bottom,
//   the top slot taking the remainder.
// Frame record: [log2 tile] [varint payload size of every tile, raster order] [tile payloads].
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace av1mi_host {

// CDF ids (shared with oracle/av1o_entropy.c and csrc/entropy_kernels.hip)
enum {
  CDF_TOK = 0,      // + (pt * 4 + band) * 3 + prev          24
  CDF_GOL = 24,     // + pt * 5 + j                          10
  CDF_EOB_HI = 34,  // + pt                                   2
  CDF_EOB_LO = 36,  // + pt * 2 + hi                          4
  CDF_MODE_HI = 40, // + which (0 luma, 1 chroma)             2
  CDF_MODE_LO = 42, // + which * 4 + hi                       8
  CDF_SKIP = 50,    //                                        1
  CDF_MV_HI = 51,   // + comp                                 2
  CDF_MV_LO = 53,   // + comp * 4 + hi                        8
  CDF_COUNT = 61
};
struct Cdf4 { uint16_t c[3]; uint16_t count; };   // P(sym <= i) * 32768 for i = 0..2; the fourth value is 32768
struct EntropyModels {
  Cdf4 cdf[CDF_COUNT];
  EntropyModels();          // entropy_init.hpp
  void set_uniform();
};

struct RangeEncoder {
  std::vector<uint8_t> out;
  uint64_t low = 0;       // interval base: 16 + pending bits, carries go into bytes already in `out`
  uint32_t rng = 0x8000;  // 16-bit range, kept in [2^15, 2^16)
  int pending = 0;
  void encode(int s, Cdf4 &cdf);                     // adaptive 4-ary symbol
  void encode_bits(unsigned v, int nbits);           // equiprobable raw bits, most significant first
  void finish();
private:
  void add(uint64_t v);
  void normalize();
};
struct RangeDecoder {
  const uint8_t *buf = nullptr; size_t len = 0, pos = 0;
  uint32_t rng = 0x8000;
  uint64_t code = 0;      // (stream value - interval base), aligned like the encoder's `low`
  int avail = 0;          // valid bits below the 16-bit comparison window
  void init(const uint8_t *p, size_t n);
  int decode(Cdf4 &cdf);
  unsigned decode_bits(int nbits);
private:
  void refill();
};
void cdf_adapt(Cdf4 &cdf, int s);

struct FrameSyms {        // what one coded frame carries besides the header (all block-raster order, see include/av1mi.h)
  int width = 0, height = 0, key = 1;
  int tile = 64;          // entropy tile edge in luma samples (power of two, 32..4096)
  const int16_t *lev_y = nullptr, *lev_u = nullptr, *lev_v = nullptr;   // 8x8 luma / 4x4 chroma blocks, row-major inside a block
  const uint8_t *modes_y = nullptr, *modes_uv = nullptr;                // key frames
  const int16_t *mvs = nullptr; const uint8_t *skip = nullptr;          // P frames
};
// one tile's range-coded payload appended to e.out (e must be fresh)
void entropy_encode_tile(const FrameSyms &f, int tx, int ty, RangeEncoder &e, EntropyModels *final_models = nullptr);
std::vector<uint8_t> entropy_assemble_frame(int tile, const std::vector<const uint8_t *> &tiles, const std::vector<size_t> &sizes);
// returns the coded record of one frame
std::vector<uint8_t> entropy_encode_frame(const FrameSyms &f);
// inverse: fills caller-provided arrays of the sizes implied by width/height/key; returns false on a corrupt stream
bool entropy_decode_frame(const uint8_t *data, size_t n, int width, int height, int key, int16_t *lev_y, int16_t *lev_u, int16_t *lev_v,
                          uint8_t *modes_y, uint8_t *modes_uv, int16_t *mvs, uint8_t *skip);
// codes `frames` on `threads` host threads; out[i] = record of frames[i]
void entropy_encode_frames(const std::vector<FrameSyms> &frames, int threads, std::vector<std::vector<uint8_t>> *out);

}  // namespace av1mi_host
