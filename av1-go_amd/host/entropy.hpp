// entropy.hpp — host-side entropy coding of the GPU pipeline's output ("stays on the host cores").
//
// What this IS: the AV1 multi-symbol arithmetic coder mechanics — 15-bit CDFs, the spec's interval partition
// ((R >> 8) * (f >> 6) >> 1 + 4 * (N - s), AV1 spec §8.2.6) and its CDF adaptation rule (rate 3 + (count > 15) +
// (count > 31) + min(log2 N, 2), §8.2.6) — driving a coefficient syntax shaped like AV1's (end-of-block class + extra
// bits, base levels 0..3 with neighbour contexts, Exp-Golomb remainder, raw sign), one independent coder state per frame so
// that frames are coded in parallel on the host cores.
// What this is NOT: a conformant AV1 bitstream.  AV1's default CDF tables and full syntax cannot be restated from memory
// and nothing in the container holds them; all CDFs here start uniform and the syntax is this project's.
// The decoder below exists to prove the stream is complete and lossless w.r.t. the levels (round-trip tests).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace av1mi_host {

struct RangeEncoder {
  std::vector<uint8_t> out;
  uint64_t low = 0;       // pending interval base, bits above `shift` are carries into already buffered bytes
  uint32_t rng = 0x8000;  // 16-bit range, kept in [2^15, 2^16)
  int pending = 0;        // bits of `low` not yet moved to `out`
  void encode(int s, uint16_t *cdf, int nsyms);     // adaptive symbol, cdf has nsyms + 1 entries (last = adaptation count)
  void encode_bits(unsigned v, int nbits);           // equiprobable raw bits, MSB first
  void finish();
private:
  void normalize();
  void put_byte_with_carry();
};
struct RangeDecoder {
  const uint8_t *buf = nullptr; size_t len = 0, pos = 0;
  uint32_t rng = 0x8000;
  uint64_t code = 0;      // (stream value - interval base), aligned like the encoder's `low`
  int avail = 0;          // valid fractional bits below the 16-bit comparison window
  void init(const uint8_t *p, size_t n);
  int decode(uint16_t *cdf, int nsyms);
  unsigned decode_bits(int nbits);
private:
  void refill();
};
void cdf_init_uniform(uint16_t *cdf, int nsyms);
void cdf_adapt(uint16_t *cdf, int s, int nsyms);

// adaptive models of one tile; every CDF is N cumulative 15-bit values followed by the adaptation counter.
// Initial values: entropy_init.hpp (this project's own constants, tools/train_cdfs.py), NOT AV1's default tables.
struct EntropyModels {
  uint16_t eob[2][9], tok[2][4][3][5], gol[2][17], mode[2][14], skip[3], mvc[2][17];   // [plane type][...]
  EntropyModels();
  void set_uniform();
};

struct FrameSyms {        // what one coded frame carries besides the header (all block-raster order, see include/av1mi.h)
  int width = 0, height = 0, key = 1;
  int tile = 64;          // entropy tile edge in luma samples (power of two, 32..4096): every tile has its own coder + CDF state
  const int16_t *lev_y = nullptr, *lev_u = nullptr, *lev_v = nullptr;   // 8x8 luma / 4x4 chroma blocks, row-major inside a block
  const uint8_t *modes_y = nullptr, *modes_uv = nullptr;                // key frames
  const int16_t *mvs = nullptr; const uint8_t *skip = nullptr;          // P frames
};
// one tile's range-coded payload appended to e.out (e must be fresh); blocks in raster order inside the tile
void entropy_encode_tile(const FrameSyms &f, int tx, int ty, RangeEncoder &e, EntropyModels *final_models = nullptr);
// frame record = [log2 tile][varint size of every tile, raster order][tile payloads]
std::vector<uint8_t> entropy_assemble_frame(int tile, const std::vector<const uint8_t *> &tiles, const std::vector<size_t> &sizes);
// returns the coded record of one frame
std::vector<uint8_t> entropy_encode_frame(const FrameSyms &f);
// inverse: fills caller-provided arrays of the sizes implied by width/height/key; returns false on a corrupt stream
bool entropy_decode_frame(const uint8_t *data, size_t n, int width, int height, int key, int16_t *lev_y, int16_t *lev_u, int16_t *lev_v,
                          uint8_t *modes_y, uint8_t *modes_uv, int16_t *mvs, uint8_t *skip);
// codes `frames` (independent coder states) on `threads` host threads; out[i] = bytes of frames[i]
void entropy_encode_frames(const std::vector<FrameSyms> &frames, int threads, std::vector<std::vector<uint8_t>> *out);

}  // namespace av1mi_host
