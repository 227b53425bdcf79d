// coder_selftest.cpp — the op-stream range coder (csrc/av1_ops.hpp: lazy byte output, a carry into bytes already stored happens
// about once in 2^16 flushes) against the host writer's eager one, on millions of random interval updates.
// The header is compiled a second time here, inside a namespace of its own and with a carry counter switched on, so the test
// can tell that the rare path really ran.  Test hook only.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "av1_bitstream.hpp"
#define AV1_CODER_STATS
namespace coder_selftest {
#include "../csrc/av1_ops.hpp"
}

extern "C" long long av1mi_host_coder_selftest(uint64_t seed, long long count, long long *carries) {
  using coder_selftest::av1ops::Coder;
  std::vector<uint32_t> fl((size_t)count), fh((size_t)count);
  std::vector<uint8_t> sy((size_t)count), ns((size_t)count);
  uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 16); };
  for (long long i = 0; i < count; i++) {
    const int n = 2 + (int)(rnd() % 15);
    int s;
    uint32_t a, b;       // inverse CDF values around the symbol: a = icdf[s - 1] > b = icdf[s], multiples of 64 apart at least
    s = (int)(rnd() % (uint32_t)n);
    const uint32_t hi = 32768u - 64u * (uint32_t)s, lo = 64u * (uint32_t)(n - 1 - s);
    a = s == 0 ? 32768u : lo + 64u + rnd() % (hi - lo - 63u);
    b = s == n - 1 ? 0u : lo + rnd() % (a - 63u - lo);
    fl[(size_t)i] = a; fh[(size_t)i] = b; sy[(size_t)i] = (uint8_t)s; ns[(size_t)i] = (uint8_t)n;
  }
  const std::vector<uint8_t> ref = av1mi_host::av1::range_code_raw(fl.data(), fh.data(), sy.data(), ns.data(), (size_t)count);
  std::vector<uint8_t> out((size_t)count * 2 + 64);
  coder_selftest::av1ops::coder_stat_carries = 0;
  Coder c;
  uint16_t stage[Coder::kStage];
  c.init(out.data(), (int)out.size(), stage);
  for (long long i = 0; i < count; i++) {
    c.encode(fl[(size_t)i], fh[(size_t)i], sy[(size_t)i], ns[(size_t)i]);
    if (c.stage_full()) c.spill();
  }
  const int sz = c.finish();
  if (carries) *carries = coder_selftest::av1ops::coder_stat_carries;
  if (sz < 0 || (size_t)sz != ref.size()) return -1;
  for (size_t i = 0; i < ref.size(); i++) if (out[i] != ref[i]) return (long long)i + 1;
  return 0;
}
