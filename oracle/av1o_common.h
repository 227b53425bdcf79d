/*
 * av1o_common.h — shared declarations of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle.so.  The product (libav1mi.so) never links or calls it.
 *
 * PARITY: the reference (IONIQ6000/av1-go) holds no codec arithmetic, no tests
 * and no golden vectors for this path: by the
 * reference's own fixtures this oracle is "parity unpinned", and no reference
 * build exists to compare with (the arithmetic lives in an un-vendored FFmpeg 8.x
 * "latest" -> av1_vaapi -> Intel hardware; internal/ffmpeg/transcode.go:120,195,
 * internal/config/config.go:33).  What pins it instead is a conformant third-party
 * DECODER: every normative function here (dequantiser, inverse transforms of all
 * 19 sizes x 16 types, intra prediction of every block size, motion compensation
 * of every block size x 4 filters, deblocking with every filter length, CDEF,
 * loop restoration) is compared bit for bit with dav1d 1.5.3 on arbitrary
 * symbols (tests/test_av1_conformance.py, tests/test_av1_blocks.py).  The
 * non-normative encoder side (forward transforms, quantiser rounding, searches)
 * has no external pin; it is checked by round-trip properties.  The oracle
 * restates the AV1 decoding process (AV1 Bitstream & Decoding Process
 * Specification §7.11-7.17) and the matching libaom C functions from knowledge;
 * neither text is in the container, so every function names the spec section /
 * libaom function it restates instead of a reference file:line.
 */
#ifndef AV1O_COMMON_H
#define AV1O_COMMON_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* TX sizes, AV1 spec §6.10.19 / libaom TX_SIZE order. */
enum {
  TX_4X4, TX_8X8, TX_16X16, TX_32X32, TX_64X64,
  TX_4X8, TX_8X4, TX_8X16, TX_16X8, TX_16X32, TX_32X16, TX_32X64, TX_64X32,
  TX_4X16, TX_16X4, TX_8X32, TX_32X8, TX_16X64, TX_64X16, TX_SIZES_ALL
};
/* TX types, AV1 spec §6.10.19 / libaom TX_TYPE order. Name = <column(vertical)>_<row(horizontal)>. */
enum {
  DCT_DCT, ADST_DCT, DCT_ADST, ADST_ADST, FLIPADST_DCT, DCT_FLIPADST,
  FLIPADST_FLIPADST, ADST_FLIPADST, FLIPADST_ADST, IDTX, V_DCT, H_DCT,
  V_ADST, H_ADST, V_FLIPADST, H_FLIPADST, TX_TYPES
};
enum { T1D_DCT, T1D_ADST, T1D_FLIPADST, T1D_IDTX };
#define AV1O_WHT_WHT 16   /* not a TX_TYPE of the spec: selects the lossless 4x4 Walsh-Hadamard path */

extern const int av1o_tx_w[TX_SIZES_ALL];
extern const int av1o_tx_h[TX_SIZES_ALL];

static inline int av1o_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int av1o_round2(int x, int n) { return n == 0 ? x : (x + (1 << (n - 1))) >> n; }
static inline int64_t av1o_round2_64(int64_t x, int n) { return n == 0 ? x : (x + ((int64_t)1 << (n - 1))) >> n; }
static inline int av1o_msb(unsigned v) { int r = -1; while (v) { v >>= 1; r++; } return r; }

#ifdef __cplusplus
}
#endif
#endif
