// av1_opstream.cpp — the CPU twin of the GPU tile entropy coder: the three-stage formulation of csrc/av1_ops.hpp (tokenize per
// block with all contexts from neighbour data; one adaptation chain per CDF slot; one serial range-coder pass per tile) run on
// the host, so that its logic is checked byte for byte against the block-sequential writer of av1_bitstream.cpp on every
// machine, GPU or not (tests/test_av1_opstream.py).  Also the reference the GPU kernels' output is compared with.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../csrc/av1_ops_cdfs.hpp"
#include "av1_bitstream.hpp"

namespace av1mi_host {
namespace av1 {

// true when the description stays inside the GPU coder's tool set (what the GPU block pipeline produces)
bool opstream_supported(const av1mi_obu_frame &f, std::string *why) {
  auto no = [&](const char *m) { if (why) *why = m; return false; };
  if (f.cdef_bits || f.reduced_tx_set || f.disable_cdf_update) return no("cdef_bits / reduced_tx_set / disable_cdf_update must be 0");
  if (f.tile_cols_log2 >= 0 || f.tile_rows_log2 >= 0) return no("one superblock per tile only");
  if (f.angle_y || f.angle_uv || f.cfl_alpha || f.tx_type || f.is_inter) return no("angle deltas, chroma from luma, transform types and intra blocks in inter frames are host-writer only");
  if (f.frame_type == 0 && f.skip) return no("key frames are coded with skip = 0");
  for (int p = 0; p < 3; p++) if (f.lr_type[p] > 1) return no("Wiener restoration only");
  if (f.lr_unit_shift || f.lr_uv_shift) return no("64x64 restoration units only");
  return true;
}

// tile payloads (range-coded, finished) of a frame through tokenize + code; tiles in raster order
// key_rows32 > 0: a key frame whose first key_rows32 luma rows (whole superblock rows) are coded in 32x32 blocks (av1_ops32.hpp; the
// symbol arrays in the session's layout: those rows' modes one per 32x32 block from entry 0, levels block-contiguous over the 32x32
// grid; the rows below in the 8x8 layout at their usual places)
bool opstream_tiles(const av1mi_obu_frame &f, std::vector<std::vector<uint8_t>> *tiles, std::string *err, int key_rows32) {
  using namespace av1ops;
  if (!opstream_supported(f, err)) return false;
  if (key_rows32 && (f.frame_type != 0 || (key_rows32 & 63) || key_rows32 > f.height || (f.width & 31))) { if (err) *err = "bad 32x32 band"; return false; }
  FrameView v;
  memset(&v, 0, sizeof(v));
  v.w8 = f.width / 8; v.h8 = f.height / 8; v.key = f.frame_type == 0;
  v.y_mode = f.y_mode; v.uv_mode = f.uv_mode; v.mv = f.mv; v.skip = f.skip; v.lev_y = f.lev_y; v.lev_u = f.lev_u; v.lev_v = f.lev_v;
  std::vector<uint8_t> zskip;
  if (!v.key && !v.skip) { zskip.assign((size_t)v.w8 * v.h8, 0); v.skip = zskip.data(); }
  for (int p = 0; p < 3; p++) {
    v.lr_on[p] = f.lr_type[p] == 1;
    const int ph = p ? (visible_height(f) + 1) >> 1 : visible_height(f), pw = p ? (visible_width(f) + 1) >> 1 : visible_width(f);
    v.lr_rows[p] = std::max((ph + 32) / 64, 1); v.lr_cols[p] = std::max((pw + 32) / 64, 1);
  }
  // the op-stream coder takes ONE unit record per plane class (what the session's policy produces): check and copy it
  for (int p = 0; p < 3; p++) {
    if (!v.lr_on[p]) continue;
    const int8_t *u = f.lr_units[p];
    const size_t n = (size_t)v.lr_rows[p] * v.lr_cols[p];
    for (size_t i = 1; i < n; i++) if (memcmp(u, u + i * 8, 8)) { if (err) *err = "restoration units must be uniform per plane"; return false; }
    if (p == 2 && v.lr_on[1] && memcmp(u, f.lr_units[1], 8)) { if (err) *err = "U and V restoration units must be equal"; return false; }
    memcpy(v.lr_unit[p ? 1 : 0], u, 8);
  }
  const size_t nb = (size_t)v.w8 * v.h8;
  std::vector<BlockInfo> info(nb);
  v.info = info.data();
  for (size_t b = 0; b < nb; b++) { memset(&info[b], 0, sizeof(BlockInfo)); block_summary(v, (int)b, &info[b]); }
  if (!v.key) for (int r = 0; r < v.h8; r++) for (int c = 0; c < v.w8; c++) inter_mode_decision(v, r, c, &info[(size_t)r * v.w8 + c]);
  const int qcat = f.base_q_idx <= 20 ? 0 : f.base_q_idx <= 60 ? 1 : f.base_q_idx <= 120 ? 2 : 3;
  SlotTable tab;
  const std::vector<uint16_t> image = default_slot_image(v.key != 0, qcat, &tab);
  const int sbr_n = (v.h8 + 7) / 8, sbc_n = (v.w8 + 7) / 8;
  tiles->assign((size_t)sbr_n * sbc_n, {});
  const int nslots = v.key ? S_KEY_END : S_INTER_END;
  std::vector<op_t> list;
  std::vector<uint32_t> grouped;
  std::vector<uint8_t> cnt((size_t)S_MAX * kBlocksPerTile);
  std::vector<uint16_t> pos((size_t)S_MAX * kBlocksPerTile), rec((size_t)kBlocksPerTile * kBlockRecords);
  ScanTables scan;
  fill_scan_tables(&scan);
  alignas(16) uint8_t mag[kMagBytes];
  const TokScratch ts = { mag, &scan };
  // AV1MI_TOK_STATS=1 (diagnostic): the capacities a frame would need — records per block, symbols of one slot in one block, list
  // words per tile — to stderr; tiles over capacity are skipped instead of failing the call
  const bool stats = getenv("AV1MI_TOK_STATS") != nullptr;
  int st_rec = 0, st_cnt = 0, st_ops = 0, st_over = 0; long st_ops_sum = 0;
  // the 32x32 band's own slot table, default CDFs, scan tables and scratch
  SlotTable tab32;
  const std::vector<uint16_t> image32 = key_rows32 ? default_slot_image_k32(qcat, &tab32) : std::vector<uint16_t>();
  std::vector<uint16_t> rec32(key_rows32 ? (size_t)kBlocksPerTile * kBlockRecords : 0), cnt32((size_t)K_END * kBlocks32);
  std::vector<ScanTables32> scan32(key_rows32 ? 1 : 0);
  if (key_rows32) fill_scan_tables32(scan32.data());
  alignas(16) uint8_t mag32[kMag32Bytes];
  for (int sbr = 0; sbr < sbr_n; sbr++)
    for (int sbc = 0; sbc < sbc_n; sbc++) {
      if (sbr * 64 < key_rows32) {
        // a tile of the 32x32 band: tokenized serially (the GPU: one lane per tile), then the same chains and the same range coder
        std::fill(cnt32.begin(), cnt32.end(), 0);
        const TokScratch32 ts32 = { mag32, scan32.data() };
        Sum32 sums[kBlocks32];
        const bool half = sbc * 8 + 4 >= v.w8;      // width % 64 == 32: the superblock's right blocks lie outside the frame
        for (int b = 0; b < kBlocks32; b++) if (!(half && (b & 1))) block_sums32(v, block_index32(v, sbr, sbc, b), &sums[b]);
        int nrec[kBlocks32], first[kBlocks32 + 1];
        first[0] = 0;
        for (int b = 0; b < kBlocks32; b++) {        // (the GPU: one lane per block)
          Sink32 k = { &rec32[(size_t)b * kBlockRecords32], cnt32.data(), b, (int)kBlockRecords32, 0, 0, false, 0, 0, 0, 0 };
          tok_block32(v, k, ts32, sbr, sbc, b, sums);
          if (k.overflow) { if (err) *err = "a block exceeds the tokenizer's record area"; return false; }
          nrec[b] = k.nrec; first[b + 1] = first[b] + k.n;
        }
        for (int b = 0; b < kBlocks32; b++)
          if (!count_block32(&rec32[(size_t)b * kBlockRecords32], nrec[b], cnt32.data(), b)) { if (err) *err = "too many symbols of one slot in a block"; return false; }
        uint16_t base[K_END], total[K_END];
        const int run = place_tile32(cnt32.data(), total, base);
        if (run > 65535) { if (err) *err = "tile too large for 16-bit entry positions"; return false; }
        const int nops = first[kBlocks32];
        list.assign((size_t)nops, 0);
        grouped.assign((size_t)run, 0);
        for (int b = 0; b < kBlocks32; b++) replay_block32(&rec32[(size_t)b * kBlockRecords32], nrec[b], cnt32.data(), b, first[b], list.data(), grouped.data());
        for (int sl = 0; sl < K_END; sl++)
          if (total[sl]) run_chain(&image32[tab32.off[sl]], tab32.nsym[sl], &grouped[(size_t)base[sl]], total[sl], list.data());
        std::vector<uint8_t> &out = (*tiles)[(size_t)sbr * sbc_n + sbc];
        out.resize((size_t)nops * 2 + 64);
        Coder c;
        uint16_t stage[Coder::kStage];
        c.init(out.data(), (int)out.size(), stage);
        for (int i = 0; i < nops; i++) code_word(c, list[(size_t)i]);
        const int n = c.finish();
        if (n < 0) { if (err) *err = "tile payload overflow"; return false; }
        out.resize((size_t)n);
        continue;
      }
      // stage 1, tokenize (the GPU: one thread per block): records + counts, place, replay
      std::fill(cnt.begin(), cnt.end(), 0);
      int first[kBlocksPerTile + 1], nrec[kBlocksPerTile];
      first[0] = 0;
      bool over = false;
      for (int zi = 0; zi < kBlocksPerTile; zi++) {
        Sink k = { &rec[(size_t)zi * kBlockRecords], cnt.data(), zi, 0, 0, false };
        tok_block(v, k, ts, sbr, sbc, zi);
        if (k.overflow && !stats) { if (err) *err = "a block exceeds the tokenizer's record area"; return false; }
        over |= k.overflow;
        nrec[zi] = k.nrec;
        first[zi + 1] = first[zi] + k.n;
        if (k.nrec > st_rec) st_rec = k.nrec;
      }
      if (stats) {
        for (uint8_t c : cnt) if (c > st_cnt) st_cnt = c;
        if (first[kBlocksPerTile] > st_ops) st_ops = first[kBlocksPerTile];
        st_ops_sum += first[kBlocksPerTile];
        if (over) { st_over++; continue; }
      }
      const int nops = first[kBlocksPerTile];
      int base[S_MAX], total[S_MAX], run = 0;
      for (int sl = 0; sl < nslots; sl++) {
        base[sl] = run;
        total[sl] = group_positions(&cnt[(size_t)sl * kBlocksPerTile], &pos[(size_t)sl * kBlocksPerTile], run);
        run = (run + total[sl] + kListAlign - 1) & ~(kListAlign - 1);
      }
      if (run > 65535) { if (err) *err = "tile too large for 16-bit entry positions"; return false; }
      list.assign((size_t)nops, 0);
      grouped.assign((size_t)run, 0);
      for (int zi = 0; zi < kBlocksPerTile; zi++)
        replay_block(&rec[(size_t)zi * kBlockRecords], nrec[zi], pos.data(), zi, first[zi], list.data(), grouped.data());
      // stage 2, one chain per slot (the GPU: the tile's threads take the slots, longest first)
      for (int sl = 0; sl < nslots; sl++)
        if (total[sl]) run_chain(&image[tab.off[sl]], tab.nsym[sl], &grouped[(size_t)base[sl]], total[sl], list.data());
      // stage 3, the serial range coder over the finished list (the GPU: one lane per tile)
      std::vector<uint8_t> &out = (*tiles)[(size_t)sbr * sbc_n + sbc];
      out.resize((size_t)nops * 2 + 64);       // a step adds at most 15 bits
      Coder c;
      uint16_t stage[Coder::kStage];
      c.init(out.data(), (int)out.size(), stage);
      for (int i = 0; i < nops; i++) code_word(c, list[(size_t)i]);
      const int n = c.finish();
      if (n < 0) { if (err) *err = "tile payload overflow"; return false; }
      out.resize((size_t)n);
    }
  if (stats)
    fprintf(stderr, "[av1mi tok stats] %d tiles: records per block <= %d (capacity %d), symbols of a slot in a block <= %d (255), list words per tile <= %d, mean %.0f; %d tiles over a block capacity\n",
            sbr_n * sbc_n, st_rec, (int)kBlockRecords, st_cnt, st_ops, (double)st_ops_sum / (sbr_n * sbc_n), st_over);
  return true;
}

}  // namespace av1
}  // namespace av1mi_host
