// backend.cpp — the part of RunTranscode that replaces the FFmpeg child (internal/ffmpeg/transcode.go:194-203): raw frames in,
// an AV1 elementary stream out.  A THIN caller: closed-GOP orchestration, filter-parameter policy and PCIe plumbing live in
// libav1mi.so's GOP session (include/av1mi.h av1mi_gop_*), entropy coding + OBU packing in av1_bitstream.cpp on the host
// cores, so nothing about the encoder is decided here.
//
// Input is Y4M (4:2:0, 8- or 10-bit), from a file or from a stream ("-i -", a FIFO: y4m.hpp), because demux / H.264 decode stay
// FFmpeg's job (SURVEY.md §8b "Gap to flag"): any decoder process can pipe its frames in.
// Output, chosen by the file name: ".obu" = Section-5 low-overhead OBU stream (what `dav1d -i x.obu` / `aomdec --obu` read),
// ".ivf" = IVF, anything else (the reference's "<base>.av1-tmp.mkv") = Matroska with one V_AV1 video track (mux.cpp);
// audio / subtitle copy (transcode.go:134-137) needs a demuxer and is not done.
//
// `segments` closed GOPs of the file are coded in lockstep (the session's batch dimension); while the host codes the
// symbols of frame t the GPU already works on frames t + 1 and t + 2 (three batches in flight).
#include "backend.hpp"
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "../../include/av1mi.h"
#include "av1_bitstream.hpp"
#include "mux.hpp"
#include "y4m.hpp"

namespace av1mi_host {

void DescribeSessionFrame(const av1mi_gop_frame &fr, int seg, int width, int height, int bit_depth, SessionFrameDesc *d, int visible_width, int visible_height) {
  const av1mi_frame_params &p = fr.params;
  av1mi_obu_frame &f = d->f;
  memset(&f, 0, sizeof(f));
  f.width = width; f.height = height; f.bit_depth = bit_depth; f.frame_type = p.frame_type; f.base_q_idx = p.base_q_idx;
  f.visible_width = visible_width; f.visible_height = visible_height;
  for (int i = 0; i < 4; i++) f.lf_level[i] = p.lf_level[i];
  f.lf_sharpness = p.lf_sharpness; f.cdef_damping = p.cdef_damping; f.cdef_bits = 0; f.cdef_y[0] = p.cdef_y; f.cdef_uv[0] = p.cdef_uv;
  auto units = [&](int n) { const int u = (n + p.lr_unit_size / 2) / p.lr_unit_size; return u > 1 ? u : 1; };
  const int vw = visible_width ? visible_width : width, vh = visible_height ? visible_height : height;      // the units tile the TRUE frame
  const size_t uy = (size_t)units(vh) * units(vw), uc = (size_t)units((vh + 1) / 2) * units((vw + 1) / 2);
  d->lr_y.resize(uy * 8); d->lr_uv.resize(uc * 8);
  for (size_t i = 0; i < uy; i++) memcpy(&d->lr_y[i * 8], p.lr_unit_y, 8);
  for (size_t i = 0; i < uc; i++) memcpy(&d->lr_uv[i * 8], p.lr_unit_uv, 8);
  // restoration per plane: the policy's type where the encoder kept it ON for this segment's frame (fr.lr_on), NONE elsewhere
  const uint8_t *on = fr.lr_on ? fr.lr_on + (size_t)seg * 3 : nullptr;
  f.lr_type[0] = (!on || on[0]) ? p.lr_unit_y[0] : 0;
  f.lr_type[1] = (!on || on[1]) ? p.lr_unit_uv[0] : 0;
  f.lr_type[2] = (!on || on[2]) ? p.lr_unit_uv[0] : 0;
  f.lr_unit_shift = p.lr_unit_size == 64 ? 0 : p.lr_unit_size == 128 ? 1 : 2; f.lr_uv_shift = 0;
  f.lr_units[0] = d->lr_y.data(); f.lr_units[1] = f.lr_units[2] = d->lr_uv.data();
  f.tile_cols_log2 = f.tile_rows_log2 = -1;      // one superblock per tile: the independence the GPU pipeline's prediction assumes
  const size_t nb = fr.blocks_per_frame, o = (size_t)seg * nb;
  if (fr.y_mode) { f.y_mode = fr.y_mode + o; f.uv_mode = fr.uv_mode + o; }       // the symbols are absent when the tiles were coded on the GPU
  if (fr.mv) { f.mv = fr.mv + o * 2; f.skip = fr.skip + o; }
  if (fr.lev_y) { f.lev_y = fr.lev_y + o * 64; f.lev_u = fr.lev_u + o * 16; f.lev_v = fr.lev_v + o * 16; }     // absent when the tiles were coded on the GPU
}

// A key frame of a key_block_size 32 session (av1mi_gop_frame.key_block_size == 32): 32x32 blocks over the complete superblock rows,
// 8x8 blocks in a last partial row — written by the general block writer (av1_blockstream.cpp) from the session's symbols.
static bool Key32TemporalUnit(const av1mi_gop_frame &fr, int seg, const SessionFrameDesc &desc, int width, int height, bool with_sequence_header, int threads,
                              std::vector<uint8_t> *out, std::string *err) {
  if (!fr.y_mode || !fr.lev_y) { if (err) *err = "a key frame in 32x32 blocks needs its symbols (gpu_entropy 0)"; return false; }
  const int hA = (height / 64) * 64, mi_rows = height / 4, mi_cols = width / 4;
  const size_t ny = (size_t)width * height, nc = ny / 4;
  const uint8_t *my = fr.y_mode + (size_t)seg * fr.key_modes_stride, *muv = fr.uv_mode + (size_t)seg * fr.key_modes_stride;
  std::vector<int16_t> lev(ny + 2 * nc);
  memcpy(lev.data(), fr.lev_y + (size_t)seg * ny, ny * 2);
  memcpy(lev.data() + ny, fr.lev_u + (size_t)seg * nc, nc * 2);
  memcpy(lev.data() + ny + nc, fr.lev_v + (size_t)seg * nc, nc * 2);
  std::vector<uint8_t> parts;
  std::vector<av1mi_obu_block> blocks;
  auto block = [&](int r, int c, int bsize, int mode_y, int mode_uv, size_t oy, size_t oc) {
    av1mi_obu_block b;
    memset(&b, 0, sizeof(b));
    b.mi_row = (uint16_t)r; b.mi_col = (uint16_t)c; b.bsize = (uint8_t)bsize; b.y_mode = (uint8_t)mode_y; b.uv_mode = (uint8_t)mode_uv;
    b.tx_type_off = 0;                                   // every luma transform is DCT_DCT: one shared entry
    b.lev_off[0] = (uint32_t)oy; b.lev_off[1] = (uint32_t)(ny + oc); b.lev_off[2] = (uint32_t)(ny + nc + oc);
    blocks.push_back(b);
  };
  const int w32 = width / 32, w8 = width / 8;
  std::vector<size_t> starts;          // per tile (= superblock): its first block, its first partition symbol; the writer's threads start there
  for (int sr = 0; sr * 16 < mi_rows; sr++)
    for (int sc = 0; sc * 16 < mi_cols; sc++) {
      starts.push_back(blocks.size()); starts.push_back(parts.size());
      if (sr * 64 < hA) {                                // PARTITION_SPLIT at 64x64, four 32x32 blocks (two where the frame ends mid-superblock)
        parts.push_back(3);
        for (int k = 0; k < 4; k++) {
          const int r32 = sr * 2 + (k >> 1), c32 = sc * 2 + (k & 1);
          if (c32 >= w32) continue;
          const size_t i = (size_t)r32 * w32 + c32;
          parts.push_back(0);
          block(r32 * 8, c32 * 8, 9 /* BLOCK_32X32 */, my[i], muv[i], i * 1024, i * 256);
        }
      } else {                                           // the last, partial superblock row: split down to 8x8 wherever the frame reaches
        const size_t oyB = (size_t)hA * width, ocB = oyB / 4;
        const uint8_t *myB = my + fr.key_modes_band, *muvB = muv + fr.key_modes_band;
        for (int k = 0; k < 64; k++) {                   // z-order over the superblock's 8x8 blocks; a level's symbol precedes its first block
          int bx = 0, by = 0;
          for (int i = 0; i < 3; i++) { bx |= ((k >> (2 * i)) & 1) << i; by |= ((k >> (2 * i + 1)) & 1) << i; }
          const int r = sr * 16 + by * 2, c = sc * 16 + bx * 2;
          for (int n8 = 8; n8 >= 2; n8 >>= 1)            // 64, 32, 16: PARTITION_SPLIT at every level that starts here, inside the frame
            if (!(bx & (n8 - 1)) && !(by & (n8 - 1)) && r < mi_rows && c < mi_cols) parts.push_back(3);
          if (r >= mi_rows || c >= mi_cols) continue;
          parts.push_back(0);
          const size_t i = (size_t)(r / 2 - hA / 8) * w8 + c / 2;
          block(r, c, 3 /* BLOCK_8X8 */, myB[i], muvB[i], oyB + i * 64, ocB + i * 16);
        }
      }
    }
  av1mi_obu_blocks d;
  memset(&d, 0, sizeof(d));
  d.hdr = desc.f;
  const uint8_t dct = 0;
  d.partition = parts.data(); d.n_partition = parts.size(); d.blocks = blocks.data(); d.n_blocks = blocks.size(); d.tx_type = &dct; d.levels = lev.data();
  starts.push_back(blocks.size()); starts.push_back(parts.size());
  std::string werr;
  if (!av1::blocks_temporal_unit(d, with_sequence_header, out, &werr, threads, reinterpret_cast<const size_t (*)[2]>(starts.data()))) {
    if (err) *err = "bitstream writer: " + werr;
    return false;
  }
  return true;
}

bool SessionTemporalUnit(const av1mi_gop_frame &fr, int seg, int width, int height, int bit_depth, int visible_width, int visible_height,
                         bool with_sequence_header, int threads, std::vector<uint8_t> *out, std::string *err) {
  SessionFrameDesc desc;
  DescribeSessionFrame(fr, seg, width, height, bit_depth, &desc, visible_width, visible_height);
  std::string werr;
  if (fr.tile_size) {      // tiles coded on the GPU: frame header + tile group around them
    const uint32_t *sz = fr.tile_size + (size_t)seg * fr.tiles_per_frame;
    size_t off = 0;
    for (size_t i = 0; i < (size_t)seg * fr.tiles_per_frame; i++) off += fr.tile_size[i];
    std::vector<uint8_t> frame;
    if (!av1::frame_obu_from_tiles(desc.f, fr.tile_payload + off, sz, fr.tiles_per_frame, &frame, &werr)) { if (err) *err = "bitstream assembly: " + werr; return false; }
    *out = av1::temporal_delimiter_obu();
    if (with_sequence_header) { const std::vector<uint8_t> sh = av1::sequence_header_obu(av1::sequence_params(desc.f)); out->insert(out->end(), sh.begin(), sh.end()); }
    out->insert(out->end(), frame.begin(), frame.end());
    return true;
  }
  if (fr.key_block_size == 32) return Key32TemporalUnit(fr, seg, desc, width, height, with_sequence_header, threads, out, err);      // symbols of a key frame in 32x32 blocks
  if (!av1::temporal_unit(desc.f, with_sequence_header, threads, out, &werr)) { if (err) *err = "bitstream writer: " + werr; return false; }
  return true;
}

int RunBackend(const BackendJob &job, std::string *err) {
  av1mi_ctx *ctx = nullptr;
  if (av1mi_device_count() <= 0 || av1mi_open(job.device, &ctx) != AV1MI_OK) {
    *err = "Error: no usable HIP device for the av1mi backend (device " + std::to_string(job.device) + ")";
    return -1;
  }
  Y4mSource y;
  av1mi_gop *gop = nullptr;
  StreamSink sink;
  int code = 0;
#define CHK(call)                                                                                   \
  do { int rc_ = (call); if (rc_ != AV1MI_OK) { *err = std::string(#call) + ": " + av1mi_last_error(ctx); code = 2; goto done; } } while (0)
  if (!y.open(job.input, err)) { code = 1; goto done; }
  {
    const int G = job.gop, w = (y.w + 7) & ~7, h = (y.h + 7) & ~7;       // the coded size; y.w x y.h is what a decoder outputs
    int S = std::max(job.segments, 1);
    if (y.known_frames() >= 0) S = (int)std::max<long>(1, std::min<long>(S, (y.known_frames() + G - 1) / G));      // no more segments than the file has GOPs
    long total_frames = 0;
    const int threads = job.threads > 0 ? job.threads : (int)std::max(1u, std::thread::hardware_concurrency());
    const size_t bps = y.bd == 8 ? 1 : 2, fy = (size_t)w * h * bps, fc = fy / 4;
    av1mi_gop_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    if (w != y.w || h != y.h) { cfg.visible_width = y.w; cfg.visible_height = y.h; }
    cfg.width = w; cfg.height = h; cfg.bit_depth = y.bd; cfg.base_q_idx = job.quality < 1 ? 1 : job.quality; cfg.gop_length = G; cfg.segments = S;
    cfg.search_range = 8;
    cfg.gpu_entropy = job.gpu_entropy ? 1 : 0;
    // key frames in 32x32 blocks where the frame allows it (the coded width a multiple of 32)
    cfg.key_block_size = (job.key_block_size == 32 && (w & 31) == 0) ? 32 : 8;
    CHK(av1mi_gop_open(ctx, &cfg, &gop));
    av1::SequenceParams sp; sp.width = y.w; sp.height = y.h; sp.bit_depth = y.bd;
    for (const std::string &side : job.tracks)
      if (!sink.add_side_file(side, err)) { code = 1; goto done; }
    if (!sink.open(job.output, sp, y.fps_n, y.fps_d, err)) { code = 1; goto done; }
    std::vector<std::vector<std::vector<uint8_t>>> units((size_t)S);   // [segment][frame] temporal units of the batch in flight
    const int lag = av1mi_gop_max_in_flight() - 1;      // batches the GPU holds while the host works on the oldest
    for (long g0 = 0;; g0 += S) {
      for (auto &u : units) u.clear();
      // the next GROUP of S GOPs: a file is read in place, a stream one group ahead of the encoder (y4m.hpp)
      const long have_frames = y.prepare(g0 * G, (long)S * G, err);
      if (have_frames < 0) { code = 1; goto done; }
      if (have_frames == 0) break;
      total_frames += have_frames;
      // frames of this group that exist: segment s, position t -> frame s * G + t of the group
      auto exists = [&](int s, int t) { return (long)s * G + t < have_frames; };
      int T = 0;
      for (int t = 0; t < G; t++) if (exists(0, t)) T = t + 1;
      av1mi_gop_frame fr;
      auto collect_oldest = [&]() -> bool {
        if (av1mi_gop_collect(gop, &fr) != AV1MI_OK) { *err = std::string("av1mi_gop_collect: ") + av1mi_last_error(ctx); return false; }
        return true;
      };
      auto assemble = [&](int t) -> bool {         // the collected batch t -> temporal units (frame header + tile group, or the host coder)
        for (int s = 0; s < S; s++) {
          if (!exists(s, t)) continue;
          std::vector<uint8_t> tu;
          if (!SessionTemporalUnit(fr, s, w, h, y.bd, cfg.visible_width, cfg.visible_height, t == 0, threads, &tu, err)) return false;
          units[(size_t)s].push_back(std::move(tu));
        }
        return true;
      };
      // The frames of batch t + 1 are read (one thread per segment) WHILE the host assembles batch t - lag: the session hands out
      // the next input buffers as soon as the oldest batch has been collected.
      struct Reads {
        std::vector<std::thread> th; std::vector<char> ok;
        bool join() { for (auto &x : th) if (x.joinable()) x.join(); th.clear(); for (char c : ok) if (!c) return false; return true; }
        ~Reads() { for (auto &x : th) if (x.joinable()) x.join(); }
      } reads;
      auto start_reads = [&](int t) -> bool {
        void *py, *pu, *pv;
        if (av1mi_gop_acquire_input(gop, &py, &pu, &pv) != AV1MI_OK) { *err = std::string("av1mi_gop_acquire_input: ") + av1mi_last_error(ctx); return false; }
        reads.ok.assign((size_t)S, 1);
        for (int s = 0; s < S; s++) {
          if (!exists(s, t)) {
            // a shorter last GOP / fewer GOPs than segments: the slot is coded (the batch is one launch) and its output dropped.  Flat
            // planes, not whatever the pinned buffer held: stale pixels could cost the GPU coder's tile capacity for the whole batch
            memset((unsigned char *)py + fy * s, 0, fy); memset((unsigned char *)pu + fc * s, 0, fc); memset((unsigned char *)pv + fc * s, 0, fc);
            continue;
          }
          reads.th.emplace_back([&, s, t, py, pu, pv]() {
            reads.ok[(size_t)s] = y.read((long)s * G + t, w, h, (unsigned char *)py + fy * s, (unsigned char *)pu + fc * s, (unsigned char *)pv + fc * s);
          });
        }
        return true;
      };
      if (T > 0 && !start_reads(0)) { code = 2; goto done; }
      for (int t = 0; t < T; t++) {
        if (!reads.join()) { *err = job.input + ": Invalid data found when processing input (truncated frame)"; code = 1; goto done; }
        CHK(av1mi_gop_submit(gop, t == 0 ? 0 : 1));
        const bool have = t >= lag;
        if (have && !collect_oldest()) { code = 2; goto done; }            // the GPU works on the frames after it meanwhile
        if (t + 1 < T && !start_reads(t + 1)) { code = 2; goto done; }
        if (have && !assemble(t - lag)) { code = 2; goto done; }
      }
      for (int t = std::max(0, T - lag); t < T; t++)
        if (!collect_oldest() || !assemble(t)) { code = 2; goto done; }
      for (int s = 0; s < S; s++)
        for (size_t t = 0; t < units[(size_t)s].size(); t++)
          if (!sink.write(units[(size_t)s][t], t == 0, err)) { code = 1; goto done; }
    }
    if (total_frames == 0) { *err = job.input + ": Invalid data found when processing input (no frames)"; code = 1; goto done; }
    if (!sink.close(err)) { code = 1; goto done; }
  }
done:
  sink.abort();
  if (gop) av1mi_gop_close(gop);
  y.close();
  av1mi_close(ctx);
  return code;
#undef CHK
}

}  // namespace av1mi_host
