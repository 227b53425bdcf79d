// backend.cpp — drives libav1mi.so over closed-GOP segments of a Y4M file (4:2:0, 8- or 10-bit).
//
// Input is raw video because demux / H.264 decode stay FFmpeg's job (SURVEY.md §8b "Gap to flag"); the output is NOT an
// AV1 bitstream: the container written here ("AV1MI2") holds, per closed-GOP segment, one record per frame: the frame type
// and the frame's symbols (modes or vectors + skip flags, quantised levels) in the syntax of entropy.hpp (SURVEY.md §8a row
// H1: AV1's range-coder arithmetic and CDF adaptation over this project's own symbols, one coder state per 64x64 tile).
// The record is produced on the GPU by the tile entropy coder (K9, av1mi_entropy_encode): only coded bytes cross PCIe;
// entropy.cpp holds the same coder for host threads (byte-identical, tests/test_entropy.py) and the decoder.
// It exists so that the job contract (output file present, size gate, atomic replace) can be exercised end to end through
// every kernel K1-K8 and H1; OBU packing and AV1's default CDFs are not built.
#include "backend.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/av1mi.h"
#include "entropy.hpp"

namespace av1mi_host {
namespace {

struct Y4m { FILE *f = nullptr; int w = 0, h = 0, bd = 8, fps_n = 30, fps_d = 1; };

bool y4m_open(const std::string &path, Y4m *y, std::string *err) {
  y->f = fopen(path.c_str(), "rb");
  if (!y->f) { *err = path + ": No such file or directory"; return false; }
  char hdr[512];
  if (!fgets(hdr, sizeof(hdr), y->f) || strncmp(hdr, "YUV4MPEG2", 9)) { *err = path + ": Invalid data found when processing input (not Y4M)"; return false; }
  std::string cs = "420jpeg";
  for (char *t = strtok(hdr, " \n"); t; t = strtok(nullptr, " \n")) {
    if (t[0] == 'W') y->w = atoi(t + 1);
    else if (t[0] == 'H') y->h = atoi(t + 1);
    else if (t[0] == 'F') sscanf(t + 1, "%d:%d", &y->fps_n, &y->fps_d);
    else if (t[0] == 'C') cs = t + 1;
  }
  if (cs.rfind("420p10", 0) == 0) y->bd = 10;
  else if (cs.rfind("420", 0) == 0 && cs.find("p1") == std::string::npos) y->bd = 8;
  else { *err = "Invalid argument: unsupported Y4M colourspace " + cs + " (4:2:0 8/10-bit only)"; return false; }
  if (y->w <= 0 || y->h <= 0 || (y->w & 7) || (y->h & 7)) { *err = "Invalid argument: frame size must be a multiple of 8"; return false; }
  return true;
}
// reads one frame into planes; returns 1 ok, 0 eof, -1 error
int y4m_frame(Y4m *y, unsigned char *Y, unsigned char *U, unsigned char *V) {
  char line[128];
  if (!fgets(line, sizeof(line), y->f)) return 0;
  if (strncmp(line, "FRAME", 5)) return -1;
  const size_t bps = y->bd == 8 ? 1 : 2, ny = (size_t)y->w * y->h * bps, nc = ny / 4;
  if (fread(Y, 1, ny, y->f) != ny || fread(U, 1, nc, y->f) != nc || fread(V, 1, nc, y->f) != nc) return -1;
  return 1;
}
void put_varint(std::vector<unsigned char> &o, unsigned v) { while (v >= 128) { o.push_back((unsigned char)(v | 128)); v >>= 7; } o.push_back((unsigned char)v); }
#define CHK(call)                                                                        \
  do { int rc_ = (call); if (rc_ != AV1MI_OK) { *err = std::string(#call) + ": " + av1mi_last_error(ctx); code = 2; goto done; } } while (0)

}  // namespace

// encoder policies shared with av1-go_amd/pipeline.py (non-normative): deblock level / CDEF strengths from the AC step
static int lf_level_from_q(int ac_q, int bd, bool key) {
  long g;
  if (bd == 8) g = key ? ((long)ac_q * 17563 - 421574 + (1 << 17)) >> 18 : ((long)ac_q * 6017 + 650707 + (1 << 17)) >> 18;
  else g = ((long)ac_q * 20723 + 4060632 + (1 << 19)) >> 20;
  return (int)(g < 0 ? 0 : g > 63 ? 63 : g);
}
static void cdef_strength_from_q(int ac_q, int bd, uint8_t st[4]) {
  const int q = ac_q >> (bd - 8);
  int y = q < 700 ? (q * q * 3 + 32768) >> 16 : 15;
  y = y > 15 ? 15 : y;
  st[0] = (uint8_t)(y + 2 > 15 ? 15 : (y + 2 < 1 ? 1 : y + 2)); st[1] = 1; st[2] = (uint8_t)(y < 1 ? 1 : y); st[3] = 1;
}

int RunBackend(const BackendJob &job, std::string *err) {
  av1mi_ctx *ctx = nullptr;
  if (av1mi_device_count() <= 0 || av1mi_open(job.device, &ctx) != AV1MI_OK) {
    *err = "Error: no usable HIP device for the av1mi backend (device " + std::to_string(job.device) + ")";
    return -1;
  }
  // One closed GOP (segment) at a time: frame 0 is a key frame, frames 1.. are P frames predicted from the previous
  // reconstructed frame after deblocking + CDEF + loop restoration.  Frames of a segment are uploaded once.
  Y4m y;
  int code = 0;
  FILE *out = nullptr;
  std::vector<unsigned char> hY, hU, hV, packed, bytes;
  std::vector<unsigned char> rec;
  enum { SY, SU, SV, RY, RU, RV, DY, DU, DV, CY, CU, CV, OY, OU, OV, LY, LU, LV, MY, MUV, MVS, SKIP, ZSKIP, MIY, MIC, CSB, LRY, LRC, ENT, EOFF, NBUF };
  void *d[NBUF] = { nullptr };
  long frames_total = 0;
  if (!y4m_open(job.input, &y, err)) { code = 1; goto done; }
  {
    const size_t bps = y.bd == 8 ? 1 : 2;
    const int G = job.gop, w = y.w, h = y.h;
    const size_t ny = (size_t)w * h, nc = ny / 4, nb = ny / 64;
    const int ac_q = av1mi_ac_q(job.quality, y.bd);
    const int nsb = ((w + 63) / 64) * ((h + 63) / 64);
    const int ury = (h + 32) / 64 > 1 ? (h + 32) / 64 : 1, ucy = (w + 32) / 64 > 1 ? (w + 32) / 64 : 1;
    const int urc = (h / 2 + 32) / 64 > 1 ? (h / 2 + 32) / 64 : 1, ucc = (w / 2 + 32) / 64 > 1 ? (w / 2 + 32) / 64 : 1;
    hY.resize(ny * bps * G); hU.resize(nc * bps * G); hV.resize(nc * bps * G);
    const size_t ent_cap = ny * 3 + 65536;      // the raw int16 size of a frame's levels: a coded frame stays far below
    size_t sizes[NBUF];
    for (int i = SY; i <= SV; i++) sizes[i] = (i == SY ? ny : nc) * bps * G;
    for (int i = RY; i <= OV; i++) sizes[i] = ((i - RY) % 3 == 0 ? ny : nc) * bps;
    sizes[LY] = ny * 2; sizes[LU] = sizes[LV] = nc * 2; sizes[MY] = sizes[MUV] = sizes[SKIP] = sizes[ZSKIP] = nb; sizes[MVS] = nb * 4;
    sizes[ENT] = ent_cap; sizes[EOFF] = 16;
    sizes[MIY] = (ny / 16) * 4; sizes[MIC] = (nc / 16) * 4; sizes[CSB] = (size_t)nsb * 4; sizes[LRY] = (size_t)ury * ucy * 8; sizes[LRC] = (size_t)urc * ucc * 8;
    for (int i = 0; i < NBUF; i++) CHK(av1mi_malloc(ctx, &d[i], sizes[i]));
    CHK(av1mi_memset(ctx, d[ZSKIP], 0, nb));
    {  // constant side information of this job
      std::vector<uint32_t> mi(ny / 16);
      std::vector<uint8_t> sb((size_t)nsb * 4);
      std::vector<int8_t> lr((size_t)(ury * ucy > urc * ucc ? ury * ucy : urc * ucc) * 8);
      uint8_t st[4];
      cdef_strength_from_q(ac_q, y.bd, st);
      for (int i = 0; i < nsb; i++) memcpy(&sb[(size_t)i * 4], st, 4);
      CHK(av1mi_upload(ctx, d[CSB], sb.data(), sb.size()));
      const int8_t wy[8] = { 1, 3, -7, 15, 3, -7, 15, 0 }, wc[8] = { 1, 0, -7, 15, 0, -7, 15, 0 };
      for (int i = 0; i < ury * ucy; i++) memcpy(&lr[(size_t)i * 8], wy, 8);
      CHK(av1mi_upload(ctx, d[LRY], lr.data(), (size_t)ury * ucy * 8));
      for (int i = 0; i < urc * ucc; i++) memcpy(&lr[(size_t)i * 8], wc, 8);
      CHK(av1mi_upload(ctx, d[LRC], lr.data(), (size_t)urc * ucc * 8));
    }
    out = fopen(job.output.c_str(), "wb");
    if (!out) { *err = job.output + ": Permission denied"; code = 1; goto done; }
    fprintf(out, "AV1MI2 W%d H%d B%d F%d:%d Q%d G%d\n", w, h, y.bd, y.fps_n, y.fps_d, job.quality, G);
    const int damping = 3 + ((ac_q >> (y.bd - 8)) > 100) + ((ac_q >> (y.bd - 8)) > 300);
    for (;;) {
      int n = 0, r = 1;
      while (n < G && (r = y4m_frame(&y, hY.data() + ny * bps * n, hU.data() + nc * bps * n, hV.data() + nc * bps * n)) == 1) n++;
      if (r < 0) { *err = job.input + ": Invalid data found when processing input (truncated frame)"; code = 1; goto done; }
      if (n == 0) break;
      CHK(av1mi_upload(ctx, d[SY], hY.data(), ny * bps * n));
      CHK(av1mi_upload(ctx, d[SU], hU.data(), nc * bps * n));
      CHK(av1mi_upload(ctx, d[SV], hV.data(), nc * bps * n));
      packed.clear();
      for (int t = 0; t < n; t++) {
        const bool key = t == 0;
        const char *sy = (const char *)d[SY] + ny * bps * t, *su = (const char *)d[SU] + nc * bps * t, *sv = (const char *)d[SV] + nc * bps * t;
        if (key) {
          av1mi_intra_job ij;
          memset(&ij, 0, sizeof(ij));
          ij.width = w; ij.height = h; ij.bit_depth = y.bd; ij.nframes = 1; ij.qindex = job.quality; ij.block_size = 8;
          ij.stride_y = w; ij.stride_uv = w / 2;
          ij.d_src_y = sy; ij.d_src_u = su; ij.d_src_v = sv; ij.d_rec_y = d[RY]; ij.d_rec_u = d[RU]; ij.d_rec_v = d[RV];
          ij.d_lev_y = (int16_t *)d[LY]; ij.d_lev_u = (int16_t *)d[LU]; ij.d_lev_v = (int16_t *)d[LV];
          ij.d_modes_y = (uint8_t *)d[MY]; ij.d_modes_uv = (uint8_t *)d[MUV];
          CHK(av1mi_intra_encode(ctx, &ij));
        } else {
          av1mi_inter_job pj;
          memset(&pj, 0, sizeof(pj));
          pj.width = w; pj.height = h; pj.bit_depth = y.bd; pj.nframes = 1; pj.qindex = job.quality; pj.search_range = 8;
          pj.stride_y = w; pj.stride_uv = w / 2;
          pj.d_src_y = sy; pj.d_src_u = su; pj.d_src_v = sv; pj.d_ref_y = d[OY]; pj.d_ref_u = d[OU]; pj.d_ref_v = d[OV];
          pj.d_rec_y = d[RY]; pj.d_rec_u = d[RU]; pj.d_rec_v = d[RV];
          pj.d_lev_y = (int16_t *)d[LY]; pj.d_lev_u = (int16_t *)d[LU]; pj.d_lev_v = (int16_t *)d[LV];
          pj.d_mvs = (int16_t *)d[MVS]; pj.d_skip = (uint8_t *)d[SKIP];
          CHK(av1mi_inter_encode(ctx, &pj));
        }
        if (t + 1 < n) {   // the next frame needs this one as its reference: deblock -> CDEF -> loop restoration
          const int lvl = lf_level_from_q(ac_q, y.bd, key);
          std::vector<uint32_t> mi(ny / 16, 3u | (3u << 4) | ((uint32_t)lvl << 8) | ((uint32_t)lvl << 16) | (3u << 25));
          CHK(av1mi_upload(ctx, d[MIY], mi.data(), mi.size() * 4));
          std::vector<uint32_t> mic(nc / 16, 2u | (2u << 4) | ((uint32_t)lvl << 8) | ((uint32_t)lvl << 16) | (3u << 25));
          CHK(av1mi_upload(ctx, d[MIC], mic.data(), mic.size() * 4));
          CHK(av1mi_deblock_plane(ctx, d[RY], w, d[DY], w, w, h, y.bd, 0, (const uint32_t *)d[MIY], w / 4, 0));
          CHK(av1mi_deblock_plane(ctx, d[RU], w / 2, d[DU], w / 2, w / 2, h / 2, y.bd, 1, (const uint32_t *)d[MIC], w / 8, 0));
          CHK(av1mi_deblock_plane(ctx, d[RV], w / 2, d[DV], w / 2, w / 2, h / 2, y.bd, 1, (const uint32_t *)d[MIC], w / 8, 0));
          av1mi_cdef_job cj;
          memset(&cj, 0, sizeof(cj));
          cj.width = w; cj.height = h; cj.bit_depth = y.bd; cj.nframes = 1; cj.damping = damping; cj.stride_y = w; cj.stride_uv = w / 2;
          cj.d_src_y = d[DY]; cj.d_src_u = d[DU]; cj.d_src_v = d[DV]; cj.d_dst_y = d[CY]; cj.d_dst_u = d[CU]; cj.d_dst_v = d[CV];
          cj.d_sb_strength = (const uint8_t *)d[CSB]; cj.d_skip8 = (const uint8_t *)(key ? d[ZSKIP] : d[SKIP]);
          CHK(av1mi_cdef_frames(ctx, &cj));
          CHK(av1mi_lr_frames(ctx, d[CY], d[DY], d[OY], w, w, h, y.bd, 0, 64, (const int8_t *)d[LRY], 0, 1));
          CHK(av1mi_lr_frames(ctx, d[CU], d[DU], d[OU], w / 2, w / 2, h / 2, y.bd, 1, 64, (const int8_t *)d[LRC], 0, 1));
          CHK(av1mi_lr_frames(ctx, d[CV], d[DV], d[OV], w / 2, w / 2, h / 2, y.bd, 1, 64, (const int8_t *)d[LRC], 0, 1));
        }
        // symbols of this frame -> coded record, on the device; only the record travels to the host
        av1mi_entropy_job ej;
        memset(&ej, 0, sizeof(ej));
        ej.width = w; ej.height = h; ej.nframes = 1; ej.key = key; ej.tile = 64;
        ej.d_lev_y = (const int16_t *)d[LY]; ej.d_lev_u = (const int16_t *)d[LU]; ej.d_lev_v = (const int16_t *)d[LV];
        ej.d_modes_y = (const uint8_t *)d[MY]; ej.d_modes_uv = (const uint8_t *)d[MUV];
        ej.d_mvs = (const int16_t *)d[MVS]; ej.d_skip = (const uint8_t *)d[SKIP];
        ej.d_out = (uint8_t *)d[ENT]; ej.out_cap = ent_cap; ej.d_frame_off = (uint64_t *)d[EOFF];
        CHK(av1mi_entropy_encode(ctx, &ej));
        uint64_t off[2];
        CHK(av1mi_download(ctx, off, d[EOFF], sizeof(off)));
        if (off[1] > ent_cap) { *err = "entropy record larger than its buffer"; code = 2; goto done; }
        rec.resize((size_t)off[1]);
        CHK(av1mi_download(ctx, rec.data(), d[ENT], rec.size()));
        packed.push_back(key ? 'K' : 'P');
        put_varint(packed, (unsigned)rec.size());
        packed.insert(packed.end(), rec.begin(), rec.end());
      }
      fprintf(out, "SEG %d %zu\n", n, packed.size());
      if (fwrite(packed.data(), 1, packed.size(), out) != packed.size()) { *err = job.output + ": No space left on device"; code = 1; goto done; }
      frames_total += n;
      if (r == 0) break;
    }
    if (frames_total == 0) { *err = job.input + ": Invalid data found when processing input (no frames)"; code = 1; }
  }
done:
  for (int i = 0; i < NBUF; i++) if (d[i]) av1mi_free(ctx, d[i]);
  if (out) fclose(out);
  if (y.f) fclose(y.f);
  av1mi_close(ctx);
  return code;
}

}  // namespace av1mi_host
