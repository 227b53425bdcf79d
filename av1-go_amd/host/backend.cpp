// backend.cpp — drives libav1mi.so over closed-GOP segments of a Y4M file (4:2:0, 8- or 10-bit).
//
// Input is raw video because demux / H.264 decode stay FFmpeg's job (SURVEY.md §8b "Gap to flag"); the output is NOT an
// AV1 bitstream: entropy coding and OBU packing are not built, so the container written here
// ("AV1MI1") holds, per segment, the mode bytes and the quantised levels run-length/varint packed on the host.  It exists
// so that the job contract (output file present, size gate, atomic replace) can be exercised end to end.
#include "backend.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/av1mi.h"

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
// zero-run + zigzag varint packing of levels (host-side stand-in for entropy coding; not AV1)
void pack_levels(const int16_t *lv, size_t n, std::vector<unsigned char> &o) {
  unsigned run = 0;
  for (size_t i = 0; i < n; i++) {
    if (lv[i] == 0) { run++; continue; }
    put_varint(o, run << 1); run = 0;
    put_varint(o, ((unsigned)(lv[i] < 0 ? -lv[i] : lv[i]) << 1 | (lv[i] < 0)) << 1 | 1);
  }
  put_varint(o, run << 1);
}
#define CHK(call)                                                                        \
  do { int rc_ = (call); if (rc_ != AV1MI_OK) { *err = std::string(#call) + ": " + av1mi_last_error(ctx); code = 2; goto done; } } while (0)

}  // namespace

int RunBackend(const BackendJob &job, std::string *err) {
  av1mi_ctx *ctx = nullptr;
  if (av1mi_device_count() <= 0 || av1mi_open(job.device, &ctx) != AV1MI_OK) {
    *err = "Error: no usable HIP device for the av1mi backend (device " + std::to_string(job.device) + ")";
    return -1;
  }
  Y4m y;
  int code = 0;
  FILE *out = nullptr;
  std::vector<unsigned char> hY, hU, hV, packed, modes;
  std::vector<int16_t> lev;
  void *d[11] = { nullptr };
  size_t ny = 0, nc = 0, nb = 0;
  long frames_total = 0;
  if (!y4m_open(job.input, &y, err)) { code = 1; goto done; }
  {
    const size_t bps = y.bd == 8 ? 1 : 2;
    const int G = job.gop;
    ny = (size_t)y.w * y.h; nc = ny / 4; nb = ny / 64;
    hY.resize(ny * bps * G); hU.resize(nc * bps * G); hV.resize(nc * bps * G);
    lev.resize(ny * G); modes.resize(nb * G);
    const size_t sizes[11] = { ny * bps * G, nc * bps * G, nc * bps * G, ny * bps * G, nc * bps * G, nc * bps * G,
                               ny * 2 * G, nc * 2 * G, nc * 2 * G, nb * G, nb * G };
    for (int i = 0; i < 11; i++) CHK(av1mi_malloc(ctx, &d[i], sizes[i]));
    out = fopen(job.output.c_str(), "wb");
    if (!out) { *err = job.output + ": Permission denied"; code = 1; goto done; }
    fprintf(out, "AV1MI1 W%d H%d B%d F%d:%d Q%d G%d\n", y.w, y.h, y.bd, y.fps_n, y.fps_d, job.quality, G);
    for (;;) {
      int n = 0, r = 1;
      while (n < G && (r = y4m_frame(&y, hY.data() + ny * bps * n, hU.data() + nc * bps * n, hV.data() + nc * bps * n)) == 1) n++;
      if (r < 0) { *err = job.input + ": Invalid data found when processing input (truncated frame)"; code = 1; goto done; }
      if (n == 0) break;
      CHK(av1mi_upload(ctx, d[0], hY.data(), ny * bps * n));
      CHK(av1mi_upload(ctx, d[1], hU.data(), nc * bps * n));
      CHK(av1mi_upload(ctx, d[2], hV.data(), nc * bps * n));
      av1mi_intra_job ij;
      memset(&ij, 0, sizeof(ij));
      ij.width = y.w; ij.height = y.h; ij.bit_depth = y.bd; ij.nframes = n; ij.qindex = job.quality; ij.block_size = 8;
      ij.stride_y = y.w; ij.stride_uv = y.w / 2;
      ij.d_src_y = d[0]; ij.d_src_u = d[1]; ij.d_src_v = d[2]; ij.d_rec_y = d[3]; ij.d_rec_u = d[4]; ij.d_rec_v = d[5];
      ij.d_lev_y = (int16_t *)d[6]; ij.d_lev_u = (int16_t *)d[7]; ij.d_lev_v = (int16_t *)d[8];
      ij.d_modes_y = (uint8_t *)d[9]; ij.d_modes_uv = (uint8_t *)d[10];
      CHK(av1mi_intra_encode(ctx, &ij));   // every frame a key frame until the inter path is wired into the host
      packed.clear();
      const size_t lev_n[3] = { ny * n, nc * n, nc * n };
      for (int p = 0; p < 3; p++) {
        CHK(av1mi_download(ctx, lev.data(), d[6 + p], lev_n[p] * 2));
        pack_levels(lev.data(), lev_n[p], packed);
      }
      for (int p = 0; p < 2; p++) {
        CHK(av1mi_download(ctx, modes.data(), d[9 + p], nb * n));
        packed.insert(packed.end(), modes.begin(), modes.begin() + nb * n);
      }
      fprintf(out, "SEG %d %zu\n", n, packed.size());
      if (fwrite(packed.data(), 1, packed.size(), out) != packed.size()) { *err = job.output + ": No space left on device"; code = 1; goto done; }
      frames_total += n;
      if (r == 0) break;
    }
    if (frames_total == 0) { *err = job.input + ": Invalid data found when processing input (no frames)"; code = 1; }
  }
done:
  for (int i = 0; i < 11; i++) if (d[i]) av1mi_free(ctx, d[i]);
  if (out) fclose(out);
  if (y.f) fclose(y.f);
  av1mi_close(ctx);
  return code;
}

}  // namespace av1mi_host
