// y4m.cpp — see y4m.hpp.
#include "y4m.hpp"
#include <sys/stat.h>
#include <unistd.h>
#include <cstdlib>
#include <cstring>

namespace av1mi_host {

Y4mSource::~Y4mSource() { close(); }

void Y4mSource::close() {
  if (reader_running_) { reader_.join(); reader_running_ = false; }
  if (f_ && own_) fclose(f_);
  f_ = nullptr;
}

static bool read_line(FILE *f, std::string *line, size_t cap = 1 << 16) {      // up to and including '\n'; no fixed header size
  line->clear();
  for (int c; (c = fgetc(f)) != EOF;) {
    line->push_back((char)c);
    if (c == '\n') return true;
    if (line->size() > cap) return false;
  }
  return false;
}

bool Y4mSource::open(const std::string &path, std::string *err) {
  path_ = path;
  const bool is_stdin = path == "-" || path == "pipe:0" || path == "pipe:" || path == "/dev/stdin";
  if (is_stdin) { f_ = stdin; own_ = false; }
  else { f_ = fopen(path.c_str(), "rb"); own_ = true; }
  if (!f_) { *err = path + ": No such file or directory"; return false; }
  std::string hdr;
  if (!read_line(f_, &hdr) || hdr.compare(0, 9, "YUV4MPEG2")) { *err = path + ": Invalid data found when processing input (not Y4M)"; return false; }
  hdr_len_ = (long)hdr.size();
  std::string cs = "420jpeg";
  for (size_t p = 0; p < hdr.size();) {
    size_t q = hdr.find_first_of(" \n", p);
    if (q == std::string::npos) q = hdr.size();
    const std::string t = hdr.substr(p, q - p);
    p = q + 1;
    if (t.empty()) continue;
    if (t[0] == 'W') w = atoi(t.c_str() + 1);
    else if (t[0] == 'H') h = atoi(t.c_str() + 1);
    else if (t[0] == 'F') sscanf(t.c_str() + 1, "%d:%d", &fps_n, &fps_d);
    else if (t[0] == 'C') cs = t.substr(1);
  }
  if (cs.rfind("420p10", 0) == 0) bd = 10;
  else if (cs.rfind("420", 0) == 0 && cs.find("p1") == std::string::npos) bd = 8;
  else { *err = "Invalid argument: unsupported Y4M colourspace " + cs + " (4:2:0 8/10-bit only)"; return false; }
  if (w < 8 || h < 8) { *err = "Invalid argument: frame size below 8x8"; return false; }
  if (w > 4096 || h > 4096) { *err = "Invalid argument: frames above 4096x4096 need more than 64 tile rows / columns"; return false; }
  if (fps_n <= 0 || fps_d <= 0) { fps_n = 30; fps_d = 1; }
  // Y4M 4:2:0 planes of a w x h picture: w * h luma and two ceil(w / 2) * ceil(h / 2) chroma planes
  frame_bytes_ = ((size_t)w * h + 2 * (size_t)((w + 1) / 2) * ((h + 1) / 2)) * (bd == 8 ? 1 : 2);
  // in place only when the file seeks AND every frame header is the bare "FRAME\n" (frame parameters make the frames unequal in size)
  struct stat st;
  seekable_ = false;
  if (!is_stdin && !fstat(fileno(f_), &st) && S_ISREG(st.st_mode)) {
    char tag[6];
    const bool bare = pread(fileno(f_), tag, 6, hdr_len_) == 6 && !memcmp(tag, "FRAME\n", 6);
    const off_t body = st.st_size - hdr_len_;
    if (bare && body % (off_t)(6 + frame_bytes_) == 0) {
      seekable_ = true;
      nframes_ = (long)(body / (off_t)(6 + frame_bytes_));
    }
  }
  return true;
}

// one group read sequentially into buf (frames back to back, planes as in the file); returns the frames read
long Y4mSource::read_group(std::vector<unsigned char> &buf, long max_frames, bool *bad) {
  *bad = false;
  if (eof_) return 0;
  if (buf.size() < (size_t)max_frames * frame_bytes_) buf.resize((size_t)max_frames * frame_bytes_);
  long n = 0;
  std::string line;
  for (; n < max_frames; n++) {
    if (!read_line(f_, &line)) {
      eof_ = true;
      if (!line.empty()) *bad = true;      // a partial FRAME line
      break;
    }
    if (line.compare(0, 5, "FRAME") || (line.size() > 6 && line[5] != ' ')) { *bad = true; eof_ = true; break; }      // "FRAME\n" or "FRAME <params>\n"
    if (fread(buf.data() + (size_t)n * frame_bytes_, 1, frame_bytes_, f_) != frame_bytes_) { *bad = true; eof_ = true; break; }
  }
  return n;
}

void Y4mSource::start_read_ahead(long first, long max_frames) {
  next_first_ = first;
  reader_running_ = true;
  reader_ = std::thread([this, max_frames]() { next_n_ = read_group(next_, max_frames, &next_bad_); });
}

long Y4mSource::prepare(long first, long max_frames, std::string *err) {
  group_first_ = first;
  if (seekable_) {
    const long left = nframes_ - first;
    return left < 0 ? 0 : left < max_frames ? left : max_frames;
  }
  if (!reader_running_) start_read_ahead(first, max_frames);      // the very first group
  reader_.join();
  reader_running_ = false;
  if (next_first_ != first) { if (err) *err = path_ + ": groups of a stream must be read in order"; return -1; }
  cur_.swap(next_);
  cur_n_ = next_n_;
  if (next_bad_) { if (err) *err = path_ + ": Invalid data found when processing input (truncated or malformed frame)"; return -1; }
  if (cur_n_ == max_frames) start_read_ahead(first + max_frames, max_frames);      // the next group, while this one is coded
  return cur_n_;
}

bool Y4mSource::read(long i, int cw, int ch, unsigned char *Y, unsigned char *U, unsigned char *V) const {
  const size_t bps = bd == 8 ? 1 : 2;
  const unsigned char *mem = nullptr;
  const int fd = seekable_ ? fileno(f_) : -1;
  off_t off = 0;
  if (seekable_) {
    off = (off_t)hdr_len_ + (off_t)(group_first_ + i) * (off_t)(6 + frame_bytes_) + 6;
  } else {
    if (i < 0 || i >= cur_n_) return false;
    mem = cur_.data() + (size_t)i * frame_bytes_;
  }
  auto rd = [&](void *dst, size_t n) {
    if (mem) { memcpy(dst, mem, n); mem += n; return true; }
    unsigned char *p = (unsigned char *)dst;
    while (n) {
      const ssize_t k = pread(fd, p, n, off);
      if (k <= 0) return false;
      p += k; off += k; n -= (size_t)k;
    }
    return true;
  };
  // one plane: pw x ph samples in the source -> dw x dh in memory
  auto plane = [&](unsigned char *dst, int pw, int ph, int dw, int dh) {
    if (pw == dw) { if (!rd(dst, (size_t)pw * ph * bps)) return false; }
    else
      for (int r = 0; r < ph; r++) {
        unsigned char *row = dst + (size_t)r * dw * bps;
        if (!rd(row, (size_t)pw * bps)) return false;
        for (int c = pw; c < dw; c++) memcpy(row + (size_t)c * bps, row + (size_t)(pw - 1) * bps, bps);
      }
    for (int r = ph; r < dh; r++) memcpy(dst + (size_t)r * dw * bps, dst + (size_t)(ph - 1) * dw * bps, (size_t)dw * bps);
    return true;
  };
  return plane(Y, w, h, cw, ch) && plane(U, (w + 1) / 2, (h + 1) / 2, cw / 2, ch / 2) && plane(V, (w + 1) / 2, (h + 1) / 2, cw / 2, ch / 2);
}

}  // namespace av1mi_host
