// mux.cpp — see mux.hpp.  Matroska element IDs / the V_AV1 mapping (CodecPrivate = AV1CodecConfigurationRecord, block payload =
// the temporal unit without its temporal delimiter) are restated from the Matroska and "AV1 in Matroska" specifications.
#include "mux.hpp"
#include <cstring>

namespace av1mi_host {
namespace {

void be(std::vector<uint8_t> &o, uint64_t v, int n) { for (int i = n - 1; i >= 0; i--) o.push_back((uint8_t)(v >> (8 * i))); }
void ebml_id(std::vector<uint8_t> &o, uint32_t id) { int n = id > 0xFFFFFF ? 4 : id > 0xFFFF ? 3 : id > 0xFF ? 2 : 1; be(o, id, n); }
void ebml_size(std::vector<uint8_t> &o, uint64_t v, int force = 0) {     // EBML variable-length size
  int n = force;
  if (!n) for (n = 1; n < 8 && v >= ((uint64_t)1 << (7 * n)) - 1; n++) {}
  be(o, v | ((uint64_t)1 << (7 * n)), n);
}
void el_uint(std::vector<uint8_t> &o, uint32_t id, uint64_t v) {
  int n = 1;
  while (n < 8 && (v >> (8 * n))) n++;
  ebml_id(o, id); ebml_size(o, (uint64_t)n); be(o, v, n);
}
void el_str(std::vector<uint8_t> &o, uint32_t id, const char *s) { ebml_id(o, id); ebml_size(o, strlen(s)); o.insert(o.end(), s, s + strlen(s)); }
void el_bin(std::vector<uint8_t> &o, uint32_t id, const std::vector<uint8_t> &b) { ebml_id(o, id); ebml_size(o, b.size()); o.insert(o.end(), b.begin(), b.end()); }
void el_master(std::vector<uint8_t> &o, uint32_t id, const std::vector<uint8_t> &body) { el_bin(o, id, body); }
void el_float(std::vector<uint8_t> &o, uint32_t id, double d) { uint64_t u; memcpy(&u, &d, 8); ebml_id(o, id); ebml_size(o, 8); be(o, u, 8); }

bool ends_with(const std::string &s, const char *suf) { const size_t n = strlen(suf); return s.size() >= n && !s.compare(s.size() - n, n, suf); }

}  // namespace


// ------------------------------------------------------------------------------------------------ MkvReader
namespace {
enum : uint32_t { ID_EBML = 0x1A45DFA3, ID_SEGMENT = 0x18538067, ID_INFO = 0x1549A966, ID_TSCALE = 0x2AD7B1, ID_TRACKS = 0x1654AE6B, ID_ENTRY = 0xAE,
                  ID_TNUM = 0xD7, ID_TUID = 0x73C5, ID_CLUSTER = 0x1F43B675, ID_CTIME = 0xE7, ID_SIMPLE = 0xA3, ID_GROUP = 0xA0, ID_BLOCK = 0xA1,
                  ID_BDUR = 0x9B, ID_REF = 0xFB, ID_CUES = 0x1C53BB6B, ID_TAGS = 0x1254C367, ID_SEEK = 0x114D9B74, ID_CHAPTERS = 0x1043A770,
                  ID_ATTACH = 0x1941A469 };
bool segment_level(uint32_t id) {
  return id == ID_CLUSTER || id == ID_CUES || id == ID_TAGS || id == ID_SEEK || id == ID_CHAPTERS || id == ID_ATTACH || id == ID_INFO || id == ID_TRACKS;
}
// an element header out of a memory buffer; false at the end / on a malformed header
bool mem_header(const std::vector<uint8_t> &b, size_t *pos, uint32_t *id, uint64_t *size) {
  if (*pos >= b.size()) return false;
  int n = 1;
  for (uint8_t m = 0x80; n <= 4 && !(b[*pos] & m); m >>= 1) n++;
  if (n > 4 || *pos + n > b.size()) return false;
  uint32_t v = 0;
  for (int i = 0; i < n; i++) v = (v << 8) | b[*pos + i];
  *pos += n; *id = v;
  if (*pos >= b.size()) return false;
  int k = 1;
  for (uint8_t m = 0x80; k <= 8 && !(b[*pos] & m); m >>= 1) k++;
  if (k > 8 || *pos + k > b.size()) return false;
  uint64_t sz = b[*pos] & (0xFFu >> k);
  for (int i = 1; i < k; i++) sz = (sz << 8) | b[*pos + i];
  *pos += k; *size = sz;
  return *pos + sz <= b.size();
}
uint64_t mem_uint(const std::vector<uint8_t> &b, size_t pos, uint64_t n) { uint64_t v = 0; for (uint64_t i = 0; i < n && i < 8; i++) v = (v << 8) | b[pos + i]; return v; }
}  // namespace

MkvReader::~MkvReader() { if (f_) fclose(f_); }
bool MkvReader::fail(std::string *err, const char *what) const { if (err) *err = path_ + ": " + what; return false; }

// element ID and size at the file position; size -1 = unknown.  false with an empty *err at a clean end of file
bool MkvReader::header(uint32_t *id, int64_t *size, std::string *err) {
  const int c = fgetc(f_);
  if (c == EOF) { if (err) err->clear(); return false; }
  int n = 1;
  for (int m = 0x80; n <= 4 && !(c & m); m >>= 1) n++;
  if (n > 4) return fail(err, "Invalid data found when processing input (element id)");
  uint32_t v = (uint32_t)c;
  for (int i = 1; i < n; i++) { const int d = fgetc(f_); if (d == EOF) return fail(err, "Invalid data found when processing input (truncated)"); v = (v << 8) | (uint32_t)d; }
  const int s0 = fgetc(f_);
  if (s0 == EOF) return fail(err, "Invalid data found when processing input (truncated)");
  int k = 1;
  for (int m = 0x80; k <= 8 && !(s0 & m); m >>= 1) k++;
  if (k > 8) return fail(err, "Invalid data found when processing input (element size)");
  uint64_t sz = (uint64_t)(s0 & (0xFF >> k));
  bool ones = sz == (uint64_t)(0xFF >> k);
  for (int i = 1; i < k; i++) { const int d = fgetc(f_); if (d == EOF) return fail(err, "Invalid data found when processing input (truncated)"); sz = (sz << 8) | (uint64_t)d; ones = ones && d == 0xFF; }
  *id = v; *size = ones ? -1 : (int64_t)sz;
  return true;
}

bool MkvReader::read_tracks(long end, std::string *err) {
  while (ftell(f_) < end) {
    uint32_t id; int64_t size;
    if (!header(&id, &size, err) || size < 0) return fail(err, "Invalid data found when processing input (tracks)");
    if (id != ID_ENTRY) { fseek(f_, (long)size, SEEK_CUR); continue; }
    std::vector<uint8_t> body((size_t)size);
    if (fread(body.data(), 1, body.size(), f_) != body.size()) return fail(err, "Invalid data found when processing input (track entry)");
    Track t;
    size_t pos = 0;
    for (;;) {
      const size_t at = pos;
      uint32_t cid; uint64_t csz;
      if (!mem_header(body, &pos, &cid, &csz)) break;
      if (cid == ID_TNUM) t.number = mem_uint(body, pos, csz);
      else if (cid != ID_TUID) t.entry.insert(t.entry.end(), body.begin() + (long)at, body.begin() + (long)(pos + csz));
      pos += csz;
    }
    if (!t.number) return fail(err, "Invalid data found when processing input (track without a number)");
    tracks_.push_back(t);
  }
  return true;
}

bool MkvReader::open(const std::string &path, std::string *err) {
  path_ = path;
  f_ = fopen(path.c_str(), "rb");
  if (!f_) return fail(err, "No such file or directory");
  uint32_t id; int64_t size;
  if (!header(&id, &size, err) || id != ID_EBML || size < 0) return fail(err, "Invalid data found when processing input (not a Matroska file)");
  fseek(f_, (long)size, SEEK_CUR);
  if (!header(&id, &size, err) || id != ID_SEGMENT) return fail(err, "Invalid data found when processing input (no segment)");
  stack_.push_back({ ID_SEGMENT, size < 0 ? -1 : ftell(f_) + (long)size });
  for (;;) {
    const long at = ftell(f_);
    if (!header(&id, &size, err)) { if (err && !err->empty()) return false; break; }       // a file without clusters: tracks only
    if (id == ID_CLUSTER) { fseek(f_, at, SEEK_SET); break; }
    if (size < 0) return fail(err, "Invalid data found when processing input (unknown size outside a cluster)");
    const long end = ftell(f_) + (long)size;
    if (id == ID_TRACKS) { if (!read_tracks(end, err)) return false; }
    else if (id == ID_INFO) {
      while (ftell(f_) < end) {
        uint32_t cid; int64_t csz;
        if (!header(&cid, &csz, err) || csz < 0) return fail(err, "Invalid data found when processing input (info)");
        if (cid == ID_TSCALE && csz <= 8) { uint64_t v = 0; for (int i = 0; i < csz; i++) v = (v << 8) | (uint64_t)fgetc(f_); scale_ns_ = v ? v : 1000000; }
        else fseek(f_, (long)csz, SEEK_CUR);
      }
    }
    fseek(f_, end, SEEK_SET);
  }
  if (tracks_.empty()) return fail(err, "Invalid data found when processing input (no tracks)");
  return true;
}

bool MkvReader::next(Block *b, bool *end, std::string *err) {
  *end = false;
  for (;;) {
    bool in_cluster = stack_.size() > 1;
    if (in_cluster && stack_.back().end >= 0 && ftell(f_) >= stack_.back().end) { stack_.pop_back(); continue; }
    const long at = ftell(f_);
    uint32_t id; int64_t size;
    if (!header(&id, &size, err)) { if (err && !err->empty()) return false; *end = true; return true; }
    if (in_cluster && segment_level(id)) { stack_.pop_back(); in_cluster = false; }       // the end of a cluster of unknown size
    if (!in_cluster) {
      if (id == ID_CLUSTER) { stack_.push_back({ ID_CLUSTER, size < 0 ? -1 : ftell(f_) + (long)size }); cluster_ts_ = 0; continue; }
      if (size < 0) return fail(err, "Invalid data found when processing input (unknown size outside a cluster)");
      fseek(f_, (long)size, SEEK_CUR);
      continue;
    }
    if (size < 0 || size > (1 << 28)) return fail(err, "Invalid data found when processing input (cluster element)");
    (void)at;
    if (id == ID_CTIME) { uint64_t v = 0; for (int i = 0; i < size; i++) v = (v << 8) | (uint64_t)fgetc(f_); cluster_ts_ = v; continue; }
    if (id != ID_SIMPLE && id != ID_GROUP) { fseek(f_, (long)size, SEEK_CUR); continue; }
    std::vector<uint8_t> body((size_t)size);
    if (fread(body.data(), 1, body.size(), f_) != body.size()) return fail(err, "Invalid data found when processing input (truncated block)");
    auto block = [&](const uint8_t *p, size_t n) -> bool {      // track vint, 16-bit relative time, flags, data
      if (n < 4) return false;
      int k = 1;
      for (int m = 0x80; k <= 8 && !(p[0] & m); m >>= 1) k++;
      if (k > 8 || n < (size_t)k + 3) return false;
      uint64_t tr = p[0] & (0xFF >> k);
      for (int i = 1; i < k; i++) tr = (tr << 8) | p[i];
      const int16_t rel = (int16_t)((p[k] << 8) | p[k + 1]);
      b->track = tr;
      b->t_ns = ((int64_t)cluster_ts_ + rel) * (int64_t)scale_ns_;
      b->tail.assign(p + k + 2, p + n);
      return true;
    };
    b->duration_ns = -1; b->key = true;
    if (id == ID_SIMPLE) {
      if (!block(body.data(), body.size())) return fail(err, "Invalid data found when processing input (block)");
      b->key = (b->tail[0] & 0x80) != 0;
      return true;
    }
    bool have = false;
    size_t pos = 0;
    for (;;) {
      uint32_t cid; uint64_t csz;
      if (!mem_header(body, &pos, &cid, &csz)) break;
      if (cid == ID_BLOCK) { if (!block(body.data() + pos, (size_t)csz)) return fail(err, "Invalid data found when processing input (block)"); have = true; }
      else if (cid == ID_BDUR) b->duration_ns = (int64_t)mem_uint(body, pos, csz) * (int64_t)scale_ns_;
      else if (cid == ID_REF) b->key = false;
      pos += csz;
    }
    if (have) return true;
  }
}

// ------------------------------------------------------------------------------------------------ StreamSink: side tracks
StreamSink::~StreamSink() { abort(); for (Side *s : sides_) delete s; }

bool StreamSink::add_side_file(const std::string &path, std::string *err) {
  if (f_) { if (err) *err = "side files must be added before the output is opened"; return false; }
  Side *s = new Side;
  if (!s->rd.open(path, err)) { delete s; return false; }
  sides_.push_back(s);
  return true;
}

bool StreamSink::put_side_block(Side &s, std::string *err) {
  const MkvReader::Block &b = s.pending;
  int out = 0;
  for (size_t i = 0; i < s.rd.tracks().size(); i++) if (s.rd.tracks()[i].number == b.track) out = s.out_number[i];
  s.have = false;
  if (!out) return true;                                       // a block of a track the file does not declare: dropped
  const long t_ms = (long)((b.t_ns + 500000) / 1000000);
  if (!cluster_open_ || t_ms - cluster_time_ms_ > 32767 || t_ms - cluster_time_ms_ < -32768) { if (!start_cluster(t_ms, false, err)) return false; }
  std::vector<uint8_t> blk;
  ebml_size(blk, (uint64_t)out);                               // the track number is coded like a size
  be(blk, (uint64_t)(uint16_t)(int16_t)(t_ms - cluster_time_ms_), 2);
  std::vector<uint8_t> o;
  if (b.duration_ns < 0 && b.key) {                            // SimpleBlock
    ebml_id(o, 0xA3); ebml_size(o, blk.size() + b.tail.size());
    o.insert(o.end(), blk.begin(), blk.end());
    o.push_back((uint8_t)(b.tail[0] | 0x80));
    o.insert(o.end(), b.tail.begin() + 1, b.tail.end());
  } else {                                                     // BlockGroup: the duration (subtitles) / the reference of a non-key block
    std::vector<uint8_t> g;
    ebml_id(g, 0xA1); ebml_size(g, blk.size() + b.tail.size());
    g.insert(g.end(), blk.begin(), blk.end());
    g.push_back((uint8_t)(b.tail[0] & 0x7F));
    g.insert(g.end(), b.tail.begin() + 1, b.tail.end());
    if (b.duration_ns >= 0) el_uint(g, 0x9B, (uint64_t)((b.duration_ns + 500000) / 1000000));
    if (!b.key) { ebml_id(g, 0xFB); ebml_size(g, 1); g.push_back(0xFF); }      // ReferenceBlock -1: "depends on an earlier block"
    el_master(o, 0xA0, g);
  }
  const long end_ms = t_ms + (b.duration_ns > 0 ? (long)(b.duration_ns / 1000000) : 0);
  if (end_ms > side_end_ms_) side_end_ms_ = end_ms;
  return put(o.data(), o.size(), err);
}

// copies, in timestamp order over all side files, the blocks due before (or at) t_ms
bool StreamSink::side_blocks_until(long t_ms, bool inclusive, std::string *err) {
  for (;;) {
    Side *best = nullptr;
    for (Side *s : sides_) {
      while (!s->have && !s->done) {
        bool end = false;
        if (!s->rd.next(&s->pending, &end, err)) return false;
        if (end) s->done = true; else s->have = true;
      }
      if (s->have && (!best || s->pending.t_ns < best->pending.t_ns)) best = s;
    }
    if (!best) return true;
    const long bt = (long)((best->pending.t_ns + 500000) / 1000000);
    if (inclusive ? bt > t_ms : bt >= t_ms) return true;
    if (!put_side_block(*best, err)) return false;
  }
}

bool StreamSink::start_cluster(long t_ms, bool cue, std::string *err) {
  close_cluster();
  cluster_start_ = ftell(f_);
  std::vector<uint8_t> c;
  ebml_id(c, 0x1F43B675); cluster_size_pos_ = cluster_start_ + (long)c.size(); be(c, 0x01FFFFFFFFFFFFFFull, 8);
  el_uint(c, 0xE7, (uint64_t)t_ms);
  if (!put(c.data(), c.size(), err)) return false;
  cluster_open_ = true; cluster_time_ms_ = t_ms;
  if (cue) cues_.push_back({ t_ms, cluster_start_ - seg_data_start_ });
  return true;
}

bool StreamSink::put(const void *p, size_t n, std::string *err) {
  if (fwrite(p, 1, n, f_) != n) { if (err) *err = path_ + ": No space left on device"; return false; }
  return true;
}

bool StreamSink::open(const std::string &path, const av1::SequenceParams &sp, int fps_n, int fps_d, std::string *err) {
  path_ = path; fps_n_ = fps_n; fps_d_ = fps_d; frames_ = 0;
  kind_ = ends_with(path, ".obu") ? OBU : ends_with(path, ".ivf") ? IVF : MKV;
  if (kind_ != MKV && !sides_.empty()) { if (err) *err = path + ": Invalid argument: copied tracks need a Matroska output"; return false; }
  f_ = fopen(path.c_str(), "wb");
  if (!f_) { if (err) *err = path + ": Permission denied"; return false; }
  std::vector<uint8_t> h;
  if (kind_ == IVF) {
    h.insert(h.end(), { 'D', 'K', 'I', 'F', 0, 0, 32, 0, 'A', 'V', '0', '1' });
    auto le = [&](uint32_t v, int n) { for (int i = 0; i < n; i++) h.push_back((uint8_t)(v >> (8 * i))); };
    le((uint32_t)sp.width, 2); le((uint32_t)sp.height, 2); le((uint32_t)fps_n, 4); le((uint32_t)fps_d, 4); le(0, 4); le(0, 4);
    return put(h.data(), h.size(), err);
  }
  if (kind_ == MKV) {
    std::vector<uint8_t> e;
    el_uint(e, 0x4286, 1); el_uint(e, 0x42F7, 1); el_uint(e, 0x42F2, 4); el_uint(e, 0x42F3, 8);
    el_str(e, 0x4282, "matroska"); el_uint(e, 0x4287, 4); el_uint(e, 0x4285, 2);
    el_master(h, 0x1A45DFA3, e);
    ebml_id(h, 0x18538067); be(h, 0x01FFFFFFFFFFFFFFull, 8);       // Segment, size patched on close
    seg_data_start_ = (long)h.size();
    std::vector<uint8_t> info;
    el_uint(info, 0x2AD7B1, 1000000);                              // TimestampScale: 1 ms
    el_str(info, 0x4D80, "av1mi"); el_str(info, 0x5741, "av1mi");  // MuxingApp, WritingApp
    const size_t dur_off = info.size();
    el_float(info, 0x4489, 0.0);                                   // Duration, patched on close
    std::vector<uint8_t> ih; ebml_id(ih, 0x1549A966); ebml_size(ih, info.size());
    duration_pos_ = (long)(h.size() + ih.size() + dur_off + 3);    // past the ID (2 bytes) and the size byte of Duration
    h.insert(h.end(), ih.begin(), ih.end()); h.insert(h.end(), info.begin(), info.end());
    // Tracks: one V_AV1 video track; CodecPrivate = av1C + the sequence header OBU
    const std::vector<uint8_t> seq = av1::sequence_header_obu(sp);
    std::vector<uint8_t> av1c = { 0x81, 0x1F, (uint8_t)((sp.bit_depth == 10 ? 0x40 : 0x00) | 0x0C), 0x00 };   // profile 0, level index 31, 4:2:0
    av1c.insert(av1c.end(), seq.begin(), seq.end());
    std::vector<uint8_t> video; el_uint(video, 0xB0, (uint64_t)sp.width); el_uint(video, 0xBA, (uint64_t)sp.height);
    std::vector<uint8_t> te;
    el_uint(te, 0xD7, 1); el_uint(te, 0x73C5, 1); el_uint(te, 0x83, 1); el_uint(te, 0x9C, 0);
    el_str(te, 0x86, "V_AV1"); el_bin(te, 0x63A2, av1c);
    el_uint(te, 0x23E383, (uint64_t)(1000000000.0 * fps_d / fps_n + 0.5));      // DefaultDuration, ns
    el_master(te, 0xE0, video);
    std::vector<uint8_t> tracks; el_master(tracks, 0xAE, te);
    int number = 1;
    for (Side *sd : sides_)                                        // the side files' tracks: entries verbatim under new numbers
      for (const MkvReader::Track &t : sd->rd.tracks()) {
        std::vector<uint8_t> e2;
        number++;
        el_uint(e2, 0xD7, (uint64_t)number); el_uint(e2, 0x73C5, (uint64_t)number);
        e2.insert(e2.end(), t.entry.begin(), t.entry.end());
        el_master(tracks, 0xAE, e2);
        sd->out_number.push_back(number);
      }
    el_master(h, 0x1654AE6B, tracks);
    return put(h.data(), h.size(), err);
  }
  return true;
}

void StreamSink::close_cluster() {
  if (!cluster_open_) return;
  const long end = ftell(f_);
  std::vector<uint8_t> sz; be(sz, (uint64_t)(end - cluster_size_pos_ - 8) | ((uint64_t)1 << 56), 8);
  fseek(f_, cluster_size_pos_, SEEK_SET); fwrite(sz.data(), 1, 8, f_); fseek(f_, end, SEEK_SET);
  cluster_open_ = false;
}

bool StreamSink::write(const std::vector<uint8_t> &tu, bool key, std::string *err) {
  if (!f_) { if (err) *err = "stream sink is not open"; return false; }
  const long idx = frames_++;
  if (kind_ == OBU) return put(tu.data(), tu.size(), err);
  if (kind_ == IVF) {
    uint8_t fh[12];
    for (int i = 0; i < 4; i++) fh[i] = (uint8_t)(tu.size() >> (8 * i));
    for (int i = 0; i < 8; i++) fh[4 + i] = (uint8_t)((uint64_t)idx >> (8 * i));
    return put(fh, 12, err) && put(tu.data(), tu.size(), err);
  }
  const long t_ms = (long)((double)idx * 1000.0 * fps_d_ / fps_n_ + 0.5);
  if (key || !cluster_open_ || t_ms - cluster_time_ms_ > 30000) {     // a cluster per closed GOP
    if (cluster_open_ && !side_blocks_until(t_ms, false, err)) return false;      // what is due before this frame stays in the old cluster
    if (!start_cluster(t_ms, key, err)) return false;
  }
  if (!sides_.empty()) {
    if (!side_blocks_until(t_ms, true, err)) return false;
    if (t_ms - cluster_time_ms_ > 32767 || t_ms < cluster_time_ms_) { if (!start_cluster(t_ms, false, err)) return false; }    // a side block moved the cluster
  }
  // the block payload is the temporal unit without its temporal delimiter (0x12 0x00)
  size_t skip = tu.size() >= 2 && tu[0] == 0x12 && tu[1] == 0x00 ? 2 : 0;
  std::vector<uint8_t> b;
  ebml_id(b, 0xA3); ebml_size(b, tu.size() - skip + 4);
  b.push_back(0x81);                                                    // track 1
  be(b, (uint64_t)(uint16_t)(t_ms - cluster_time_ms_), 2);
  b.push_back(key ? 0x80 : 0x00);
  return put(b.data(), b.size(), err) && put(tu.data() + skip, tu.size() - skip, err);
}

bool StreamSink::close(std::string *err) {
  if (!f_) return true;
  bool ok = true;
  if (kind_ == IVF) {
    uint8_t n[4];
    for (int i = 0; i < 4; i++) n[i] = (uint8_t)((uint32_t)frames_ >> (8 * i));
    ok = !fseek(f_, 24, SEEK_SET) && fwrite(n, 1, 4, f_) == 4;
  } else if (kind_ == MKV) {
    ok = side_blocks_until(0x7FFFFFFFL, true, err);               // what the side files hold beyond the last video frame
    close_cluster();
    std::vector<uint8_t> cues;
    for (auto &c : cues_) {
      std::vector<uint8_t> pos, pt;
      el_uint(pos, 0xF7, 1); el_uint(pos, 0xF1, (uint64_t)c.second);
      el_uint(pt, 0xB3, (uint64_t)c.first); el_master(pt, 0xB7, pos);
      el_master(cues, 0xBB, pt);
    }
    std::vector<uint8_t> ce; el_master(ce, 0x1C53BB6B, cues);
    ok = put(ce.data(), ce.size(), err) && ok;
    const long end = ftell(f_);
    std::vector<uint8_t> sz; be(sz, (uint64_t)(end - seg_data_start_) | ((uint64_t)1 << 56), 8);
    fseek(f_, seg_data_start_ - 8, SEEK_SET); ok = ok && fwrite(sz.data(), 1, 8, f_) == 8;
    double dur = (double)frames_ * 1000.0 * fps_d_ / fps_n_;
    if ((double)side_end_ms_ > dur) dur = (double)side_end_ms_;
    std::vector<uint8_t> d; uint64_t u; memcpy(&u, &dur, 8); be(d, u, 8);
    fseek(f_, duration_pos_, SEEK_SET); ok = ok && fwrite(d.data(), 1, 8, f_) == 8;
  }
  ok = !fclose(f_) && ok;
  f_ = nullptr;
  if (!ok && err && err->empty()) *err = path_ + ": write failed";
  return ok;
}

void StreamSink::abort() {
  if (f_) { fclose(f_); f_ = nullptr; }
}

}  // namespace av1mi_host
