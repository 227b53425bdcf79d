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

bool StreamSink::put(const void *p, size_t n, std::string *err) {
  if (fwrite(p, 1, n, f_) != n) { if (err) *err = path_ + ": No space left on device"; return false; }
  return true;
}

bool StreamSink::open(const std::string &path, const av1::SequenceParams &sp, int fps_n, int fps_d, std::string *err) {
  path_ = path; fps_n_ = fps_n; fps_d_ = fps_d; frames_ = 0;
  kind_ = ends_with(path, ".obu") ? OBU : ends_with(path, ".ivf") ? IVF : MKV;
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
    close_cluster();
    cluster_start_ = ftell(f_);
    std::vector<uint8_t> c;
    ebml_id(c, 0x1F43B675); cluster_size_pos_ = cluster_start_ + (long)c.size(); be(c, 0x01FFFFFFFFFFFFFFull, 8);
    el_uint(c, 0xE7, (uint64_t)t_ms);
    if (!put(c.data(), c.size(), err)) return false;
    cluster_open_ = true; cluster_time_ms_ = t_ms;
    if (key) cues_.push_back({ t_ms, cluster_start_ - seg_data_start_ });
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
    close_cluster();
    std::vector<uint8_t> cues;
    for (auto &c : cues_) {
      std::vector<uint8_t> pos, pt;
      el_uint(pos, 0xF7, 1); el_uint(pos, 0xF1, (uint64_t)c.second);
      el_uint(pt, 0xB3, (uint64_t)c.first); el_master(pt, 0xB7, pos);
      el_master(cues, 0xBB, pt);
    }
    std::vector<uint8_t> ce; el_master(ce, 0x1C53BB6B, cues);
    ok = put(ce.data(), ce.size(), err);
    const long end = ftell(f_);
    std::vector<uint8_t> sz; be(sz, (uint64_t)(end - seg_data_start_) | ((uint64_t)1 << 56), 8);
    fseek(f_, seg_data_start_ - 8, SEEK_SET); ok = ok && fwrite(sz.data(), 1, 8, f_) == 8;
    const double dur = (double)frames_ * 1000.0 * fps_d_ / fps_n_;
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
