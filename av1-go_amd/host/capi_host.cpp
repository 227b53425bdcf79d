// capi_host.cpp — extern "C" hooks over the host mirror so that the CPU test-suite (ctypes) can pin it against
// values read off the reference source.
#include <cstring>
#include "daemon.hpp"
#include "av1_bitstream.hpp"
#include "mux.hpp"
#include "y4m.hpp"
#include "backend.hpp"

using namespace av1mi_host;

extern "C" {
int av1mi_host_determine_quality(int height) { return DetermineQuality(height); }
int av1mi_host_check_size_gate(long long orig, long long neu, double ratio) { return CheckSizeGate(orig, neu, ratio) ? 1 : 0; }
// argv joined with '\n' into buf; returns the number of arguments, -1 on the reference's error (text in buf)
int av1mi_host_transcode_args(const char *in, const char *out, int has_video, int index, int height, int webrip, char *buf, int cap) {
  ProbeResult pr; pr.has_video_stream = has_video != 0; pr.VideoStream.Index = index; pr.VideoStream.Height = height;
  std::vector<std::string> a; std::string err;
  const bool ok = TranscodeArgs("ffmpeg", in, out, pr, webrip != 0, &a, &err);
  std::string j;
  if (ok) for (size_t i = 0; i < a.size(); i++) j += (i ? "\n" : "") + a[i]; else j = err;
  strncpy(buf, j.c_str(), cap - 1); buf[cap - 1] = 0;
  return ok ? (int)a.size() : -1;
}
// Y4mSource (y4m.hpp) alone, for the CPU tests: reads the whole input in groups of `group` frames the way RunBackend does and
// returns the number of frames (-1 on error, text in err); *sum = a checksum over every frame's padded planes; *seekable = the mode
long long av1mi_host_y4m_scan(const char *path, int group, unsigned long long *sum, int *seekable, int *geometry, char *err, int cap) {
  Y4mSource y;
  std::string e;
  auto fail = [&]() { strncpy(err, e.c_str(), cap - 1); err[cap - 1] = 0; return -1LL; };
  if (!y.open(path, &e)) return fail();
  const int cw = (y.w + 7) & ~7, ch = (y.h + 7) & ~7;
  const size_t bps = y.bd == 8 ? 1 : 2;
  std::vector<unsigned char> Y((size_t)cw * ch * bps), U((size_t)cw * ch * bps / 4), V((size_t)cw * ch * bps / 4);
  unsigned long long acc = 1469598103934665603ull;
  long long total = 0;
  for (long first = 0;; first += group) {
    const long n = y.prepare(first, group, &e);
    if (n < 0) return fail();
    if (n == 0) break;
    for (long i = 0; i < n; i++) {
      if (!y.read(i, cw, ch, Y.data(), U.data(), V.data())) { e = "read failed"; return fail(); }
      for (const auto *pl : { &Y, &U, &V }) for (unsigned char b : *pl) acc = (acc ^ b) * 1099511628211ull;
    }
    total += n;
  }
  *sum = acc; *seekable = y.seekable() ? 1 : 0;
  geometry[0] = y.w; geometry[1] = y.h; geometry[2] = y.bd; geometry[3] = y.fps_n; geometry[4] = y.fps_d;
  return total;
}
// StreamSink (mux.hpp) alone, for the CPU tests: n_units byte strings (unit i = bytes [off[i], off[i + 1]) of `units`, standing in for
// temporal units; unit i is a key frame when i % gop == 0) muxed at fps_n / fps_d together with the tracks of the Matroska side files
// `sides` (paths separated by '\n').  Returns 0, or -1 with the text in err.
int av1mi_host_mux_selftest(const char *out_path, int width, int height, int bit_depth, int fps_n, int fps_d, const unsigned char *units,
                            const long long *off, int n_units, int gop, const char *sides, char *err, int cap) {
  std::string e;
  auto fail = [&]() { strncpy(err, e.c_str(), cap - 1); err[cap - 1] = 0; return -1; };
  StreamSink sink;
  for (std::string rest = sides ? sides : ""; !rest.empty();) {
    const size_t nl = rest.find('\n');
    const std::string one = rest.substr(0, nl);
    rest = nl == std::string::npos ? "" : rest.substr(nl + 1);
    if (!one.empty() && !sink.add_side_file(one, &e)) return fail();
  }
  av1::SequenceParams sp;
  sp.width = width; sp.height = height; sp.bit_depth = bit_depth;
  if (!sink.open(out_path, sp, fps_n, fps_d, &e)) return fail();
  for (int i = 0; i < n_units; i++) {
    const std::vector<uint8_t> tu(units + off[i], units + off[i + 1]);
    if (!sink.write(tu, i % gop == 0, &e)) { sink.abort(); return fail(); }
  }
  if (!sink.close(&e)) return fail();
  return 0;
}
// The drop-in for internal/ffmpeg/transcode.go:194 `RunTranscode(ffmpegPath string, args []string) (int, error)`: what the cgo
// shim of INTEGRATION.md binds.  Returns the exit code of the contract (0 = output written, -1 = could not run, else failed);
// the error text (<= 800 chars + "...", transcode.go:295-297) goes to err.  Declared in include/av1mi_host.h.
int av1mi_run_transcode(int argc, const char *const *argv, char *err, size_t errcap) {
  std::vector<std::string> a;
  for (int i = 0; i < argc; i++) a.push_back(argv[i] ? argv[i] : "");
  const RunResult rr = RunTranscode("av1mi", a);
  if (err && errcap) { strncpy(err, rr.err.c_str(), errcap - 1); err[errcap - 1] = 0; }
  return rr.exitCode;
}
// runs the replacement RunTranscode on an argv joined with '\n'; error text into buf; returns the exit code
int av1mi_host_run_transcode(const char *joined, char *buf, int cap) {
  std::vector<std::string> a; std::string s = joined; size_t p = 0, q;
  while ((q = s.find('\n', p)) != std::string::npos) { a.push_back(s.substr(p, q - p)); p = q + 1; }
  a.push_back(s.substr(p));
  const RunResult rr = RunTranscode("av1mi", a);
  strncpy(buf, rr.err.c_str(), cap - 1); buf[cap - 1] = 0;
  return rr.exitCode;
}
// ProcessJob on a source file; returns 0 when the reference would return nil; status and reason into the buffers
int av1mi_host_process_job(const char *source, long long orig_size, double ratio, const char *state_dir, int wait_s, int replace_source,
                           char *status, char *reason, int cap) {
  Job job; job.ID = "test"; job.SourcePath = source; job.OriginalSize = orig_size;
  TranscodeConfig cfg; cfg.MaxSizeRatio = ratio; cfg.JobStateDir = state_dir ? state_dir : ""; cfg.StableWaitSeconds = wait_s;
  cfg.ReplaceSource = replace_source != 0;
  ProbeResult pr; pr.HasVideo = true; pr.has_video_stream = true; pr.VideoStream.Height = 720;
  const std::string e = ProcessJob(&job, "av1mi", pr, cfg);
  strncpy(status, job.Status.c_str(), cap - 1); status[cap - 1] = 0;
  strncpy(reason, job.Reason.c_str(), cap - 1); reason[cap - 1] = 0;
  return e.empty() ? 0 : 1;
}
// AV1 bitstream writer (av1_bitstream.hpp): one temporal unit (delimiter [+ sequence header] + frame) for a frame description.
// Returns the size in bytes (copied into out when it fits in cap), -1 on a description the writer cannot code (text in err).
long long av1mi_obu_write_temporal_unit(const av1mi_obu_frame *f, int with_sequence_header, int threads, uint8_t *out, long long cap,
                                        char *err, int errcap) {
  std::vector<uint8_t> b; std::string e;
  if (!av1::temporal_unit(*f, with_sequence_header != 0, threads, &b, &e)) {
    if (err && errcap > 0) { strncpy(err, e.c_str(), errcap - 1); err[errcap - 1] = 0; }
    return -1;
  }
  if ((long long)b.size() <= cap && out) memcpy(out, b.data(), b.size());
  return (long long)b.size();
}
// One segment of a collected session batch -> its temporal unit (include/av1mi_host.h; backend.cpp SessionTemporalUnit)
long long av1mi_session_temporal_unit(const av1mi_gop_frame *fr, int seg, int width, int height, int bit_depth, int visible_width, int visible_height,
                                      int with_sequence_header, int threads, uint8_t *out, long long cap, char *err, int errcap) {
  std::vector<uint8_t> b; std::string e;
  if (!fr || seg < 0 || seg >= fr->segments) e = "bad batch / segment";
  else if (SessionTemporalUnit(*fr, seg, width, height, bit_depth, visible_width, visible_height, with_sequence_header != 0, threads, &b, &e)) {
    if ((long long)b.size() <= cap && out) memcpy(out, b.data(), b.size());
    return (long long)b.size();
  }
  if (err && errcap > 0) { strncpy(err, e.c_str(), errcap - 1); err[errcap - 1] = 0; }
  return -1;
}
// The general block description (av1_blockstream.cpp): any block / transform size, any partition
long long av1mi_obu_write_blocks_temporal_unit(const av1mi_obu_blocks *f, int with_sequence_header, uint8_t *out, long long cap, char *err, int errcap) {
  std::vector<uint8_t> b; std::string e;
  if (!f || !av1::blocks_temporal_unit(*f, with_sequence_header != 0, &b, &e)) {
    if (err && errcap > 0) { strncpy(err, f ? e.c_str() : "null description", errcap - 1); err[errcap - 1] = 0; }
    return -1;
  }
  if ((long long)b.size() <= cap && out) memcpy(out, b.data(), b.size());
  return (long long)b.size();
}
// One temporal unit from tile payloads coded by the GPU tile entropy coder (include/av1mi.h av1mi_av1_entropy_encode / the GOP
// session with gpu_entropy): `f` needs only its header fields (geometry, quantiser, filter parameters); payloads = the frame's
// ntiles finished tile payloads back to back in raster order, sizes[t] bytes each.  Same return convention as
// av1mi_obu_write_temporal_unit.  Declared in include/av1mi_host.h.
long long av1mi_obu_assemble_temporal_unit(const av1mi_obu_frame *f, const uint8_t *payloads, const uint32_t *sizes, int ntiles,
                                           int with_sequence_header, uint8_t *out, long long cap, char *err, int errcap) {
  std::vector<uint8_t> fr; std::string e;
  if (!av1::frame_obu_from_tiles(*f, payloads, sizes, ntiles, &fr, &e)) {
    if (err && errcap > 0) { strncpy(err, e.c_str(), errcap - 1); err[errcap - 1] = 0; }
    return -1;
  }
  std::vector<uint8_t> b = av1::temporal_delimiter_obu();
  if (with_sequence_header) {
    const av1::SequenceParams sp = av1::sequence_params(*f);
    const std::vector<uint8_t> sh = av1::sequence_header_obu(sp);
    b.insert(b.end(), sh.begin(), sh.end());
  }
  b.insert(b.end(), fr.begin(), fr.end());
  if ((long long)b.size() <= cap && out) memcpy(out, b.data(), b.size());
  return (long long)b.size();
}
// the op-stream path on the host (av1_opstream.cpp): same contract as av1mi_obu_write_temporal_unit; -2 = outside its tool set
// key_rows32 > 0: a key frame whose first key_rows32 luma rows are coded in 32x32 blocks (symbol arrays in the session's layout, see
// av1_opstream.cpp opstream_tiles)
long long av1mi_host_opstream_key32_temporal_unit(const av1mi_obu_frame *f, int key_rows32, int with_sequence_header, uint8_t *out, long long cap, char *err,
                                                  int errcap) {
  std::vector<std::vector<uint8_t>> tiles; std::string e;
  auto fail = [&](long long code) { if (err && errcap > 0) { strncpy(err, e.c_str(), errcap - 1); err[errcap - 1] = 0; } return code; };
  if (!av1::opstream_supported(*f, &e)) return fail(-2);
  if (!av1::opstream_tiles(*f, &tiles, &e, key_rows32)) return fail(-1);
  std::vector<uint8_t> cat; std::vector<uint32_t> sizes;
  for (auto &t : tiles) { sizes.push_back((uint32_t)t.size()); cat.insert(cat.end(), t.begin(), t.end()); }
  std::vector<uint8_t> fr;
  if (!av1::frame_obu_from_tiles(*f, cat.data(), sizes.data(), (int)sizes.size(), &fr, &e)) return fail(-1);
  std::vector<uint8_t> b = av1::temporal_delimiter_obu();
  if (with_sequence_header) {
    const av1::SequenceParams sp = av1::sequence_params(*f);
    const std::vector<uint8_t> sh = av1::sequence_header_obu(sp);
    b.insert(b.end(), sh.begin(), sh.end());
  }
  b.insert(b.end(), fr.begin(), fr.end());
  if ((long long)b.size() <= cap && out) memcpy(out, b.data(), b.size());
  return (long long)b.size();
}
long long av1mi_host_opstream_temporal_unit(const av1mi_obu_frame *f, int with_sequence_header, uint8_t *out, long long cap, char *err, int errcap) {
  return av1mi_host_opstream_key32_temporal_unit(f, 0, with_sequence_header, out, cap, err, errcap);
}
// job record + GPU probe hooks for the CPU tests
int av1mi_host_job_json(const char *id, const char *source, const char *status, const char *reason, long long orig, long long neu, char *buf, int cap) {
  Job j; j.ID = id; j.SourcePath = source; j.Status = status; j.Reason = reason; j.OriginalSize = orig; j.NewSize = neu; j.CreatedAt = "2026-01-02T03:04:05Z";
  const std::string s = JobToJSON(j);
  strncpy(buf, s.c_str(), cap - 1); buf[cap - 1] = 0;
  return (int)s.size();
}
double av1mi_host_gpu_usage(int device, const char *sysfs_root) { return GetGPUUsage(device, sysfs_root ? sysfs_root : "/sys"); }
// container hook for the CPU tests: writes `n` temporal units (concatenated in data, sizes[i] bytes each) to path; 0 = OK
int av1mi_host_mux_units(const char *path, int w, int h, int bd, int fps_n, int fps_d, const uint8_t *data, const long long *sizes,
                         const uint8_t *keys, int n) {
  StreamSink sink; std::string err;
  av1::SequenceParams sp; sp.width = w; sp.height = h; sp.bit_depth = bd;
  if (!sink.open(path, sp, fps_n, fps_d, &err)) return 1;
  for (int i = 0; i < n; i++) {
    std::vector<uint8_t> tu(data, data + sizes[i]);
    data += sizes[i];
    if (!sink.write(tu, keys[i] != 0, &err)) { sink.abort(); return 2; }
  }
  return sink.close(&err) ? 0 : 3;
}
// RunJobPool over `n` source files joined with '\n'; statuses joined with '\n' into buf; returns the number of successes
int av1mi_host_job_pool(const char *sources, int workers, int ngpus, double ratio, const char *state_dir, char *buf, int cap) {
  std::vector<Job> jobs; std::string s = sources; size_t p = 0, q;
  auto add = [&](const std::string &path) {
    Job j; j.ID = "pool" + std::to_string(jobs.size()); j.SourcePath = path;
    FILE *f = fopen(path.c_str(), "rb");
    if (f) { fseek(f, 0, SEEK_END); j.OriginalSize = ftell(f); fclose(f); }
    jobs.push_back(j);
  };
  while ((q = s.find('\n', p)) != std::string::npos) { add(s.substr(p, q - p)); p = q + 1; }
  add(s.substr(p));
  TranscodeConfig cfg; cfg.MaxSizeRatio = ratio; cfg.JobStateDir = state_dir ? state_dir : ""; cfg.StableWaitSeconds = 0;
  ProbeResult pr; pr.HasVideo = true; pr.has_video_stream = true; pr.VideoStream.Height = 720;
  std::vector<std::string> errs;
  const PoolStats ps = RunJobPool(&jobs, workers, ngpus, pr, cfg, &errs);
  std::string out;
  for (size_t i = 0; i < jobs.size(); i++) out += (i ? "\n" : "") + jobs[i].Status + (errs[i].empty() ? "" : ": " + errs[i]);
  strncpy(buf, out.c_str(), cap - 1); buf[cap - 1] = 0;
  return ps.succeeded;
}
}
