// daemon.cpp — see daemon.hpp.
#include "daemon.hpp"
#include <atomic>
#include <chrono>
#include <thread>
#include <sys/stat.h>
#include <unistd.h>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>

namespace av1mi_host {
namespace {
bool stat_size(const std::string &p, int64_t *sz) { struct stat st; if (stat(p.c_str(), &st)) return false; *sz = st.st_size; return true; }
void split(const std::string &path, std::string *dir, std::string *stem, std::string *ext) {
  const size_t sl = path.find_last_of('/');
  *dir = sl == std::string::npos ? "." : path.substr(0, sl);
  const std::string base = sl == std::string::npos ? path : path.substr(sl + 1);
  const size_t dot = base.find_last_of('.');
  *stem = dot == std::string::npos || dot == 0 ? base : base.substr(0, dot);
  *ext = dot == std::string::npos || dot == 0 ? "" : base.substr(dot);
}
void write_text(const std::string &path, const std::string &text) { std::ofstream f(path); f << text; }
std::string json_escape(const std::string &s) {           // what encoding/json does for the strings of jobs.SaveJob
  std::string o;
  for (unsigned char c : s) {
    if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
    else if (c == '\n') o += "\\n";
    else if (c == '\r') o += "\\r";
    else if (c == '\t') o += "\\t";
    else if (c < 0x20) { char b[8]; snprintf(b, sizeof(b), "\\u%04x", c); o += b; }
    else o += (char)c;
  }
  return o;
}
void save_job(Job &j, const std::string &dir) {      // terminal states carry finished_at (daemon.go sets it before every final SaveJob)
  if ((j.Status == "success" || j.Status == "failed" || j.Status == "skipped") && j.FinishedAt.empty()) j.FinishedAt = NowRFC3339();
  SaveJob(j, dir);
}
void write_why(const std::string &source, const std::string &reason) {   // metadata.WriteWhyFile, probe.go:398
  std::string dir, stem, ext; split(source, &dir, &stem, &ext);
  write_text(dir + "/" + stem + ".av1qsvd-why.txt", reason);
}
}  // namespace

std::string NowRFC3339() {
  char buf[40];
  const time_t t = time(nullptr);
  struct tm tmv;
  gmtime_r(&t, &tmv);
  strftime(buf, sizeof(buf), "%Y-%m-%dT%H:%M:%SZ", &tmv);
  return buf;
}
std::string JobToJSON(const Job &j) {     // encoding/json with the struct tags of jobs.go:25-46 (omitempty where tagged)
  std::string o = "{\n";
  bool first = true;
  auto sep = [&] { if (!first) o += ",\n"; first = false; };
  auto str = [&](const char *k, const std::string &v, bool omitempty) { if (omitempty && v.empty()) return; sep(); o += std::string("  \"") + k + "\": \"" + json_escape(v) + "\""; };
  auto num = [&](const char *k, int64_t v, bool omitempty) { if (omitempty && v == 0) return; sep(); o += std::string("  \"") + k + "\": " + std::to_string(v); };
  str("id", j.ID, false); str("source_path", j.SourcePath, false); str("output_path", j.OutputPath, true);
  str("created_at", j.CreatedAt.empty() ? "0001-01-01T00:00:00Z" : j.CreatedAt, false);      // time.Time zero value
  str("started_at", j.StartedAt, true); str("finished_at", j.FinishedAt, true);
  str("status", j.Status, false); str("reason", j.Reason, true);
  num("original_bytes", j.OriginalSize, true); num("new_bytes", j.NewSize, true); num("estimated_bytes", j.EstimatedSize, true);
  sep(); o += std::string("  \"is_webrip_like\": ") + (j.IsWebRipLike ? "true" : "false");
  str("source_codec", j.SourceCodec, true); str("resolution", j.Resolution, true); num("bit_depth", j.BitDepth, true);
  str("frame_rate", j.FrameRate, true); str("container", j.Container, true); str("video_codec", j.VideoCodec, true);
  num("audio_streams", j.AudioStreams, true); num("subtitle_streams", j.SubStreams, true);
  return o + "\n}";
}
void SaveJob(const Job &j, const std::string &dir) {
  if (dir.empty() || j.ID.empty()) return;
  mkdir(dir.c_str(), 0755);              // os.MkdirAll, jobs.go:63
  std::ofstream f(dir + "/" + j.ID + ".json");
  f << JobToJSON(j);
}
double GetGPUUsage(int device, const std::string &sysfs_root) {
  // cards are numbered in probe order; the device-th card that has the amdgpu busy file is the device-th HIP device on a box
  // whose only GPUs are the MI355X (render-only nodes carry no cardN entry and are skipped)
  int seen = 0;
  for (int card = 0; card < 64; card++) {
    std::ifstream f(sysfs_root + "/class/drm/card" + std::to_string(card) + "/device/gpu_busy_percent");
    if (!f) continue;
    double v = -1;
    f >> v;
    if (seen++ == device) return v < 0 ? -1 : v > 100 ? 100 : v;
  }
  return -1;
}

bool CheckSizeGate(int64_t origBytes, int64_t newBytes, double maxRatio) { return (double)newBytes <= (double)origBytes * maxRatio; }

bool AtomicReplaceFile(const std::string &originalPath, const std::string &newPath, std::string *err) {
  std::string dir, stem, ext; split(originalPath, &dir, &stem, &ext);
  const std::string tmp = dir + "/" + stem + ".av1-tmp.mkv";
  if (newPath != tmp && rename(newPath.c_str(), tmp.c_str())) { *err = std::string("failed to move new file to temp: ") + strerror(errno); return false; }
  int64_t sz;
  if (!stat_size(tmp, &sz)) { *err = std::string("temp file does not exist: ") + strerror(errno); return false; }
  if (rename(tmp.c_str(), originalPath.c_str())) { *err = std::string("failed to replace original file: ") + strerror(errno); return false; }
  return true;
}

bool CheckFileStable(const std::string &path, int waitSeconds, bool *stable, std::string *err) {
  int64_t s0, s1;
  if (!stat_size(path, &s0)) { *err = std::string("failed to stat file: ") + strerror(errno); return false; }
  if (waitSeconds > 0) sleep((unsigned)waitSeconds);
  if (!stat_size(path, &s1)) { *err = std::string("failed to stat file after wait: ") + strerror(errno); return false; }
  *stable = s0 == s1;
  return true;
}

std::string ProcessJob(Job *job, const std::string &backendPath, const ProbeResult &probeResult, const TranscodeConfig &cfg) {
  bool stable = false;
  std::string err;
  if (!CheckFileStable(job->SourcePath, cfg.StableWaitSeconds, &stable, &err)) return "failed to check file stability: " + err;   // :59-62
  if (!stable) { job->Status = "skipped"; job->Reason = "file still copying"; write_why(job->SourcePath, job->Reason); return ""; }       // :63-71
  job->Status = "running";                                                                                                             // :74-79
  job->StartedAt = NowRFC3339();
  if (job->CreatedAt.empty()) job->CreatedAt = job->StartedAt;
  job->VideoCodec = "av1";
  save_job(*job, cfg.JobStateDir);
  std::string dir, stem, ext; split(job->SourcePath, &dir, &stem, &ext);
  const std::string outputPath = dir + "/" + stem + ".av1-tmp.mkv";                                                                     // :82-87
  job->OutputPath = outputPath;
  std::vector<std::string> args;
  if (!TranscodeArgs(backendPath, job->SourcePath, outputPath, probeResult, job->IsWebRipLike, &args, &err)) {                          // :90-98
    job->Status = "failed"; job->Reason = "failed to build ffmpeg args: " + err; save_job(*job, cfg.JobStateDir);
    return "failed to build transcode args: " + err;
  }
  if (cfg.Device) { const std::string out = args.back(); args.back() = "-av1mi_device"; args.push_back(std::to_string(cfg.Device)); args.push_back(out); }
  const RunResult rr = RunTranscode(backendPath, args);                                                                                 // :101
  if (!rr.err.empty() || rr.exitCode != 0) {                                                                                            // :102-112
    job->Status = "failed"; job->Reason = "ffmpeg exit code " + std::to_string(rr.exitCode) + ": " + rr.err;
    save_job(*job, cfg.JobStateDir); write_why(job->SourcePath, job->Reason); remove(outputPath.c_str());
    return "transcode failed: " + rr.err;
  }
  int64_t newSize = 0;
  if (!stat_size(outputPath, &newSize)) {                                                                                               // :115-124
    job->Status = "failed"; job->Reason = std::string("failed to stat output file: ") + strerror(errno);
    save_job(*job, cfg.JobStateDir); remove(outputPath.c_str());
    return std::string("output file not found: ") + strerror(errno);
  }
  job->NewSize = newSize;
  if (!CheckSizeGate(job->OriginalSize, job->NewSize, cfg.MaxSizeRatio)) {                                                              // :129-150
    char buf[160];
    snprintf(buf, sizeof(buf), "size gate: new %.1f MB vs orig %.1f MB (>%.0f%%)", job->NewSize / 1048576.0, job->OriginalSize / 1048576.0,
             cfg.MaxSizeRatio * 100);
    job->Status = "skipped"; job->Reason = buf;
    write_why(job->SourcePath, job->Reason);
    write_text(dir + "/" + stem + ".av1qsvd-skip", "skip");
    remove(outputPath.c_str());
    save_job(*job, cfg.JobStateDir);
    return "";
  }
  if (!cfg.ReplaceSource) {
    // NOT the reference's step (daemon.go:154 renames the output over the source): this backend's file has no audio / subtitle
    // streams, so the source stays and the coded file is kept beside it.  See TranscodeConfig::ReplaceSource.
    const std::string kept = dir + "/" + stem + ".av1mi.mkv";
    if (rename(outputPath.c_str(), kept.c_str())) {
      job->Status = "failed"; job->Reason = std::string("failed to move output: ") + strerror(errno); save_job(*job, cfg.JobStateDir); remove(outputPath.c_str());
      return job->Reason;
    }
    job->OutputPath = kept;
    job->Status = "success"; job->Reason = "source kept: video-only output, replace step disabled";
    save_job(*job, cfg.JobStateDir);
    return "";
  }
  if (!AtomicReplaceFile(job->SourcePath, outputPath, &err)) {                                                                          // :154-163
    job->Status = "failed"; job->Reason = "failed to replace file: " + err; save_job(*job, cfg.JobStateDir); remove(outputPath.c_str());
    return "failed to replace file: " + err;
  }
  int64_t sz;
  if (!stat_size(job->SourcePath, &sz)) {                                                                                               // :166-172
    job->Status = "failed"; job->Reason = std::string("replaced file verification failed: ") + strerror(errno);
    save_job(*job, cfg.JobStateDir);
    return job->Reason;
  }
  job->Status = "success";                                                                                                             // :176-179
  save_job(*job, cfg.JobStateDir);
  return "";
}

PoolStats RunJobPool(std::vector<Job> *jobs, int workers, int ngpus, const ProbeResult &probeResult, const TranscodeConfig &cfg,
                     std::vector<std::string> *errors) {
  PoolStats st;
  if (workers < 1) workers = 1;
  if (ngpus < 1) ngpus = 1;
  errors->assign(jobs->size(), "");
  std::atomic<size_t> next{0};
  const auto t0 = std::chrono::steady_clock::now();
  auto work = [&](int w) {
    TranscodeConfig mine = cfg;
    mine.Device = w % ngpus;
    for (size_t i; (i = next.fetch_add(1)) < jobs->size();) (*errors)[i] = ProcessJob(&(*jobs)[i], "av1mi", probeResult, mine);
  };
  std::vector<std::thread> pool;
  for (int w = 1; w < workers; w++) pool.emplace_back(work, w);
  work(0);
  for (auto &t : pool) t.join();
  st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (const Job &j : *jobs) { st.succeeded += j.Status == "success"; st.skipped += j.Status == "skipped"; st.failed += j.Status == "failed"; }
  return st;
}

}  // namespace av1mi_host
