// capi_host.cpp — extern "C" hooks over the host mirror so that the CPU test-suite (ctypes) can pin it against
// values read off the reference source.
#include <cstring>
#include "daemon.hpp"

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
int av1mi_host_process_job(const char *source, long long orig_size, double ratio, const char *state_dir, int wait_s, char *status,
                           char *reason, int cap) {
  Job job; job.ID = "test"; job.SourcePath = source; job.OriginalSize = orig_size;
  TranscodeConfig cfg; cfg.MaxSizeRatio = ratio; cfg.JobStateDir = state_dir ? state_dir : ""; cfg.StableWaitSeconds = wait_s;
  ProbeResult pr; pr.HasVideo = true; pr.has_video_stream = true; pr.VideoStream.Height = 720;
  const std::string e = ProcessJob(&job, "av1mi", pr, cfg);
  strncpy(status, job.Status.c_str(), cap - 1); status[cap - 1] = 0;
  strncpy(reason, job.Reason.c_str(), cap - 1); reason[cap - 1] = 0;
  return e.empty() ? 0 : 1;
}
}
