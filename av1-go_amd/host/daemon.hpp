// daemon.hpp — mirror of internal/daemon/daemon.go (the only caller of the transcode API) and of the job record
// fields it touches (internal/jobs/jobs.go:25-46).  File-stability wait (internal/scan/scan.go:13) included.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "transcode.hpp"

namespace av1mi_host {

bool CheckSizeGate(int64_t origBytes, int64_t newBytes, double maxRatio);                 // daemon.go:18-21
bool AtomicReplaceFile(const std::string &originalPath, const std::string &newPath, std::string *err);   // daemon.go:25-53
bool CheckFileStable(const std::string &path, int waitSeconds, bool *stable, std::string *err);          // scan.go:13-33

struct Job {                       // jobs.Job, jobs.go:25-46: all 20 fields, so that the records av1top reads stay complete
  std::string ID, SourcePath, OutputPath, Status = "pending", Reason;
  std::string CreatedAt, StartedAt, FinishedAt;      // RFC 3339 (time.Time in the reference); empty = omitted like a nil pointer
  int64_t OriginalSize = 0, NewSize = 0, EstimatedSize = 0;
  bool IsWebRipLike = false;
  std::string SourceCodec, Resolution, FrameRate, Container, VideoCodec;
  int BitDepth = 0, AudioStreams = 0, SubStreams = 0;
};
// jobs.SaveJob (jobs.go:61-79): <dir>/<id>.json, two-space indent, field names and omitempty rules of the struct tags
void SaveJob(const Job &j, const std::string &jobsDir);
std::string JobToJSON(const Job &j);
// RFC 3339 UTC timestamp of now
std::string NowRFC3339();

// AMD replacement of the reference's Intel-only GPU utilisation probe (internal/tui/gpu.go:16 getGPUUsage reads i915's
// rps_act_freq_mhz / rps_max_freq_mhz): amdgpu exposes the busy percentage directly in sysfs,
// /sys/class/drm/card<N>/device/gpu_busy_percent.  Returns 0..100, or -1 when no amdgpu card exposes it.
// `sysfs_root` is "/sys" outside tests.
double GetGPUUsage(int device = 0, const std::string &sysfs_root = "/sys");
struct TranscodeConfig {           // daemon.go:185-188; Device is this backend's addition (which GPU runs the job)
  std::string JobStateDir; double MaxSizeRatio = 0.90; int StableWaitSeconds = 10; int Device = 0;
  // The reference renames its output over the source (daemon.go:154): there the output carries the copied audio and
  // subtitle streams.  This backend's output is a VIDEO-ONLY AV1 file (no demuxer, no stream copy), so replacing the source
  // would destroy its other streams: the step is off unless explicitly asked for, and the coded file is then kept beside
  // the source as "<base>.av1mi.mkv".
  bool ReplaceSource = false;
};

// daemon.go:57-182.  Returns "" where the reference returns nil, else the error text; job.Status / job.Reason are
// updated exactly as upstream ("running" -> "success" | "failed" | "skipped").
std::string ProcessJob(Job *job, const std::string &backendPath, const ProbeResult &probeResult, const TranscodeConfig &cfg);

// The job loop of cmd/av1d/main.go:291-349, which the reference runs serially on one device, as a pool of `workers`
// threads (BASELINE config 5: 8 concurrent jobs, one per GPU).  Worker i owns GPU
// i % ngpus (its own av1mi context and stream, nothing shared, no GPU<->GPU traffic) and pulls the next pending job.
// Every job goes through ProcessJob unchanged, so the job JSON files keep being written the way av1top expects.
// errors[i] = what ProcessJob returned for jobs[i].  Returns the number of jobs that ended in "success".
struct PoolStats { int succeeded = 0, skipped = 0, failed = 0; double seconds = 0; };
PoolStats RunJobPool(std::vector<Job> *jobs, int workers, int ngpus, const ProbeResult &probeResult, const TranscodeConfig &cfg,
                     std::vector<std::string> *errors);

}  // namespace av1mi_host
