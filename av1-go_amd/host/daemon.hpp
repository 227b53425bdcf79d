// daemon.hpp — mirror of internal/daemon/daemon.go (the only caller of the transcode API) and of the job record
// fields it touches (internal/jobs/jobs.go:25-46).  File-stability wait (internal/scan/scan.go:13) included.
#pragma once
#include <cstdint>
#include <string>
#include "transcode.hpp"

namespace av1mi_host {

bool CheckSizeGate(int64_t origBytes, int64_t newBytes, double maxRatio);                 // daemon.go:18-21
bool AtomicReplaceFile(const std::string &originalPath, const std::string &newPath, std::string *err);   // daemon.go:25-53
bool CheckFileStable(const std::string &path, int waitSeconds, bool *stable, std::string *err);          // scan.go:13-33

struct Job {                       // jobs.Job, the fields ProcessJob reads or writes
  std::string ID, SourcePath, OutputPath, Status = "pending", Reason;
  int64_t OriginalSize = 0, NewSize = 0;
  bool IsWebRipLike = false;
};
struct TranscodeConfig { std::string JobStateDir; double MaxSizeRatio = 0.90; int StableWaitSeconds = 10; };   // daemon.go:185-188

// daemon.go:57-182.  Returns "" where the reference returns nil, else the error text; job.Status / job.Reason are
// updated exactly as upstream ("running" -> "success" | "failed" | "skipped").
std::string ProcessJob(Job *job, const std::string &backendPath, const ProbeResult &probeResult, const TranscodeConfig &cfg);

}  // namespace av1mi_host
