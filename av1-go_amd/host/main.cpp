// av1mi_transcode — CLI with the reference's process contract (exit code, stderr text, output file last):
//   av1mi_transcode [ffmpeg-style args] -i in.y4m [-global_quality:v:0 Q] [-g GOP] out.av1-tmp.mkv
//   av1mi_transcode --job in.y4m [--ratio 0.9] [--state DIR] [--wait S] [--replace-source 1]   (the ProcessJob lifecycle;
//                   the source is only replaced on request: the output is video-only)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/stat.h>
#include "daemon.hpp"

using namespace av1mi_host;

int main(int argc, char **argv) {
  if (argc >= 2 && !strcmp(argv[1], "--gpu-usage")) {     // the AMD twin of internal/tui/gpu.go getGPUUsage
    printf("%.1f\n", GetGPUUsage(argc >= 3 ? atoi(argv[2]) : 0));
    return 0;
  }
  if (argc >= 3 && !strcmp(argv[1], "--job")) {
    Job job; job.ID = "cli"; job.SourcePath = argv[2];
    TranscodeConfig cfg; cfg.StableWaitSeconds = 0;
    for (int i = 3; i + 1 < argc; i += 2) {
      if (!strcmp(argv[i], "--ratio")) cfg.MaxSizeRatio = atof(argv[i + 1]);
      else if (!strcmp(argv[i], "--state")) cfg.JobStateDir = argv[i + 1];
      else if (!strcmp(argv[i], "--wait")) cfg.StableWaitSeconds = atoi(argv[i + 1]);
      else if (!strcmp(argv[i], "--replace-source")) { cfg.ReplaceSource = atoi(argv[i + 1]) != 0; }
    }
    struct stat st;
    if (!stat(job.SourcePath.c_str(), &st)) job.OriginalSize = st.st_size;
    ProbeResult pr; pr.HasVideo = true; pr.has_video_stream = true; pr.VideoStream.Height = 1080;
    const std::string e = ProcessJob(&job, "av1mi", pr, cfg);
    fprintf(stderr, "job %s: %s%s%s\n", job.Status.c_str(), job.Reason.c_str(), e.empty() ? "" : " | ", e.c_str());
    return e.empty() ? 0 : 1;
  }
  if (argc >= 3 && !strcmp(argv[1], "--jobs")) {   // --jobs a.y4m b.y4m ... [--gpus N] [--workers W] [--ratio R] [--state DIR]
    std::vector<Job> jobs;
    TranscodeConfig cfg; cfg.StableWaitSeconds = 0;
    int gpus = 1, workers = 0;
    for (int i = 2; i < argc; i++) {
      if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
      else if (!strcmp(argv[i], "--workers") && i + 1 < argc) workers = atoi(argv[++i]);
      else if (!strcmp(argv[i], "--ratio") && i + 1 < argc) cfg.MaxSizeRatio = atof(argv[++i]);
      else if (!strcmp(argv[i], "--state") && i + 1 < argc) cfg.JobStateDir = argv[++i];
      else {
        Job j; j.ID = "job" + std::to_string(jobs.size()); j.SourcePath = argv[i];
        struct stat st;
        if (!stat(argv[i], &st)) j.OriginalSize = st.st_size;
        jobs.push_back(j);
      }
    }
    ProbeResult pr; pr.HasVideo = true; pr.has_video_stream = true; pr.VideoStream.Height = 1080;
    std::vector<std::string> errs;
    const PoolStats ps = RunJobPool(&jobs, workers > 0 ? workers : gpus, gpus, pr, cfg, &errs);
    for (size_t i = 0; i < jobs.size(); i++)
      fprintf(stderr, "%s %s: %s%s%s\n", jobs[i].ID.c_str(), jobs[i].Status.c_str(), jobs[i].Reason.c_str(), errs[i].empty() ? "" : " | ", errs[i].c_str());
    fprintf(stderr, "%d succeeded, %d skipped, %d failed in %.2f s\n", ps.succeeded, ps.skipped, ps.failed, ps.seconds);
    return ps.failed ? 1 : 0;
  }
  std::vector<std::string> args(argv + 1, argv + argc);
  const RunResult rr = RunTranscode(argv[0], args);
  if (rr.exitCode != 0) fprintf(stderr, "%s\n", rr.err.c_str());
  return rr.exitCode < 0 ? 255 : rr.exitCode;
}
