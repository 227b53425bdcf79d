// transcode.cpp — see transcode.hpp.  Argument vector and error strings follow internal/ffmpeg/transcode.go.
#include "transcode.hpp"
#include "backend.hpp"
#include <cstdio>

namespace av1mi_host {

int DetermineQuality(int height) {   // transcode.go:157-165
  if (height >= 1440) return 23;
  if (height >= 1080) return 24;
  return 25;
}
std::string determineSurfaceFormat(int bitDepth) { return bitDepth >= 10 ? "p010" : "nv12"; }
std::string joinFilterParts(const std::vector<std::string> &parts) {
  std::string out;
  for (size_t i = 0; i < parts.size(); i++) out += (i ? "," : "") + parts[i];
  return out;
}

bool TranscodeArgs(const std::string &ffmpegPath, const std::string &inputPath, const std::string &outputPath,
                   const ProbeResult &probeResult, bool isWebRipLike, std::vector<std::string> *args, std::string *err) {
  (void)ffmpegPath;   // unused upstream as well
  if (!probeResult.has_video_stream) {
    if (err) *err = "no video stream found in probe result";   // transcode.go:19
    return false;
  }
  std::vector<std::string> &a = *args;
  a = { "-hide_banner", "-analyzeduration", "50M", "-probesize", "50M",                        // :40-44
        "-init_hw_device", "vaapi=va", "-hwaccel", "vaapi", "-hwaccel_output_format", "vaapi",   // :48-52
        "-filter_hw_device", "va" };
  if (isWebRipLike) for (const char *s : { "-fflags", "+genpts", "-copyts", "-start_at_zero" }) a.push_back(s);   // :59-65
  a.push_back("-i"); a.push_back(inputPath);                                                   // :68
  for (const char *s : { "-map", "0", "-map", "-0:v", "-map", "-0:t" }) a.push_back(s);         // :71-75
  a.push_back("-map"); a.push_back("0:v:" + std::to_string(probeResult.VideoStream.Index));     // :76
  for (const char *s : { "-map", "0:a?", "-map", "-0:a:m:language:rus", "-map", "-0:a:m:language:ru", "-map", "0:s?",
                         "-map", "-0:s:m:language:rus", "-map", "-0:s:m:language:ru", "-map_chapters", "0" })
    a.push_back(s);                                                                              // :77-83
  const int quality = DetermineQuality(probeResult.VideoStream.Height);                          // :86
  std::vector<std::string> vf;
  if (isWebRipLike) vf.push_back("scale_vaapi=w='if(gt(iw,iw*sar),iw,iw*sar)':h='if(gt(iw,iw*sar),iw/sar,ih)'");   // :96
  for (const char *s : { "scale_vaapi=w=ceil(iw/2)*2:h=ceil(ih/2)*2", "hwdownload,format=nv12", "setsar=1", "format=nv12", "hwupload" })
    vf.push_back(s);                                                                             // :97-112
  a.push_back("-vf:v:0"); a.push_back(joinFilterParts(vf));                                      // :115
  a.push_back("-c:v:0"); a.push_back("av1_vaapi");                                               // :120
  a.push_back("-global_quality:v:0"); a.push_back(std::to_string(quality));                      // :121
  a.push_back("-compression_level"); a.push_back("2");                                           // :122
  if (isWebRipLike) for (const char *s : { "-vsync", "0", "-avoid_negative_ts", "make_zero" }) a.push_back(s);     // :126-131
  for (const char *s : { "-c:a", "copy", "-c:s", "copy", "-max_muxing_queue_size", "2048", "-map_metadata", "0", "-f", "matroska",
                         "-movflags", "+faststart" })
    a.push_back(s);                                                                              // :134-145
  a.push_back(outputPath);                                                                       // :148
  return true;
}

bool ParseBackendJob(const std::vector<std::string> &args, BackendJob *job, std::string *err) {
  if (args.size() < 3) { if (err) *err = "Invalid argument: too few arguments"; return false; }
  job->output = args.back();
  bool have_in = false;
  for (size_t i = 0; i + 1 < args.size(); i++) {
    if (args[i] == "-i") { job->input = args[i + 1]; have_in = true; }
    else if (args[i] == "-global_quality:v:0") job->quality = std::atoi(args[i + 1].c_str());
    else if (args[i] == "-g") job->gop = std::atoi(args[i + 1].c_str());
    else if (args[i] == "-av1mi_device") job->device = std::atoi(args[i + 1].c_str());
    else if (args[i] == "-av1mi_segments") job->segments = std::atoi(args[i + 1].c_str());
    else if (args[i] == "-threads") job->threads = std::atoi(args[i + 1].c_str());
    else if (args[i] == "-av1mi_gpu_entropy") job->gpu_entropy = std::atoi(args[i + 1].c_str()) != 0;
    else if (args[i] == "-av1mi_tracks") job->tracks.push_back(args[i + 1]);
    else if (args[i] == "-av1mi_key_block_size") job->key_block_size = std::atoi(args[i + 1].c_str());
  }
  if (!have_in) { if (err) *err = "Invalid argument: no input (-i) given"; return false; }
  if (job->quality < 0 || job->quality > 255 || job->gop < 1 || job->gop > 256 || job->segments < 1 || job->segments > 256 || job->threads < 0 || (job->key_block_size != 8 && job->key_block_size != 32)) { if (err) *err = "Invalid argument: quality/gop/key block size out of range"; return false; }
  return true;
}

RunResult RunTranscode(const std::string &backendPath, const std::vector<std::string> &args) {
  (void)backendPath;   // the library is linked, there is no child process to locate
  BackendJob job;
  std::string err;
  if (!ParseBackendJob(args, &job, &err)) return { 1, "av1mi failed with exit code 1: " + err };
  const int code = RunBackend(job, &err);
  if (code == 0) return { 0, "" };
  if (err.size() > 800) err = err.substr(0, 800) + "...";      // transcode.go:295-297
  if (code < 0) return { -1, "av1mi execution failed: " + err };   // transcode.go:311 (could not run)
  return { code, "av1mi failed with exit code " + std::to_string(code) + ": " + err };   // transcode.go:299
}

}  // namespace av1mi_host
