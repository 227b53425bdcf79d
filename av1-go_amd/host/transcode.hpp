// transcode.hpp — host-side mirror of the reference's transcode-job API for the MI355X backend.
//
// The reference is Go and the image has no Go toolchain, so the host side above the C ABI is C++
// with the same names, argument meaning and error behaviour as
//   internal/ffmpeg/transcode.go:17   TranscodeArgs(ffmpegPath, inputPath, outputPath, probeResult, isWebRipLike)
//   internal/ffmpeg/transcode.go:157  DetermineQuality(height)
//   internal/ffmpeg/transcode.go:194  RunTranscode(ffmpegPath, args) (int, error)
//   internal/metadata/probe.go:14-46  ProbeResult / StreamInfo (the fields the path consumes)
// RunTranscode here does not spawn ffmpeg: it drives libav1mi.so (include/av1mi.h) on raw frames.
#pragma once
#include <string>
#include <vector>

namespace av1mi_host {

struct StreamInfo {       // metadata.StreamInfo, probe.go:35-46 (consumed fields only)
  int Index = 0;
  std::string CodecName, CodecType;
  int Width = 0, Height = 0;
  std::string AvgFrameRate;
  int BitDepth = 0;
};
struct ProbeResult {      // metadata.ProbeResult, probe.go:14-22
  bool HasVideo = false, HasAV1 = false, IsWebRipLike = false;
  bool has_video_stream = false;   // VideoStream != nil
  StreamInfo VideoStream;
};

// transcode.go:157-165
int DetermineQuality(int height);
// transcode.go:174-179 (dead code upstream; kept because a 10-bit capable backend needs it)
std::string determineSurfaceFormat(int bitDepth);
// transcode.go:182-191
std::string joinFilterParts(const std::vector<std::string> &parts);

// transcode.go:17-151.  Returns false and sets *err ("no video stream found in probe result") when VideoStream is nil,
// otherwise fills `args` with exactly the argv the reference builds.
bool TranscodeArgs(const std::string &ffmpegPath, const std::string &inputPath, const std::string &outputPath,
                   const ProbeResult &probeResult, bool isWebRipLike, std::vector<std::string> *args, std::string *err);

// What the MI355X backend takes from that argv: input (last "-i"), output (last argument), quality
// ("-global_quality:v:0"), and the bit depth policy.  The reference forces 8-bit NV12 (transcode.go:99-110).
struct BackendJob {
  std::string input, output;
  int quality = 25;        // FFmpeg global_quality; av1_vaapi uses it directly as the AV1 base_q_idx [ext]
  int gop = 30;            // closed-GOP segment length
  int device = 0;
  int segments = 4;        // closed GOPs coded in lockstep (the GOP session's batch)
  int threads = 0;         // host threads for entropy coding; 0 = all cores
  int gpu_entropy = 1;     // 1 = the AV1 tile entropy coder runs on the GPU (the host only assembles frames); 0 = north_star's split:
                           // symbols are downloaded and coded on the host cores.  Same bytes either way.
  int key_block_size = 32; // -av1mi_key_block_size 8 | 32: key frames in 32x32 blocks where the coded width (the source's rounded up to 8) is a multiple of 32 (av1mi_gop_config.key_block_size;
                           // +3.7 dB at equal size on the synthetic key frames at q 128 for ~9 % of the throughput), else 8x8 like every other frame
  std::vector<std::string> tracks;   // -av1mi_tracks <file.mka> (repeatable): Matroska side files whose audio / subtitle tracks are copied
                                     // next to the video (the reference's `-c:a copy -c:s copy`, transcode.go:134-137, after an external demux)
};
bool ParseBackendJob(const std::vector<std::string> &args, BackendJob *job, std::string *err);

// transcode.go:194-315 contract: (0, "") on success AND the output file exists; (code, text <= 800 chars) on failure;
// (-1, text) when the backend could not run at all (no HIP device, library error before any frame).
struct RunResult { int exitCode; std::string err; };
RunResult RunTranscode(const std::string &backendPath, const std::vector<std::string> &args);

}  // namespace av1mi_host
