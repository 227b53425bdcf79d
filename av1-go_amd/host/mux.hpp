// mux.hpp — container side of the transcode job: where the coded temporal units go.  The reference muxes with FFmpeg
// (`-f matroska`, internal/ffmpeg/transcode.go:140-145); FFmpeg is not in this image, so the three containers an AV1 decoder
// or player reads directly are written here: Section-5 OBU stream (.obu), IVF (.ivf) and Matroska with one V_AV1 video track
// (everything else, i.e. the reference's "<base>.av1-tmp.mkv").  Audio / subtitle stream copy (transcode.go:71-83,134-137)
// needs a demuxer for the source and is NOT done: the Matroska file is video only.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "av1_bitstream.hpp"

namespace av1mi_host {

// a frame description whose restoration-unit arrays are owned here (backend.cpp DescribeSessionFrame)
struct SessionFrameDesc { av1mi_obu_frame f; std::vector<int8_t> lr_y, lr_uv; };

class StreamSink {
 public:
  bool open(const std::string &path, const av1::SequenceParams &sp, int fps_n, int fps_d, std::string *err);
  // one temporal unit (temporal delimiter first) in presentation order
  bool write(const std::vector<uint8_t> &temporal_unit, bool key, std::string *err);
  bool close(std::string *err);     // finishes headers / indexes
  void abort();                     // closes the file without finishing (no-op after close)
 private:
  enum Kind { OBU, IVF, MKV } kind_ = OBU;
  FILE *f_ = nullptr;
  std::string path_;
  int fps_n_ = 30, fps_d_ = 1;
  long frames_ = 0;
  // Matroska state
  long seg_data_start_ = 0, duration_pos_ = 0, cluster_start_ = 0, cluster_size_pos_ = 0;
  long cluster_time_ms_ = 0;
  bool cluster_open_ = false;
  std::vector<std::pair<long, long>> cues_;    // (time ms, cluster position relative to the segment data)
  bool put(const void *p, size_t n, std::string *err);
  void close_cluster();
};

}  // namespace av1mi_host
