// mux.hpp — container side of the transcode job: where the coded temporal units go.  The reference muxes with FFmpeg
// (`-f matroska`, internal/ffmpeg/transcode.go:140-145); FFmpeg is not in this image, so the three containers an AV1 decoder
// or player reads directly are written here: Section-5 OBU stream (.obu), IVF (.ivf) and Matroska with one V_AV1 video track
// (everything else, i.e. the reference's "<base>.av1-tmp.mkv").  Audio / subtitle STREAM COPY (transcode.go:71-83,134-137:
// `-map 0:a? -map 0:s? -c:a copy -c:s copy`) is done for tracks that were demuxed beforehand into Matroska side files
// (`ffmpeg -i movie.mkv -map 0:a? -map 0:s? -c copy side.mka`, or mkvmerge): add_side_file() copies every track of such a file —
// TrackEntry verbatim (codec id, codec private, language, audio / video settings), blocks re-timed into this file's clusters and
// interleaved with the video by timestamp, lacing and block durations kept.  Demuxing the source itself stays with FFmpeg.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "av1_bitstream.hpp"

namespace av1mi_host {

// a frame description whose restoration-unit arrays are owned here (backend.cpp DescribeSessionFrame)
struct SessionFrameDesc { av1mi_obu_frame f; std::vector<int8_t> lr_y, lr_uv; };

// sequential reader of a Matroska file's tracks and blocks (side files of add_side_file; the file is read as the output advances,
// never held in memory)
class MkvReader {
 public:
  struct Track { uint64_t number = 0; std::vector<uint8_t> entry; };      // entry: the TrackEntry's children except TrackNumber / TrackUID
  struct Block { uint64_t track = 0; int64_t t_ns = 0; int64_t duration_ns = -1; bool key = true; std::vector<uint8_t> tail; };   // tail: flags byte onwards
  ~MkvReader();
  bool open(const std::string &path, std::string *err);      // reads up to the first cluster: the tracks are known afterwards
  const std::vector<Track> &tracks() const { return tracks_; }
  bool next(Block *b, bool *end, std::string *err);           // blocks in file order
 private:
  struct Level { uint32_t id; long end; };                    // end < 0: unknown size
  FILE *f_ = nullptr;
  std::string path_;
  std::vector<Track> tracks_;
  std::vector<Level> stack_;
  uint64_t scale_ns_ = 1000000, cluster_ts_ = 0;
  bool header(uint32_t *id, int64_t *size, std::string *err);
  bool read_tracks(long end, std::string *err);
  bool fail(std::string *err, const char *what) const;
};

class StreamSink {
 public:
  // before open(): every track of a Matroska side file is copied next to the video track (Matroska output only)
  bool add_side_file(const std::string &path, std::string *err);
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
  // side tracks
  struct Side { MkvReader rd; std::vector<int> out_number; MkvReader::Block pending; bool have = false, done = false; };
  std::vector<Side *> sides_;
  long side_end_ms_ = 0;
  bool put(const void *p, size_t n, std::string *err);
  void close_cluster();
  bool start_cluster(long t_ms, bool cue, std::string *err);
  bool side_blocks_until(long t_ms, bool inclusive, std::string *err);     // copies the side files' blocks that are due
  bool put_side_block(Side &s, std::string *err);
 public:
  ~StreamSink();
};

}  // namespace av1mi_host
