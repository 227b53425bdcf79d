// y4m.hpp — the source side of the transcode job: raw 4:2:0 frames in YUV4MPEG2 framing.  The reference hands its FFmpeg child a
// container (`-i <file>`, internal/ffmpeg/transcode.go:68) that the child demuxes and decodes itself; this backend takes what such a
// decoder emits, from a FILE or from a STREAM: "-" / "pipe:0" (stdin) or a FIFO, so that any decoder process can feed it
// (`ffmpeg -i movie.mkv -f yuv4mpegpipe - | av1mi_transcode -i - ... out.mkv`).
//
// The encoder codes S closed GOPs of G frames in lockstep, i.e. it needs frames g*G + t of S different GOPs at the same time.  A
// seekable file with fixed-size frame headers is read in place (pread, one thread per segment).  Anything else is read
// sequentially, one GROUP of S*G frames ahead of the encoder, into host memory (two group buffers: S*G frames each).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace av1mi_host {

class Y4mSource {
 public:
  ~Y4mSource();
  int w = 0, h = 0, bd = 8, fps_n = 30, fps_d = 1;
  // false + *err (FFmpeg-style text) on a missing file, a header that is not Y4M, an unsupported colourspace or size
  bool open(const std::string &path, std::string *err);
  bool seekable() const { return seekable_; }
  long known_frames() const { return seekable_ ? nframes_ : -1; }      // -1: a stream, the end is found by reading
  // the group of up to max_frames frames that starts at frame `first` (groups must be asked for in order, without gaps): returns how
  // many of them exist (0 = end of input), -1 on a malformed / truncated frame (*err)
  long prepare(long first, long max_frames, std::string *err);
  // frame i of the prepared group -> the three planes of the CODED size cw x ch (the true size rounded up to 8: the last column / row
  // replicated into the padding).  Thread-safe for distinct destinations.
  bool read(long i, int cw, int ch, unsigned char *Y, unsigned char *U, unsigned char *V) const;
  void close();

 private:
  FILE *f_ = nullptr;
  bool own_ = false, seekable_ = false;
  std::string path_;
  long hdr_len_ = 0, nframes_ = 0, group_first_ = 0;
  size_t frame_bytes_ = 0;
  // sequential mode: the current group and the one being read ahead
  std::vector<unsigned char> cur_, next_;
  long cur_n_ = 0, next_n_ = 0, next_first_ = 0;
  bool next_bad_ = false, eof_ = false;
  std::thread reader_;
  bool reader_running_ = false;
  void start_read_ahead(long first, long max_frames);
  long read_group(std::vector<unsigned char> &buf, long max_frames, bool *bad);
};

}  // namespace av1mi_host
