// backend.hpp — the part of RunTranscode that replaces the ffmpeg child: raw frames in, an AV1 stream out; everything
// per-pixel on the GPU through the GOP session of libav1mi.so, entropy coding + OBU packing on the host cores.
#pragma once
#include <string>
#include "../../include/av1mi.h"
#include "mux.hpp"
#include "transcode.hpp"

namespace av1mi_host {
// 0 = OK and job.output written; > 0 = failed after starting (bad input, I/O, device error mid-run);
// < 0 = could not run (no HIP device / library unusable).  *err carries the text.
int RunBackend(const BackendJob &job, std::string *err);
// the bitstream writer's frame description for segment `seg` of a collected session batch (pointers into the batch)
// visible_*: the true frame size when width x height is it rounded up to 8 (0 = the coded size)
void DescribeSessionFrame(const av1mi_gop_frame &fr, int seg, int width, int height, int bit_depth, SessionFrameDesc *d, int visible_width = 0,
                          int visible_height = 0);
// the temporal unit of segment `seg` of a collected batch: GPU-coded tiles wrapped, or the symbols coded on `threads` host threads
bool SessionTemporalUnit(const av1mi_gop_frame &fr, int seg, int width, int height, int bit_depth, int visible_width, int visible_height,
                         bool with_sequence_header, int threads, std::vector<uint8_t> *out, std::string *err);
}
