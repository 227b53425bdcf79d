// backend.hpp — the part of RunTranscode that replaces the ffmpeg child: raw frames in, coded segments out,
// everything per-pixel on the GPU through the C ABI of libav1mi.so.
#pragma once
#include <string>
#include "transcode.hpp"

namespace av1mi_host {
// 0 = OK and job.output written; > 0 = failed after starting (bad input, I/O, device error mid-run);
// < 0 = could not run (no HIP device / library unusable).  *err carries the text.
int RunBackend(const BackendJob &job, std::string *err);
}
