#!/usr/bin/env python3
"""GPU diagnostic: time of the tile entropy coder (K9) alone for several batch sizes / tile sizes / quantisers."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "av1-go_amd"))
import av1mi      # noqa: E402
import pipeline   # noqa: E402

ctx = av1mi.Context(0)
cases = [(48, 64, 128), (8, 64, 128), (1, 64, 128), (48, 32, 128), (48, 128, 128), (48, 64, 220), (48, 64, 60), (48, 32, 220)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for frames, tile, q in cases:
    pipe = pipeline.IntraPipeline(ctx, 1920, 1080, 8, frames, q, entropy_tile=tile)
    pipe.entropy_in_step = False
    pipe.step()
    ctx.entropy_encode(pipe.ent_job)
    ctx.sync()
    ctx.prof_reset()
    ctx.prof_enable(True)
    for _ in range(3):
        ctx.entropy_encode(pipe.ent_job)
    ctx.sync()
    ctx.prof_enable(False)
    prof = ctx.prof_get()
    recs = pipe.coded_records()
    ms = prof["entropy_code"][1] / prof["entropy_code"][0]
    tk = prof["entropy_tokens"][1] / prof["entropy_tokens"][0]
    print("frames %3d tile %3d q %3d: tokens %.3f ms  code %.3f ms  pack %.3f ms  %.0f fps  %.0f bytes/frame" %
          (frames, tile, q, tk, ms, prof["entropy_pack"][1] / prof["entropy_pack"][0], frames / (ms + tk) * 1e3, sum(map(len, recs)) / frames), flush=True)
    pipe.close()
ctx.close()
