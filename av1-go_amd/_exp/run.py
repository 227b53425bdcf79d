# timing experiment only (wrong pixels): k_inter_pipe with one phase removed
import sys, os, json
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, ".."))
sys.path.insert(0, os.path.join(here, "..", ".."))
import av1mi
n = sys.argv[1]
if n != "0":
    av1mi.LIB_PATH = os.path.join(here, "libav1mi_%s.so" % n)
import pipeline
with av1mi.Context(0) as ctx:
    gp = pipeline.GopPipeline(ctx, 3840, 2160, 10, segments=12, gop=4, qindex=128, first_frame=1, search_range=8)
    ctx.prof_enable(1) if hasattr(ctx, "prof_enable") else None
    for _ in range(2):
        gp.step()
    ctx.sync()
    ctx.lib.av1mi_prof_reset(ctx.h)
    for _ in range(3):
        gp.step()
    ctx.sync()
    import ctypes as C
    ln, ms = C.c_int(), C.c_double()
    for kind in range(16):
        ctx.lib.av1mi_kernel_kind_name.restype = C.c_char_p
        ctx.lib.av1mi_prof_get(ctx.h, kind, C.byref(ln), C.byref(ms))
        if ln.value:
            print(n, ctx.lib.av1mi_kernel_kind_name(kind).decode(), ln.value, "%.4f ms avg" % (ms.value / ln.value))
