"""CPU tests of the drop-in boundary: libav1mi.so loads, exports every symbol include/av1mi.h
declares, refuses to run without a GPU (no CPU fallback), and agrees with the oracle on which
(size,type) pairs exist.  No compute calls here."""
import ctypes as C
import os

import pytest


def test_library_exports_every_declared_symbol(av1mi):
    lib = av1mi.load()
    names = av1mi.exported_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libav1mi.so does not export %s" % n


def test_host_library_exports_every_declared_symbol():
    """include/av1mi_host.h: the RunTranscode drop-in and the bitstream writer, exported by libav1mi_host.so"""
    import re
    import av1stream
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "av1mi_host.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(av1mi_[a-z0-9_]+)\s*\(", hdr)))
    assert names == ["av1mi_obu_assemble_temporal_unit", "av1mi_obu_write_blocks_temporal_unit", "av1mi_obu_write_temporal_unit", "av1mi_run_transcode",
                     "av1mi_session_temporal_unit"]
    lib = av1stream.lib()
    for n in names:
        assert hasattr(lib, n), "libav1mi_host.so does not export %s" % n
    # the ctypes mirror of av1mi_obu_frame has the size the C compiler gives the header's struct
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "sz.c")
        open(src, "w").write('#include "av1mi_host.h"\n#include <stdio.h>\nint main(void){printf("%zu %zu %zu", sizeof(av1mi_obu_frame), '
                             'sizeof(av1mi_obu_block), sizeof(av1mi_obu_blocks));return 0;}\n')
        subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), src, "-o", os.path.join(d, "sz")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "sz")]).split()]
        assert sizes == [C.sizeof(av1stream.ObuFrame), av1stream.BLOCK_DTYPE.itemsize, C.sizeof(av1stream.ObuBlocks)]


def test_policy_is_exported_and_frame_type_dependent(av1mi):
    """the filter-parameter policy lives in libav1mi.so only (round 1 kept two copies that disagreed on the P-frame level)"""
    k, p = av1mi.policy_frame_params(128, 8, 0), av1mi.policy_frame_params(128, 8, 1)
    assert list(k.lf_level) == [10] * 4 and list(p.lf_level) == [7] * 4      # libaom's key / inter fits for 8-bit
    # CDEF: libaom's pick-from-q fits, one for intra-only and one for inter frames (weaker, no secondary strength at this q)
    assert (k.cdef_y >> 2, k.cdef_y & 3) == (2, 1) and (p.cdef_y >> 2, p.cdef_y & 3) == (1, 0) and k.cdef_damping == p.cdef_damping == 5
    assert list(k.lr_unit_y) == [1, 3, -7, 15, 3, -7, 15, 0]
    k10, p10 = av1mi.policy_frame_params(128, 10, 0), av1mi.policy_frame_params(128, 10, 1)
    assert list(k10.lf_level) == list(p10.lf_level)


def test_version_and_geometry(av1mi, O):
    lib = av1mi.load()
    assert lib.av1mi_version().decode().startswith("av1mi ")
    for ts in range(19):
        assert lib.av1mi_tx_width(ts) == O.TX_W[ts] and lib.av1mi_tx_height(ts) == O.TX_H[ts]
        for tt in range(16):
            assert bool(lib.av1mi_txfm_valid(ts, tt)) == O.txfm_valid(ts, tt)
    assert lib.av1mi_tx_width(19) == 0 and lib.av1mi_txfm_valid(-1, 0) == 0


def test_qtables_match_oracle(av1mi, O):
    lib = av1mi.load()
    for bd in (8, 10):
        for q in range(256):
            assert lib.av1mi_dc_q(q, bd) == O.dc_q(q, bd) and lib.av1mi_ac_q(q, bd) == O.ac_q(q, bd)


def test_no_cpu_fallback_without_gpu(av1mi):
    lib = av1mi.load()
    if lib.av1mi_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert lib.av1mi_open(0, C.byref(h)) == -4 and not h.value  # AV1MI_E_NODEV
    with pytest.raises(av1mi.Av1miError):
        av1mi.Context(0)


def test_null_context_is_rejected(av1mi):
    lib = av1mi.load()
    assert lib.av1mi_sync(None) == -1
    assert lib.av1mi_last_error(None) == b"null context"


@pytest.mark.gpu
def test_device_copy_and_memset(ctx):
    """av1mi_copy / av1mi_memset on the context's stream: bytes arrive, null pointers are refused with an error text"""
    import numpy as np
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, 1 << 20, dtype=np.uint8)
    d_a, d_b = ctx.to_device(a), ctx.alloc(a.nbytes)
    ctx.memset(d_b, 7, a.nbytes)
    assert (d_b.download(a.shape, a.dtype) == 7).all()
    ctx.copy(d_b, d_a, a.nbytes)
    assert (d_b.download(a.shape, a.dtype) == a).all()
    rc = ctx.lib.av1mi_copy(ctx.h, None, None, C.c_size_t(16))
    assert rc < 0 and b"null" in ctx.lib.av1mi_last_error(ctx.h)
    d_a.free()
    d_b.free()


def test_off_screen_deblocking_units_are_marked_unfiltered():
    """policy_arrays(visible=...): 4x4 units that start at or beyond the true size carry "skipped inter block, no block edge" (spec
    7.14.2 onScreen; a chroma unit is 8 luma samples wide), the others the frame's levels — what the session builds in C++"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "av1-go_amd"))
    import pipeline as P
    a = P.policy_arrays(120, 8, 1, 104, 80, visible=(98, 75))
    y, c = a["mi_y"], a["mi_c"]
    on_y, on_c = y[0, 0], c[0, 0]
    assert (on_y >> 24) == 6 and (on_c >> 24) == 6              # both edges are block edges, not skipped
    assert (y[:19, :25] == on_y).all() and (y[19:, :] >> 24 == 1).all() and (y[:, 25:] >> 24 == 1).all()      # 4 * 19 = 76 >= 75, 4 * 25 = 100 >= 98
    assert (c[:10, :13] == on_c).all() and (c[10:, :] >> 24 == 1).all() and (c[:, 13:] >> 24 == 1).all()      # 8 * 10 = 80 >= 75, 8 * 13 = 104 >= 98
    b = P.policy_arrays(120, 8, 1, 104, 80)
    assert (b["mi_y"] == on_y).all() and (b["mi_c"] == on_c).all()
