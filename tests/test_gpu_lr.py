"""GPU parity for K7 (loop restoration: Wiener + self-guided) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from lf_util import test_image as make_image

pytestmark = pytest.mark.gpu


def _random_units(rng, O, unit, h, w):
    u = np.zeros((O.lr_units(unit, h), O.lr_units(unit, w), 8), np.int8)
    for r in range(u.shape[0]):
        for c in range(u.shape[1]):
            t = int(rng.integers(0, 3))
            if t == 1:
                u[r, c] = O.lr_unit_wiener((rng.integers(-5, 11), rng.integers(-23, 9), rng.integers(-17, 47)),
                                           (rng.integers(-5, 11), rng.integers(-23, 9), rng.integers(-17, 47)))
            elif t == 2:
                u[r, c] = O.lr_unit_sgr(int(rng.integers(0, 16)), int(rng.integers(-96, 32)), int(rng.integers(-32, 96)))
    return u


def _run(ctx, cdef, dbl, bd, ss, unit, units):
    nf, h, w = cdef.shape
    d_c, d_d, d_u = ctx.to_device(cdef), ctx.to_device(dbl), ctx.to_device(units)
    d_o = ctx.alloc(cdef.nbytes)
    ctx.lr_frames(d_c, d_d, d_o, w, w, h, bd, ss, unit, d_u, 0 if units.ndim == 3 else units.shape[1] * units.shape[2], nf)
    out = d_o.download(cdef.shape, cdef.dtype)
    for b in (d_c, d_d, d_u, d_o):
        b.free()
    return out


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("ss", [0, 1])
def test_lr_random_units(ctx, O, bd, ss):
    rng = np.random.default_rng(90 + bd + ss)
    for (h, w), unit in (((136, 200), 64), ((72, 100), 32 if ss else 64), ((300, 260), 128), ((40, 24), 64), ((130, 520), 256)):
        nf = 2
        cdef = np.stack([make_image(rng, h, w, bd) for _ in range(nf)])
        dbl = np.stack([make_image(rng, h, w, bd) for _ in range(nf)])   # deliberately different from cdef
        units = np.stack([_random_units(rng, O, unit, h, w) for _ in range(nf)])
        got = _run(ctx, cdef, dbl, bd, ss, unit, units)
        for f in range(nf):
            exp = O.lr_plane(cdef[f], dbl[f], bd, ss, unit, units[f])
            assert (got[f] == exp).all(), ((h, w), unit, bd, ss, f, np.argwhere(got[f] != exp)[:4])


def test_lr_every_sgr_set_and_extreme_wiener(ctx, O):
    rng = np.random.default_rng(95)
    h, w = 96, 16 * 64
    cdef = make_image(rng, h, w, 10)[None]
    dbl = make_image(rng, h, w, 10)[None]
    units = np.zeros((O.lr_units(64, h), 16, 8), np.int8)
    for s in range(16):
        units[:, s] = O.lr_unit_sgr(s, -96 if s & 1 else 31, 95 if s & 2 else -32)
    got = _run(ctx, cdef, dbl, 10, 0, 64, units)
    assert (got[0] == O.lr_plane(cdef[0], dbl[0], 10, 0, 64, units)).all()
    units[:] = O.lr_unit_wiener((10, 8, 46), (-5, -23, -17))
    got = _run(ctx, cdef, dbl, 10, 0, 64, units)
    assert (got[0] == O.lr_plane(cdef[0], dbl[0], 10, 0, 64, units)).all()


def test_lr_1080p_fixed_point(ctx, O):
    """BASELINE size, size-independent property: a flat 1920x1080 plane is a fixed point of every filter type"""
    h, w = 1080, 1920
    flat = np.full((1, h, w), 131, np.uint8)
    rng = np.random.default_rng(96)
    units = _random_units(rng, O, 64, h, w)
    assert (_run(ctx, flat, flat, 8, 0, 64, units) == 131).all()


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("ss", [0, 1])
def test_lr_on_off_decision_equals_the_oracle(ctx, O, bd, ss):
    """av1mi_lr_frames_decide: the restored plane as before, and per frame the ON / OFF flag = av1o_lr_keep (squared error against
    the source, strictly smaller) — frames built so that both outcomes occur, every unit type, odd sizes, chroma stripes"""
    rng = np.random.default_rng(190 + bd + ss)
    for (h, w), unit in (((136, 200), 64), ((72, 100), 32 if ss else 64), ((300, 260), 128), ((40, 24), 64), ((200, 712), 64)):      # the last one: >= 32 tiles, sampled
        nf = 4
        cdef = np.stack([make_image(rng, h, w, bd) for _ in range(nf)])
        dbl = cdef.copy()
        units = np.stack([_random_units(rng, O, unit, h, w) for _ in range(nf)])
        lr = np.stack([O.lr_plane(cdef[f], dbl[f], bd, ss, unit, units[f]) for f in range(nf)])
        # sources: frames 0, 1 lie at the CDEF samples (restoration loses), 2 at the restored ones (it wins unless no unit filters), 3 between
        mx = (1 << bd) - 1
        noise = rng.integers(-1, 2, cdef.shape)
        src = np.stack([np.clip(cdef[0].astype(int) + noise[0], 0, mx), cdef[1], lr[2],
                        (cdef[3].astype(int) + lr[3].astype(int) + 1) >> 1]).astype(cdef.dtype)
        d_c, d_d, d_u, d_s = ctx.to_device(cdef), ctx.to_device(dbl), ctx.to_device(units), ctx.to_device(src)
        d_o, d_on = ctx.alloc(cdef.nbytes), ctx.to_device(np.full(3 * nf, 7, np.uint8))
        d_scr = ctx.alloc(ctx.lr_decide_scratch_bytes(h, ss, nf))
        ctx.lr_frames_decide(d_c, d_d, d_o, w, w, h, bd, ss, unit, d_u, units.shape[1] * units.shape[2], nf, d_s, d_scr, d_on, on_offset=1, on_stride=3)
        got, on = d_o.download(cdef.shape, cdef.dtype), d_on.download((nf, 3), np.uint8)
        for b in (d_c, d_d, d_u, d_s, d_o, d_on, d_scr):
            b.free()
        assert (got == lr).all()
        exp = [O.lr_select((src[f],), (cdef[f],), (lr[f],), bd, ss)[1][0] for f in range(nf)]
        assert on[:, 1].tolist() == exp and (on[:, 0] == 7).all() and (on[:, 2] == 7).all(), ((h, w), unit, on.tolist(), exp)
        assert exp[1] == 0 and (exp[2] == 1 or (lr[2] == cdef[2]).mean() > 0.2)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,bd", [(200, 136, 8), (136, 72, 10), (576, 328, 10)])      # the last: luma has 54 tiles, sampled sparsely
def test_three_plane_decide_equals_three_single_plane_calls(ctx, av1mi, O, w, h, bd):
    """av1mi_lr_yuv_decide (sampled tiles, decision, the rest of the frames that stay ON: what the GOP session uses) against
    av1mi_lr_frames_decide per plane: the same restored planes and the same ON / OFF flags"""
    rng = np.random.default_rng(w * 7 + bd)
    nf, unit = 3, 64
    dims = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
    cdef = [np.stack([make_image(rng, hh, ww, bd) for _ in range(nf)]) for hh, ww in dims]
    dbl = [c.copy() for c in cdef]
    mx = (1 << bd) - 1
    src = [np.clip(c.astype(int) + rng.integers(-2, 3, c.shape), 0, mx).astype(c.dtype) for c in cdef]
    uy, uc = _random_units(rng, O, unit, h, w), _random_units(rng, O, unit, h // 2, w // 2)
    for p in range(3):       # frame 1: the source IS the restored plane, so restoration stays on there (unless no unit filters)
        src[p][1] = O.lr_plane(cdef[p][1], dbl[p][1], bd, int(p > 0), unit, uc if p else uy)
    d = {k: [ctx.to_device(a) for a in v] for k, v in (("cdef", cdef), ("dbl", dbl), ("src", src))}
    d_uy, d_uc = ctx.to_device(uy), ctx.to_device(uc)
    out1 = [ctx.alloc(c.nbytes) for c in cdef]
    out3 = [ctx.alloc(c.nbytes) for c in cdef]
    on1, on3 = ctx.to_device(np.full(3 * nf + 1, 9, np.uint8)), ctx.to_device(np.full(3 * nf + 1, 9, np.uint8))
    for p in range(3):
        scr = ctx.alloc(ctx.lr_decide_scratch_bytes(dims[p][0], p > 0, nf))
        ctx.lr_frames_decide(d["cdef"][p], d["dbl"][p], out1[p], dims[p][1], dims[p][1], dims[p][0], bd, p > 0, unit, d_uc if p else d_uy, 0, nf,
                             d["src"][p], scr, on1, on_offset=p, on_stride=3)
        ctx.sync()
        scr.free()
    scr = ctx.to_device(np.full(ctx.lr_yuv_decide_scratch_bytes(h, nf), 0xA5, np.uint8))      # the call zeroes it itself
    ctx.lr_yuv_decide(av1mi.LrDecideJob(w, h, bd, nf, unit, w, w // 2, *[b.ptr for b in d["cdef"] + d["dbl"] + out3 + d["src"]], d_uy.ptr, d_uc.ptr, 0, 0,
                                        scr.ptr, on3.ptr))
    a, b = on1.download((3 * nf + 1,), np.uint8), on3.download((3 * nf + 1,), np.uint8)
    assert a.tolist() == b.tolist() and a[-1] == 9 and set(a[:-1].tolist()) <= {0, 1} and a[3] == 1 and a[0] == 0
    for p in range(3):       # the restored planes where restoration stays ON (elsewhere the two-pass form writes the sampled tiles only)
        x, y = out1[p].download(cdef[p].shape, cdef[p].dtype), out3[p].download(cdef[p].shape, cdef[p].dtype)
        for f in range(nf):
            assert not a[3 * f + p] or (x[f] == y[f]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("bd", [8, 10])
def test_extend_frames_replicates_the_true_edge(ctx, bd):
    """av1mi_extend_frames == numpy edge replication (pipeline.extend_visible), stacked frames, every padding width 0 .. 7"""
    import pipeline as P
    rng = np.random.default_rng(bd)
    dt = np.uint8 if bd == 8 else np.uint16
    for (w, h, vw, vh) in ((64, 48, 64, 48), (64, 48, 57, 48), (64, 48, 64, 41), (72, 40, 66, 35), (16, 16, 9, 9), (104, 80, 100, 76)):
        a = rng.integers(0, 1 << bd, (3, h, w)).astype(dt)
        d = ctx.to_device(a)
        ctx.extend_frames(d, w, w, h, vw, vh, bd, 3)
        got = d.download(a.shape, dt)
        d.free()
        assert (got == P.extend_visible(a.copy(), vw, vh)).all(), (w, h, vw, vh)
        assert (got[:, :vh, :vw] == a[:, :vh, :vw]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,bd", [(576, 328, 10), (200, 136, 8)])
def test_wiener_only_kernel_equals_the_general_one(ctx, av1mi, O, w, h, bd):
    """av1mi_lr_decide_job.no_self_guided_units: the kernel without the self-guided path (half the LDS, what the session runs) gives
    the planes and decisions of the general kernel when no unit is self-guided"""
    rng = np.random.default_rng(w + bd)
    nf, unit = 2, 64
    dims = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
    cdef = [np.stack([make_image(rng, hh, ww, bd) for _ in range(nf)]) for hh, ww in dims]
    mx = (1 << bd) - 1
    src = [np.clip(c.astype(int) + rng.integers(-2, 3, c.shape), 0, mx).astype(c.dtype) for c in cdef]
    uy, uc = _random_units(rng, O, unit, h, w), _random_units(rng, O, unit, h // 2, w // 2)
    for u in (uy, uc):
        u[u[..., 0] == 2] = 0                       # self-guided units -> none
    for p in range(3):
        src[p][1] = O.lr_plane(cdef[p][1], cdef[p][1], bd, int(p > 0), unit, uc if p else uy)       # frame 1 keeps its restoration
    d_c, d_s = [ctx.to_device(a) for a in cdef], [ctx.to_device(a) for a in src]
    d_uy, d_uc = ctx.to_device(uy), ctx.to_device(uc)
    res = []
    for flag in (0, 1):
        out = [ctx.alloc(c.nbytes) for c in cdef]
        on = ctx.to_device(np.full(3 * nf + 1, 9, np.uint8))
        scr = ctx.alloc(ctx.lr_yuv_decide_scratch_bytes(h, nf))
        ctx.lr_yuv_decide(av1mi.LrDecideJob(w, h, bd, nf, unit, w, w // 2, *[b.ptr for b in d_c + d_c + out + d_s], d_uy.ptr, d_uc.ptr, 0, 0,
                                            scr.ptr, on.ptr, flag))
        res.append((on.download((3 * nf + 1,), np.uint8), [o.download(c.shape, c.dtype) for o, c in zip(out, cdef)]))
    assert res[0][0].tolist() == res[1][0].tolist() and res[0][0][3] == 1
    for p in range(3):
        for f in range(nf):
            if res[0][0][3 * f + p]:
                assert (res[0][1][p][f] == res[1][1][p][f]).all()
                assert (res[1][1][p][f] == O.lr_plane(cdef[p][f], cdef[p][f], bd, int(p > 0), unit, uc if p else uy)).all()
