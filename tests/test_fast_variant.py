"""The tolerance-bounded FAST variant (include/lutr.h lutr_ctx_set_precision, csrc/lutr_tile2.hip V_FAST).

north_star allows <= 1 LSB at 8 bit and <= 2 LSB at 10 bit against FFmpeg's lut3d.  The strict kernels are the
bit-exact restatement of FFmpeg's scalar C; the fast variant may differ from strict by at most ONE code at 8 AND at
10 bit -- which leaves the second 10-bit code for FFmpeg-vs-restatement differences (its AVX2 tetrahedral path fuses
multiply-adds).  Three things are pinned here:
  (1) the fp32 -> fp16 round-to-nearest-even used for the fast lattice (oracle vs numpy),
  (2) CPU: |fast oracle - strict oracle| <= 1 over modes x formats x LUTs x content, and how often they differ,
  (3) GPU: the fast kernels == the fast oracle bit for bit, and <= 1 code from the strict oracle.
The fast variant: lattice as fp16 of value * (2^depth - 1), blend as an fma chain with fp32 accumulation (1 mul + 3 fma
per channel instead of FFmpeg's 4 mul + 3 add and a final `* M`), same coordinates, same truncation.
"""
import numpy as np
import pytest

from lut_renderer_amd import cube, frames

MODES = ("tetrahedral", "trilinear", "nearest")
FORMATS = (("yuv420p10le", 10, 1, 1), ("yuv420p", 8, 1, 1), ("yuv422p10le", 10, 1, 0), ("yuv444p", 8, 0, 0))


def test_half_conversion_matches_numpy(orc):
    lib = orc.load()
    rng = np.random.default_rng(11)
    vals = np.concatenate([
        rng.uniform(0, 1023.75, 20000), rng.uniform(0, 1e-3, 2000), rng.uniform(-300, 300, 2000),
        np.array([0.0, 1023.0, 255.0, 1022.75, 1023.25, 511.9999, 2048.5, 2049.5, 65504.0, 65519.9, 65520.0, 1e9,
                  6.1e-5, 6.0e-5, 5.96e-8, 2.98e-8, 2.99e-8, 8.9e-8, -0.0, -6.0e-5]),
    ]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16)
    for v, w in zip(vals, want):
        h = lib.orc_f2h(float(v))
        assert h == int(w.view(np.uint16)), (float(v), hex(h), hex(int(w.view(np.uint16))))
        if np.isfinite(w):
            assert lib.orc_h2f(h) == float(w)


def _luts():
    rng = np.random.default_rng(3)
    yield "log709_33", cube.log709_lattice(33)
    yield "identity_17", cube.identity_lattice(17)
    yield "random_9", rng.uniform(0.0, 1.0, size=(9, 9, 9, 3)).astype(np.float32)        # steep, inside [0, 1]
    yield "log709_65", cube.log709_lattice(65)


@pytest.mark.parametrize("fmt,depth,csx,csy", FORMATS)
def test_fast_oracle_is_within_one_code_of_strict(orc, fmt, depth, csx, csy):
    one = np.ones(3, np.float32)
    k = orc.yuv_constants("bt709", "tv", "bt709", "tv", depth, depth, depth, 1 << (csx + csy))
    worst, differ, total = 0, 0, 0
    for name, lat in _luts():
        for dist in ("natural", "uniform", "noise16"):
            src = frames.make_yuv(dist, 128, 64, depth, csx, csy, k=5)
            for mode in MODES:
                a = orc.apply_yuv(lat, one, mode, k, depth, depth, depth, csx, csy, src)
                b = orc.apply_yuv(lat, one, mode, k, depth, depth, depth, csx, csy, src, fast=True)
                for x, y in zip(a, b):
                    d = np.abs(x.astype(np.int32) - y.astype(np.int32))
                    worst = max(worst, int(d.max()))
                    differ += int((d > 0).sum())
                    total += d.size
                assert worst <= 1, (name, dist, mode, worst)
    assert worst <= 1
    assert differ / total < 0.25, differ / total          # most samples are identical; the rest are off by exactly one


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,depth,csx,csy", FORMATS)
def test_fast_kernels_match_the_fast_oracle_and_stay_within_one_code_of_strict(engine, orc, cube_dir, fmt, depth, csx, csy):
    import torch
    one = np.ones(3, np.float32)
    k = orc.yuv_constants("bt709", "tv", "bt709", "tv", depth, depth, depth, 1 << (csx + csy))
    dt = np.uint16 if depth > 8 else np.uint8
    engine.set_variant("vec_lds")
    engine.set_precision("fast")
    try:
        for name, lat in _luts():
            engine.set_lut(cube.CubeLut(lat.shape[0], one, lat))
            for dist in ("natural", "uniform", "noise16"):
                src = frames.make_yuv(dist, 256, 72, depth, csx, csy, k=6)
                dev = [torch.from_numpy(p.view(np.int16) if p.dtype == np.uint16 else p).to(engine.device) for p in src]
                for mode in MODES:
                    got = [t.cpu().numpy().view(dt) for t in engine.apply_yuv(dev, pix_fmt=fmt, interp=mode)]
                    if mode != "nearest":
                        assert "fast" in engine.last_kernel, engine.last_kernel
                    fast = orc.apply_yuv(lat, one, mode, k, depth, depth, depth, csx, csy, src, fast=True)
                    strict = orc.apply_yuv(lat, one, mode, k, depth, depth, depth, csx, csy, src)
                    for i, (g, f, s) in enumerate(zip(got, fast, strict)):
                        if mode != "nearest":       # nearest runs the strict kernel: one node, nothing to blend
                            assert np.array_equal(g, f), (name, dist, mode, i, int(np.abs(g.astype(int) - f.astype(int)).max()))
                        assert np.abs(g.astype(np.int32) - s.astype(np.int32)).max() <= 1, (name, dist, mode, i)
        # a lattice outside [0, 1] cannot take the clip-free fast kernels: strict runs instead, bit-exact
        rng = np.random.default_rng(9)
        lat = rng.uniform(-0.2, 1.2, size=(9, 9, 9, 3)).astype(np.float32)
        engine.set_lut(cube.CubeLut(9, one, lat))
        src = frames.make_yuv("uniform", 256, 72, depth, csx, csy, k=7)
        dev = [torch.from_numpy(p.view(np.int16) if p.dtype == np.uint16 else p).to(engine.device) for p in src]
        got = [t.cpu().numpy().view(dt) for t in engine.apply_yuv(dev, pix_fmt=fmt)]
        assert "fast" not in engine.last_kernel
        for g, s in zip(got, orc.apply_yuv(lat, one, "tetrahedral", k, depth, depth, depth, csx, csy, src)):
            assert np.array_equal(g, s)
    finally:
        engine.set_precision("strict")
        engine.set_variant("auto")


@pytest.mark.gpu
def test_fast_full_uhd_frame_and_config5_prologue(engine, orc, cube_dir):
    """Headline size under the fast variant: whole UHD frame within one code of the strict kernels, a strip bit-exact
    against the fast oracle; plus the pc -> tv prologue (BASELINE config 5) with 10-bit in, 8-bit LUT, 10- and 8-bit out."""
    import torch
    lut = cube.read_cube(cube_dir / "log709_33.cube")
    engine.set_lut(lut)
    engine.set_variant("vec_lds")
    w, h = 3840, 2160
    src = frames.natural_yuv(w, h, 10, 1, 1, k=3)
    dev = [torch.from_numpy(p.view(np.int16)).to(engine.device) for p in src]
    try:
        strict = [t.clone() for t in engine.apply_yuv(dev, pix_fmt="yuv420p10le")]
        engine.set_precision("fast")
        fast = engine.apply_yuv(dev, pix_fmt="yuv420p10le")
        assert "fast" in engine.last_kernel
        for a, b in zip(strict, fast):
            assert (a.to(torch.int32) - b.to(torch.int32)).abs().max().item() <= 1
        strip = [src[0][640:704], src[1][320:352], src[2][320:352]]
        k = orc.yuv_constants(din=10)
        want = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 10, 1, 1, strip, nthreads=8, fast=True)
        got = [t.cpu().numpy().view(np.uint16) for t in fast]
        for g, wv in zip([got[0][640:704], got[1][320:352], got[2][320:352]], want):
            assert np.array_equal(g, wv)
        for out_fmt, dout in (("yuv420p10le", 10), ("yuv420p", 8)):
            s5 = frames.make_yuv("natural", 256, 72, 10, 1, 1, k=4, full_range=True)
            d5 = [torch.from_numpy(p.view(np.int16)).to(engine.device) for p in s5]
            k5 = orc.yuv_constants("bt709", "tv", "bt709", "tv", 10, 8, dout, 4, prologue=True)
            got5 = engine.apply_yuv(d5, pix_fmt="yuv420p10le", out_pix_fmt=out_fmt, range_src="pc", range_in="tv", lut_depth=8)
            assert "pre" in engine.last_kernel and "fast" in engine.last_kernel, engine.last_kernel
            want5 = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k5, 10, 8, dout, 1, 1, s5, fast=True)
            for g, wv in zip(got5, want5):
                g = g.cpu().numpy()
                assert np.array_equal(g.view(np.uint16) if dout > 8 else g, wv)
    finally:
        engine.set_precision("strict")
        engine.set_variant("auto")
