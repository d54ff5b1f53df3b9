"""GPU parity: liblutr's HIP kernels (through the C-ABI) against the CPU oracle, bit-exact.

The oracle restates FFmpeg lut3d -- the filter the reference emits at
/root/reference/src/lut_renderer/ffmpeg.py:246 -- and the engine's YUV contract.  All work
here is integer codes in / integer codes out, so the bar is exact equality; the <=1 LSB
(8-bit) / <=2 LSB (10-bit) allowance of the north star is for a live ffmpeg binary
(tests/test_ffmpeg_live.py), not for our own oracle.
"""
import numpy as np
import pytest
import torch

from lut_renderer_amd import cube, frames

pytestmark = pytest.mark.gpu

MODES3 = ("nearest", "trilinear", "tetrahedral")
MODES5 = MODES3 + ("pyramid", "prism")


def _to_dev(planes, eng):
    out = []
    for p in planes:
        t = torch.from_numpy(p.view(np.int16) if p.dtype == np.uint16 else p)
        out.append(t.to(eng.device))
    return out


def _to_np(tensors, like_dtype):
    out = []
    for t in tensors:
        a = t.cpu().numpy()
        out.append(a.view(np.uint16) if like_dtype == np.uint16 else a)
    return out


def _load(eng, cube_dir, name):
    lut = cube.read_cube(cube_dir / name)
    eng.set_lut(lut)
    return lut


def _assert_equal(got, want, what):
    for i, (a, b) in enumerate(zip(got, want)):
        if not np.array_equal(a, b):
            diff = np.abs(a.astype(np.int64) - b.astype(np.int64))
            bad = np.argwhere(diff > 0)
            raise AssertionError(f"{what}: plane {i} differs at {len(bad)} samples, max |d|={diff.max()}, "
                                 f"first {bad[0].tolist()} got {a[tuple(bad[0])]} want {b[tuple(bad[0])]}")


@pytest.mark.parametrize("variant", ["generic", "vec_global", "vec_lds"])
@pytest.mark.parametrize("depth", [8, 10, 12, 16])
@pytest.mark.parametrize("lutname", ["log709_33.cube", "random_9.cube", "domain_2.cube"])
def test_rgb_parity(engine, orc, cube_dir, variant, depth, lutname):
    lut = _load(engine, cube_dir, lutname)
    engine.set_variant(variant)
    for dist_k, mk in ((0, frames.uniform_rgb), (1, frames.natural_rgb)):
        src = mk(128, 36, depth, k=dist_k)
        for mode in (MODES5 if variant == "generic" else MODES3):
            want = orc.apply_rgb(lut.table, lut.scale, depth, mode, src)
            got = _to_np(engine.apply_rgb(_to_dev(src, engine), depth=depth, interp=mode), src[0].dtype)
            _assert_equal(got, want, f"rgb {variant} d{depth} {mode} {lutname}")
            assert variant == "generic" or any(t in engine.last_kernel for t in ("vec", "tile", "tube"))
    engine.set_variant("auto")


@pytest.mark.parametrize("variant", ["generic", "vec_global", "vec_lds"])
@pytest.mark.parametrize("fmt", ["yuv420p", "yuv420p10le", "yuv422p10le", "yuv444p10le", "yuv422p", "yuv444p",
                                 "yuv420p12le"])
def test_yuv_parity(engine, orc, cube_dir, variant, fmt):
    from lut_renderer_amd.engine import parse_pix_fmt
    pf = parse_pix_fmt(fmt)
    lut = _load(engine, cube_dir, "log709_33.cube")
    engine.set_variant(variant)
    for matrix, rng_in, rng_out in (("bt709", "tv", "tv"), ("bt2020nc", "tv", "tv"), ("smpte170m", "pc", "tv"),
                                    ("bt709", "pc", "pc")):
        k = orc.yuv_constants(matrix, rng_in, matrix, rng_out, pf.depth, None, None, 1 << (pf.csx + pf.csy))
        for dist in ("uniform", "natural"):
            src = frames.make_yuv(dist, 128, 36, pf.depth, pf.csx, pf.csy, k=3, full_range=(rng_in == "pc"))
            for mode in (MODES5 if variant == "generic" else MODES3):
                want = orc.apply_yuv(lut.table, lut.scale, mode, k, pf.depth, pf.depth, pf.depth, pf.csx, pf.csy, src)
                got = engine.apply_yuv(_to_dev(src, engine), pix_fmt=fmt, interp=mode, matrix_in=matrix,
                                       range_src=rng_in, range_out=rng_out)
                _assert_equal(_to_np(got, src[0].dtype), want, f"yuv {variant} {fmt} {mode} {matrix} {rng_in}->{rng_out} {dist}")
    engine.set_variant("auto")


@pytest.mark.parametrize("w,h", [(2, 2), (6, 4), (31, 17), (130, 7), (1, 1), (33, 2)])
def test_ragged_sizes_go_generic(engine, orc, cube_dir, w, h):
    """Odd and tiny sizes (edge chroma blocks replicate the last sample) -- generic kernel."""
    lut = _load(engine, cube_dir, "log709_33.cube")
    for fmt, depth, cs in (("yuv420p", 8, (1, 1)), ("yuv420p10le", 10, (1, 1)), ("yuv422p10le", 10, (1, 0))):
        k = orc.yuv_constants("bt709", "tv", None, "tv", depth, None, None, 1 << sum(cs))
        src = frames.uniform_yuv(w, h, depth, cs[0], cs[1], k=5)
        want = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, depth, depth, depth, cs[0], cs[1], src)
        got = engine.apply_yuv(_to_dev(src, engine), pix_fmt=fmt)
        _assert_equal(_to_np(got, src[0].dtype), want, f"ragged {fmt} {w}x{h}")
    src = frames.uniform_rgb(w, h, 10, k=6)
    want = orc.apply_rgb(lut.table, lut.scale, 10, "trilinear", src)
    got = _to_np(engine.apply_rgb(_to_dev(src, engine), depth=10, interp="trilinear"), np.uint16)
    _assert_equal(got, want, f"ragged rgb {w}x{h}")


def test_empty_inputs_are_noops(engine, cube_dir):
    _load(engine, cube_dir, "identity_17.cube")
    z = [torch.empty((0, 16), dtype=torch.uint8, device=engine.device) for _ in range(3)]
    engine.apply_rgb(z, depth=8)
    y = [torch.zeros((4, 16), dtype=torch.uint8, device=engine.device),
         torch.zeros((2, 8), dtype=torch.uint8, device=engine.device),
         torch.zeros((2, 8), dtype=torch.uint8, device=engine.device)]
    out = engine.apply_yuv(y, pix_fmt="yuv420p", rows=0)
    assert out[0].shape == (4, 16)


def test_row_shards_and_batches_match_whole_frame(engine, orc, cube_dir):
    """Row-block shards (the multi-GPU partition) and frame batches give the same pixels."""
    lut = _load(engine, cube_dir, "log709_33.cube")
    w, h, nf = 256, 48, 3
    frames_np = [frames.natural_yuv(w, h, 10, 1, 1, k=i) for i in range(nf)]
    k = orc.yuv_constants(din=10)
    want = [orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 10, 1, 1, f) for f in frames_np]
    batch = [torch.stack([_to_dev(f, engine)[i] for f in frames_np]) for i in range(3)]
    got = engine.apply_yuv(batch, pix_fmt="yuv420p10le")
    for i in range(nf):
        _assert_equal(_to_np([g[i] for g in got], np.uint16), want[i], f"batch frame {i}")
    # shard frame 0 into 4 row blocks, even-aligned, applied one by one into one output
    src = _to_dev(frames_np[0], engine)
    dst = [torch.zeros_like(t) for t in src]
    from lut_renderer_amd.shard import row_blocks
    for r0, r1 in row_blocks(h, 4, align=2):
        engine.apply_yuv(src, dst, pix_fmt="yuv420p10le", row0=r0, rows=r1 - r0)
    _assert_equal(_to_np(dst, np.uint16), want[0], "row shards")


def test_mixed_depth_and_prologue(engine, orc, cube_dir):
    """Reference cases K (10-bit LUT, 8-bit output) and A/D (full-range prologue to 8-bit), SURVEY App. D."""
    lut = _load(engine, cube_dir, "log709_33.cube")
    # K: yuv420p10le tv source, lut3d at 10 bit, format=yuv420p
    src = frames.natural_yuv(64, 36, 10, 1, 1, k=9)
    k = orc.yuv_constants("bt2020nc", "tv", "bt2020nc", "tv", 10, 10, 8, 4)
    want = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 8, 1, 1, src)
    got = engine.apply_yuv(_to_dev(src, engine), pix_fmt="yuv420p10le", out_pix_fmt="yuv420p", matrix_in="bt2020nc")
    _assert_equal(_to_np(got, np.uint8), want, "case K")
    # D: yuv422p10le pc source -> scale pc->tv + format=yuv422p (8 bit) -> lut3d -> format=yuv422p10le
    src = frames.uniform_yuv(64, 36, 10, 1, 0, k=10, full_range=True)
    k = orc.yuv_constants("smpte170m", "tv", "smpte170m", "tv", 10, 8, 10, 2, prologue=True)
    want = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 8, 10, 1, 0, src)
    got = engine.apply_yuv(_to_dev(src, engine), pix_fmt="yuv422p10le", matrix_in="smpte170m", range_src="pc",
                           range_in="tv", lut_depth=8)
    _assert_equal(_to_np(got, np.uint16), want, "case D")
    # A: yuvj420p (8-bit pc) -> tv 8 bit -> lut3d
    src = frames.uniform_yuv(64, 36, 8, 1, 1, k=11, full_range=True)
    k = orc.yuv_constants("bt709", "tv", "bt709", "tv", 8, 8, 8, 4, prologue=True)
    want = orc.apply_yuv(lut.table, lut.scale, "trilinear", k, 8, 8, 8, 1, 1, src)
    got = engine.apply_yuv(_to_dev(src, engine), pix_fmt="yuv420p", interp="trilinear", range_src="pc", range_in="tv")
    _assert_equal(_to_np(got, np.uint8), want, "case A")


def test_full_size_properties_uhd(engine, orc, cube_dir):
    """BASELINE config 2 size (3840x2160 yuv420p10le, 33^3, tetrahedral): properties that need
    no full-frame oracle run, plus an oracle check on a strip."""
    w, h = 3840, 2160
    src_np = frames.natural_yuv(w, h, 10, 1, 1, k=0)
    src = _to_dev(src_np, engine)
    # (1) identity lattice: output within 1 code of the input (A.6 #1 composed with the YUV round trip)
    _load(engine, cube_dir, "identity_33.cube")
    out = _to_np(engine.apply_yuv(src, pix_fmt="yuv420p10le"), np.uint16)
    for a, b in zip(out, src_np):
        assert np.abs(a.astype(np.int32) - b.astype(np.int32)).max() <= 2
    # (2) vector kernel == generic kernel on the whole frame (two independent code paths)
    lut = _load(engine, cube_dir, "log709_33.cube")
    engine.set_variant("generic")
    ref = engine.apply_yuv(src, pix_fmt="yuv420p10le")
    engine.set_variant("auto")
    fast = engine.apply_yuv(src, pix_fmt="yuv420p10le")
    assert "tile" in engine.last_kernel
    for a, b in zip(ref, fast):
        assert torch.equal(a, b)
    # (3) idempotence of sharding: 8 row blocks == whole frame
    from lut_renderer_amd.shard import row_blocks
    dst = [torch.zeros_like(t) for t in src]
    for r0, r1 in row_blocks(h, 8, align=2):
        engine.apply_yuv(src, dst, pix_fmt="yuv420p10le", row0=r0, rows=r1 - r0)
    for a, b in zip(dst, fast):
        assert torch.equal(a, b)
    # (4) oracle on a 64-row strip of the same frame
    strip = [src_np[0][512:576], src_np[1][256:288], src_np[2][256:288]]
    k = orc.yuv_constants(din=10)
    want = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 10, 1, 1, strip, nthreads=8)
    got = _to_np(fast, np.uint16)
    _assert_equal([got[0][512:576], got[1][256:288], got[2][256:288]], want, "uhd strip")


def test_apply_lut_api_follows_the_reference_options(engine, orc, cube_dir):
    """`apply_lut` with the reference's option vocabulary (models.py:45-56) == the oracle chain the
    same options select in the reference's filter string (SURVEY.md Appendix D cases A, E, F)."""
    from lut_renderer_amd.api import apply_lut
    lut = cube.read_cube(cube_dir / "log709_33.cube")
    # E: tv 10-bit bt2020nc source, input_matrix=auto -> bt2020nc both ways, 10-bit LUT, tags bt709/tv
    src = frames.natural_yuv(128, 72, 10, 1, 1, k=12)
    out, tags = apply_lut(_to_dev(src, engine), cube=cube_dir / "log709_33.cube", pix_fmt="yuv420p10le",
                          colorspace="bt2020nc", color_range="tv", engine=engine)
    k = orc.yuv_constants("bt2020nc", "tv", "bt2020nc", "tv", 10, 10, 10, 4)
    _assert_equal(_to_np(out, np.uint16), orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 10, 1, 1, src),
                  "apply_lut case E")
    assert tags == {"color_primaries": "bt709", "color_trc": "bt709", "colorspace": "bt709", "color_range": "tv"}
    # F: forced bt709 matrix, trilinear
    out, _ = apply_lut(_to_dev(src, engine), cube=None, interp="trilinear", pix_fmt="yuv420p10le",
                       input_matrix="bt709", colorspace="bt2020nc", engine=engine)
    k = orc.yuv_constants("bt709", "tv", "bt709", "tv", 10, 10, 10, 4)
    _assert_equal(_to_np(out, np.uint16), orc.apply_yuv(lut.table, lut.scale, "trilinear", k, 10, 10, 10, 1, 1, src),
                  "apply_lut case F")
    # A: yuvj420p full-range 8-bit -> prologue to tv, 8-bit LUT, yuv420p out; 'inherit' returns no tags
    src8 = frames.uniform_yuv(128, 72, 8, 1, 1, k=13, full_range=True)
    out, tags = apply_lut(_to_dev(src8, engine), cube=None, pix_fmt="yuvj420p", colorspace="bt709", color_range="pc",
                          output_tags="bt709", engine=engine)
    k = orc.yuv_constants("bt709", "tv", "bt709", "tv", 8, 8, 8, 4, prologue=True)
    _assert_equal(_to_np(out, np.uint8), orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 8, 8, 8, 1, 1, src8),
                  "apply_lut case A")
    _, tags = apply_lut(_to_dev(src8, engine), cube=None, pix_fmt="yuvj420p", color_range="pc",
                        output_tags="inherit", engine=engine)
    assert tags is None
    with pytest.raises(ValueError):
        apply_lut(_to_dev(src, engine), cube=None, interp="cubic", pix_fmt="yuv420p10le", engine=engine)
    # zscale_dither=error_diffusion with a 10-bit source going to 8-bit yuv420p (force_8bit policy, ffmpeg.py:288-291)
    out, _ = apply_lut(_to_dev(src, engine), cube=None, pix_fmt="yuv420p10le", colorspace="bt2020nc", color_range="tv",
                       out_pix_fmt="yuv420p", zscale_dither="error_diffusion", engine=engine)
    k = orc.yuv_constants("bt2020nc", "tv", "bt2020nc", "tv", 10, 10, 8, 4)
    _assert_equal([t.cpu().numpy() for t in out],
                  orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 8, 1, 1, src, dither="error_diffusion"),
                  "apply_lut dithered")


def test_tile_window_statistics(engine, cube_dir):
    """Natural content mostly hits the LDS window; uniform noise never fits and is all gather."""
    _load(engine, cube_dir, "log709_33.cube")
    engine.set_variant("vec_lds")
    for dist, expect_global in (("natural", False), ("uniform", True)):
        one = _to_dev(frames.make_yuv(dist, 1920, 1080, 10, 1, 1, k=0), engine)
        src = [t.unsqueeze(0).repeat(16, 1, 1) for t in one]      # a batch: waves get long strips to walk
        engine.tile_stats(True)
        engine.apply_yuv(src, pix_fmt="yuv420p10le")
        st = engine.tile_stats(False)
        assert st["tiles"] > 0
        if expect_global:
            assert st["global_tiles"] == st["tiles"]
        else:
            assert st["global_tiles"] + st["misses"] < 0.5 * st["tiles"]
    engine.set_variant("auto")


def test_distributed_lut_load_single_rank_rccl(orc, cube_dir):
    """The RCCL path of `set_lut_distributed` (broadcast straight into the device lattice) with a
    one-rank nccl group: everything bench.py --gpus N does except having peers."""
    import os
    import socket
    import torch.distributed as dist
    from lut_renderer_amd.engine import LutEngine
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        lut = cube.read_cube(cube_dir / "log709_33.cube")
        with LutEngine(0) as eng:
            eng.set_lut_distributed(lut, src=0)
            lat = eng.lattice_tensor()
            assert lat.numel() == 34 ** 3 * 4 and lat.is_cuda
            # padded layout: node (r,g,b) at ((r*34+g)*34+b)*4, last node replicated at index 33
            host = lat.cpu().numpy().reshape(34, 34, 34, 4)
            assert np.array_equal(host[:33, :33, :33, :3], lut.table)
            assert np.array_equal(host[33, 33, 33, :3], lut.table[32, 32, 32])
            src = frames.natural_yuv(256, 64, 10, 1, 1, k=1)
            out = eng.apply_yuv(_to_dev(src, eng), pix_fmt="yuv420p10le")
            k = orc.yuv_constants(din=10)
            _assert_equal(_to_np(out, np.uint16),
                          orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 10, 1, 1, src), "rccl lut load")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [2, 3, 17, 65, 128, 256])
def test_lattice_sizes_min_to_max(engine, orc, n):
    """LUT_3D_SIZE from the smallest (2) to FFmpeg's MAX_LEVEL (256): every kernel family, both modes.
    (257^3 float4 nodes = 271 MB on the device; addresses beyond 2^24 need the integer index path.)"""
    from lut_renderer_amd.cube import CubeLut
    rng = np.random.default_rng(n)
    tab = rng.uniform(0.0, 1.0, size=(n, n, n, 3)).astype(np.float32)
    lut = CubeLut(n, np.ones(3, dtype=np.float32), tab)
    engine.set_lut(lut)
    src = frames.uniform_yuv(128, 36, 10, 1, 1, k=n)
    rgb = frames.uniform_rgb(128, 36, 10, k=n)
    rgb[0][0, :8] = 1023; rgb[1][0, :8] = 1023; rgb[2][0, :8] = 1023        # the last node exactly
    rgb[0][1, :8] = 0; rgb[1][1, :8] = 1023; rgb[2][1, :8] = 0
    k = orc.yuv_constants(din=10)
    for variant in ("generic", "vec_global", "vec_lds"):
        engine.set_variant(variant)
        for mode in ("trilinear", "tetrahedral", "nearest"):
            want = orc.apply_yuv(tab, lut.scale, mode, k, 10, 10, 10, 1, 1, src)
            got = engine.apply_yuv(_to_dev(src, engine), pix_fmt="yuv420p10le", interp=mode)
            _assert_equal(_to_np(got, np.uint16), want, f"N={n} {variant} {mode} yuv")
            want = orc.apply_rgb(tab, lut.scale, 10, mode, rgb)
            got = engine.apply_rgb(_to_dev(rgb, engine), depth=10, interp=mode)
            _assert_equal(_to_np(got, np.uint16), want, f"N={n} {variant} {mode} rgb")
    engine.set_variant("auto")


def test_padded_strides_in_place_and_bottom_up(engine, orc, cube_dir):
    """Layouts FFmpeg frames really have: linesize > width (padding), in-place filtering (lut3d writes into a
    writable input frame) and negative linesize (bottom-up), which must fall back to the generic kernel."""
    lut = _load(engine, cube_dir, "log709_33.cube")
    w, h = 256, 40
    src = frames.natural_yuv(w, h, 10, 1, 1, k=21)
    k = orc.yuv_constants(din=10)
    want = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 10, 1, 1, src)

    def padded(p, pad):
        big = torch.full((p.shape[0], p.shape[1] + pad), -1, dtype=torch.int16, device=engine.device)
        big[:, : p.shape[1]] = torch.from_numpy(p.view(np.int16)).to(engine.device)
        return big[:, : p.shape[1]]                      # a view with row stride = width + pad

    for variant in ("generic", "vec_global", "vec_lds"):
        engine.set_variant(variant)
        s = [padded(src[0], 64), padded(src[1], 32), padded(src[2], 32)]
        d = [padded(np.zeros_like(src[0]), 128), padded(np.zeros_like(src[1]), 64), padded(np.zeros_like(src[2]), 64)]
        engine.apply_yuv(s, d, pix_fmt="yuv420p10le")
        _assert_equal(_to_np([t.contiguous() for t in d], np.uint16), want, f"padded strides {variant}")
        pad_ok = bool((d[0].as_strided((h, 128), (w + 128, 1), w) == -1).all())
        assert int(d[0].storage_offset()) == 0 and pad_ok, "padding bytes beyond the row must stay untouched"
        # in place: dst planes are the src planes
        s2 = _to_dev(src, engine)
        engine.apply_yuv(s2, s2, pix_fmt="yuv420p10le")
        _assert_equal(_to_np(s2, np.uint16), want, f"in place {variant}")
    engine.set_variant("auto")
    # bottom-up source (negative row stride): torch cannot express it, so go through the C-ABI directly
    import ctypes as C
    from lut_renderer_amd import _native
    lib = _native.load()
    s3 = _to_dev(src, engine)
    d3 = [torch.zeros_like(t) for t in s3]
    sp, dp = _native.Planes(), _native.Planes()
    for i, (a, b) in enumerate(zip(s3, d3)):
        rows, rb = a.shape[0], a.shape[1] * 2
        sp.data[i] = a.data_ptr() + (rows - 1) * rb
        sp.stride[i] = -rb
        dp.data[i] = b.data_ptr() + (rows - 1) * rb
        dp.stride[i] = -rb
    p = _native.YuvParams(_native.fmt_code(10, 1, 1), _native.fmt_code(10, 1, 1), 10, 0, 0, 0, 0, 0)
    flipped = [np.ascontiguousarray(x[::-1]) for x in src]
    want_f = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 10, 1, 1, flipped)
    _native.check(lib.lutr_apply_yuv(engine._ctx, C.byref(p), 2, w, h, 1, C.byref(sp), C.byref(dp), 0, h))
    torch.cuda.synchronize()
    assert engine.last_kernel == "k_yuv_generic"
    _assert_equal([x[::-1] for x in _to_np(d3, np.uint16)], want_f, "bottom-up")


@pytest.mark.parametrize("pix_fmt", ["rgb24", "bgr24", "rgba", "bgra", "argb", "abgr", "rgb0", "0bgr",
                                     "rgb48le", "bgr48le", "rgba64le", "bgra64le"])
def test_packed_rgb_parity(engine, orc, cube_dir, pix_fmt):
    """the interleaved formats lut3d takes, vector and scalar kernels, every mode."""
    bits, nc = orc.PACKED[pix_fmt][:2]
    for lutname in ("log709_33.cube", "domain_2.cube"):
        lut = _load(engine, cube_dir, lutname)
        for (w, h) in ((128, 20), (37, 9)):                      # 37 wide -> scalar kernel
            rng = np.random.default_rng(0xC0BE + w)
            img = rng.integers(0, 1 << bits, size=(h, w, nc), dtype=np.uint16 if bits == 16 else np.uint8)
            if w == 128:                                         # smooth half: neighbouring cells, all six tetrahedra
                ramp = (np.add.outer(np.arange(h), np.arange(w)) * ((1 << bits) - 1) // (h + w)).astype(img.dtype)
                img[:, : w // 2, :3] = ramp[:, : w // 2, None] ^ rng.integers(0, 8, size=(h, w // 2, 3), dtype=img.dtype)
            dev = torch.from_numpy(img.view(np.int16) if bits == 16 else img).to(engine.device)
            for variant in ("auto", "generic"):
                engine.set_variant(variant)
                for mode in (MODES5 if variant == "generic" or w == 37 else MODES3):
                    want = orc.apply_packed(lut.table, lut.scale, pix_fmt, mode, img)
                    got = engine.apply_packed(dev, pix_fmt=pix_fmt, interp=mode).cpu().numpy()
                    got = got.view(np.uint16) if bits == 16 else got
                    _assert_equal([got], [want], f"packed {pix_fmt} {variant} {mode} {w}x{h} {lutname}")
                    # auto: the round-3 tube kernel (16-byte aligned rows of whole units), scalar kernel otherwise
                    expect = "k_rgb_tube" if (variant == "auto" and w == 128 and mode in MODES3) else "k_packed_generic"
                    assert engine.last_kernel.startswith(expect), engine.last_kernel
    engine.set_variant("auto")


def test_packed_rgb_batches_shards_in_place_and_padding(engine, orc, cube_dir):
    lut = _load(engine, cube_dir, "log709_33.cube")
    rng = np.random.default_rng(7)
    f, h, w = 3, 24, 64
    img = rng.integers(0, 256, size=(f, h, w, 4), dtype=np.uint8)
    want = np.stack([orc.apply_packed(lut.table, lut.scale, "bgra", "tetrahedral", img[i]) for i in range(f)])
    # padded rows: a [F,H,W+8,4] buffer viewed at W
    pad = torch.zeros((f, h, w + 8, 4), dtype=torch.uint8, device=engine.device)
    view = pad[:, :, :w, :]
    view.copy_(torch.from_numpy(img))
    out = torch.full_like(pad, 0x55)
    engine.apply_packed(view, out[:, :, :w, :], pix_fmt="bgra")
    assert np.array_equal(out[:, :, :w, :].cpu().numpy(), want)
    assert (out[:, :, w:, :] == 0x55).all()                      # padding untouched
    # two row shards, in place
    engine.apply_packed(view, view, pix_fmt="bgra", row0=0, rows=10)
    engine.apply_packed(view, view, pix_fmt="bgra", row0=10, rows=14)
    assert np.array_equal(view.cpu().numpy(), want)
    # wrong element size / unknown name fail loudly
    with pytest.raises(ValueError):
        engine.apply_packed(view, pix_fmt="rgb48le")
    with pytest.raises(ValueError):
        engine.apply_packed(view, pix_fmt="rgb565")


def test_unit_range_lattices_use_the_clip_free_kernels_and_others_do_not(engine, orc, cube_dir):
    """A lattice inside [0,1] selects the tile kernels without the output clip (provably dead code);
    anything else keeps it.  Both against the oracle, including the lattices that sit on the bounds."""
    from lut_renderer_amd.cube import CubeLut
    rng = np.random.default_rng(5)
    base = cube.log709_lattice(17)
    edge = rng.choice(np.array([0.0, 1.0, 1e-30, 1.0 - 2.0 ** -24, 0.5], dtype=np.float32), size=(9, 9, 9, 3))
    cases = [("ones", np.ones((5, 5, 5, 3), np.float32), True), ("zeros", np.zeros((5, 5, 5, 3), np.float32), True),
             ("edge", edge, True), ("log709", base, True), ("identity", cube.identity_lattice(33), True),
             ("wide", (base * 1.3 - 0.15).astype(np.float32), False),
             ("just_above", np.where(base >= 1.0, np.float32(1.0 + 2.0 ** -23), base).astype(np.float32), False),
             ("just_below", np.where(base <= 0.0, np.float32(-1e-38), base).astype(np.float32), False)]
    k10, k8 = orc.yuv_constants(din=10), orc.yuv_constants(din=8)
    for name, tab, unit in cases:
        lut = CubeLut(tab.shape[0], np.ones(3, dtype=np.float32), tab)
        engine.set_lut(lut)
        for mode in MODES3:
            for depth, kk, fmt in ((10, k10, "yuv420p10le"), (8, k8, "yuv420p")):
                src = frames.uniform_yuv(256, 64, depth, 1, 1, k=3)
                got = engine.apply_yuv(_to_dev(src, engine), pix_fmt=fmt, interp=mode)
                assert ("unit" in engine.last_kernel) == unit, (name, engine.last_kernel)
                want = orc.apply_yuv(tab, lut.scale, mode, kk, depth, depth, depth, 1, 1, src)
                _assert_equal(_to_np(got, src[0].dtype), want, f"unit/{name} {mode} {fmt}")
            rgb = frames.uniform_rgb(256, 36, 10, k=4)
            got = engine.apply_rgb(_to_dev(rgb, engine), depth=10, interp=mode)
            assert ("unit" in engine.last_kernel) == unit
            _assert_equal(_to_np(got, np.uint16), orc.apply_rgb(tab, lut.scale, 10, mode, rgb), f"unit/{name} {mode} rgb")


def test_seal_records_range_and_rejects_non_finite(cube_dir):
    """lutr_ctx_lut_alloc + a write through lutr_ctx_lut_device (what a broadcast does) + lutr_ctx_lut_seal."""
    import ctypes as C
    from lut_renderer_amd import _native
    from lut_renderer_amd.engine import LutEngine
    lut = cube.read_cube(cube_dir / "log709_33.cube")
    src = frames.natural_yuv(256, 64, 10, 1, 1, k=1)
    with LutEngine(0) as a, LutEngine(0) as b:
        a.set_lut(lut)
        want = a.apply_yuv(_to_dev(src, a), pix_fmt="yuv420p10le")
        assert "unit" in a.last_kernel
        scale = (C.c_float * 3)(1.0, 1.0, 1.0)
        _native.check(b._lib.lutr_ctx_lut_alloc(b._ctx, 33, scale))
        b.n, b.scale = 33, np.ones(3, np.float32)
        b.lattice_tensor().copy_(a.lattice_tensor())
        torch.cuda.synchronize()
        got = b.apply_yuv(_to_dev(src, b), pix_fmt="yuv420p10le")
        assert "tile" in b.last_kernel and "unit" not in b.last_kernel      # unsealed: general kernel
        _native.check(b._lib.lutr_ctx_lut_seal(b._ctx))
        got2 = b.apply_yuv(_to_dev(src, b), pix_fmt="yuv420p10le")
        assert "unit" in b.last_kernel
        for x, y, z in zip(want, got, got2):
            assert torch.equal(x, y) and torch.equal(x, z)
        b.lattice_tensor()[5] = float("nan")
        torch.cuda.synchronize()
        with pytest.raises(_native.LutrError) as ei:
            _native.check(b._lib.lutr_ctx_lut_seal(b._ctx))
        assert ei.value.code == _native.EINVAL


def test_non_cube_lut_files_end_to_end(engine, orc, tmp_path):
    """a .3dl and a .dat file through the product reader and the kernels, against the
    oracle's reader and pixel path."""
    rng = np.random.default_rng(12)
    codes = np.sort(rng.integers(0, 4096, size=(17, 17, 17, 3)), axis=0)
    p3 = tmp_path / "look.3dl"
    p3.write_text(" ".join(str(min(64 * i, 1023)) for i in range(17)) + "\n" +
                  "".join("%d %d %d\n" % tuple(codes[r, g, b]) for r in range(17) for g in range(17) for b in range(17)))
    tab = cube.log709_lattice(9)
    pd = tmp_path / "look.dat"
    pd.write_text("3DLUTSIZE 9\n" + "".join("%.7f %.7f %.7f\n" % tuple(tab[r, g, b]) for r in range(9)
                                             for g in range(9) for b in range(9)))
    src = frames.natural_yuv(256, 64, 10, 1, 1, k=2)
    k = orc.yuv_constants(din=10)
    for path in (p3, pd):
        lut = engine.load_cube(path)
        n, scale, table = orc.parse_lut_file(path)
        assert lut.n == n and np.array_equal(lut.table, table)
        for mode in ("tetrahedral", "trilinear"):
            got = engine.apply_yuv(_to_dev(src, engine), pix_fmt="yuv420p10le", interp=mode)
            _assert_equal(_to_np(got, np.uint16), orc.apply_yuv(table, scale, mode, k, 10, 10, 10, 1, 1, src),
                          f"{path.suffix} {mode}")


def _torchrun(nproc, script_args, extra_env, timeout=300):
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, LUTR_DIST_BACKEND="gloo", LUTR_FORCE_DEVICE="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_rank_lut_broadcast_and_row_blocks(orc, cube_dir, tmp_path):
    """The N>1 control flow with two ranks sharing the one GPU of the test box (collective over gloo): the
    receiving rank's alloc -> broadcast -> seal path and per-rank row blocks against the whole-frame oracle."""
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    res = _torchrun(2, [str(root / "tests" / "_gpu_dist_worker.py"), str(tmp_path), str(cube_dir / "log709_33.cube")], {})
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lut = cube.read_cube(cube_dir / "log709_33.cube")
    src = frames.natural_yuv(256, 72, 10, 1, 1, k=9)
    want = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", orc.yuv_constants(din=10), 10, 10, 10, 1, 1, src)
    rows = 0
    for rank in range(2):
        d = np.load(tmp_path / f"rank{rank}.npz")
        r0, r1 = int(d["r0"]), int(d["r1"])
        rows += r1 - r0
        assert "unit" in str(d["kernel"])                      # the received lattice was sealed on rank 1 too
        assert np.array_equal(d["y"][r0:r1], want[0][r0:r1])
        assert np.array_equal(d["cb"][r0 // 2:r1 // 2], want[1][r0 // 2:r1 // 2])
        assert np.array_equal(d["cr"][r0 // 2:r1 // 2], want[2][r0 // 2:r1 // 2])
    assert rows == 72


def test_bench_two_ranks_rehearsal():
    """bench.py's N>1 path end to end (torchrun, barriers, max/sum reductions, one JSON line from rank 0)."""
    import json
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    res = _torchrun(2, [str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "2",
                        "--size", "1080p", "--pipeline", "host", "--host-frames", "20"], {})
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["cpu_baseline"] is None
    assert d["config"]["parallelism"] == "row-block x2" and "tile" in d["config"]["kernel"]
    # BASELINE config 5 over the ranks: whole frames round-robin, every rank drives its own pinned ring
    assert d["host_pipeline"]["gpus"] == 2 and d["host_pipeline"]["frames"] == 20 and d["host_pipeline"]["fps"] > 0


@pytest.mark.parametrize("fmt,out_fmt,w,h,nframes", [
    ("yuv420p10le", "yuv420p10le", 256, 200, 2),     # 4 bands of 64 rows: bands pipelined over several waves
    ("yuv420p10le", "yuv420p", 320, 136, 1),         # 10-bit LUT path, 8-bit dithered output (the reference's use)
    ("yuv420p", "yuv420p", 128, 64, 3),
    ("yuv444p10le", "yuv444p10le", 64, 130, 1),
    ("yuv422p10le", "yuv422p10le", 74, 70, 1),
    ("yuv420p10le", "yuv420p10le", 37, 19, 2),       # odd sizes: edge replication into the last chroma block
    ("yuv420p", "yuv420p", 2, 2, 1),
])
def test_error_diffusion_dither_parity(engine, orc, cube_dir, fmt, out_fmt, w, h, nframes):
    """`zscale_dither=error_diffusion`.  Floyd-Steinberg is sequential, so any slip in
    the skewed-wave pipeline shows up as a different bit pattern; the bar is exact equality."""
    from lut_renderer_amd.engine import parse_pix_fmt
    lut = _load(engine, cube_dir, "log709_33.cube")
    fin, fout = parse_pix_fmt(fmt), parse_pix_fmt(out_fmt)
    k = orc.yuv_constants(din=fin.depth, dl=fin.depth, dout=fout.depth, chroma_n=1 << (fin.csx + fin.csy))
    batch = [frames.natural_yuv(w, h, fin.depth, fin.csx, fin.csy, k=i) for i in range(nframes)]
    for mode in ("tetrahedral", "trilinear"):
        want = [orc.apply_yuv(lut.table, lut.scale, mode, k, fin.depth, fin.depth, fout.depth, fin.csx, fin.csy, f,
                              dither="error_diffusion") for f in batch]
        src = [torch.from_numpy(np.stack([f[i] for f in batch]).view(np.int16 if fin.depth > 8 else np.uint8))
               .to(engine.device) for i in range(3)]
        got = engine.apply_yuv(src, pix_fmt=fmt, out_pix_fmt=out_fmt, interp=mode, dither="error_diffusion")
        assert "dither" in engine.last_kernel
        for i in range(3):
            g = got[i].cpu().numpy()
            g = g.view(np.uint16) if fout.depth > 8 else g
            for fr in range(nframes):
                _assert_equal([g[fr]], [want[fr][i]], f"dither {fmt}->{out_fmt} {mode} plane {i} frame {fr}")
    with pytest.raises(ValueError):
        engine.apply_yuv(src, pix_fmt=fmt, out_pix_fmt=out_fmt, dither="error_diffusion", row0=0, rows=h // 2 & ~1)
    with pytest.raises(ValueError):
        engine.apply_yuv(src, pix_fmt=fmt, out_pix_fmt=out_fmt, dither="ordered")


def test_error_diffusion_dither_1080p_frame(engine, orc, cube_dir):
    """A full 1080p frame: 17 bands, wide error rows (the LDS budget picks the number of waves per plane)."""
    lut = _load(engine, cube_dir, "log709_33.cube")
    src = frames.natural_yuv(1920, 1080, 10, 1, 1, k=1)
    k = orc.yuv_constants(din=10, dl=10, dout=8)
    want = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 8, 1, 1, src, nthreads=8, dither="error_diffusion")
    got = engine.apply_yuv(_to_dev(src, engine), pix_fmt="yuv420p10le", out_pix_fmt="yuv420p", dither="error_diffusion")
    _assert_equal([t.cpu().numpy() for t in got], want, "dither 1080p")


@pytest.mark.parametrize("fmt,w", [("yuv420p10le", 166), ("yuv420p10le", 165), ("yuv420p", 89), ("yuv422p10le", 70),
                                   ("yuv444p10le", 43), ("gbrp10le", 77), ("gbrp", 50)])
def test_ragged_width_on_padded_rows_splits_between_fast_and_scalar_kernels(engine, orc, cube_dir, fmt, w):
    """Decoder-style frames: a width that is not a multiple of the fast kernel's unit (1366, DCI 1998, ...) in
    rows padded to an aligned stride.  The tile kernel takes the whole units, the scalar kernel the rest."""
    from lut_renderer_amd.engine import parse_pix_fmt
    lut = _load(engine, cube_dir, "log709_33.cube")
    pf = parse_pix_fmt(fmt)
    h, nf = 36, 2
    esz = 2 if pf.depth > 8 else 1
    tdt = torch.int16 if esz == 2 else torch.uint8
    if pf.family == "gbr":
        frames_np = [frames.natural_rgb(w, h, pf.depth, k=i) for i in range(nf)]
    else:
        frames_np = [frames.natural_yuv(w, h, pf.depth, pf.csx, pf.csy, k=i) for i in range(nf)]
    src, dst = [], []
    for i in range(3):
        ph, pw = pf.plane_shape(i, w, h)
        stride = ((pw * esz + 8 + 63) // 64) * 64 // esz            # rows padded to a 64-byte multiple, like a decoder's
        buf = torch.zeros((nf, ph, stride), dtype=tdt, device=engine.device)
        arr = np.stack([f[i] for f in frames_np])
        buf[:, :, :pw] = torch.from_numpy(arr.view(np.int16) if esz == 2 else arr).to(engine.device)
        src.append(buf[:, :, :pw])
        dst.append(torch.full_like(buf, 0x55)[:, :, :pw])
    for mode in ("tetrahedral", "trilinear"):
        if pf.family == "gbr":
            got = engine.apply_rgb(src, dst, depth=pf.depth, interp=mode)
            want = [orc.apply_rgb(lut.table, lut.scale, pf.depth, mode, f) for f in frames_np]
        else:
            got = engine.apply_yuv(src, dst, pix_fmt=fmt, interp=mode)
            k = orc.yuv_constants(din=pf.depth, chroma_n=1 << (pf.csx + pf.csy))
            want = [orc.apply_yuv(lut.table, lut.scale, mode, k, pf.depth, pf.depth, pf.depth, pf.csx, pf.csy, f)
                    for f in frames_np]
        assert "tile" in engine.last_kernel, engine.last_kernel
        for i in range(3):
            g = got[i].cpu().numpy()
            g = g.view(np.uint16) if esz == 2 else g
            for fr in range(nf):
                _assert_equal([g[fr]], [want[fr][i]], f"ragged {fmt} w={w} {mode} plane {i} frame {fr}")


def test_randomised_differential_sweep(engine, orc, tmp_path):
    """120 seeded random cases over formats, depths, sizes (whole units, ragged, odd), padded / dense rows, batches,
    row shards, matrices, ranges, lattice sizes (incl. out-of-[0,1] lattices) and modes -- every one bit-exact."""
    from lut_renderer_amd.cube import CubeLut
    from lut_renderer_amd.engine import parse_pix_fmt
    rng = np.random.default_rng(20260204)
    fmts = ["yuv420p", "yuv422p", "yuv444p", "yuv420p10le", "yuv422p10le", "yuv444p10le", "yuv420p12le", "gbrp", "gbrp10le",
            "gbrp12le", "gbrp16le"]
    mats = ["bt709", "smpte170m", "bt2020nc"]
    kernels = set()
    for case in range(120):
        fmt = fmts[rng.integers(len(fmts))]
        pf = parse_pix_fmt(fmt)
        n = int(rng.choice([2, 5, 9, 17, 33]))
        tab = rng.random((n, n, n, 3)).astype(np.float32)
        if rng.random() < 0.3:
            tab = (tab * 1.4 - 0.2).astype(np.float32)                      # values outside [0,1]: clipping kernels
        scale = np.ones(3, np.float32) if rng.random() < 0.7 else rng.choice([0.25, 0.5, 1.0], size=3).astype(np.float32)
        engine.set_lut(CubeLut(n, scale, tab))
        unit = 16 if pf.depth <= 8 else 8
        w = int(rng.choice([unit * rng.integers(1, 12), unit * rng.integers(1, 12) + rng.integers(1, unit), rng.integers(1, 40)]))
        h = int(rng.choice([2 * rng.integers(1, 40), rng.integers(1, 50)]))
        nf = int(rng.choice([1, 1, 2, 3]))
        mode = MODES5[rng.integers(5)] if rng.random() < 0.2 else MODES3[rng.integers(3)]
        esz = 2 if pf.depth > 8 else 1
        tdt = torch.int16 if esz == 2 else torch.uint8
        padded = rng.random() < 0.6
        if pf.family == "gbr":
            fr_np = [frames.uniform_rgb(w, h, pf.depth, k=case * 7 + i) if rng.random() < 0.5 else
                     frames.natural_rgb(w, h, pf.depth, k=case * 7 + i) for i in range(nf)]
        else:
            fr_np = [frames.make_yuv("uniform" if rng.random() < 0.5 else "natural", w, h, pf.depth, pf.csx, pf.csy,
                                     k=case * 7 + i) for i in range(nf)]
        src, dst, guards = [], [], []
        for i in range(3):
            ph, pw = pf.plane_shape(i, w, h)
            stride = ((pw * esz + 63) // 64) * 64 // esz if padded else pw
            buf = torch.zeros((nf, ph, stride), dtype=tdt, device=engine.device)
            arr = np.stack([f[i] for f in fr_np])
            buf[:, :, :pw] = torch.from_numpy(arr.view(np.int16) if esz == 2 else arr).to(engine.device)
            src.append(buf[:, :, :pw])
            # destination with a guard row above and below every frame and (when padded) guard columns to the right
            guard = torch.full((nf, ph + 2, stride), 0x5A5A if esz == 2 else 0x5A, dtype=tdt, device=engine.device)
            guards.append(guard)
            dst.append(guard[:, 1:ph + 1, :pw])
        what = f"case {case}: {fmt} {w}x{h}x{nf} N={n} {mode} padded={padded}"
        if pf.family == "gbr":
            got = engine.apply_rgb(src, dst, depth=pf.depth, interp=mode)
            want = [orc.apply_rgb(tab, scale, pf.depth, mode, f) for f in fr_np]
        else:
            m_in, m_out = mats[rng.integers(3)], mats[rng.integers(3)]
            r_out = "tv" if rng.random() < 0.7 else "pc"
            bh = 1 << pf.csy
            shard = h % bh == 0 and h >= 4 * bh and rng.random() < 0.3
            k = orc.yuv_constants(m_in, "tv", m_out, r_out, pf.depth, pf.depth, pf.depth, 1 << (pf.csx + pf.csy))
            kw = dict(pix_fmt=fmt, interp=mode, matrix_in=m_in, matrix_out=m_out, range_out=r_out)
            if shard:
                cut = (h // 2) // bh * bh
                engine.apply_yuv(src, dst, row0=0, rows=cut, **kw)
                got = engine.apply_yuv(src, dst, row0=cut, rows=h - cut, **kw)
            else:
                got = engine.apply_yuv(src, dst, **kw)
            want = [orc.apply_yuv(tab, scale, mode, k, pf.depth, pf.depth, pf.depth, pf.csx, pf.csy, f) for f in fr_np]
            what += f" {m_in}->{m_out} {r_out} shard={shard}"
        kernels.add(engine.last_kernel.split("<")[0])
        for i in range(3):                                                      # nothing outside the planes was written
            ph, pw = pf.plane_shape(i, w, h)
            gv = 0x5A5A if esz == 2 else 0x5A
            assert (guards[i][:, 0, :] == gv).all() and (guards[i][:, ph + 1, :] == gv).all(), what + " guard rows"
            assert (guards[i][:, :, pw:] == gv).all(), what + " guard columns"
        for i in range(3):
            g = got[i].cpu().numpy()
            g = g.view(np.uint16) if esz == 2 else g
            for fr in range(nf):
                _assert_equal([g[fr]], [want[fr][i]], f"{what} plane {i} frame {fr} ({engine.last_kernel})")
    assert {"k_yuv_tile2", "k_rgb_tile", "k_yuv_generic", "k_rgb_generic"} <= kernels, kernels


def test_contexts_on_concurrent_threads(orc, cube_dir):
    """The reference runs up to 16 TaskRunners at once (task_manager.py:229-235): one context per thread, all
    launching concurrently, each on its own stream; every result must still be exact."""
    import threading
    from lut_renderer_amd.engine import LutEngine
    lut = cube.read_cube(cube_dir / "log709_33.cube")
    k = orc.yuv_constants(din=10)
    jobs = []
    for t in range(6):
        src = frames.natural_yuv(256, 64 + 16 * t, 10, 1, 1, k=20 + t)
        jobs.append((src, orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 10, 1, 1, src)))
    errors = []

    def work(i):
        try:
            src, want = jobs[i]
            with LutEngine(0, use_torch_stream=False) as eng:          # the context's own stream
                eng.set_lut(lut)
                dev = [torch.from_numpy(p.view(np.int16)).to(eng.device) for p in src]
                torch.cuda.synchronize()
                for _ in range(20):
                    out = eng.apply_yuv(dev, pix_fmt="yuv420p10le")
                eng.sync()
                got = [t.cpu().numpy().view(np.uint16) for t in out]
                _assert_equal(got, want, f"thread {i}")
        except Exception as exc:  # noqa: BLE001
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_small_jobs_take_the_low_latency_kernels(engine, orc, cube_dir, monkeypatch):
    """Under "auto" a launch below the small-job boundary (33 Mpx by default since round 3's two-level chunk queue) runs on the
    plain vector kernels (lower latency for one frame), anything larger on the tile / tube kernels; same pixels either way."""
    lut = _load(engine, cube_dir, "log709_33.cube")
    src = frames.natural_yuv(512, 128, 10, 1, 1, k=5)
    rgb = frames.natural_rgb(512, 128, 10, k=5)
    k = orc.yuv_constants(din=10)
    want = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 10, 1, 1, src)
    want_rgb = orc.apply_rgb(lut.table, lut.scale, 10, "tetrahedral", rgb)
    for mpx, expect in (("95", "k_yuv_vec"), ("0", "k_yuv_tile")):
        monkeypatch.setenv("LUTR_SMALL_JOB_MPX", mpx)
        got = engine.apply_yuv(_to_dev(src, engine), pix_fmt="yuv420p10le")
        assert engine.last_kernel.startswith(expect), engine.last_kernel
        _assert_equal(_to_np(got, np.uint16), want, f"small job boundary {mpx}")
        got = engine.apply_rgb(_to_dev(rgb, engine), depth=10)
        assert engine.last_kernel.startswith("k_rgb_vec" if mpx == "95" else "k_rgb_tube"), engine.last_kernel
        _assert_equal(_to_np(got, np.uint16), want_rgb, f"small job boundary {mpx} rgb")
