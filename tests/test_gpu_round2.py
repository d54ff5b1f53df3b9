"""GPU tests added in round 2: the configs the first round left unexercised (BASELINE configs 4 and 5), the
single-process multi-GPU entry, stream changes, layout limits, the self-launching bench and the CLI's SIGTERM branch.
All pixel comparisons are bit-exact against the CPU oracle unless a test says otherwise."""
import json
import os
import signal
import subprocess
import sys
import threading
import time
from pathlib import Path

import ctypes as C
import numpy as np
import pytest
import torch

from lut_renderer_amd import _native, cube, frames

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _to_dev(planes, eng):
    return [torch.from_numpy(p.view(np.int16) if p.dtype == np.uint16 else p).to(eng.device) for p in planes]


def _to_np(tensors, like_dtype):
    return [t.cpu().numpy().view(np.uint16) if like_dtype == np.uint16 else t.cpu().numpy() for t in tensors]


def _assert_equal(got, want, what):
    for i, (a, b) in enumerate(zip(got, want)):
        if not np.array_equal(a, b):
            diff = np.abs(a.astype(np.int64) - b.astype(np.int64))
            bad = np.argwhere(diff > 0)
            raise AssertionError(f"{what}: plane {i} differs at {len(bad)} samples, max |d|={diff.max()}, "
                                 f"first {bad[0].tolist()} got {a[tuple(bad[0])]} want {b[tuple(bad[0])]}")


def _load(eng, cube_dir, name):
    lut = cube.read_cube(cube_dir / name)
    eng.set_lut(lut)
    return lut


# ------------------------------------------------------------------ BASELINE config 5: pc -> tv prologue, 4:2:0 10 bit
@pytest.mark.parametrize("mode", ["tetrahedral", "trilinear"])
@pytest.mark.parametrize("out_fmt,dout", [("yuv420p10le", 10), ("yuv420p", 8)])
def test_config5_prologue_runs_on_the_tile_kernels(engine, orc, cube_dir, out_fmt, dout, mode):
    """yuv420p10le FULL-range source -> scale=in_range=pc:out_range=tv,format=yuv420p (8 bit) -> lut3d at 8 bit ->
    format=<10 or 8 bit> (ffmpeg.py:212-233, :304-310): the fused prologue of the LDS-window tile kernels."""
    lut = _load(engine, cube_dir, "log709_33.cube")
    k = orc.yuv_constants("bt709", "tv", "bt709", "tv", 10, 8, dout, 4, prologue=True)
    for dist in ("natural", "uniform"):
        src = frames.make_yuv(dist, 256, 72, 10, 1, 1, k=21, full_range=True)
        want = orc.apply_yuv(lut.table, lut.scale, mode, k, 10, 8, dout, 1, 1, src)
        for variant in ("vec_lds", "vec_global", "generic"):
            engine.set_variant(variant)
            got = engine.apply_yuv(_to_dev(src, engine), pix_fmt="yuv420p10le", out_pix_fmt=out_fmt, interp=mode,
                                   range_src="pc", range_in="tv", lut_depth=8)
            if variant == "vec_lds":
                assert "tile" in engine.last_kernel and "pre" in engine.last_kernel, engine.last_kernel
            _assert_equal(_to_np(got, np.uint16 if dout > 8 else np.uint8), want, f"config 5 {dist} {variant} -> {out_fmt}")
    engine.set_variant("auto")


def test_depth_changing_output_takes_the_tile_kernels(engine, orc, cube_dir):
    """The reference's default codec is libx264, so a 10-bit source gets format=yuv420p (ffmpeg.py:287-302, App. D case K);
    8-bit sources go to 10 bit with x265 main10.  Both directions, every chroma layout."""
    lut = _load(engine, cube_dir, "log709_33.cube")
    engine.set_variant("vec_lds")
    for fmt, out_fmt, cs in (("yuv420p10le", "yuv420p", (1, 1)), ("yuv422p10le", "yuv422p", (1, 0)),
                             ("yuv444p10le", "yuv444p", (0, 0)), ("yuv420p12le", "yuv420p", (1, 1)),
                             ("yuv420p", "yuv420p10le", (1, 1))):
        din = 8 if fmt.endswith("p") else int(fmt.rstrip("le")[-2:])
        dout = 8 if out_fmt.endswith("p") else int(out_fmt.rstrip("le")[-2:])
        src = frames.natural_yuv(256, 72, din, cs[0], cs[1], k=22)
        k = orc.yuv_constants("bt2020nc", "tv", "bt2020nc", "tv", din, din, dout, 1 << sum(cs))
        for mode in ("tetrahedral", "trilinear", "nearest"):
            want = orc.apply_yuv(lut.table, lut.scale, mode, k, din, din, dout, cs[0], cs[1], src)
            # 8 -> 10 bit has no tile instance (an 8-bit source keeps 8 bit under the reference's "preserve" policy)
            engine.set_variant("vec_lds" if din > 8 else "auto")
            got = engine.apply_yuv(_to_dev(src, engine), pix_fmt=fmt, out_pix_fmt=out_fmt, interp=mode, matrix_in="bt2020nc")
            assert "tile" in engine.last_kernel or din == 8, engine.last_kernel
            _assert_equal(_to_np(got, np.uint16 if dout > 8 else np.uint8), want, f"{fmt} -> {out_fmt} {mode}")
    engine.set_variant("auto")


# ------------------------------------------------------------------ BASELINE config 4 at full size
@pytest.mark.parametrize("mode", ["tetrahedral", "trilinear"])
def test_full_size_properties_8k(engine, orc, cube_dir, mode):
    """7680x4320 yuv420p10le, 33^3: tile == generic on the whole frame, 8 row blocks == whole frame, oracle on a strip."""
    w, h = 7680, 4320
    src_np = frames.natural_yuv(w, h, 10, 1, 1, k=1)
    src = _to_dev(src_np, engine)
    lut = _load(engine, cube_dir, "log709_33.cube")
    engine.set_variant("generic")
    ref = engine.apply_yuv(src, pix_fmt="yuv420p10le", interp=mode)
    engine.set_variant("vec_lds")
    fast = engine.apply_yuv(src, pix_fmt="yuv420p10le", interp=mode)
    assert "tile" in engine.last_kernel
    for a, b in zip(ref, fast):
        assert torch.equal(a, b)
    from lut_renderer_amd.shard import row_blocks
    dst = [torch.zeros_like(t) for t in src]
    for r0, r1 in row_blocks(h, 8, align=2):
        assert r1 - r0 == 540
        engine.apply_yuv(src, dst, pix_fmt="yuv420p10le", interp=mode, row0=r0, rows=r1 - r0)
    for a, b in zip(dst, fast):
        assert torch.equal(a, b)
    y0 = 2048
    strip = [src_np[0][y0:y0 + 32], src_np[1][y0 // 2:y0 // 2 + 16], src_np[2][y0 // 2:y0 // 2 + 16]]
    want = orc.apply_yuv(lut.table, lut.scale, mode, orc.yuv_constants(din=10), 10, 10, 10, 1, 1, strip, nthreads=8)
    got = _to_np(fast, np.uint16)
    _assert_equal([got[0][y0:y0 + 32], got[1][y0 // 2:y0 // 2 + 16], got[2][y0 // 2:y0 // 2 + 16]], want, "8k strip")
    engine.set_variant("auto")


def test_lattice_65_full_uhd_frame(engine, orc, cube_dir):
    """BASELINE config 3's lattice (65^3) on a whole UHD frame: tile == generic, oracle strip."""
    w, h = 3840, 2160
    src_np = frames.natural_yuv(w, h, 10, 1, 1, k=2)
    src = _to_dev(src_np, engine)
    lut = _load(engine, cube_dir, "log709_65.cube")
    engine.set_variant("generic")
    ref = engine.apply_yuv(src, pix_fmt="yuv420p10le")
    engine.set_variant("vec_lds")
    fast = engine.apply_yuv(src, pix_fmt="yuv420p10le")
    for a, b in zip(ref, fast):
        assert torch.equal(a, b)
    strip = [src_np[0][1000:1032], src_np[1][500:516], src_np[2][500:516]]
    want = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", orc.yuv_constants(din=10), 10, 10, 10, 1, 1, strip, nthreads=8)
    got = _to_np(fast, np.uint16)
    _assert_equal([got[0][1000:1032], got[1][500:516], got[2][500:516]], want, "65^3 strip")
    engine.set_variant("auto")


# ------------------------------------------------------------------ one context, two streams
def test_applies_from_two_streams_on_one_engine_serialise(engine, orc, cube_dir):
    """A context owns ONE work queue: applies issued from different torch streams must not overlap on the GPU
    (lutr_ctx_set_stream orders the new stream behind the pending work).  Two tile-kernel launches, two streams."""
    lut = _load(engine, cube_dir, "log709_33.cube")
    engine.set_variant("vec_lds")
    w, h, nf = 1920, 1080, 6
    a_np = frames.natural_yuv(w, h, 10, 1, 1, k=31)
    b_np = frames.make_yuv("noise16", w, h, 10, 1, 1, k=32)
    a = [t.unsqueeze(0).repeat(nf, 1, 1) for t in _to_dev(a_np, engine)]
    b = [t.unsqueeze(0).repeat(nf, 1, 1) for t in _to_dev(b_np, engine)]
    ref_a = [t.clone() for t in engine.apply_yuv(a, pix_fmt="yuv420p10le")]
    ref_b = [t.clone() for t in engine.apply_yuv(b, pix_fmt="yuv420p10le")]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(engine.device), torch.cuda.Stream(engine.device)
    for _ in range(4):
        out_a = [torch.zeros_like(t) for t in a]
        out_b = [torch.zeros_like(t) for t in b]
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            engine.apply_yuv(a, out_a, pix_fmt="yuv420p10le")
        with torch.cuda.stream(s2):
            engine.apply_yuv(b, out_b, pix_fmt="yuv420p10le")
        with torch.cuda.stream(s1):
            engine.apply_yuv(a, out_a, pix_fmt="yuv420p10le")
        torch.cuda.synchronize()
        for x, y in zip(out_a + out_b, ref_a + ref_b):
            assert torch.equal(x, y)
    engine.set_variant("auto")


# ------------------------------------------------------------------ one process, several contexts (devices=[...])
def test_engine_group_two_contexts_split_the_rows(orc, cube_dir):
    """`LutEngineGroup([0, 0])`: two contexts, lattice copied device to device by lutr_lut_broadcast, two row blocks
    == the whole-frame oracle.  (On an 8-GPU node the same code runs with devices=[0..7].)"""
    from lut_renderer_amd.multigpu import LutEngineGroup
    lut = cube.read_cube(cube_dir / "log709_33.cube")
    with LutEngineGroup([0, 0]) as grp:
        grp.set_lut(lut)
        grp.set_variant("vec_lds")
        for fmt, depth, cs, h in (("yuv420p10le", 10, (1, 1), 74), ("yuv422p", 8, (1, 0), 37)):
            src = frames.natural_yuv(256, h, depth, cs[0], cs[1], k=41)
            k = orc.yuv_constants("bt709", "tv", "bt709", "tv", depth, depth, depth, 1 << sum(cs))
            want = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, depth, depth, depth, cs[0], cs[1], src)
            got = grp.apply_yuv(_to_dev(src, grp.engines[0]), pix_fmt=fmt)
            grp.sync()
            assert len(grp.last_blocks) == 2 and grp.last_blocks[0][1] == grp.last_blocks[1][0] > 0
            assert all("unit" in name for name in grp.last_kernels), grp.last_kernels     # the copy inherited the seal
            _assert_equal(_to_np(got, np.uint16 if depth > 8 else np.uint8), want, f"group {fmt}")


def test_apply_lut_with_a_device_list(orc, cube_dir):
    from lut_renderer_amd import api
    lut = cube.read_cube(cube_dir / "log709_33.cube")
    src = frames.natural_yuv(256, 72, 10, 1, 1, k=42)
    dev = [torch.from_numpy(p.view(np.int16)).to("cuda:0") for p in src]
    k = orc.yuv_constants("bt2020nc", "tv", "bt2020nc", "tv", 10, 10, 10, 4)
    want = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 10, 1, 1, src)
    try:
        for devices in ([0], [0, 0], [0, 0, 0]):
            out, tags = api.apply_lut(dev, cube=cube_dir / "log709_33.cube", pix_fmt="yuv420p10le", colorspace="bt2020nc",
                                      color_range="tv", devices=devices)
            _assert_equal(_to_np(out, np.uint16), want, f"apply_lut devices={devices}")
            assert tags["color_range"] == "tv"
        assert set(api._engine_cache) == {(0,), (0, 0), (0, 0, 0)}       # contexts are kept between calls
        with pytest.raises(ValueError):
            api.apply_lut(dev, cube=None, pix_fmt="yuv420p10le", devices=[])
    finally:
        api.close_cached_engines()


# ------------------------------------------------------------------ layout limits
def test_packed_component_orders_outside_rgb_bgr_take_the_scalar_kernel(engine, orc, cube_dir):
    lut = _load(engine, cube_dir, "log709_33.cube")
    lib = _native.load()
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(16, 64, 3), dtype=np.uint8)
    for ro, go, bo in ((1, 0, 2), (0, 2, 1), (2, 0, 1), (1, 2, 0)):
        g, b, r = orc.apply_rgb(lut.table, lut.scale, 8, "tetrahedral", (img[..., go], img[..., bo], img[..., ro]))
        want = img.copy()
        want[..., ro], want[..., go], want[..., bo] = r, g, b
        s = torch.from_numpy(img).to(engine.device)
        d = torch.empty_like(s)
        ps, pd = _native.Packed(), _native.Packed()
        ps.data, ps.stride, ps.frame_stride = s.data_ptr(), 64 * 3, 0
        pd.data, pd.stride, pd.frame_stride = d.data_ptr(), 64 * 3, 0
        engine._bind_stream()
        _native.check(lib.lutr_apply_packed_rgb(engine._ctx, _native.packed_code(8, 3, ro, go, bo), 2, 64, 16, 1,
                                                C.byref(ps), C.byref(pd), 0, 16))
        assert engine.last_kernel == "k_packed_generic"
        assert np.array_equal(d.cpu().numpy(), want), (ro, go, bo)


def test_strides_beyond_the_tile_kernels_reach_take_the_scalar_kernel(engine, orc, cube_dir):
    """The tile kernels form row offsets in 32 bits; a stride they cannot address must select k_*_generic."""
    lut = _load(engine, cube_dir, "log709_33.cube")
    stride = 1 << 27                                  # 134,217,728 B: 16-byte aligned, above (2^32 - 1 - 1024) / 32
    w, h = 64, 4
    src = frames.natural_rgb(w, h, 8, k=7)
    want = orc.apply_rgb(lut.table, lut.scale, 8, "tetrahedral", src)
    bufs = [torch.zeros(stride * (h - 1) + w, dtype=torch.uint8, device=engine.device) for _ in range(6)]
    sp = [torch.as_strided(b, (h, w), (stride, 1)) for b in bufs[:3]]
    dp = [torch.as_strided(b, (h, w), (stride, 1)) for b in bufs[3:]]
    for v, p in zip(sp, src):
        v.copy_(torch.from_numpy(p).to(engine.device))
    engine.set_variant("auto")
    engine.apply_rgb(sp, dp, depth=8)
    assert engine.last_kernel == "k_rgb_generic"
    _assert_equal([t.cpu().numpy() for t in dp], want, "huge stride")
    with pytest.raises(_native.LutrError):
        engine.set_variant("vec_lds")
        engine.apply_rgb(sp, dp, depth=8)
    engine.set_variant("auto")


# ------------------------------------------------------------------ bench.py starts its own ranks
def test_bench_starts_its_own_ranks_and_reports_the_collective():
    """`python bench.py --gpus 2` with no launcher around it (how the driver calls it): the parent spawns two ranks
    before touching the GPU and relays rank 0's line.  Both ranks share this box's one GPU; collectives over gloo."""
    env = dict(os.environ, LUTR_DIST_BACKEND="gloo", LUTR_FORCE_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--frames", "2", "--size", "1080p"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["collective"]["world"] == 2 and d["collective"]["bcast_bytes"] > 600000 and d["collective"]["bcast_us"] > 0
    assert d["collective"]["data_path_collectives"] == 0
    assert d["strong"]["frames"] == 2 and d["strong"]["rows_per_gpu"] == 540 and d["strong"]["value"] > 0
    assert "rows [0,540)" in d["config"]["workload"]


def test_bench_single_gpu_line_has_every_contract_field():
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--frames", "4",
                          "--cpu-seconds", "1", "--pipeline", "host", "--host-frames", "16", "--range-src", "pc"],
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "extra_Mpx_s", "host_pipeline"):
        assert key in d, key
    assert d["config"]["range_src"] == "pc" and "pre" in d["config"]["kernel"]
    assert "pre" in d["host_pipeline"]["kernel"] and d["host_pipeline"]["prologue"].startswith("scale=in_range=pc")
    assert set(d["extra_Mpx_s"]) == {"vivid", "noise8", "noise16", "noise64", "uniform"}
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1


# ------------------------------------------------------------------ the CLI's SIGTERM branch (task_manager.py:38-44, :180-183)
def test_cli_stops_on_sigterm_with_whole_frames_written(cube_dir, tmp_path):
    w, h, nframes = 640, 360, 4000
    fb = w * h * 3 // 2
    src = tmp_path / "in.yuv"
    with open(src, "wb") as f:
        f.truncate(fb * nframes)                      # sparse: reads as zeros
    fifo = tmp_path / "out.yuv"
    os.mkfifo(fifo)
    proc = subprocess.Popen([sys.executable, "-m", "lut_renderer_amd.cli", "-i", str(src), "-o", str(fifo), "-y",
                             "--size", f"{w}x{h}", "--pix-fmt", "yuv420p", "--cube", str(cube_dir / "log709_33.cube"),
                             "--colorspace", "bt709", "--color-range", "tv", "--batch", "4"],
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, bufsize=1, cwd=str(ROOT))
    got = {"bytes": 0}
    slow = threading.Event()
    slow.set()

    def reader():
        with open(fifo, "rb") as f:
            while True:
                chunk = f.read(1 << 20)
                if not chunk:
                    return
                got["bytes"] += len(chunk)
                if slow.is_set():
                    time.sleep(0.02)                  # back-pressure: the CLI blocks in write() and cannot finish early

    t = threading.Thread(target=reader, daemon=True)
    t.start()
    lines = []
    for line in proc.stdout:                          # what TaskRunner._run_stage reads
        lines.append(line)
        if "time=" in line:
            break
    proc.terminate()                                  # TaskRunner.cancel(): terminate(), then kill() after a grace period
    slow.clear()
    rest, _ = proc.communicate(timeout=120)
    t.join(timeout=60)
    out = "".join(lines) + rest
    assert proc.returncode == 255, out[-2000:]
    assert "Duration: 00:02:40.00" in out and "received signal 15" in out
    assert 0 < got["bytes"] < fb * nframes and got["bytes"] % fb == 0


# ------------------------------------------------------------------ whole lattice in LDS (small N)
@pytest.mark.parametrize("n,precision,expect_whole", [(17, "strict", True), (17, "fast", True), (21, "strict", True),
                                                        (25, "fast", True), (25, "strict", False), (33, "fast", False)])
def test_small_lattices_are_staged_whole_and_content_stops_mattering(engine, orc, n, precision, expect_whole):
    """A lattice that fits one workgroup's LDS beside the coordinate table (N <= 21 strict, N <= 25 fast at 10 bit; .3dl files
    are always 17^3) is staged once per workgroup: no windows, no validity tests, so uniform-random frames -- the worst case
    of the window kernels -- take exactly the same path as natural ones."""
    lat = cube.log709_lattice(n)
    one = np.ones(3, np.float32)
    engine.set_lut(cube.CubeLut(n, one, lat))
    engine.set_variant("vec_lds")
    engine.set_precision(precision)
    try:
        k = orc.yuv_constants(din=10)
        for dist in ("uniform", "natural"):
            src = frames.make_yuv(dist, 256, 72, 10, 1, 1, k=51)
            for mode in ("tetrahedral", "trilinear"):
                engine.tile_stats(True)
                got = engine.apply_yuv(_to_dev(src, engine), pix_fmt="yuv420p10le", interp=mode)
                st = engine.tile_stats(False)
                whole = "whole-lattice" in engine.last_kernel
                if mode == "tetrahedral":
                    assert whole == expect_whole, (engine.last_kernel, n, precision)
                if whole:
                    assert st["global_tiles"] == 0 and st["misses"] == 0 and st["staged"] == 0
                want = orc.apply_yuv(lat, one, mode, k, 10, 10, 10, 1, 1, src, fast="fast" in engine.last_kernel)
                _assert_equal(_to_np(got, np.uint16), want, f"whole {n} {precision} {dist} {mode}")
        # raw codes above 2^10 - 1 in the 16-bit containers: the table does not cover them, the clamping body takes the tile
        src = frames.make_yuv("uniform", 256, 72, 10, 1, 1, k=52)
        src[0][5, 7] = 60000
        src[1][3, 3] = 2047
        got = engine.apply_yuv(_to_dev(src, engine), pix_fmt="yuv420p10le")
        want = orc.apply_yuv(lat, one, "tetrahedral", k, 10, 10, 10, 1, 1, src, fast="fast" in engine.last_kernel)
        _assert_equal(_to_np(got, np.uint16), want, f"whole {n} {precision} wild codes")
    finally:
        engine.set_precision("strict")
        engine.set_variant("auto")


# ------------------------------------------------------------------ decode | engine | encode (pipe.py) with stand-in codecs
def test_three_process_stage_streams_frames_through_the_engine(orc, cube_dir, tmp_path):
    """`pipe.run_stage` with `cat` as decoder and encoder (no ffmpeg on this image): frames enter the engine CLI on stdin, leave
    on stdout, the report (Duration: / time= lines) is relayed for TaskRunner._run_stage, and the bytes equal the oracle's."""
    import io
    from lut_renderer_amd.command import engine_command
    from lut_renderer_amd.params import ProcessingParams, VideoInfo
    from lut_renderer_amd.pipe import StageCommands, run_stage
    w, h, nf = 256, 72, 21
    lut = cube.read_cube(cube_dir / "log709_33.cube")
    fr = [frames.natural_yuv(w, h, 10, 1, 1, k=60 + i) for i in range(nf)]
    raw = tmp_path / "in.yuv"
    with open(raw, "wb") as f:
        for planes in fr:
            for p in planes:
                f.write(np.ascontiguousarray(p).tobytes())
    info = VideoInfo(width=w, height=h, bit_depth=10, pix_fmt="yuv420p10le", color_range="tv", colorspace="bt709", fps=25.0,
                     duration=nf / 25.0)
    params = ProcessingParams(video_codec="libx264")          # 10-bit source, 8-bit codec: format=yuv420p (App. D case K)
    out = tmp_path / "out.yuv"
    eng = engine_command(Path("-"), Path("-"), params, cube_dir / "log709_33.cube", info) + ["--duration", f"{nf / 25.0:.3f}", "--batch", "4"]
    cmds = StageCommands(["cat", str(raw)], eng, ["sh", "-c", f"cat > '{out}'"], [])
    log = io.StringIO()
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        rc = run_stage(cmds, out=log)
    finally:
        os.chdir(cwd)
    text = log.getvalue()
    assert rc == 0, text
    assert "Duration: 00:00:00.84" in text and "time=00:00:00.84" in text and "LUT: " in text
    k = orc.yuv_constants("bt709", "tv", "bt709", "tv", 10, 10, 8, 4)
    got = np.frombuffer(out.read_bytes(), dtype=np.uint8)
    fb8 = w * h * 3 // 2
    assert got.size == nf * fb8
    for i, planes in enumerate(fr):
        want = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 8, 1, 1, planes)
        assert np.array_equal(got[i * fb8:(i + 1) * fb8], np.concatenate([p.ravel() for p in want])), f"frame {i}"
    # a failing stage is reported and fails the whole stage
    bad = StageCommands(["cat", str(tmp_path / "missing.yuv")], eng, ["sh", "-c", "cat > /dev/null"], [])
    log2 = io.StringIO()
    os.chdir(ROOT)
    try:
        assert run_stage(bad, out=log2) != 0
    finally:
        os.chdir(cwd)
    assert "decoder exited" in log2.getvalue()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["strict", "fast"])
def test_grey_tube_serves_low_saturation_tiles_and_saturated_ones_fall_through(engine, orc, cube_dir, precision):
    """The workgroup-shared grey tube (csrc/lutr_tile2.hip tube_holds): near-neutral content of ANY luma is served without a
    window -- luma noise that would break every raw box does not matter --, saturated content goes through the per-wave
    windows as before, and both are bit-exact.  Frames: (a) grey ramp with heavy luma noise and +-6 codes of chroma,
    (b) the same luma with chroma pushed to +-300 codes (10 bit) in four flat quadrants."""
    lut = cube.read_cube(cube_dir / "log709_33.cube")
    engine.set_lut(lut)
    engine.set_variant("vec_lds")
    engine.set_precision(precision)
    rng = np.random.default_rng(77)
    w, h = 1024, 256
    y = np.clip(np.linspace(64, 940, w)[None, :] + rng.normal(0, 60, size=(h, w)), 64, 940).astype(np.uint16)
    grey = [y, (512 + rng.integers(-6, 7, size=(h // 2, w // 2))).astype(np.uint16),
            (512 + rng.integers(-6, 7, size=(h // 2, w // 2))).astype(np.uint16)]
    cb = np.full((h // 2, w // 2), 512, np.int32); cr = cb.copy()
    cb[:, : w // 4] += 300; cb[:, w // 4:] -= 300
    cr[: h // 4, :] += 300; cr[h // 4:, :] -= 300
    sat = [y, cb.astype(np.uint16), cr.astype(np.uint16)]
    k = orc.yuv_constants(din=10)
    try:
        for name, src, expect_tube in (("grey", grey, True), ("saturated", sat, False)):
            engine.tile_stats(True)
            got = engine.apply_yuv(_to_dev(src, engine), pix_fmt="yuv420p10le")
            st = engine.tile_stats(False)
            assert "tube" in engine.last_kernel, engine.last_kernel
            if expect_tube:
                assert st["tube_tiles"] == st["tiles"] and st["misses"] == 0 and st["global_tiles"] == 0, st
            else:
                assert st["tube_tiles"] == 0, st
            want = orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 10, 1, 1, src, fast=precision == "fast")
            _assert_equal(_to_np(got, np.uint16), want, f"tube {precision} {name}")
    finally:
        engine.set_precision("strict")
        engine.set_variant("auto")


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["strict", "fast"])
@pytest.mark.parametrize("n", [22, 27, 40, 41])
def test_tube_and_windows_across_lattice_sizes(engine, orc, n, precision):
    """Lattice sizes around the modes' limits (whole-lattice up to 21 / 25, tube up to 40, windows only above), three kinds of
    content, two interpolations, 10- and 8-bit 4:2:0: every combination bit-exact against the oracle (the tube's width and its
    chroma bound scale with the lattice)."""
    lat = cube.log709_lattice(n)
    one = np.ones(3, np.float32)
    engine.set_lut(cube.CubeLut(n, one, lat))
    engine.set_variant("vec_lds")
    engine.set_precision(precision)
    try:
        for fmt, depth in (("yuv420p10le", 10), ("yuv420p", 8)):
            k = orc.yuv_constants(din=depth, dl=depth, dout=depth)
            dt = np.uint16 if depth > 8 else np.uint8
            for dist in ("natural", "vivid", "uniform"):
                src = frames.make_yuv(dist, 512, 96, depth, 1, 1, k=n)
                for mode in ("tetrahedral", "trilinear"):
                    got = engine.apply_yuv(_to_dev(src, engine), pix_fmt=fmt, interp=mode)
                    name = engine.last_kernel
                    assert ("tube" in name) == (n <= 40 and "whole" not in name), (name, n)
                    want = orc.apply_yuv(lat, one, mode, k, depth, depth, depth, 1, 1, src, fast="fast" in name)
                    _assert_equal(_to_np(got, dt), want, f"n={n} {precision} {fmt} {dist} {mode} {name}")
    finally:
        engine.set_precision("strict")
        engine.set_variant("auto")


@pytest.mark.gpu
def test_csp_prelut_parity_on_every_path_that_takes_one(engine, orc, tmp_path):
    """f4: a cineSpace LUT with a shaper on all three channels (lut3d's prelut).  Planar RGB and fused YUV, generic and vector
    kernels, 8 and 10 bit, three interpolations: bit-exact against the oracle's apply_prelut + cube.  The tile kernels and the
    fast precision do not read a prelut -- the router must keep such a LUT away from them."""
    from tests.test_lut_formats import _csp_with_prelut
    rng = np.random.default_rng(21)
    n = 17
    tab = cube.log709_lattice(n)
    xs = [np.array([0.0, 0.05, 0.2, 0.5, 0.8, 1.0]), np.linspace(0.0, 1.0, 33), np.array([0.0, 0.3, 0.6, 1.0])]
    ys = [np.array([0.0, 0.2, 0.45, 0.7, 0.9, 1.0]), np.linspace(0.0, 1.0, 33) ** 0.6, np.array([0.0, 0.25, 0.7, 1.0])]
    p = tmp_path / "shaped.csp"
    _csp_with_prelut(p, n, tab, list(zip(xs, ys)))
    lut = cube.read_lut(p)
    n2, s2, t2, pre = orc.parse_lut_file_ex(p)
    assert lut.prelut is not None and np.array_equal(lut.prelut.table, pre.table)
    engine.set_lut(lut)
    try:
        for variant in ("auto", "generic", "vec_global", "vec_lds"):
            engine.set_variant(variant)
            for depth, fmt in ((10, "yuv420p10le"), (8, "yuv420p")):
                dt = np.uint16 if depth > 8 else np.uint8
                k = orc.yuv_constants(din=depth, dl=depth, dout=depth)
                src = frames.make_yuv("uniform", 256, 64, depth, 1, 1, k=3)
                for mode in ("tetrahedral", "trilinear", "nearest"):
                    got = engine.apply_yuv(_to_dev(src, engine), pix_fmt=fmt, interp=mode)
                    assert "tile" not in engine.last_kernel, engine.last_kernel
                    want = orc.apply_yuv(t2, s2, mode, k, depth, depth, depth, 1, 1, src, prelut=pre)
                    _assert_equal(_to_np(got, dt), want, f"prelut yuv {variant} {fmt} {mode} {engine.last_kernel}")
                rgb = frames.make_rgb("uniform", 256, 64, depth, k=4)
                got = engine.apply_rgb(_to_dev(rgb, engine), depth=depth, interp="tetrahedral")
                assert "tile" not in engine.last_kernel, engine.last_kernel
                want = orc.apply_rgb(t2, s2, depth, "tetrahedral", rgb, prelut=pre)
                _assert_equal(_to_np(got, dt), want, f"prelut rgb {variant} {depth}")
        # a big batch under auto routing would take the tile kernels: with a prelut it must not, and the fast precision falls back
        engine.set_variant("auto")
        engine.set_precision("fast")
        src = frames.make_yuv("natural", 1920, 1080, 10, 1, 1, k=5)
        dev = [t.unsqueeze(0).repeat(24, 1, 1) for t in _to_dev(src, engine)]
        got = engine.apply_yuv(dev, pix_fmt="yuv420p10le")
        assert "tile" not in engine.last_kernel and "fast" not in engine.last_kernel, engine.last_kernel
        k = orc.yuv_constants(din=10)
        want = orc.apply_yuv(t2, s2, "tetrahedral", k, 10, 10, 10, 1, 1, src, nthreads=8, prelut=pre)
        _assert_equal([g[7].cpu().numpy().view(np.uint16) for g in got], want, "prelut batch")
        # one process, two contexts (LutEngineGroup): the prelut is host-side state and must reach every context
        from lut_renderer_amd.multigpu import LutEngineGroup
        with LutEngineGroup([0, 0]) as grp:
            grp.set_lut(lut)
            got = grp.apply_yuv(_to_dev(src, engine), pix_fmt="yuv420p10le")
            _assert_equal(_to_np(got, np.uint16), want, "prelut group")
        # a new lattice drops the prelut
        engine.set_precision("strict")
        engine.set_lut(cube.CubeLut(n, np.ones(3, np.float32), tab))
        got = engine.apply_yuv(_to_dev(src, engine), pix_fmt="yuv420p10le")
        want = orc.apply_yuv(tab, np.ones(3, np.float32), "tetrahedral", k, 10, 10, 10, 1, 1, src, nthreads=8)
        _assert_equal(_to_np(got, np.uint16), want, "prelut dropped")
    finally:
        engine.set_precision("strict")
        engine.set_variant("auto")


@pytest.mark.gpu
def test_soak_tile_kernels_against_the_generic_kernel_on_the_validity_bounds():
    """tools/soak.py for a few seconds: chroma swept radially across the tube's bound, luma over the whole range, random lattices,
    domains, matrices, ranges, formats -- the tile kernels (tube, windows, gather) must equal the scalar kernel bit for bit."""
    res = subprocess.run([sys.executable, str(ROOT / "tools" / "soak.py"), "8"], capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert res.returncode == 0 and "soak ok" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]
    runs = int(res.stdout.split("soak ok:")[1].split("runs")[0])
    assert runs >= 20, res.stdout
