"""Host-side additions of round 2 (no GPU): plane validation ahead of the C-ABI, the engine twin of build_command,
the noise-swept frame generator."""
from pathlib import Path

import numpy as np
import pytest
import torch

from lut_renderer_amd import frames
from lut_renderer_amd.command import build_command, engine_command
from lut_renderer_amd.engine import _check_planes, parse_pix_fmt
from lut_renderer_amd.params import ProcessingParams, VideoInfo


def _planes(fmt, w, h, dtype=None, nframes=None):
    pf = parse_pix_fmt(fmt)
    dt = dtype or (torch.uint8 if pf.depth <= 8 else torch.int16)
    lead = () if nframes is None else (nframes,)
    return pf, [torch.zeros(lead + pf.plane_shape(i, w, h), dtype=dt) for i in range(3)]


def test_plane_validation_accepts_what_the_format_implies():
    for fmt in ("yuv420p", "yuv420p10le", "yuv422p10le", "yuv444p12le", "gbrp10le"):
        pf, planes = _planes(fmt, 64, 36)
        _check_planes(planes, pf, 64, 36, "source")
        pf, planes = _planes(fmt, 65, 37, nframes=3)              # odd sizes: chroma planes round up
        _check_planes(planes, pf, 65, 37, "source")


def test_plane_validation_rejects_what_would_run_off_the_buffers():
    """The C-ABI takes bare pointers (include/lutr.h): a wrong element size or a short chroma plane would make the
    kernels read or write outside the tensors, so the Python layer refuses them before any call into liblutr."""
    pf, planes = _planes("yuv420p10le", 64, 36, dtype=torch.uint8)       # 8-bit tensors named as a 10-bit format
    with pytest.raises(ValueError, match="16-bit"):
        _check_planes(planes, pf, 64, 36, "source")
    pf, planes = _planes("yuv420p", 64, 36, dtype=torch.int16)
    with pytest.raises(ValueError, match="8-bit"):
        _check_planes(planes, pf, 64, 36, "destination")
    pf, planes = _planes("yuv420p10le", 64, 36)
    planes[1] = planes[1][:-1]                                           # chroma plane one row short
    with pytest.raises(ValueError, match="plane 1"):
        _check_planes(planes, pf, 64, 36, "source")
    pf, planes = _planes("yuv422p", 64, 36)
    planes[2] = torch.zeros((36, 64), dtype=torch.uint8)                 # 4:4:4 sized plane in a 4:2:2 frame
    with pytest.raises(ValueError, match="plane 2"):
        _check_planes(planes, pf, 64, 36, "source")
    pf, planes = _planes("yuv420p", 64, 36)
    with pytest.raises(ValueError):
        _check_planes(planes[:2], pf, 64, 36, "source")
    with pytest.raises(ValueError):
        _check_planes([p.float() for p in planes], pf, 64, 36, "source")
    with pytest.raises(TypeError):
        _check_planes([p.numpy() for p in planes], pf, 64, 36, "source")


PC = VideoInfo(width=1920, height=1080, bit_depth=8, pix_fmt="yuvj420p", color_range="pc", colorspace="bt709", fps=25.0)
PC10 = VideoInfo(width=1920, height=1080, bit_depth=10, pix_fmt="yuv422p10le", color_range="pc", colorspace="bt2020nc")
TV10 = VideoInfo(width=3840, height=2160, bit_depth=10, pix_fmt="yuv420p10le", color_range="tv", colorspace="bt2020nc", fps=24.0)


@pytest.mark.parametrize("info,kw", [
    (PC, dict(video_codec="libx265")),
    (PC, dict(video_codec="libx265", lut_output_tags="inherit")),
    (PC10, dict(video_codec="libx265", lut_input_matrix="none")),
    (TV10, dict(video_codec="libx265")),
    (TV10, dict(video_codec="libx265", lut_input_matrix="bt709", lut_interp="trilinear")),
    (TV10, dict(video_codec="libx264")),                                                   # App. D case K: 10 -> 8 bit
    (TV10, dict(video_codec="libx264", bit_depth_policy="force_8bit", zscale_dither="error_diffusion")),
    (TV10, dict(video_codec="prores_ks")),
    (TV10, dict(video_codec="libx265", lut_interp="bogus")),
])
def test_engine_command_renders_the_same_plan_as_build_command(info, kw):
    """The engine CLI's argv, parsed back by the CLI's own parser, resolves to the filter chain build_command puts into
    -vf (ffmpeg.py:195-247) and to the pixel format it passes as -pix_fmt (ffmpeg.py:287-310)."""
    from lut_renderer_amd import cli
    params = ProcessingParams(**kw)
    notes_f, notes_e = [], []
    ff = build_command(Path("in.mov"), Path("out.mp4"), params, lut_path=Path("l u't/look.cube"), source_info=info,
                       notes=notes_f)
    cmd = engine_command(Path("in.yuv"), Path("out.yuv"), params, Path("l u't/look.cube"), info, python_bin="python3",
                         notes=notes_e)
    assert cmd[:3] == ["python3", "-m", "lut_renderer_amd.cli"] and cmd[0:1] == ["python3"]
    args = cli.build_parser().parse_args(cmd[3:])
    plan, call, w, h = cli.plan_from_args(args)
    vf = ff[ff.index("-vf") + 1].split(",")
    n = next(i for i, f in enumerate(vf) if f.startswith("lut3d=")) + 1
    assert plan.filters() == vf[:n]
    assert (w, h) == (info.width, info.height)
    if "-pix_fmt" in ff:
        assert call["out_pix_fmt"] == ff[ff.index("-pix_fmt") + 1]
    assert ("zscale=dither=error_diffusion" in vf) == (call.get("dither") == "error_diffusion")
    keep = lambda ns: [x for x in ns if x.startswith(("LUT:", "LUT 输入", "Range"))]      # the plan's own notes
    assert keep(notes_f) == keep(notes_e) and keep(notes_e)


def test_engine_command_keeps_the_copy_guard_and_needs_geometry():
    with pytest.raises(ValueError, match="copy"):
        engine_command(Path("a"), Path("b"), ProcessingParams(video_codec="copy"), Path("x.cube"), TV10)
    with pytest.raises(ValueError):
        engine_command(Path("a"), Path("b"), ProcessingParams(), Path("x.cube"), VideoInfo(pix_fmt="yuv420p"))
    with pytest.raises(ValueError):
        engine_command(Path("a"), Path("b"), ProcessingParams(), None, TV10)


def test_noise_sweep_distributions():
    a = frames.make_yuv("noise2", 128, 64, 10, 1, 1, k=0)
    b = frames.make_yuv("natural", 128, 64, 10, 1, 1, k=0)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))                # sigma 2 IS the natural frame
    c = frames.make_yuv("noise64", 128, 64, 10, 1, 1, k=0)
    assert np.abs(c[0].astype(int) - b[0].astype(int)).std() > 20
    assert c[1].min() >= 64 and c[1].max() <= 960 and c[0].max() <= 940
    v = frames.make_yuv("vivid", 128, 64, 10, 1, 1, k=0)                  # the natural frame with three times the chroma
    assert np.array_equal(v[0], b[0])
    assert np.abs(v[1].astype(int) - 512).mean() > 2.0 * np.abs(b[1].astype(int) - 512).mean()
    g = frames.make_rgb("noise16", 64, 32, 8, k=1)
    assert g[0].dtype == np.uint8 and g[0].shape == (32, 64)
    for bad in ("noise", "noisex", "noise-4", "plasma"):
        with pytest.raises(ValueError):
            frames.make_yuv(bad, 16, 16, 8, 1, 1)


def test_engine_stage_commands_split_build_command_around_the_engine():
    """decode | engine | encode (pipe.py): the decoder carries the source, the engine the LutPlan, the encoder everything
    build_command emits except the -vf chain, with the colour tags of the LUT policy (ffmpeg.py:348-383)."""
    from lut_renderer_amd.pipe import engine_stage_commands
    params = ProcessingParams(video_codec="libx264", audio_codec="aac", crf="18", preset="slow")
    info = VideoInfo(width=3840, height=2160, bit_depth=10, pix_fmt="yuv420p10le", color_range="tv", colorspace="bt2020nc",
                     fps=24.0, duration=12.5)
    c = engine_stage_commands(Path("in.mov"), Path("out.mp4"), params, Path("look.cube"), info, python_bin="python3")
    full = build_command(Path("in.mov"), Path("out.mp4"), params, lut_path=Path("look.cube"), source_info=info)
    assert c.decoder[0] == "ffmpeg" and c.decoder[c.decoder.index("-i") + 1] == "in.mov"
    assert c.decoder[c.decoder.index("-pix_fmt") + 1] == "yuv420p10le" and c.decoder[-1] == "pipe:1" and "-vf" not in c.decoder
    assert c.engine[:3] == ["python3", "-m", "lut_renderer_amd.cli"] and c.engine[c.engine.index("-i") + 1] == "-"
    assert c.engine[c.engine.index("--out-pix-fmt") + 1] == "yuv420p"          # libx264 cannot do 10 bit: App. D case K
    assert c.engine[c.engine.index("--duration") + 1] == "12.500"
    e = c.encoder
    assert "-vf" not in e and e[-1] == "out.mp4" and e[e.index("-i") + 1] == "pipe:0"
    assert e[e.index("-f") + 1] == "rawvideo" and e[e.index("-s") + 1] == "3840x2160" and e.index("-f") < e.index("-i")
    assert e[e.index("-pix_fmt") + 1] == "yuv420p"                              # raw input format = what the engine writes
    for flag in ("-c:v", "-c:a", "-crf", "-preset", "-color_primaries", "-color_trc", "-colorspace", "-color_range"):
        assert flag in e and e[e.index(flag) + 1] == full[full.index(flag) + 1], flag
    with pytest.raises(ValueError, match="copy"):
        engine_stage_commands(Path("a"), Path("b"), ProcessingParams(video_codec="copy"), Path("x.cube"), info)


def test_encoder_of_the_engine_stage_carries_audio_subtitles_metadata_and_the_rational_rate():
    """the reference's one ffmpeg process carries the source's audio / subtitles / metadata into the output
    (ffmpeg.py:385-414) -- a raw pipe has none of them, so the encoder reads the source as a second input; and the frame rate
    travels as ffprobe's rational, not as a six-digit float (30000/1001 vs 29.97: timestamps drift)."""
    from lut_renderer_amd.command import fps_rational
    from lut_renderer_amd.pipe import engine_stage_commands
    params = ProcessingParams(video_codec="libx265", audio_codec="copy")
    info = VideoInfo(width=1920, height=1080, bit_depth=10, pix_fmt="yuv420p10le", color_range="tv", colorspace="bt709",
                     fps=30000 / 1001, duration=2.0)
    c = engine_stage_commands(Path("clip.mov"), Path("out.mp4"), params, Path("look.cube"), info, python_bin="python3",
                              precision="fast")
    e = c.encoder
    ins = [i for i, t in enumerate(e) if t == "-i"]
    assert len(ins) == 2 and e[ins[0] + 1] == "pipe:0" and e[ins[1] + 1] == "clip.mov"
    assert e[e.index("-r") + 1] == "30000/1001" and e.index("-r") < ins[0]            # an INPUT option of the raw pipe
    maps = [e[i + 1] for i, t in enumerate(e) if t == "-map"]
    assert maps == ["0:v:0", "1:a?", "1:s?"]
    assert e[e.index("-map_metadata") + 1] == "1" and e[e.index("-map_chapters") + 1] == "1"
    assert e[e.index("-c:a") + 1] == "copy" and e[-1] == "out.mp4"
    assert c.engine[c.engine.index("--fps") + 1] == "30000/1001"
    assert c.engine[c.engine.index("--precision") + 1] == "fast"
    for value, text in ((25.0, "25"), (24000 / 1001, "24000/1001"), (59.94005994, "60000/1001"), (12.5, "25/2"), (50, "50")):
        assert fps_rational(value) == text


def test_engine_command_precision_is_an_engine_option_with_a_strict_default():
    """everything the reference's caller can reach must be able to select FAST; default strict."""
    from lut_renderer_amd import cli
    info = VideoInfo(width=64, height=32, bit_depth=10, pix_fmt="yuv420p10le", color_range="tv", colorspace="bt709", fps=25.0)
    base = engine_command(Path("a.yuv"), Path("b.yuv"), ProcessingParams(), Path("x.cube"), info, python_bin="py")
    assert "--precision" not in base
    fast = engine_command(Path("a.yuv"), Path("b.yuv"), ProcessingParams(), Path("x.cube"), info, python_bin="py", precision="fast")
    assert fast[fast.index("--precision") + 1] == "fast" and [t for t in fast if t not in ("--precision", "fast")] == base
    with pytest.raises(ValueError):
        engine_command(Path("a.yuv"), Path("b.yuv"), ProcessingParams(), Path("x.cube"), info, precision="bf16")
    args = cli.build_parser().parse_args(fast[3:])
    assert args.precision == "fast" and args.fps == 25.0
    assert cli.build_parser().parse_args(base[3:]).precision == "strict"
    assert abs(cli.build_parser().parse_args(base[3:] + ["--fps", "30000/1001"]).fps - 29.97002997) < 1e-6
