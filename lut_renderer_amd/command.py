"""argv twins of the reference's command builder, rendered from a LutPlan.

Call-compatible with
  build_command   /root/reference/src/lut_renderer/ffmpeg.py:179-414
  build_pipeline  /root/reference/src/lut_renderer/ffmpeg.py:436-487
  CommandStage    /root/reference/src/lut_renderer/ffmpeg.py:14-25
(same positional/keyword arguments, `notes` mutated in place, ValueError with the
reference's messages), so /root/reference/src/lut_renderer/task_manager.py:55,73-81 can use
them unchanged.  tests/test_dropin_argv.py pins the output against argv captured from the
reference itself (tests/golden/argv_cases.json).

The argv is assembled section by section (time structure, rate control, colour tags, ...)
from small helpers instead of one long function; the LUT part comes from plan.LutPlan.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional, Tuple

from .params import ProcessingParams, Task, VideoInfo
from .plan import LutPlan, output_color_tags, resolve_lut_plan, resolve_pix_fmt

_RATE = re.compile(r"^\s*(\d+(?:\.\d+)?)([kKmMgG]?)\s*$")


@dataclass
class CommandStage:
    name: str
    source_path: Path
    output_path: Path
    params: ProcessingParams
    lut_path: Optional[Path] = None
    cleanup_on_success: bool = False
    notes: List[str] = field(default_factory=list)
    probe_source: bool = False      # probe the stage input right before building its command


# ---------------------------------------------------------------- small value helpers
def _trim_float(value: float) -> str:
    return f"{value:.3f}".rstrip("0").rstrip(".")


def _fraction(text: str) -> Optional[float]:
    text = (text or "").strip()
    if not text:
        return None
    try:
        if "/" in text:
            num, den = text.split("/", 1)
            return float(num) / float(den) if float(den) != 0 else None
        return float(text)
    except ValueError:
        return None


def _rate(text: Optional[str]) -> Optional[Tuple[float, str]]:
    m = _RATE.match(text) if text else None
    if not m or float(m.group(1)) <= 0:
        return None
    return float(m.group(1)), m.group(2) or ""


def _double_rate(text: str) -> Optional[str]:
    parsed = _rate(text)
    if not parsed:
        return None
    number, unit = parsed[0] * 2, parsed[1]
    return f"{int(round(number))}{unit}" if abs(number - round(number)) < 1e-6 else f"{number:g}{unit}"


def _kbps(text: Optional[str]) -> Optional[float]:
    parsed = _rate(text)
    if not parsed:
        return None
    return {"k": parsed[0], "m": parsed[0] * 1e3, "g": parsed[0] * 1e6}.get(parsed[1].lower())


def _inherit_tags(cmd: List[str], info: Optional[VideoInfo], notes: List[str]) -> None:
    if not info:
        return
    shown = []
    for flag, label, value in (("-color_primaries", "primaries", info.color_primaries),
                               ("-color_trc", "trc", info.color_trc),
                               ("-colorspace", "colorspace", info.colorspace),
                               ("-color_range", "range", info.color_range)):
        if value:
            cmd += [flag, value]
            shown.append(f"{label}={value}")
    if shown:
        notes.append(f"继承色彩元数据: {', '.join(shown)}")


# ---------------------------------------------------------------- argv sections
def _time_structure(cmd, params, info, notes) -> Optional[float]:
    """-fps_mode / -r (ffmpeg.py:259-285); returns the fps used for the automatic GOP."""
    if params.fps:
        fps_value, fps_text = _fraction(params.fps), params.fps
    elif info and info.fps:
        fps_value, fps_text = info.fps, _trim_float(info.fps)
    else:
        fps_value, fps_text = None, None
    if params.fps:
        cmd += ["-fps_mode", "cfr", "-r", params.fps]
        notes.append(f"时间结构: fps_mode=cfr, 输出帧率={params.fps}")
        return fps_value
    vfr = bool(info and info.is_vfr)
    if vfr and params.force_cfr:
        cmd += ["-fps_mode", "cfr"]
        if fps_text:
            cmd += ["-r", fps_text]
            notes.append(f"时间结构: 源为 VFR，已强制 CFR，输出帧率={fps_text}")
        else:
            notes.append("时间结构: 源为 VFR，已强制 CFR（未检测到帧率）")
    elif params.force_cfr and info is None:
        cmd += ["-fps_mode", "cfr"]
        notes.append("时间结构: fps_mode=cfr（未读取源信息）")
    else:
        cmd += ["-fps_mode", "passthrough"]
        notes.append("时间结构: 源为 VFR，fps_mode=passthrough（不重写时间戳）" if vfr
                     else "时间结构: 源为 CFR/未知，fps_mode=passthrough（避免时间戳重写）")
    return fps_value


def _rate_control(cmd, params, fps_value, notes) -> None:
    if params.resolution:
        cmd += ["-s", params.resolution]
    if params.bitrate:
        cmd += ["-b:v", params.bitrate]
        bufsize = _double_rate(params.bitrate)
        if bufsize:
            cmd += ["-maxrate", params.bitrate, "-bufsize", bufsize]
            notes.append(f"码率稳定: maxrate={params.bitrate}, bufsize={bufsize}")
    for flag, value in (("-crf", params.crf), ("-preset", params.preset), ("-tune", params.tune)):
        if value:
            cmd += [flag, value]
    if params.gop:
        cmd += ["-g", params.gop]
    elif fps_value:
        gop = max(1, round(fps_value))
        cmd += ["-g", str(gop)]
        notes.append(f"自动 GOP={gop} (fps={_trim_float(fps_value)})")
    for flag, value in (("-profile:v", params.profile), ("-level", params.level), ("-threads", params.threads)):
        if value:
            cmd += [flag, value]


def _colour_tags(cmd, params, info, plan: Optional[LutPlan], notes) -> None:
    """ffmpeg.py:348-386"""
    if plan is None:
        if params.inherit_color_metadata:
            _inherit_tags(cmd, info, notes)
        return
    tags = output_color_tags(plan.output_policy)
    if tags is not None:
        for key in ("color_primaries", "color_trc", "colorspace", "color_range"):
            cmd += [f"-{key}", tags[key]]
        notes.append("LUT 输出标记: bt709/bt709/bt709, range=tv" +
                     ("" if plan.output_policy == "bt709" else "（回退）"))
    elif plan.output_policy == "inherit":
        if params.inherit_color_metadata:
            _inherit_tags(cmd, info, notes)
    else:
        notes.append("LUT 输出标记: none（不写色彩元数据）")


# ---------------------------------------------------------------- public twins
def build_command(source: Path, output: Path, params: ProcessingParams, lut_path: Optional[Path] = None,
                  ffmpeg_bin: str = "ffmpeg", source_info: Optional[VideoInfo] = None,
                  notes: Optional[List[str]] = None) -> List[str]:
    notes = notes if notes is not None else []
    cmd = [ffmpeg_bin, "-hide_banner"]
    if params.overwrite:
        cmd.append("-y")
    cmd += ["-i", str(source)]

    plan = resolve_lut_plan(params, lut_path, source_info) if lut_path else None
    filters: List[str] = []
    if plan is not None:
        filters += plan.filters()
        notes.extend(plan.notes)

    if params.video_codec:
        cmd += ["-c:v", params.video_codec]
    if params.audio_codec:
        cmd += ["-c:a", params.audio_codec]
    if filters and params.video_codec == "copy":
        raise ValueError("启用 LUT/滤镜时不能使用视频 copy（streamcopy 与滤镜不可同时使用）。")

    if params.video_codec and params.video_codec != "copy":
        fps_value = _time_structure(cmd, params, source_info, notes)
        pix_fmt = resolve_pix_fmt(params, source_info, notes)
        if pix_fmt:
            if getattr(params, "zscale_dither", "none") == "error_diffusion":
                filters.append("zscale=dither=error_diffusion")
                notes.append("抖动: zscale=dither=error_diffusion")
            if plan is not None:
                filters.append(f"format={pix_fmt}")
            cmd += ["-pix_fmt", pix_fmt]
        _rate_control(cmd, params, fps_value, notes)
        _colour_tags(cmd, params, source_info, plan, notes)
        if "videotoolbox" in params.video_codec:
            kbps = _kbps(params.bitrate or (source_info.bitrate if source_info else ""))
            if kbps and kbps >= 50_000:
                notes.append("提示: h264_videotoolbox 在高码率/重负载时可能出现 PTS 重建/帧重排的节奏错觉；"
                             "如需更稳定建议用 libx264 或切到“专业母带”。")

    if filters:
        cmd += ["-vf", ",".join(filters)]
    if params.audio_codec and params.audio_codec != "copy":
        for flag, value in (("-b:a", params.audio_bitrate), ("-ar", params.sample_rate), ("-ac", params.channels)):
            if value:
                cmd += [flag, value]
    if params.faststart:
        cmd += ["-movflags", "+faststart"]
    cmd.append(str(output))
    return cmd


def _master_params(params: ProcessingParams) -> ProcessingParams:
    """ProRes 422 HQ mezzanine settings of the two-stage mode (ffmpeg.py:417-433)."""
    master = ProcessingParams.from_dict(params.to_dict())
    master.video_codec, master.audio_codec = "prores_ks", "copy"
    master.pix_fmt, master.profile = "yuv422p10le", "3"
    for name in ("level", "crf", "preset", "tune", "bitrate", "audio_bitrate", "sample_rate", "channels"):
        setattr(master, name, "")
    master.faststart = False
    master.bit_depth_policy = "preserve"
    return master


def build_pipeline(task: Task, ffmpeg_bin: str = "ffmpeg") -> List[CommandStage]:
    """One stage with the LUT ("fast"), or ProRes master WITH the LUT followed by a
    distribution encode WITHOUT it ("pro"): the LUT is applied exactly once per task."""
    params = task.params
    if params.processing_mode != "pro":
        return [CommandStage("快速交付", task.source_path, task.output_path, params, lut_path=task.lut_path)]
    if not task.intermediate_path:
        raise ValueError("专业母带模式需要显式设置中间文件路径（请在界面中设置母带缓存目录）。")
    return [
        CommandStage("ProRes 母带", task.source_path, task.intermediate_path, _master_params(params),
                     lut_path=task.lut_path, cleanup_on_success=True,
                     notes=["母带固定为 ProRes 422 HQ (yuv422p10le)"]),
        CommandStage("分发编码", task.intermediate_path, task.output_path, params, lut_path=None,
                     probe_source=True),
    ]


# ---------------------------------------------------------------- engine twin
def engine_command(source: Path, output: Path, params: ProcessingParams, lut_path: Path,
                   source_info: VideoInfo, python_bin: Optional[str] = None, device: int = 0,
                   notes: Optional[List[str]] = None, precision: str = "strict") -> List[str]:
    """`build_command`'s twin for the LUT stage alone: the argv of the ENGINE CLI (`python -m lut_renderer_amd.cli`)
    that applies exactly the chain `build_command` would put into `-vf` -- the same `LutPlan`, rendered as CLI options
    instead of as a filter string (ffmpeg.py:195-247, :287-310).  `source` / `output` are rawvideo files (or `-`) in
    `source_info.pix_fmt` and the pixel format `resolve_pix_fmt` picks; `task_manager.py:145-151` can Popen the result
    unchanged (same `Duration:` / `time=` / exit-code / SIGTERM contract).  `notes` receives the plan's notes, like
    `build_command`'s out-parameter.  The copy guard of ffmpeg.py:255-256 applies: a LUT stage cannot be a stream copy.
    `precision` is the engine's own setting (`--precision`, default strict; the reference's records have no such field)."""
    import sys as _sys
    if precision not in ("strict", "fast"):
        raise ValueError(f"unknown precision '{precision}' (strict | fast)")
    if lut_path is None:
        raise ValueError("engine_command renders the LUT stage: lut_path is required")
    if source_info is None or not source_info.pix_fmt or not source_info.width or not source_info.height:
        raise ValueError("engine_command needs source_info with pix_fmt, width and height (raw frames carry no header)")
    if params.video_codec == "copy":
        raise ValueError("启用 LUT/滤镜时不能使用视频 copy（streamcopy 与滤镜不可同时使用）。")
    notes = notes if notes is not None else []
    plan = resolve_lut_plan(params, lut_path, source_info)
    notes.extend(plan.notes)
    cmd = [python_bin or _sys.executable, "-m", "lut_renderer_amd.cli"]
    if params.overwrite:
        cmd.append("-y")
    cmd += ["-i", str(source), "-o", str(output), "--size", f"{source_info.width}x{source_info.height}",
            "--pix-fmt", source_info.pix_fmt]
    pix_fmt = resolve_pix_fmt(params, source_info, notes) if params.video_codec else ""
    if pix_fmt:
        cmd += ["--out-pix-fmt", pix_fmt]
    cmd += ["--cube", str(plan.lut_path), "--interp", plan.interp, "--input-matrix", plan.matrix_policy,
            "--output-tags", plan.output_policy]
    if getattr(params, "zscale_dither", "none") == "error_diffusion" and pix_fmt:
        cmd += ["--zscale-dither", "error_diffusion"]
        notes.append("抖动: zscale=dither=error_diffusion")
    if source_info.colorspace:
        cmd += ["--colorspace", str(source_info.colorspace)]
    if source_info.color_range:
        cmd += ["--color-range", str(source_info.color_range)]
    if source_info.fps:
        cmd += ["--fps", fps_rational(source_info.fps)]
    if device:
        cmd += ["--device", str(int(device))]
    if precision != "strict":
        cmd += ["--precision", precision]
    return cmd


def fps_rational(fps) -> str:
    """A frame rate as the rational ffprobe reported (`r_frame_rate`): `VideoInfo.fps` is that fraction parsed to a float
    (media_info.py:135-137), and printing it with six digits (29.97) makes timestamps drift against the source.  NTSC rates
    come back as N000/1001, everything else as the closest fraction with a denominator up to 1001."""
    from fractions import Fraction
    value = float(fps)
    for den in (1, 1001):
        num = round(value * den)
        if num > 0 and abs(num / den - value) <= 1e-6 * value:
            return str(num) if den == 1 else f"{num}/{den}"
    frac = Fraction(value).limit_denominator(1001)
    return str(frac.numerator) if frac.denominator == 1 else f"{frac.numerator}/{frac.denominator}"
