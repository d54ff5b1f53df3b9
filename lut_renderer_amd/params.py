"""Option and task records of the drop-in layer.

Same names, defaults and dict round-trip as the reference's records so that its GUI and
controller can hand theirs over unchanged:
  ProcessingParams  /root/reference/src/lut_renderer/models.py:19-122
  Task, TaskStatus  /root/reference/src/lut_renderer/models.py:11-17, :125-143
  VideoInfo         /root/reference/src/lut_renderer/media_info.py:12-52
The LUT path only reads `lut_interp`, `zscale_dither`, `lut_input_matrix`, `lut_output_tags`,
`bit_depth_policy`, `video_codec`, `pix_fmt` (models.py:45-56) and, from VideoInfo, `pix_fmt`,
`bit_depth`, `colorspace`, `color_range` (media_info.py:25-34).
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from enum import Enum
from pathlib import Path
from typing import Any, Dict, Optional


@dataclass
class VideoInfo:
    width: Optional[int] = None
    height: Optional[int] = None
    sar: Optional[str] = None
    dar: Optional[str] = None
    bitrate: Optional[str] = None
    container_bitrate: Optional[str] = None
    fps: Optional[float] = None
    avg_fps: Optional[float] = None
    r_fps: Optional[float] = None
    is_vfr: bool = False
    duration: Optional[float] = None
    pix_fmt: Optional[str] = None
    bit_depth: Optional[int] = None
    codec_name: Optional[str] = None
    codec_long_name: Optional[str] = None
    profile: Optional[str] = None
    level: Optional[str] = None
    color_primaries: Optional[str] = None
    color_trc: Optional[str] = None
    colorspace: Optional[str] = None
    color_range: Optional[str] = None
    format_name: Optional[str] = None
    format_long_name: Optional[str] = None
    file_size: Optional[int] = None
    audio_codec: Optional[str] = None
    audio_codec_long_name: Optional[str] = None
    audio_channels: Optional[int] = None
    audio_channel_layout: Optional[str] = None
    audio_sample_rate: Optional[int] = None
    audio_bitrate: Optional[str] = None
    format_tags: Optional[dict] = None
    video_tags: Optional[dict] = None
    audio_tags: Optional[dict] = None

    @property
    def resolution(self) -> Optional[str]:
        return f"{self.width}x{self.height}" if self.width and self.height else None


def infer_bit_depth(pix_fmt: Optional[str], bits_per_raw_sample: Optional[str] = None) -> Optional[int]:
    """media_info.py:86-110: bits_per_raw_sample wins; else the digits after the 'p' of the
    pixel-format name (yuv420p10le -> 10); plain names (yuv420p) give None."""
    if bits_per_raw_sample:
        try:
            bits = int(float(bits_per_raw_sample))
            if bits > 0:
                return bits
        except ValueError:
            pass
    for token in (pix_fmt or "").split(":"):
        head, sep, tail = token.partition("p")
        if not sep:
            continue
        digits = ""
        for ch in tail:
            if not ch.isdigit():
                break
            digits += ch
        if digits:
            return int(digits)
    return None


@dataclass
class ProcessingParams:
    video_codec: str = "libx264"
    audio_codec: str = "aac"
    pix_fmt: str = ""              # empty: let the bit-depth policy / encoder decide
    resolution: str = ""
    bitrate: str = ""
    fps: str = ""
    crf: str = ""
    preset: str = ""
    tune: str = ""
    gop: str = ""
    profile: str = ""
    level: str = ""
    threads: str = ""
    audio_bitrate: str = ""
    sample_rate: str = ""
    channels: str = ""
    faststart: bool = False
    overwrite: bool = True
    generate_cover: bool = False
    processing_mode: str = "fast"
    bit_depth_policy: str = "preserve"
    force_cfr: bool = True
    inherit_color_metadata: bool = True
    lut_interp: str = "tetrahedral"
    zscale_dither: str = "none"
    lut_input_matrix: str = "auto"     # auto | bt709 | none | <matrix name>
    lut_output_tags: str = "bt709"     # bt709 | inherit | none

    def to_dict(self) -> Dict[str, Any]:
        return dataclasses.asdict(self)

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "ProcessingParams":
        out = cls()
        for f in dataclasses.fields(cls):
            if f.name in data:
                value = data[f.name]
                setattr(out, f.name, bool(value) if f.type in ("bool", bool) else value)
        return out


class TaskStatus(str, Enum):
    PENDING = "pending"
    RUNNING = "running"
    COMPLETED = "completed"
    FAILED = "failed"
    CANCELED = "canceled"


@dataclass
class Task:
    task_id: str
    source_path: Path
    output_path: Path
    lut_path: Optional[Path]
    cover_path: Optional[Path]
    params: ProcessingParams
    source_info: Optional[VideoInfo] = None
    intermediate_path: Optional[Path] = None
    status: TaskStatus = TaskStatus.PENDING
    progress: int = 0
    error: str = ""
    started_at: Optional[float] = None
    finished_at: Optional[float] = None
    metadata: dict = field(default_factory=dict)

    def display_name(self) -> str:
        return self.source_path.name
