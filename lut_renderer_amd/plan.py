"""What the LUT stage of a task does, decided once and rendered twice.

The reference decides and prints in one pass while it assembles the ffmpeg argv
(/root/reference/src/lut_renderer/ffmpeg.py:195-247, :287-310, :348-383).  Here the same
decisions are taken by `resolve_lut_plan()` into a `LutPlan` record, which is then rendered
either as ffmpeg filter strings (`command.build_command`, for argv parity with the reference)
or as a `lutr_apply_yuv` call (`api.apply_lut`, the MI355X engine).  Notes are the same strings
the reference appends to its out-parameter list (ffmpeg.py:226-247), in the same order.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional

from .params import ProcessingParams, VideoInfo

#: lut3d modes the reference lets through (ffmpeg.py:243); anything else becomes tetrahedral.
#: "cubic" passes the whitelist although FFmpeg's lut3d has no such mode.
INTERP_WHITELIST = ("nearest", "trilinear", "tetrahedral", "pyramid", "prism", "cubic")
#: matrices `scale=in_color_matrix=` can be given (ffmpeg.py:119-125)
SCALE_MATRICES = ("bt709", "smpte170m", "bt470bg", "bt2020nc", "bt2020c")
#: codecs the reference trusts with 10-bit output (ffmpeg.py:109-110)
TEN_BIT_CODECS = ("prores_ks", "libx265", "hevc_videotoolbox")


def normalize_matrix(value: Optional[str]) -> Optional[str]:
    """ffmpeg.py:113-126: lower-cased name if it is one `scale` understands, else None."""
    text = str(value).strip().lower() if value else ""
    return text if text in SCALE_MATRICES else None


def is_full_range(info: Optional[VideoInfo]) -> bool:
    """ffmpeg.py:129-134: yuvj* pixel formats or color_range == pc."""
    if info is None:
        return False
    if info.pix_fmt and str(info.pix_fmt).startswith("yuvj"):
        return True
    return bool(info.color_range) and str(info.color_range).lower() == "pc"


def eight_bit_intermediate(info: Optional[VideoInfo]) -> str:
    """ffmpeg.py:137-143: the 8-bit format the full-range prologue converts to (always 8 bit,
    even for 10-bit sources -- a quirk the drop-in keeps)."""
    name = str(info.pix_fmt) if info is not None and info.pix_fmt else ""
    for tag in ("444", "422"):
        if tag in name:
            return f"yuv{tag}p"
    return "yuv420p"


def escape_filter_path(path) -> str:
    """ffmpeg.py:28-35: backslashes first, then single quotes."""
    return str(path).replace("\\", "\\\\").replace("'", "\\'")


@dataclass
class LutPlan:
    lut_path: Path
    interp: str                        # what goes after interp= (may be "cubic", see above)
    matrix: Optional[str]              # forced YUV<->RGB matrix, None = not forced
    matrix_policy: str
    output_policy: str                 # bt709 | inherit | none | <other -> bt709 fallback>
    prologue: bool                     # full-range source: scale=in_range=pc:out_range=..,format=<8 bit>
    prologue_out_range: Optional[str]  # tv | pc
    intermediate_pix_fmt: Optional[str]
    notes: List[str] = field(default_factory=list)

    def filters(self) -> List[str]:
        """The -vf fragments ahead of any zscale/format=<pix_fmt> (ffmpeg.py:211-246)."""
        out: List[str] = []
        if self.prologue:
            parts = ["in_range=pc", f"out_range={self.prologue_out_range}"]
            if self.matrix:
                parts += [f"in_color_matrix={self.matrix}", f"out_color_matrix={self.matrix}"]
            out.append("scale=" + ":".join(parts))
            out.append(f"format={self.intermediate_pix_fmt}")
        elif self.matrix:
            out.append(f"scale=in_color_matrix={self.matrix}:out_color_matrix={self.matrix}")
        out.append(f"lut3d=file='{escape_filter_path(self.lut_path)}':interp={self.interp}")
        return out


def resolve_lut_plan(params: ProcessingParams, lut_path, source_info: Optional[VideoInfo] = None) -> LutPlan:
    output_policy = (getattr(params, "lut_output_tags", "") or "bt709").strip().lower()
    matrix_policy = (getattr(params, "lut_input_matrix", "") or "auto").strip().lower()
    if matrix_policy == "bt709":
        matrix = "bt709"
    elif matrix_policy == "auto":
        matrix = normalize_matrix(source_info.colorspace if source_info else None)
    elif matrix_policy == "none":
        matrix = None
    else:
        matrix = normalize_matrix(matrix_policy)

    notes: List[str] = []
    prologue = is_full_range(source_info)
    out_range = intermediate = None
    if prologue:
        if output_policy == "bt709":
            out_range = "tv"
        elif output_policy == "inherit":
            src = source_info.color_range if source_info else None
            out_range = str(src).lower().strip() if src else "pc"
        else:
            out_range = "pc"
        intermediate = eight_bit_intermediate(source_info)
        notes.append(f"Range: 检测到 full-range(pc)，已按 out_range={out_range} 规范化，"
                     f"避免 yuvj* 旧像素格式（format={intermediate}）")
        if matrix:
            notes.append(f"LUT 输入矩阵: {matrix}（{matrix_policy}）")
    elif matrix:
        notes.append(f"LUT 输入矩阵: {matrix}（{matrix_policy}）")
    else:
        notes.append("LUT 输入矩阵: 未强制（auto/none 或无法识别源 colorspace）")

    interp = params.lut_interp or "tetrahedral"
    if interp not in INTERP_WHITELIST:
        interp = "tetrahedral"
    notes.append(f"LUT: 使用 lut3d（interp={interp}）")
    return LutPlan(Path(lut_path), interp, matrix, matrix_policy, output_policy, prologue, out_range,
                   intermediate, notes)


def resolve_pix_fmt(params: ProcessingParams, source_info: Optional[VideoInfo], notes: List[str]) -> str:
    """Output pixel format (ffmpeg.py:287-302); '' = leave it to the encoder."""
    pix_fmt = params.pix_fmt
    if params.bit_depth_policy == "force_8bit":
        if pix_fmt != "yuv420p":
            notes.append("位深策略=强制8bit: pix_fmt=yuv420p")
        return "yuv420p"
    if params.bit_depth_policy in ("preserve", "auto") and not pix_fmt:
        if source_info and source_info.bit_depth and source_info.bit_depth >= 10:
            if params.video_codec in TEN_BIT_CODECS:
                pix_fmt = "yuv422p10le" if params.video_codec == "prores_ks" else "yuv420p10le"
                notes.append(f"位深策略=保持10bit: pix_fmt={pix_fmt}")
            else:
                pix_fmt = "yuv420p"
                notes.append("位深策略=保持10bit: 编码器不支持10bit，回退 yuv420p")
    return pix_fmt


def output_color_tags(plan_policy: str) -> Optional[dict]:
    """Tags written after a LUT (ffmpeg.py:348-383): Rec.709/tv for 'bt709' and for unknown
    policies (fallback); None for 'inherit' / 'none' (handled by the caller)."""
    if plan_policy in ("inherit", "none"):
        return None
    return {"color_primaries": "bt709", "color_trc": "bt709", "colorspace": "bt709", "color_range": "tv"}
