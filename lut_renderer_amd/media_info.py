"""Import-compatible name for the reference's `lut_renderer.media_info` record (media_info.py:12-52).
Probing itself (ffprobe) stays with the reference: SURVEY.md 8 marks it out of scope."""
from .params import VideoInfo, infer_bit_depth as _infer_bit_depth  # noqa: F401
