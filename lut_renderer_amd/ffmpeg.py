"""Import-compatible name for the reference's `lut_renderer.ffmpeg` module
(/root/reference/src/lut_renderer/ffmpeg.py): `from lut_renderer_amd.ffmpeg import build_command`."""
from .command import CommandStage, build_command, build_pipeline  # noqa: F401
from .plan import escape_filter_path as _escape_filter_path  # noqa: F401
