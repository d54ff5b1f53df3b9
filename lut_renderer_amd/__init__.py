"""MI355X-native 3D-LUT apply engine: a drop-in for the one per-pixel path of
ionlz/LUT-renderer (its `lut3d` filter chain, /root/reference/src/lut_renderer/ffmpeg.py:195-247).

Importing the package does not need a GPU; creating a `LutEngine` does.
"""
from . import _native  # noqa: F401  (fails loudly at first use if liblutr.so is missing)
from .cube import CubeLut, read_cube, read_lut, write_cube, identity_lattice, log709_lattice  # noqa: F401

__all__ = ["CubeLut", "read_cube", "read_lut", "write_cube", "identity_lattice", "log709_lattice", "LutEngine"]


def __getattr__(name):
    # torch is imported lazily so that `import lut_renderer_amd` stays cheap for host-only use
    if name in ("LutEngine", "parse_pix_fmt", "yuv_constants"):
        from . import engine
        return getattr(engine, name)
    if name == "LutEngineGroup":
        from . import multigpu
        return multigpu.LutEngineGroup
    if name == "apply_lut":
        from . import api
        return api.apply_lut
    raise AttributeError(name)
