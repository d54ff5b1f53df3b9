"""ctypes binding of liblutr.so (the C-ABI declared in include/lutr.h).

The product path has no CPU fallback: if the shared library is missing or cannot be
loaded this module raises, and every compute entry point needs a GPU context.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "liblutr.so"

# error codes of include/lutr.h
OK, ENOENT, EIO, ENOMEM, EINVAL, EILSEQ = 0, -2, -5, -12, -22, -84

INTERP = {"nearest": 0, "trilinear": 1, "tetrahedral": 2, "pyramid": 3, "prism": 4}
MATRIX = {"bt709": 0, "smpte170m": 1, "bt470bg": 1, "bt601": 1, "bt2020nc": 2, "bt2020c": 2}
RANGE = {"tv": 0, "pc": 1}
DITHER = {"none": 0, "error_diffusion": 1}
VARIANT = {"auto": 0, "generic": 1, "vec_global": 2, "vec_lds": 3}
PRECISION = {"strict": 0, "fast": 1}
BCAST_FORCE_PEER_COPY = 1

#: every symbol include/lutr.h declares (tests check the library exports each one)
SYMBOLS = (
    "lutr_version", "lutr_last_error",
    "lutr_cube_parse", "lutr_cube_free", "lutr_lut_parse", "lutr_lut_parse_ex", "lutr_ctx_set_prelut",
    "lutr_ctx_create", "lutr_ctx_destroy", "lutr_ctx_set_stream", "lutr_ctx_sync",
    "lutr_ctx_set_lut", "lutr_ctx_lut_alloc", "lutr_ctx_lut_device", "lutr_ctx_lut_seal",
    "lutr_lattice_bytes", "lutr_lut_broadcast", "lutr_lut_broadcast_ex",
    "lutr_apply_planar_rgb", "lutr_apply_packed_rgb", "lutr_apply_yuv", "lutr_apply_yuv_dither",
    "lutr_ctx_set_variant", "lutr_ctx_set_precision", "lutr_ctx_last_kernel", "lutr_ctx_tile_stats", "lutr_yuv_constants",
)


def fmt_code(depth: int, csx: int, csy: int) -> int:
    """LUTR_FMT(depth, csx, csy)"""
    return depth | (csx << 8) | (csy << 9)


class YuvParams(C.Structure):
    """struct lutr_yuv_params"""
    _fields_ = [(name, C.c_int32) for name in (
        "fmt_in", "fmt_out", "lut_depth", "matrix_in", "matrix_out", "range_src", "range_in", "range_out")]


class Planes(C.Structure):
    """struct lutr_planes"""
    _fields_ = [("data", C.c_void_p * 3), ("stride", C.c_ssize_t * 3), ("frame_stride", C.c_int64 * 3)]


class Packed(C.Structure):
    """struct lutr_packed"""
    _fields_ = [("data", C.c_void_p), ("stride", C.c_ssize_t), ("frame_stride", C.c_int64)]


def packed_code(bits: int, ncomp: int, ro: int, go: int, bo: int) -> int:
    """LUTR_PACKED(bits, ncomp, ro, go, bo)"""
    return bits | (ncomp << 8) | (ro << 12) | (go << 16) | (bo << 20)


#: FFmpeg names of the packed RGB formats lut3d takes -> (bits, components, R, G, B component index)
PACKED_FORMATS = {
    "rgb24": (8, 3, 0, 1, 2), "bgr24": (8, 3, 2, 1, 0),
    "rgba": (8, 4, 0, 1, 2), "rgb0": (8, 4, 0, 1, 2), "bgra": (8, 4, 2, 1, 0), "bgr0": (8, 4, 2, 1, 0),
    "argb": (8, 4, 1, 2, 3), "0rgb": (8, 4, 1, 2, 3), "abgr": (8, 4, 3, 2, 1), "0bgr": (8, 4, 3, 2, 1),
    "rgb48le": (16, 3, 0, 1, 2), "bgr48le": (16, 3, 2, 1, 0),
    "rgba64le": (16, 4, 0, 1, 2), "bgra64le": (16, 4, 2, 1, 0),
}


class LutrError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"liblutr error {code}: {message}")
        self.code = code
        self.message = message


_lib = None


def load() -> C.CDLL:
    """Load liblutr.so once; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("LUTR_LIBRARY", LIB_PATH))
    if not path.exists():
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C lut_renderer_amd/csrc`. The LUT engine has no CPU fallback.")
    lib = C.CDLL(str(path))
    vp, ci, cp = C.c_void_p, C.c_int, C.c_char_p
    lib.lutr_version.restype = cp
    lib.lutr_last_error.restype = cp
    lib.lutr_cube_parse.argtypes = [cp, C.POINTER(C.POINTER(C.c_float)), C.POINTER(ci), C.POINTER(C.c_float)]
    lib.lutr_lut_parse.argtypes = lib.lutr_cube_parse.argtypes
    lib.lutr_lut_parse_ex.argtypes = list(lib.lutr_cube_parse.argtypes) + [C.POINTER(C.POINTER(C.c_float)), C.POINTER(ci),
                                                                           C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.lutr_ctx_set_prelut.argtypes = [vp, C.POINTER(C.c_float), ci, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.lutr_cube_free.argtypes = [C.POINTER(C.c_float)]
    lib.lutr_cube_free.restype = None
    lib.lutr_ctx_create.argtypes = [ci, C.POINTER(vp)]
    lib.lutr_ctx_destroy.argtypes = [vp]
    lib.lutr_ctx_destroy.restype = None
    lib.lutr_ctx_set_stream.argtypes = [vp, vp]
    lib.lutr_ctx_sync.argtypes = [vp]
    lib.lutr_ctx_set_lut.argtypes = [vp, C.POINTER(C.c_float), ci, C.POINTER(C.c_float)]
    lib.lutr_ctx_lut_alloc.argtypes = [vp, ci, C.POINTER(C.c_float)]
    lib.lutr_ctx_lut_device.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.lutr_ctx_lut_seal.argtypes = [vp]
    lib.lutr_lut_broadcast.argtypes = [C.POINTER(vp), ci, ci]
    lib.lutr_lut_broadcast_ex.argtypes = [C.POINTER(vp), ci, ci, C.c_uint]
    lib.lutr_lattice_bytes.argtypes = [ci]
    lib.lutr_lattice_bytes.restype = C.c_size_t
    lib.lutr_apply_planar_rgb.argtypes = [vp, ci, ci, ci, ci, ci, C.POINTER(Planes), C.POINTER(Planes), ci, ci]
    lib.lutr_apply_packed_rgb.argtypes = [vp, ci, ci, ci, ci, ci, C.POINTER(Packed), C.POINTER(Packed), ci, ci]
    lib.lutr_apply_yuv.argtypes = [vp, C.POINTER(YuvParams), ci, ci, ci, ci, C.POINTER(Planes), C.POINTER(Planes),
                                   ci, ci]
    lib.lutr_apply_yuv_dither.argtypes = [vp, C.POINTER(YuvParams), ci, ci, ci, ci, ci, C.POINTER(Planes),
                                          C.POINTER(Planes)]
    lib.lutr_ctx_set_variant.argtypes = [vp, ci]
    lib.lutr_ctx_set_precision.argtypes = [vp, ci]
    lib.lutr_ctx_last_kernel.argtypes = [vp]
    lib.lutr_ctx_last_kernel.restype = cp
    lib.lutr_ctx_tile_stats.argtypes = [vp, ci, C.POINTER(C.c_uint64)]
    lib.lutr_yuv_constants.argtypes = [C.POINTER(YuvParams), C.POINTER(C.c_float)]
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise LutrError(rc, load().lutr_last_error().decode("utf-8", "replace"))
