"""ctypes binding of the CPU oracle (oracle/_build/liblut3d_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (lut_renderer_amd).  See the header of
oracle/lut3d_oracle.h for what the oracle restates and why parity is "unpinned".
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "_build" / "liblut3d_oracle.so"

INTERP = {"nearest": 0, "trilinear": 1, "tetrahedral": 2, "pyramid": 3, "prism": 4}
MATRIX = {"bt709": 0, "smpte170m": 1, "bt470bg": 1, "bt601": 1, "bt2020nc": 2, "bt2020c": 2}
RANGE = {"tv": 0, "pc": 1}


class OrcLut(C.Structure):
    _fields_ = [("n", C.c_int), ("scale", C.c_float * 3), ("rgb", C.POINTER(C.c_float)),
                ("pre_size", C.c_int), ("pre_min", C.c_float * 3), ("pre_scale", C.c_float * 3),
                ("prelut", C.POINTER(C.c_float))]


class Prelut:
    """lut3d's 1D shaper ahead of the cube (cineSpace .csp): table[3, size], sampled at min[c] + i / scale[c]."""

    def __init__(self, table, vmin, scale):
        self.table = np.ascontiguousarray(table, dtype=np.float32)
        self.min = np.asarray(vmin, dtype=np.float32)
        self.scale = np.asarray(scale, dtype=np.float32)


class YuvConsts(C.Structure):
    _fields_ = (
        [(k, C.c_float) for k in ("ky", "yb", "coff", "krv", "kgu", "kgv", "kbu", "max_l",
                                  "cyr", "cyg", "cyb", "yob", "cbr", "cbg", "cbb",
                                  "crr", "crg", "crb", "cob", "max_o")]
        + [("pre", C.c_int)]
        + [(k, C.c_float) for k in ("py", "pyb", "pc", "pcb", "pre_max")]
    )

    def as_block(self) -> np.ndarray:
        """Same 32-float layout liblutr's lutr_yuv_constants() returns."""
        vals = [getattr(self, k) for k, _ in self._fields_[:20]]
        vals.append(float(self.pre))
        vals += [self.py, self.pyb, self.pc, self.pcb, self.pre_max]
        vals += [0.0] * (32 - len(vals))
        return np.array(vals, dtype=np.float32)


class OracleError(RuntimeError):
    def __init__(self, code: int):
        super().__init__(f"oracle error {code}")
        self.code = code


_lib = None


def build() -> Path:
    subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    return LIB_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        lib = C.CDLL(str(LIB_PATH))
        P3 = C.c_void_p * 3
        S3 = C.c_ssize_t * 3
        lib.orc_cube_parse.argtypes = [C.c_char_p, C.POINTER(OrcLut)]
        lib.orc_lut_file_parse.argtypes = [C.c_char_p, C.POINTER(OrcLut)]
        lib.orc_lut_free.argtypes = [C.POINTER(OrcLut)]
        lib.orc_lut_free.restype = None
        lib.orc_apply_planar_rgb.argtypes = [C.POINTER(OrcLut), C.c_int, C.c_int, C.c_int, C.c_int,
                                             P3, S3, P3, S3, C.c_int]
        lib.orc_apply_pixel.argtypes = [C.POINTER(OrcLut), C.c_int, C.c_int, C.c_int * 3, C.c_int * 3]
        lib.orc_yuv_constants.argtypes = [C.c_int] * 9 + [C.POINTER(YuvConsts)]
        lib.orc_apply_yuv.argtypes = [C.POINTER(OrcLut), C.c_int, C.POINTER(YuvConsts)] + [C.c_int] * 7 + \
                                     [P3, S3, P3, S3, C.c_int]
        lib.orc_apply_yuv_dither.argtypes = lib.orc_apply_yuv.argtypes
        lib.orc_apply_yuv_fast.argtypes = lib.orc_apply_yuv.argtypes
        lib.orc_f2h.argtypes = [C.c_float]
        lib.orc_f2h.restype = C.c_uint16
        lib.orc_h2f.argtypes = [C.c_uint16]
        lib.orc_h2f.restype = C.c_float
        lib.orc_dither_plane.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p,
                                         C.c_ssize_t]
        lib.orc_dither_plane.restype = None
        _lib = lib
    return _lib


def parse_cube(path):
    """-> (n, scale float32[3], table float32[n,n,n,3]); raises OracleError(code)."""
    return _parse_with("orc_cube_parse", path)


def parse_lut_file(path):
    """Any lut3d file format by extension (.cube .dat .3dl .m3d .csp); same return as parse_cube."""
    return _parse_with("orc_lut_file_parse", path)


def parse_lut_file_ex(path):
    """parse_lut_file plus the file's prelut (a Prelut, or None): (n, scale, table, prelut)."""
    return _parse_with("orc_lut_file_parse", path, want_prelut=True)


def _parse_with(symbol, path, want_prelut=False):
    lib = load()
    lut = OrcLut()
    rc = getattr(lib, symbol)(str(path).encode(), C.byref(lut))
    if rc:
        raise OracleError(rc)
    try:
        n = lut.n
        table = np.ctypeslib.as_array(lut.rgb, shape=(n * n * n * 3,)).astype(np.float32, copy=True)
        scale = np.array(list(lut.scale), dtype=np.float32)
        pre = None
        if lut.pre_size > 0:
            pt = np.ctypeslib.as_array(lut.prelut, shape=(3 * lut.pre_size,)).astype(np.float32, copy=True)
            pre = Prelut(pt.reshape(3, lut.pre_size), list(lut.pre_min), list(lut.pre_scale))
    finally:
        lib.orc_lut_free(C.byref(lut))
    if want_prelut:
        return n, scale, table.reshape(n, n, n, 3), pre
    if pre is not None:
        raise OracleError(-22)        # a caller that cannot carry the prelut must not silently drop it
    return n, scale, table.reshape(n, n, n, 3)


def _lut_struct(table: np.ndarray, scale, prelut=None) -> tuple:
    table = np.ascontiguousarray(table, dtype=np.float32)
    lut = OrcLut()
    lut.n = table.shape[0]
    for i in range(3):
        lut.scale[i] = float(scale[i])
    lut.rgb = table.ctypes.data_as(C.POINTER(C.c_float))
    if prelut is not None:
        lut.pre_size = prelut.table.shape[1]
        for i in range(3):
            lut.pre_min[i] = float(prelut.min[i])
            lut.pre_scale[i] = float(prelut.scale[i])
        lut.prelut = prelut.table.ctypes.data_as(C.POINTER(C.c_float))
    return lut, (table, prelut)      # keep the arrays alive alongside the struct


def _plane_args(planes):
    ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
    strides = (C.c_ssize_t * 3)(*[p.strides[0] for p in planes])
    return ptrs, strides


def apply_pixel(table, scale, depth: int, interp: str, rgb, prelut=None) -> tuple:
    lut, _keep = _lut_struct(table, scale, prelut)
    inp = (C.c_int * 3)(*[int(v) for v in rgb])
    out = (C.c_int * 3)()
    rc = load().orc_apply_pixel(C.byref(lut), depth, INTERP[interp], inp, out)
    if rc:
        raise OracleError(rc)
    return tuple(out)


def apply_rgb(table, scale, depth: int, interp: str, planes, nthreads: int = 1, prelut=None):
    """planes: (G, B, R) arrays [H,W] uint8 (depth 8) or uint16; returns new (G, B, R)."""
    lut, _keep = _lut_struct(table, scale, prelut)
    src = [np.ascontiguousarray(p) for p in planes]
    dst = [np.empty_like(p) for p in src]
    h, w = src[0].shape
    sp, ss = _plane_args(src)
    dp, ds = _plane_args(dst)
    rc = load().orc_apply_planar_rgb(C.byref(lut), depth, INTERP[interp], w, h, sp, ss, dp, ds, nthreads)
    if rc:
        raise OracleError(rc)
    return dst


#: packed RGB formats of FFmpeg's lut3d: name -> (bits, components, index of R, G, B);
#: rgba_map semantics of libavfilter/drawutils ff_fill_rgba_map [FFmpeg-recall]
PACKED = {
    "rgb24": (8, 3, 0, 1, 2), "bgr24": (8, 3, 2, 1, 0),
    "rgba": (8, 4, 0, 1, 2), "rgb0": (8, 4, 0, 1, 2), "bgra": (8, 4, 2, 1, 0), "bgr0": (8, 4, 2, 1, 0),
    "argb": (8, 4, 1, 2, 3), "0rgb": (8, 4, 1, 2, 3), "abgr": (8, 4, 3, 2, 1), "0bgr": (8, 4, 3, 2, 1),
    "rgb48le": (16, 3, 0, 1, 2), "bgr48le": (16, 3, 2, 1, 0),
    "rgba64le": (16, 4, 0, 1, 2), "bgra64le": (16, 4, 2, 1, 0),
}


def apply_packed(table, scale, pix_fmt: str, interp: str, img, nthreads: int = 1):
    """lut3d on one packed image [H,W,C] (uint8 / uint16): the same per-pixel arithmetic as the
    planar path at depth 8 / 16 (A.3: M = 255 / 65535), components addressed through the format's
    rgba map; the fourth component is copied (vf_lut3d.c `dst[x + a] = src[x + a]`)."""
    bits, nc, ro, go, bo = PACKED[pix_fmt]
    img = np.ascontiguousarray(img)
    assert img.ndim == 3 and img.shape[2] == nc and img.dtype.itemsize * 8 == bits
    g, b, r = apply_rgb(table, scale, bits, interp, (img[..., go], img[..., bo], img[..., ro]), nthreads)
    out = img.copy()
    out[..., ro], out[..., go], out[..., bo] = r, g, b
    return out


def yuv_constants(matrix_in="bt709", range_in="tv", matrix_out=None, range_out="tv",
                  din=8, dl=None, dout=None, chroma_n=4, prologue=False) -> YuvConsts:
    k = YuvConsts()
    dl = din if dl is None else dl
    dout = dl if dout is None else dout
    rc = load().orc_yuv_constants(MATRIX[matrix_in], RANGE[range_in], MATRIX[matrix_out or matrix_in],
                                  RANGE[range_out], din, dl, dout, chroma_n, int(bool(prologue)), C.byref(k))
    if rc:
        raise OracleError(rc)
    return k


def apply_yuv(table, scale, interp: str, consts: YuvConsts, din: int, dl: int, dout: int,
              csx: int, csy: int, planes, nthreads: int = 1, dither: str = "none", fast: bool = False, prelut=None):
    """planes: (Y, Cb, Cr) arrays; returns new (Y, Cb, Cr) with the output container dtype.
    dither="error_diffusion": Floyd-Steinberg on the final quantisation (orc_apply_yuv_dither).
    fast=True: the product's tolerance-bounded FAST variant (orc_apply_yuv_fast), not FFmpeg's arithmetic."""
    if dither not in ("none", "error_diffusion"):
        raise ValueError(dither)
    if fast and dither != "none":
        raise ValueError("the fast variant has no dither path")
    if fast and prelut is not None:
        raise ValueError("the fast variant has no prelut path")
    lut, _keep = _lut_struct(table, scale, prelut)
    src = [np.ascontiguousarray(p) for p in planes]
    odt = np.uint8 if dout <= 8 else np.uint16
    dst = [np.zeros(p.shape, dtype=odt) for p in src]
    h, w = src[0].shape
    sp, ss = _plane_args(src)
    dp, ds = _plane_args(dst)
    fn = load().orc_apply_yuv_fast if fast else (load().orc_apply_yuv if dither == "none" else load().orc_apply_yuv_dither)
    rc = fn(C.byref(lut), INTERP[interp], C.byref(consts), din, dl, dout, csx, csy, w, h, sp, ss, dp, ds, nthreads)
    if rc:
        raise OracleError(rc)
    return dst


def dither_plane(x, maxv: float, wide: bool):
    """Floyd-Steinberg error diffusion of one float plane [H,W] to integer codes (orc_dither_plane)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    h, w = x.shape
    dst = np.zeros((h, w), dtype=np.uint16 if wide else np.uint8)
    load().orc_dither_plane(x.ctypes.data_as(C.POINTER(C.c_float)), w, h, C.c_float(maxv), int(bool(wide)),
                            dst.ctypes.data_as(C.c_void_p), C.c_ssize_t(dst.strides[0]))
    return dst
