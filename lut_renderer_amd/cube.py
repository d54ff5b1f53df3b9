""".cube lattices: reading (through liblutr's parser), writing, and generated test LUTs.

The reference never reads a .cube itself: it hands the path to FFmpeg's lut3d
(`/root/reference/src/lut_renderer/ffmpeg.py:246`) and only checks the extension in its
dialogs (`lut_manager.py:121`).  Reading therefore follows FFmpeg's parse_cube
 and is implemented in `csrc/cube_parse.cpp`.

No .cube file ships with the reference, so the LUTs used by tests and bench.py are
generated here from fixed formulas: `identity`, and `log709`, an
S-Log3-like -> Rec.709 transfer composed with a saturation matrix.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from . import _native


@dataclass
class Prelut:
    """lut3d's prelut, the 1D shaper a cineSpace .csp file may put ahead of the cube: `table[c, i]`, sampled at
    `min[c] + i / scale[c]` (FFmpeg's Lut3DPreLut)."""
    table: np.ndarray      # float32[3, size]
    min: np.ndarray        # float32[3]
    scale: np.ndarray      # float32[3]

    def __post_init__(self) -> None:
        self.table = np.ascontiguousarray(self.table, dtype=np.float32)
        self.min = np.ascontiguousarray(self.min, dtype=np.float32)
        self.scale = np.ascontiguousarray(self.scale, dtype=np.float32)
        if self.table.ndim != 2 or self.table.shape[0] != 3:
            raise ValueError(f"prelut table shape {self.table.shape}, expected (3, size)")


@dataclass
class CubeLut:
    """A parsed 3D LUT: `table[r, g, b] = (R, G, B)` floats, blue fastest in memory."""
    n: int
    scale: np.ndarray      # float32[3] = clip(1/(DOMAIN_MAX-DOMAIN_MIN), 0, 1)
    table: np.ndarray      # float32[n, n, n, 3]
    prelut: Optional[Prelut] = None

    def __post_init__(self) -> None:
        self.table = np.ascontiguousarray(self.table, dtype=np.float32)
        self.scale = np.ascontiguousarray(self.scale, dtype=np.float32)
        if self.table.shape != (self.n, self.n, self.n, 3):
            raise ValueError(f"table shape {self.table.shape} does not match n={self.n}")


def read_cube(path) -> CubeLut:
    """Parse `path` with liblutr (FFmpeg parse_cube semantics).  Raises LutrError."""
    return _read_with("lutr_cube_parse", path)


def read_lut(path) -> CubeLut:
    """Parse any 3D LUT file lut3d accepts (.cube, .dat, .3dl, .m3d, .csp -- with its pre-LUT, if it has one), picked by
    extension like FFmpeg's file= option.  Raises LutrError."""
    lib = _native.load()
    rgb, pre = C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
    n, psize = C.c_int(0), C.c_int(0)
    scale, pmin, pscale = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    _native.check(lib.lutr_lut_parse_ex(str(path).encode(), C.byref(rgb), C.byref(n), scale, C.byref(pre), C.byref(psize),
                                        pmin, pscale))
    try:
        table = np.ctypeslib.as_array(rgb, shape=(n.value ** 3 * 3,)).astype(np.float32, copy=True)
        prelut = None
        if psize.value > 0:
            pt = np.ctypeslib.as_array(pre, shape=(3 * psize.value,)).astype(np.float32, copy=True)
            prelut = Prelut(pt.reshape(3, psize.value), np.array(list(pmin), np.float32), np.array(list(pscale), np.float32))
    finally:
        lib.lutr_cube_free(rgb)
        if psize.value > 0:
            lib.lutr_cube_free(pre)
    return CubeLut(n.value, np.array(list(scale), dtype=np.float32), table.reshape(n.value, n.value, n.value, 3), prelut)


def _read_with(symbol: str, path) -> CubeLut:
    lib = _native.load()
    rgb = C.POINTER(C.c_float)()
    n = C.c_int(0)
    scale = (C.c_float * 3)()
    _native.check(getattr(lib, symbol)(str(path).encode(), C.byref(rgb), C.byref(n), scale))
    try:
        count = n.value ** 3 * 3
        table = np.ctypeslib.as_array(rgb, shape=(count,)).astype(np.float32, copy=True)
    finally:
        lib.lutr_cube_free(rgb)
    return CubeLut(n.value, np.array(list(scale), dtype=np.float32), table.reshape(n.value, n.value, n.value, 3))


def write_cube(path, table: np.ndarray, *, title: Optional[str] = None,
               domain_min: Optional[Sequence[float]] = None,
               domain_max: Optional[Sequence[float]] = None, fmt: str = "%.6f") -> Path:
    """Write `table[r, g, b]` as an Adobe/Resolve .cube: red varies fastest in the file."""
    table = np.asarray(table, dtype=np.float32)
    n = table.shape[0]
    path = Path(path)
    with open(path, "w") as f:
        if title is not None:
            f.write(f'TITLE "{title}"\n')
        f.write(f"LUT_3D_SIZE {n}\n")
        if domain_min is not None:
            f.write("DOMAIN_MIN " + " ".join(fmt % v for v in domain_min) + "\n")
        if domain_max is not None:
            f.write("DOMAIN_MAX " + " ".join(fmt % v for v in domain_max) + "\n")
        # file order: for b: for g: for r  -> transpose so r is the fastest axis
        flat = np.transpose(table, (2, 1, 0, 3)).reshape(-1, 3)
        np.savetxt(f, flat, fmt=fmt)
    return path


def identity_lattice(n: int) -> np.ndarray:
    ax = (np.arange(n, dtype=np.float64) / (n - 1)).astype(np.float32)
    r, g, b = np.meshgrid(ax, ax, ax, indexing="ij")
    return np.stack([r, g, b], axis=-1).astype(np.float32)


def _slog3_to_linear(x: np.ndarray) -> np.ndarray:
    code = x * 1023.0
    hi = np.power(10.0, (code - 420.0) / 261.5) * 0.19 - 0.01
    lo = (code - 95.0) * 0.01125 / (171.2102946929 - 95.0)
    return np.where(code >= 171.2102946929, hi, lo)


def _rec709_oetf(v: np.ndarray) -> np.ndarray:
    v = np.clip(v, 0.0, None)
    return np.where(v < 0.018, 4.5 * v, 1.099 * np.power(v, 0.45) - 0.099)


def log709_lattice(n: int, saturation: float = 1.25, exposure: float = 1.0) -> np.ndarray:
    """S-Log3-like log -> Rec.709 display look: per-channel transfer, a 3x3 saturation
    matrix around Rec.709 luma, a soft shoulder, then the 709 OETF.  Deterministic."""
    lat = identity_lattice(n).astype(np.float64)
    lin = _slog3_to_linear(lat) * exposure
    w = np.array([0.2126, 0.7152, 0.0722])
    luma = (lin * w).sum(axis=-1, keepdims=True)
    lin = luma + saturation * (lin - luma)
    lin = np.clip(lin, 0.0, None)
    lin = lin / (1.0 + 0.18 * lin)          # soft shoulder keeps highlights off the clip
    out = _rec709_oetf(lin * 1.18)
    return np.clip(out, 0.0, 1.0).astype(np.float32)


def generate(name: str, n: int) -> np.ndarray:
    if name == "identity":
        return identity_lattice(n)
    if name == "log709":
        return log709_lattice(n)
    raise ValueError(f"unknown generated LUT '{name}'")
