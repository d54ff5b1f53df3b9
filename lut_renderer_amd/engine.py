"""Python host of the LUT engine: a thin wrapper over the C-ABI (include/lutr.h).

PyTorch is used only as plumbing: device memory (tensors), streams and
torch.distributed (RCCL) for the one collective this path has, the broadcast of the
lattice at LUT load.  All pixel work happens in liblutr's HIP kernels; nothing here
computes pixels, and nothing falls back to the CPU.

Replaces: the `ffmpeg` child process the reference starts per task
(`/root/reference/src/lut_renderer/task_manager.py:145-151`) for the filters built at
`/root/reference/src/lut_renderer/ffmpeg.py:195-247` and `:304-310`.
"""
from __future__ import annotations

import ctypes as C
import re
import threading
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native
from .cube import CubeLut, read_cube, read_lut

_PIXFMT_RE = re.compile(r"^(yuvj?|gbr)(420|422|444)?p(\d+)?(le)?$")


@dataclass(frozen=True)
class PixFmt:
    """Parsed FFmpeg planar pixel-format name (the ones this path can meet)."""
    name: str
    family: str      # "yuv" or "gbr"
    depth: int
    csx: int
    csy: int
    full_range: bool  # yuvj* (legacy full-range marker, media_info.py:145-147)

    @property
    def code(self) -> int:
        return _native.fmt_code(self.depth, self.csx, self.csy)

    @property
    def np_dtype(self):
        return np.uint8 if self.depth <= 8 else np.uint16

    def plane_shape(self, plane: int, w: int, h: int) -> Tuple[int, int]:
        if self.family == "gbr" or plane == 0:
            return h, w
        return (h + (1 << self.csy) - 1) >> self.csy, (w + (1 << self.csx) - 1) >> self.csx


def parse_pix_fmt(name: str) -> PixFmt:
    m = _PIXFMT_RE.match(name or "")
    if not m:
        raise ValueError(f"unsupported pixel format '{name}'")
    fam, sub, depth, _le = m.groups()
    depth_i = int(depth) if depth else 8
    if not 8 <= depth_i <= 16:
        raise ValueError(f"unsupported bit depth in '{name}'")
    if fam == "gbr":
        if sub:
            raise ValueError(f"unsupported pixel format '{name}'")
        return PixFmt(name, "gbr", depth_i, 0, 0, True)
    if not sub:
        raise ValueError(f"unsupported pixel format '{name}'")
    csx, csy = {"420": (1, 1), "422": (1, 0), "444": (0, 0)}[sub]
    return PixFmt(name, "yuv", depth_i, csx, csy, fam == "yuvj")


def _check_planes(planes: Sequence[torch.Tensor], fmt: PixFmt, w: int, h: int, what: str) -> None:
    """The C-ABI takes bare pointers and cannot know buffer sizes: every plane must have exactly the shape and the
    element size `fmt` implies for a w x h frame, or the kernels would read or write outside it."""
    if len(planes) != 3:
        raise ValueError("expected three planes")
    esize = 1 if fmt.depth <= 8 else 2
    for i, t in enumerate(planes):
        if not isinstance(t, torch.Tensor):
            raise TypeError("planes must be torch tensors resident on the engine's GPU")
        if t.dim() not in (2, 3):
            raise ValueError("planes must be [H,W] or [F,H,W]")
        if t.is_floating_point() or t.element_size() != esize:
            raise ValueError(f"{what} plane {i}: '{fmt.name}' takes {8 * esize}-bit integer samples, got {t.dtype}")
        want = fmt.plane_shape(i, w, h)
        if tuple(t.shape[-2:]) != want:
            raise ValueError(f"{what} plane {i} is {tuple(t.shape[-2:])}, '{fmt.name}' at {w}x{h} needs {want}")


def _planes_struct(planes: Sequence[torch.Tensor], device: torch.device) -> Tuple[_native.Planes, int]:
    """Describe three [H,W] or [F,H,W] tensors as struct lutr_planes; returns (struct, nframes)."""
    if len(planes) != 3:
        raise ValueError("expected three planes")
    st = _native.Planes()
    nframes = None
    for i, t in enumerate(planes):
        if not isinstance(t, torch.Tensor):
            raise TypeError("planes must be torch tensors resident on the engine's GPU")
        if t.device != device:
            raise ValueError(f"plane {i} is on {t.device}, engine is on {device}")
        if t.dim() == 2:
            f, fs = 1, 0
        elif t.dim() == 3:
            f, fs = t.shape[0], t.stride(0) * t.element_size()
        else:
            raise ValueError("planes must be [H,W] or [F,H,W]")
        if t.stride(-1) != 1:
            raise ValueError("planes must be dense along the row")
        if nframes is None:
            nframes = f
        elif nframes != f:
            raise ValueError("planes disagree on the number of frames")
        st.data[i] = t.data_ptr()
        st.stride[i] = t.stride(-2) * t.element_size()
        st.frame_stride[i] = fs
    return st, nframes


class LutEngine:
    """One GPU context: a device lattice plus the stream its kernels run on."""

    def __init__(self, device: int = 0, use_torch_stream: bool = True):
        if not torch.cuda.is_available():
            raise RuntimeError("LutEngine needs a HIP GPU; there is no CPU fallback")
        self._lib = _native.load()
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        handle = C.c_void_p()
        _native.check(self._lib.lutr_ctx_create(self.device_index, C.byref(handle)))
        self._ctx = handle
        self.n = 0
        self.scale = None
        self.use_torch_stream = use_torch_stream
        # One context = one stream, one lattice, one work queue (include/lutr.h: thread-safe per context, contexts are not
        # shared between threads).  The reference runs up to 16 tasks on a thread pool (task_manager.py:229-235) and ctypes
        # releases the GIL, so every call that touches the context takes this lock; `api.apply_lut` holds it across
        # set_lut + apply so that a cached engine cannot render one task with another task's lattice.
        self._lock = threading.RLock()
        self.precision = "strict"
        self._applied_lut = None          # the CubeLut object apply_lut uploaded last (its upload-skipping shortcut)

    # -- lifetime ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_ctx", None):
            self._lib.lutr_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- lattice ----------------------------------------------------------
    def set_lut(self, lut: CubeLut) -> None:
        table = np.ascontiguousarray(lut.table, dtype=np.float32)
        scale = (C.c_float * 3)(*[float(v) for v in lut.scale])
        with self._lock:
            self._applied_lut = None      # any direct upload invalidates apply_lut's "same LUT as last time" shortcut
            _native.check(self._lib.lutr_ctx_set_lut(
                self._ctx, table.ctypes.data_as(C.POINTER(C.c_float)), int(lut.n), scale))
            self.n, self.scale = int(lut.n), np.array(lut.scale, dtype=np.float32)
            self.set_prelut(getattr(lut, "prelut", None))

    def set_prelut(self, pre) -> None:
        """lut3d's prelut (a cineSpace shaper, `cube.Prelut`) for the lattice just set, or None to remove it.  Uploading a
        lattice drops the prelut of the previous one."""
        if pre is None:
            _native.check(self._lib.lutr_ctx_set_prelut(self._ctx, None, 0, None, None))
            return
        table = np.ascontiguousarray(pre.table, dtype=np.float32)
        pmin = (C.c_float * 3)(*[float(v) for v in pre.min])
        pscale = (C.c_float * 3)(*[float(v) for v in pre.scale])
        _native.check(self._lib.lutr_ctx_set_prelut(
            self._ctx, table.ctypes.data_as(C.POINTER(C.c_float)), int(table.shape[1]), pmin, pscale))

    def load_cube(self, path) -> CubeLut:
        lut = read_lut(path)
        self.set_lut(lut)
        return lut

    def lattice_tensor(self) -> torch.Tensor:
        """The device lattice viewed as a float32 tensor [(n+1)^3 * 4] (no copy)."""
        ptr, size = C.c_void_p(), C.c_size_t()
        _native.check(self._lib.lutr_ctx_lut_device(self._ctx, C.byref(ptr), C.byref(size)))
        return _tensor_from_ptr(ptr.value, size.value // 4, self.device)

    def set_lut_distributed(self, lut: Optional[CubeLut], src: int = 0, group=None) -> None:
        """Rank `src` uploads the lattice; every other rank receives it with ONE broadcast
        (RCCL over xGMI on GPUs).  No other collective exists on this path."""
        import torch.distributed as dist
        self._applied_lut = None
        rank = dist.get_rank(group)
        meta = torch.zeros(4, dtype=torch.float32, device=self.device)
        if rank == src:
            if lut is None:
                raise ValueError("the source rank must pass the LUT")
            self.set_lut(lut)
            meta = torch.tensor([float(lut.n), *[float(v) for v in lut.scale]], dtype=torch.float32,
                                device=self.device)
        dist.broadcast(meta, src=src, group=group)
        if rank != src:
            n = int(meta[0].item())
            scale = (C.c_float * 3)(*[float(v) for v in meta[1:].tolist()])
            _native.check(self._lib.lutr_ctx_lut_alloc(self._ctx, n, scale))
            self.n, self.scale = n, np.array(list(scale), dtype=np.float32)
        dist.broadcast(self.lattice_tensor(), src=src, group=group)
        if rank != src:
            torch.cuda.current_stream(self.device).synchronize()
            _native.check(self._lib.lutr_ctx_lut_seal(self._ctx))      # finiteness + value range of what arrived
        # a cineSpace prelut travels with the lattice: its size first (0 = none), then table and ranges in one tensor
        pre = getattr(lut, "prelut", None) if rank == src else None
        size = torch.tensor([0 if pre is None else int(pre.table.shape[1])], dtype=torch.int32, device=self.device)
        dist.broadcast(size, src=src, group=group)
        nsz = int(size.item())
        if nsz:
            from .cube import Prelut
            if rank == src:
                flat = np.concatenate([pre.table.reshape(-1), pre.min, pre.scale]).astype(np.float32)
                buf = torch.from_numpy(flat).to(self.device)
            else:
                buf = torch.empty(3 * nsz + 6, dtype=torch.float32, device=self.device)
            dist.broadcast(buf, src=src, group=group)
            if rank != src:
                host = buf.cpu().numpy()
                self.set_prelut(Prelut(host[:3 * nsz].reshape(3, nsz), host[3 * nsz:3 * nsz + 3], host[3 * nsz + 3:]))

    # -- control ----------------------------------------------------------
    def set_variant(self, name: str) -> None:
        _native.check(self._lib.lutr_ctx_set_variant(self._ctx, _native.VARIANT[name]))

    def set_precision(self, name: str) -> None:
        """"strict" (default): bit-exact with FFmpeg's scalar C.  "fast": allow the tolerance-bounded tile kernels
        (<= 1 code from strict at 8 and 10 bit; include/lutr.h lutr_ctx_set_precision)."""
        if name not in _native.PRECISION:
            raise ValueError(f"unknown precision '{name}' (strict | fast)")
        with self._lock:
            _native.check(self._lib.lutr_ctx_set_precision(self._ctx, _native.PRECISION[name]))
            self.precision = name

    @property
    def last_kernel(self) -> str:
        return self._lib.lutr_ctx_last_kernel(self._ctx).decode()

    def tile_stats(self, enable: bool = True) -> dict:
        """Counters of the LDS-window kernels since the previous call; (re)arms collection."""
        out = (C.c_uint64 * 8)()
        _native.check(self._lib.lutr_ctx_tile_stats(self._ctx, int(enable), out))
        return {"tiles": out[0], "misses": out[1], "global_tiles": out[2], "staged": out[3],
                "tube_tiles": out[6], "level2_tiles": out[7], "mixed_tiles": out[4]}

    def sync(self) -> None:
        _native.check(self._lib.lutr_ctx_sync(self._ctx))

    def _bind_stream(self) -> None:
        if self.use_torch_stream:
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _native.check(self._lib.lutr_ctx_set_stream(self._ctx, C.c_void_p(stream)))

    # -- apply ------------------------------------------------------------
    def apply_rgb(self, src: Sequence[torch.Tensor], dst: Optional[Sequence[torch.Tensor]] = None, *,
                  depth: int, interp: str = "tetrahedral", row0: int = 0, rows: Optional[int] = None):
        """lut3d on planar RGB; planes in gbrp order (G, B, R), each [H,W] or [F,H,W]."""
        if dst is None:
            dst = [torch.empty_like(t) for t in src]
        if not 8 <= int(depth) <= 16:
            raise ValueError(f"unsupported depth {depth}")
        h, w = src[0].shape[-2], src[0].shape[-1]
        fmt = PixFmt(f"gbrp{depth}", "gbr", int(depth), 0, 0, True)
        _check_planes(src, fmt, w, h, "source")
        _check_planes(dst, fmt, w, h, "destination")
        s, nf = _planes_struct(src, self.device)
        d, nfd = _planes_struct(dst, self.device)
        if nf != nfd:
            raise ValueError("src and dst disagree on the number of frames")
        rows = h - row0 if rows is None else rows
        with self._lock:
            self._bind_stream()
            _native.check(self._lib.lutr_apply_planar_rgb(
                self._ctx, depth, _native.INTERP[interp], w, h, nf, C.byref(s), C.byref(d), row0, rows))
        return dst

    def apply_packed(self, src: torch.Tensor, dst: Optional[torch.Tensor] = None, *, pix_fmt: str,
                     interp: str = "tetrahedral", row0: int = 0, rows: Optional[int] = None) -> torch.Tensor:
        """lut3d on packed RGB: `src` is [H,W,C] or [F,H,W,C] (C = 3 or 4, uint8, or int16/uint16 for the
        48/64-bit formats), `pix_fmt` an FFmpeg name from `_native.PACKED_FORMATS`.  The fourth component
        is carried over.  `dst` may be `src` (in place)."""
        if pix_fmt not in _native.PACKED_FORMATS:
            raise ValueError(f"unsupported packed pixel format '{pix_fmt}'")
        bits, nc, ro, go, bo = _native.PACKED_FORMATS[pix_fmt]
        if dst is None:
            dst = torch.empty_like(src)
        descs = []
        for t in (src, dst):
            if not isinstance(t, torch.Tensor) or t.device != self.device:
                raise ValueError("packed images must be torch tensors resident on the engine's GPU")
            if t.dim() not in (3, 4) or t.shape[-1] != nc or t.element_size() * 8 != bits:
                raise ValueError(f"'{pix_fmt}' takes [H,W,{nc}] or [F,H,W,{nc}] tensors of {bits}-bit elements")
            if t.stride(-1) != 1 or t.stride(-2) != nc:
                raise ValueError("pixels must be dense along the row")
            st = _native.Packed()
            st.data = t.data_ptr()
            st.stride = t.stride(-3) * t.element_size()
            st.frame_stride = t.stride(0) * t.element_size() if t.dim() == 4 else 0
            descs.append(st)
        if src.shape != dst.shape:
            raise ValueError("src and dst shapes differ")
        h, w = src.shape[-3], src.shape[-2]
        nf = src.shape[0] if src.dim() == 4 else 1
        rows = h - row0 if rows is None else rows
        with self._lock:
            self._bind_stream()
            _native.check(self._lib.lutr_apply_packed_rgb(
                self._ctx, _native.packed_code(bits, nc, ro, go, bo), _native.INTERP[interp], w, h, nf,
                C.byref(descs[0]), C.byref(descs[1]), row0, rows))
        return dst

    def apply_yuv(self, src: Sequence[torch.Tensor], dst: Optional[Sequence[torch.Tensor]] = None, *,
                  pix_fmt: str, interp: str = "tetrahedral", matrix_in: str = "bt709",
                  matrix_out: Optional[str] = None, range_src: str = "tv", range_in: Optional[str] = None,
                  range_out: str = "tv", lut_depth: Optional[int] = None, out_pix_fmt: Optional[str] = None,
                  row0: int = 0, rows: Optional[int] = None, dither: str = "none"):
        """Fused YUV -> RGB -> lut3d -> RGB -> YUV on planar frames (Y, Cb, Cr).
        dither="error_diffusion" (the reference's `zscale_dither`) dithers the final quantisation; whole frames only."""
        if dither not in _native.DITHER:
            raise ValueError(f"unknown dither mode '{dither}'")
        fin = parse_pix_fmt(pix_fmt)
        fout = parse_pix_fmt(out_pix_fmt or pix_fmt)
        if fin.family != "yuv" or fout.family != "yuv":
            raise ValueError("apply_yuv takes planar YUV formats")
        p = _native.YuvParams()
        p.fmt_in, p.fmt_out = fin.code, fout.code
        p.lut_depth = lut_depth if lut_depth is not None else fin.depth
        p.matrix_in = _native.MATRIX[matrix_in]
        p.matrix_out = _native.MATRIX[matrix_out or matrix_in]
        p.range_src = _native.RANGE[range_src]
        p.range_in = _native.RANGE[range_in or range_src]
        p.range_out = _native.RANGE[range_out]
        h, w = src[0].shape[-2], src[0].shape[-1]
        if dst is None:
            dt = torch.uint8 if fout.depth <= 8 else src[0].dtype if src[0].element_size() == 2 else torch.int16
            lead = tuple(src[0].shape[:-2])
            dst = [torch.empty(lead + fout.plane_shape(i, w, h), dtype=dt, device=self.device) for i in range(3)]
        _check_planes(src, fin, w, h, "source")
        _check_planes(dst, fout, w, h, "destination")
        s, nf = _planes_struct(src, self.device)
        d, nfd = _planes_struct(dst, self.device)
        if nf != nfd:
            raise ValueError("src and dst disagree on the number of frames")
        rows = h - row0 if rows is None else rows
        if dither != "none" and (row0 != 0 or rows != h):
            raise ValueError("error-diffusion dither couples the rows of a frame: whole frames only")
        with self._lock:
            self._bind_stream()
            if dither != "none":
                _native.check(self._lib.lutr_apply_yuv_dither(
                    self._ctx, C.byref(p), _native.INTERP[interp], _native.DITHER[dither], w, h, nf, C.byref(s), C.byref(d)))
            else:
                _native.check(self._lib.lutr_apply_yuv(
                    self._ctx, C.byref(p), _native.INTERP[interp], w, h, nf, C.byref(s), C.byref(d), row0, rows))
        return dst


def yuv_constants(**kw) -> np.ndarray:
    """The 32-float constant block liblutr derives for a lutr_yuv_params (host only, no GPU)."""
    p = _native.YuvParams()
    for k, v in kw.items():
        setattr(p, k, v)
    out = (C.c_float * 32)()
    _native.check(_native.load().lutr_yuv_constants(C.byref(p), out))
    return np.array(list(out), dtype=np.float32)


def _tensor_from_ptr(ptr: int, count: int, device: torch.device) -> torch.Tensor:
    """Wrap `count` device floats at `ptr` as a tensor without copying (__cuda_array_interface__)."""

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {
        "shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2, "strides": None}
    return torch.as_tensor(h, device=device)
