"""One process, several GPUs: row-block sharding of every frame behind the drop-in API.

The reference is ONE GUI process whose tasks run on a thread pool
(`/root/reference/src/lut_renderer/task_manager.py:229-235`, spawn at `:145-151`); it cannot start
one rank per GPU.  `LutEngineGroup` is what `apply_lut(..., devices=[0..7])` uses instead: one
`LutEngine` (C-ABI context) per device, the lattice uploaded once and copied GPU to GPU over xGMI
(`lutr_lut_broadcast`), and the rows of every frame split with FFmpeg's own slice rule
(`shard.row_blocks`).  Block g is launched on device g's stream; nothing on the
host waits between the launches, so the devices run concurrently.

Frames usually live on ONE device (the caller's).  Row blocks owned by another device travel there
and back as peer copies (torch plumbing, asynchronous, ordered by torch's streams) -- 3.1 MB per
peer and UHD 10-bit frame each way, a few tens of microseconds of xGMI per link.  Callers that
keep frames sharded already (one tensor list per device) pass them as such and nothing is copied.

The one-rank-per-GPU path (`LutEngine.set_lut_distributed`, `bench.py --gpus N`) stays the way to
scale a batch job; this class exists so that the untouched caller of the reference can use every GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _native
from .cube import CubeLut, read_lut
from .engine import LutEngine, parse_pix_fmt
from .shard import row_blocks


class LutEngineGroup:
    """Contexts on `devices` (repeats allowed: two contexts on one GPU split its frames in two launches)."""

    def __init__(self, devices: Sequence[int], treat_as_remote: Optional[bool] = None):
        if not devices:
            raise ValueError("at least one device")
        self.devices = tuple(int(d) for d in devices)
        # Test hook for boxes with one GPU: every engine after the first behaves as if it sat on ANOTHER device -- its row
        # block is sliced, copied (a same-device `.to(copy=True)` stands in for the peer copy), applied as a short frame of its
        # own and copied back, and `lutr_lut_broadcast` takes its hipMemcpyPeerAsync branch (a self-peer copy is legal).
        # LUTR_GROUP_FORCE_REMOTE=1 sets it from the environment.  Never set in production.
        self.treat_as_remote = bool(int(os.environ.get("LUTR_GROUP_FORCE_REMOTE", "0"))) if treat_as_remote is None \
            else bool(treat_as_remote)
        self._lock = threading.RLock()
        self._applied_lut = None
        self.precision = "strict"
        self.engines: List[LutEngine] = []
        try:
            for d in self.devices:
                self.engines.append(LutEngine(d))
        except Exception:
            self.close()
            raise
        self._lib = _native.load()
        self.last_blocks: List[tuple] = []

    # -- lifetime ---------------------------------------------------------
    def close(self) -> None:
        for e in self.engines:
            e.close()
        self.engines = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __len__(self):
        return len(self.engines)

    # -- lattice ----------------------------------------------------------
    def set_lut(self, lut: CubeLut) -> None:
        """Upload on the first device, then ONE broadcast call: every other context receives the lattice GPU to GPU."""
        with self._lock:
            self._applied_lut = None
            root = self.engines[0]
            root.set_lut(lut)
            if len(self.engines) > 1:
                for e in self.engines:
                    e._bind_stream()
                arr = (C.c_void_p * len(self.engines))(*[e._ctx for e in self.engines])
                flags = _native.BCAST_FORCE_PEER_COPY if self.treat_as_remote else 0
                _native.check(self._lib.lutr_lut_broadcast_ex(arr, len(self.engines), 0, flags))
                for e in self.engines[1:]:
                    e._applied_lut = None
                    e.n, e.scale = root.n, np.array(root.scale, dtype=np.float32)
                    e.set_prelut(getattr(lut, "prelut", None))     # host-side state: it does not travel with the lattice copy

    def load_cube(self, path) -> CubeLut:
        lut = read_lut(path)
        self.set_lut(lut)
        return lut

    def set_variant(self, name: str) -> None:
        for e in self.engines:
            e.set_variant(name)

    def set_precision(self, name: str) -> None:
        for e in self.engines:
            e.set_precision(name)
        self.precision = name

    @property
    def last_kernels(self) -> List[str]:
        return [e.last_kernel for e in self.engines]

    def sync(self) -> None:
        for e in self.engines:
            e.sync()

    # -- apply ------------------------------------------------------------
    def apply_yuv(self, src: Sequence[torch.Tensor], dst: Optional[Sequence[torch.Tensor]] = None, *, pix_fmt: str,
                  out_pix_fmt: Optional[str] = None, **kw):
        with self._lock:
            return self._apply_yuv(src, dst, pix_fmt=pix_fmt, out_pix_fmt=out_pix_fmt, **kw)

    def _apply_yuv(self, src: Sequence[torch.Tensor], dst: Optional[Sequence[torch.Tensor]] = None, *, pix_fmt: str,
                   out_pix_fmt: Optional[str] = None, **kw):
        """`LutEngine.apply_yuv` with the rows of every frame split over the group's devices.
        `src` planes live on one device (any); `dst`, if given, on the same one."""
        if kw.get("dither", "none") != "none":
            raise ValueError("error-diffusion dither couples the rows of a frame: it cannot be row-sharded")
        if "row0" in kw or "rows" in kw:
            raise ValueError("the group owns the row partition")
        fin, fout = parse_pix_fmt(pix_fmt), parse_pix_fmt(out_pix_fmt or pix_fmt)
        h, w = src[0].shape[-2], src[0].shape[-1]
        home = src[0].device
        if dst is None:
            dt = torch.uint8 if fout.depth <= 8 else (src[0].dtype if src[0].element_size() == 2 else torch.int16)
            lead = tuple(src[0].shape[:-2])
            dst = [torch.empty(lead + fout.plane_shape(i, w, h), dtype=dt, device=home) for i in range(3)]
        bh = 1 << fin.csy
        blocks = row_blocks(h, len(self.engines), align=bh)
        self.last_blocks = blocks
        pending = []
        self.last_remote = 0                                       # row blocks that took the copy-there-and-back path
        for k, (eng, (r0, r1)) in enumerate(zip(self.engines, blocks)):
            if r1 <= r0:
                continue
            if eng.device == home and not (self.treat_as_remote and k > 0):
                # same GPU: launch on the caller's planes, rows [r0, r1)
                eng.apply_yuv(src, dst, pix_fmt=pix_fmt, out_pix_fmt=out_pix_fmt, row0=r0, rows=r1 - r0, **kw)
                continue
            # another GPU: its row block travels there (peer copy), is processed as a frame of r1-r0 rows, and comes back
            c0, c1 = r0 >> fin.csy, (r1 + bh - 1) >> fin.csy
            rng = [(r0, r1), (c0, c1), (c0, c1)]
            with torch.cuda.device(eng.device):
                part = [p[..., a:b, :].to(eng.device, non_blocking=True, copy=True).contiguous()
                        for p, (a, b) in zip(src, rng)]
                out = eng.apply_yuv(part, None, pix_fmt=pix_fmt, out_pix_fmt=out_pix_fmt, **kw)
            self.last_remote += 1
            pending.append((out, rng))
        for out, rng in pending:                                   # copies back: queued after every launch was issued
            for d, o, (a, b) in zip(dst, out, rng):
                d[..., a:b, :].copy_(o, non_blocking=True)
        return dst
