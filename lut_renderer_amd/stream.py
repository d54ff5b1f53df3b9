"""Host-resident frame queue -> GPU -> host, overlapped (BASELINE.json config 5).

What the reference gets from ffmpeg's internal frame queue when it runs
`ffmpeg -i in -vf ...lut3d... out` (/root/reference/src/lut_renderer/task_manager.py:145-151),
rebuilt for frames that live in host memory: a ring of pinned host buffers, three HIP streams
(host->device, kernels, device->host) and events between them, so the PCIe copies of batch
i+1 / i-1 run under the LUT kernel of batch i (`hipMemcpyAsync` double buffering).  The LUT
kernel itself is ~60x faster than PCIe Gen5 x16 can feed it, so this pipeline is PCIe-bound by
construction; `bench.py --pipeline host` reports its rate beside (never as) the HBM-resident metric.

Frames are rawvideo-style: planes back to back (Y, Cb, Cr), frames back to back in a batch.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterable, Iterator, List, Optional

import numpy as np
import torch

from .engine import LutEngine, PixFmt, parse_pix_fmt


@dataclass
class FrameLayout:
    """Byte layout of one planar frame in a rawvideo stream."""
    fmt: PixFmt
    width: int
    height: int

    @property
    def itemsize(self) -> int:
        return 1 if self.fmt.depth <= 8 else 2

    @property
    def plane_shapes(self) -> List[tuple]:
        return [self.fmt.plane_shape(i, self.width, self.height) for i in range(3)]

    @property
    def plane_bytes(self) -> List[int]:
        return [h * w * self.itemsize for h, w in self.plane_shapes]

    @property
    def frame_bytes(self) -> int:
        return sum(self.plane_bytes)

    def plane_views(self, buf: torch.Tensor, nframes: int) -> List[torch.Tensor]:
        """[F,H,W] views of the three planes inside a flat uint8 buffer of `nframes` frames."""
        dt = torch.uint8 if self.itemsize == 1 else torch.int16
        typed = buf.view(dt)
        fe = self.frame_bytes // self.itemsize
        out, off = [], 0
        for (h, w), nbytes in zip(self.plane_shapes, self.plane_bytes):
            out.append(torch.as_strided(typed, (nframes, h, w), (fe, w, 1), off))
            off += nbytes // self.itemsize
        return out


class HostPipeline:
    """Apply the LUT to batches of host frames with copies overlapped against compute."""

    def __init__(self, engine: LutEngine, pix_fmt: str, width: int, height: int, batch: int = 8, slots: int = 3,
                 out_pix_fmt: Optional[str] = None, **apply_kw):
        self.eng = engine
        self.fin = FrameLayout(parse_pix_fmt(pix_fmt.replace("yuvj", "yuv")), width, height)
        self.fout = FrameLayout(parse_pix_fmt(out_pix_fmt or pix_fmt.replace("yuvj", "yuv")), width, height)
        self.batch, self.slots = int(batch), int(slots)
        self.kw = dict(apply_kw, pix_fmt=self.fin.fmt.name, out_pix_fmt=self.fout.fmt.name)
        dev = engine.device
        self.h_in = [torch.empty(self.batch * self.fin.frame_bytes, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.h_out = [torch.empty(self.batch * self.fout.frame_bytes, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.d_in = [torch.empty(self.batch * self.fin.frame_bytes, dtype=torch.uint8, device=dev) for _ in range(slots)]
        self.d_out = [torch.empty(self.batch * self.fout.frame_bytes, dtype=torch.uint8, device=dev) for _ in range(slots)]
        self.s_h2d, self.s_run, self.s_d2h = (torch.cuda.Stream(dev) for _ in range(3))
        self.e_in = [torch.cuda.Event() for _ in range(slots)]
        self.e_run = [torch.cuda.Event() for _ in range(slots)]
        self.e_out = [torch.cuda.Event() for _ in range(slots)]

    def host_in(self, slot: int) -> np.ndarray:
        return self.h_in[slot].numpy()

    def host_out(self, slot: int) -> np.ndarray:
        return self.h_out[slot].numpy()

    def _submit(self, slot: int, nframes: int) -> None:
        nb_in, nb_out = nframes * self.fin.frame_bytes, nframes * self.fout.frame_bytes
        with torch.cuda.stream(self.s_h2d):
            self.d_in[slot][:nb_in].copy_(self.h_in[slot][:nb_in], non_blocking=True)
            self.e_in[slot].record()
        with torch.cuda.stream(self.s_run):
            self.s_run.wait_event(self.e_in[slot])
            src = self.fin.plane_views(self.d_in[slot], nframes)
            dst = self.fout.plane_views(self.d_out[slot], nframes)
            self.eng.apply_yuv(src, dst, **self.kw)          # launches on the current (s_run) stream
            self.e_run[slot].record()
        with torch.cuda.stream(self.s_d2h):
            self.s_d2h.wait_event(self.e_run[slot])
            self.h_out[slot][:nb_out].copy_(self.d_out[slot][:nb_out], non_blocking=True)
            self.e_out[slot].record()

    def run(self, fill: Callable[[np.ndarray, int], int], drain: Callable[[np.ndarray, int], None],
            total_frames: Optional[int] = None, stop: Optional[Callable[[], bool]] = None) -> int:
        """`fill(host_in_bytes, max_frames) -> frames written` (0 = end of stream) produces input,
        `drain(host_out_bytes, nframes)` consumes output, both on the calling thread.  Returns the
        number of frames processed."""
        pending: List[tuple] = []       # (slot, nframes) in submission order
        done = 0
        i = 0
        while True:
            slot = i % self.slots
            if len(pending) == self.slots:                       # ring full: retire the oldest first
                s, n = pending.pop(0)
                self.e_out[s].synchronize()
                drain(self.host_out(s)[: n * self.fout.frame_bytes], n)
                done += n
            if stop is not None and stop():
                break
            want = self.batch if total_frames is None else min(self.batch, total_frames - done - sum(n for _, n in pending))
            if want <= 0:
                break
            n = fill(self.host_in(slot)[: want * self.fin.frame_bytes], want)
            if n <= 0:
                break
            self._submit(slot, n)
            pending.append((slot, n))
            i += 1
        for s, n in pending:
            self.e_out[s].synchronize()
            drain(self.host_out(s)[: n * self.fout.frame_bytes], n)
            done += n
        return done
