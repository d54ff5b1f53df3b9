#!/usr/bin/env python3
"""Per-call latency of the fast path on one small frame: engine.apply_yuv (Python wrapper) vs the bare C-ABI call
with prebuilt structs.  python tools/latency_probe.py [1080p|uhd]"""
import ctypes as C
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lut_renderer_amd import _native, cube, frames  # noqa: E402
from lut_renderer_amd.engine import LutEngine, _planes_struct, parse_pix_fmt  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "1080p"
w, h = {"1080p": (1920, 1080), "uhd": (3840, 2160)}[size]
fmt = "yuv420p" if size == "1080p" else "yuv420p10le"
pf = parse_pix_fmt(fmt)
eng = LutEngine(0)
eng.set_lut(cube.CubeLut(33, np.ones(3, np.float32), cube.log709_lattice(33)))
src_np = frames.natural_yuv(w, h, pf.depth, 1, 1, k=0)
src = [torch.from_numpy(p.view(np.int16) if pf.depth > 8 else p).cuda() for p in src_np]
dst = [torch.empty_like(t) for t in src]
N = 3000
for _ in range(300):
    eng.apply_yuv(src, dst, pix_fmt=fmt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    eng.apply_yuv(src, dst, pix_fmt=fmt)
torch.cuda.synchronize()
t_py = (time.perf_counter() - t0) / N
p = _native.YuvParams(pf.code, pf.code, pf.depth, 0, 0, 0, 0, 0)
s, _ = _planes_struct(src, eng.device)
d, _ = _planes_struct(dst, eng.device)
lib, ctx = eng._lib, eng._ctx
t0 = time.perf_counter()
for _ in range(N):
    lib.lutr_apply_yuv(ctx, C.byref(p), 2, w, h, 1, C.byref(s), C.byref(d), 0, h)
torch.cuda.synchronize()
t_c = (time.perf_counter() - t0) / N
# CPU-side cost alone: time the calls without waiting for the GPU in between (queue depth permitting)
print(f"{size} {fmt}: apply_yuv {t_py * 1e6:.1f} us/call, bare C-ABI {t_c * 1e6:.1f} us/call "
      f"({w * h / t_c / 1e9:.1f} Gpx/s single-frame)")
