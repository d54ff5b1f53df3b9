#!/usr/bin/env python3
"""Fixed cost of a tile-kernel launch: GPU time per launch (HIP events over 300 back-to-back launches) of the fused
yuv420p10le kernel on frames from a few tiles to one 1080p frame.  python tools/startup_probe.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lut_renderer_amd import cube, frames  # noqa: E402
from lut_renderer_amd.engine import LutEngine  # noqa: E402

eng = LutEngine(0)
eng.set_lut(cube.CubeLut(33, np.ones(3, np.float32), cube.log709_lattice(33)))
for variant in ("vec_lds", "vec_global"):
    eng.set_variant(variant)
    for (w, h) in ((128, 8), (256, 64), (1024, 64), (1920, 270), (1920, 1080), (3840, 2160)):
        src = [torch.from_numpy(p.view(np.int16)).cuda() for p in frames.natural_yuv(w, h, 10, 1, 1, k=0)]
        dst = [torch.empty_like(t) for t in src]
        for _ in range(50):
            eng.apply_yuv(src, dst, pix_fmt="yuv420p10le")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 300
        e0.record()
        for _ in range(n):
            eng.apply_yuv(src, dst, pix_fmt="yuv420p10le")
        e1.record()
        torch.cuda.synchronize()
        print(f"{variant:10s} {w:5d}x{h:<5d} {w * h / 1e6:6.2f} Mpx  {e0.elapsed_time(e1) / n * 1e3:7.1f} us/launch  {eng.last_kernel}")
