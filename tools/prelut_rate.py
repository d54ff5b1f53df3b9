#!/usr/bin/env python3
"""tools/prelut_rate.py -- throughput of a cineSpace LUT with a shared shaper (lut3d's prelut) on 64 UHD yuv420p10le frames:
the fused tile kernels (round 3) against the vector kernel (LUTR_NO_TILE2_PRELUT=1), by shaper and content."""
import os
import pathlib
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from lut_renderer_amd import cube, frames  # noqa: E402
from lut_renderer_amd.engine import LutEngine  # noqa: E402
from tests.test_lut_formats import _csp_with_prelut  # noqa: E402

eng = LutEngine(0)
n = 33
tab = cube.log709_lattice(n)
xs = np.linspace(0, 1, 33)
d = pathlib.Path(tempfile.mkdtemp())
for name, curve in (("log-like shaper x^0.55", (xs, xs ** 0.55)), ("mild shaper", (xs, xs + 0.4 * xs * (1 - xs)))):
    p = d / "s.csp"
    _csp_with_prelut(p, n, tab, [curve] * 3)
    eng.set_lut(cube.read_lut(p))
    for dist in ("natural", "noise16"):
        src = frames.make_yuv(dist, 3840, 2160, 10, 1, 1, k=3)
        dev = [torch.from_numpy(a.view(np.int16)).to(eng.device).unsqueeze(0).repeat(64, 1, 1) for a in src]
        out = [torch.empty_like(t) for t in dev]
        for _ in range(3):
            eng.apply_yuv(dev, out, pix_fmt="yuv420p10le")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.apply_yuv(dev, out, pix_fmt="yuv420p10le")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        eng.tile_stats(True)
        eng.apply_yuv(dev, out, pix_fmt="yuv420p10le")
        st = eng.tile_stats(False)
        print("LUTR_NO_TILE2_PRELUT=%s  %-24s %-8s %6.1f Gpx/s  %s  tube %s mixed %s level2 %s gather %s of %s" % (
            os.environ.get("LUTR_NO_TILE2_PRELUT", "0"), name, dist, 64 * 3840 * 2160 / dt / 1e9, eng.last_kernel,
            st.get("tube_tiles"), st.get("mixed_tiles"), st.get("level2_tiles"), st.get("global_tiles"), st.get("tiles")))
