import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from lut_renderer_amd import cube, frames
from lut_renderer_amd.engine import LutEngine
eng = LutEngine(0); eng.set_variant("vec_lds"); eng.set_precision("fast")
eng.set_lut(cube.CubeLut(33, np.ones(3, np.float32), cube.log709_lattice(33)))
src = frames.natural_yuv(3840, 2160, 10, 1, 1, k=0)
dev = [torch.from_numpy(p.view(np.int16)).to(eng.device).unsqueeze(0).repeat(64, 1, 1) for p in src]
eng.tile_stats(True)
eng.apply_yuv(dev, pix_fmt="yuv420p10le"); torch.cuda.synchronize()
import ctypes as C
from lut_renderer_amd import _native
print(eng.tile_stats(False), eng.last_kernel)
