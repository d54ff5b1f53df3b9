#!/usr/bin/env python3
"""Regenerates tests/golden/pixels.npz: small seeded frames and what the CPU oracle makes of them,
one case per kernel family (planar RGB / fused YUV, each depth, subsampling, mode, matrix, range,
prologue, mixed depth).  The oracle is a restatement of FFmpeg lut3d (PARITY
UNPINNED against a live ffmpeg), so these vectors pin the oracle and the kernels against drift
between rounds, not against FFmpeg.

    python tests/golden/make_pixel_fixtures.py
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from lut_renderer_amd import cube, frames  # noqa: E402
from oracle import binding as orc  # noqa: E402

W, H = 64, 36
CASES = []
for depth in (8, 10, 12, 16):
    for mode in ("nearest", "trilinear", "tetrahedral", "pyramid", "prism"):
        CASES.append(dict(kind="rgb", depth=depth, mode=mode, lut="log709_17"))
CASES.append(dict(kind="rgb", depth=10, mode="tetrahedral", lut="random_9_domain"))
for fmt, depth, csx, csy in (("yuv420p", 8, 1, 1), ("yuv420p10le", 10, 1, 1), ("yuv422p10le", 10, 1, 0),
                             ("yuv444p10le", 10, 0, 0), ("yuv420p12le", 12, 1, 1)):
    for mode in ("trilinear", "tetrahedral"):
        for matrix, rin, rout in (("bt709", "tv", "tv"), ("bt2020nc", "tv", "tv"), ("smpte170m", "pc", "pc")):
            CASES.append(dict(kind="yuv", fmt=fmt, din=depth, dl=depth, dout=depth, csx=csx, csy=csy, mode=mode,
                              matrix=matrix, rin=rin, rout=rout, prologue=False, lut="log709_17"))
CASES.append(dict(kind="yuv", fmt="yuv420p10le", din=10, dl=10, dout=8, csx=1, csy=1, mode="tetrahedral",
                  matrix="bt2020nc", rin="tv", rout="tv", prologue=False, lut="log709_17"))      # App. D case K
CASES.append(dict(kind="yuv", fmt="yuv422p10le", din=10, dl=8, dout=10, csx=1, csy=0, mode="tetrahedral",
                  matrix="smpte170m", rin="tv", rout="tv", prologue=True, lut="log709_17"))      # case D
CASES.append(dict(kind="yuv", fmt="yuv420p", din=8, dl=8, dout=8, csx=1, csy=1, mode="trilinear",
                  matrix="bt709", rin="tv", rout="tv", prologue=True, lut="log709_17"))          # case A


def lattices():
    rng = np.random.default_rng(9)
    return {"log709_17": (cube.log709_lattice(17), np.ones(3, np.float32)),
            "random_9_domain": (rng.uniform(-0.2, 1.2, size=(9, 9, 9, 3)).astype(np.float32),
                                np.array([1.0, 0.5, 0.75], np.float32))}


def main():
    lats = lattices()
    arrays, meta = {}, []
    for name, (tab, sc) in lats.items():
        arrays[f"lat/{name}"] = tab
        arrays[f"scale/{name}"] = sc
    for i, c in enumerate(CASES):
        tab, sc = lats[c["lut"]]
        if c["kind"] == "rgb":
            src = frames.uniform_rgb(W, H, c["depth"], k=100 + i)
            dst = orc.apply_rgb(tab, sc, c["depth"], c["mode"], src)
        else:
            src = frames.make_yuv("uniform" if i % 2 else "natural", W, H, c["din"], c["csx"], c["csy"], k=100 + i,
                                  full_range=c["prologue"] or c["rin"] == "pc")
            k = orc.yuv_constants(c["matrix"], c["rin"], c["matrix"], c["rout"], c["din"], c["dl"], c["dout"],
                                  1 << (c["csx"] + c["csy"]), prologue=c["prologue"])
            dst = orc.apply_yuv(tab, sc, c["mode"], k, c["din"], c["dl"], c["dout"], c["csx"], c["csy"], src)
        for p in range(3):
            arrays[f"{i}/src{p}"] = src[p]
            arrays[f"{i}/dst{p}"] = dst[p]
        meta.append(c)
    out = Path(__file__).with_name("pixels.npz")
    np.savez_compressed(out, meta=np.array(json.dumps(meta)), **arrays)
    print(f"wrote {out}: {len(CASES)} cases, {out.stat().st_size / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
