#!/usr/bin/env python3
"""tools/soak.py [seconds] -- GPU soak of the window / tube validity logic: tile kernels vs the scalar generic kernel, bit for bit.

The tube and the raw boxes rest on conservative bounds (csrc/lutr_tile2.hip map_box, tube_lane): if a bound were ever too
optimistic a pixel would read a node that is not staged and come out wrong.  This drives the two kernels with frames built to
sit ON those bounds -- chroma swept radially from neutral to far outside the tube in every direction, luma uniform over the
whole code range (clipping included), random lattice sizes, domains, matrices, ranges, formats, depths, interpolations -- and
compares every sample (and holds the fast kernels to one code from that).  Both sides are the product's strict kernels (GPU against GPU, so sizes can be large); the generic
kernel is itself pinned against the oracle by tests/.
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from lut_renderer_amd import cube  # noqa: E402
from lut_renderer_amd.engine import LutEngine  # noqa: E402


def frame(rng, w, h, depth, csx, csy, full):
    m = (1 << depth) - 1
    cw, ch = w >> csx, h >> csy
    yy, xx = np.mgrid[0:ch, 0:cw].astype(np.float32)
    theta = rng.uniform(0, 2 * np.pi) + 2 * np.pi * xx / cw * rng.integers(1, 5)
    rad = (yy / ch) * rng.uniform(0.05, 0.6) * m                       # neutral at the top, far out at the bottom
    rad = rad + rng.normal(0, rng.choice([0.0, 0.5, 2.0, 8.0]) * (m / 1023.0), size=rad.shape)
    mid = (m + 1) / 2
    cb = np.clip(np.rint(mid + rad * np.cos(theta)), 0, m)
    cr = np.clip(np.rint(mid + rad * np.sin(theta)), 0, m)
    kind = rng.integers(0, 3)
    if kind == 0:
        y = rng.integers(0, m + 1, size=(h, w))
    elif kind == 1:
        y = np.clip(np.rint(np.linspace(0, m, w)[None, :] + rng.normal(0, 3, size=(h, w))), 0, m)
    else:
        y = np.full((h, w), rng.integers(0, m + 1))
    dt = np.uint16 if depth > 8 else np.uint8
    return [y.astype(dt), cb.astype(dt), cr.astype(dt)]


def main():
    import os
    os.environ.setdefault("LUTR_SMALL_JOB_MPX", "0")          # "auto" must take the tile kernels whatever the launch size
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    eng = LutEngine(0)
    rng = np.random.default_rng(20261004)
    t0, runs, px, tube_tiles, tiles, fast_runs, pre_runs = time.time(), 0, 0, 0, 0, 0, 0
    fmts = [("yuv420p10le", 10, 1, 1), ("yuv420p", 8, 1, 1), ("yuv422p10le", 10, 1, 0), ("yuv444p10le", 10, 0, 0), ("yuv444p", 8, 0, 0)]
    while time.time() - t0 < budget:
        n = int(rng.choice([9, 17, 19, 20, 21, 22, 26, 29, 33, 33, 33, 37, 40, 41, 65]))      # (up to 21 / 25: the whole lattice in LDS)
        lat = rng.uniform(0.0, 1.0, size=(n, n, n, 3)).astype(np.float32) if rng.random() < 0.3 else cube.log709_lattice(n)
        scale = np.array([1.0, 1.0, 1.0], np.float32) if rng.random() < 0.7 else np.full(3, rng.uniform(0.6, 1.0), np.float32)
        pre = None
        if rng.random() < 0.3:
            # a shaper shared by the three channels (what the fused tile kernels take since round 3): random non-decreasing curve with
            # flat and steep stretches, so the slope bound the tube and the windows rest on (LutConsts::pre_kappa) is far from the mean slope
            size = int(rng.choice([17, 64, 256, 1024]))
            steps = rng.gamma(0.4, 1.0, size=size - 1) * (rng.random(size - 1) < 0.8)
            curve = np.concatenate([[0.0], np.cumsum(steps)])
            curve = (curve / max(curve[-1], 1e-9) * rng.uniform(0.7, 1.0)).astype(np.float32)
            pre = cube.Prelut(np.stack([curve] * 3), np.zeros(3, np.float32), np.full(3, size - 1, np.float32))
            pre_runs += 1
        eng.set_lut(cube.CubeLut(n, scale, lat, prelut=pre))
        fmt, depth, csx, csy = fmts[rng.integers(0, len(fmts))]
        w, h = int(rng.choice([512, 1024, 1920])), int(rng.choice([64, 136, 270]) * 2)
        ragged = rng.random() < 0.25          # a width the 16-byte kernels cannot take whole: "auto" splits it, tile kernels + scalar rest
        if ragged:
            w -= 2 * int(rng.integers(1, 7))
        rg = str(rng.choice(["tv", "pc"]))
        kw = dict(pix_fmt=fmt, interp=str(rng.choice(["tetrahedral", "trilinear"])),
                  matrix_in=str(rng.choice(["bt709", "bt601", "bt2020nc"])), range_src=rg, range_in=rg)
        if rng.random() < 0.25:                # the reference's full-range prologue (BASELINE config 5): pc source, tv into an 8-bit LUT
            kw.update(range_src="pc", range_in="tv", lut_depth=8)
        src = frame(rng, w, h, depth, csx, csy, kw["range_in"] == "pc")
        dev = [torch.from_numpy(p.view(np.int16) if p.dtype == np.uint16 else p).to(eng.device).unsqueeze(0).repeat(4, 1, 1) for p in src]
        if rng.random() < 0.3:                # a row shard of the frames (FFmpeg's slice rule aligns to the chroma block)
            r0 = 2 * int(rng.integers(0, h // 4)); kw.update(row0=r0, rows=2 * int(rng.integers(1, (h - r0) // 2 + 1)))
        eng.set_variant("auto" if ragged else "vec_lds")
        eng.tile_stats(True)
        a = [t.clone() for t in eng.apply_yuv(dev, **kw)]
        st = eng.tile_stats(False)
        name = eng.last_kernel
        if pre is not None and not ragged and "tile2" not in name:        # (a ragged width on unpadded rows is the scalar kernel's)
            raise SystemExit(f"run {runs}: a shared prelut did not take the tile kernels: {name} {fmt} {kw}")
        eng.set_variant("generic")
        b = eng.apply_yuv(dev, **kw)
        r0, rn = kw.get("row0", 0), kw.get("rows", h)
        rows_of = lambda i: slice(r0 >> (csy if i else 0), (r0 + rn) >> (csy if i else 0))     # only the shard's rows are written
        a = [t[:, rows_of(i)] for i, t in enumerate(a)]
        b = [t[:, rows_of(i)] for i, t in enumerate(b)]
        for i, (x, y) in enumerate(zip(a, b)):
            if not torch.equal(x, y):
                d = (x.to(torch.int32) - y.to(torch.int32)).abs()
                raise SystemExit(f"MISMATCH run {runs}: n={n} {fmt} {kw} {name} plane {i}: {int((d > 0).sum())} samples, max {int(d.max())}")
        # the fast kernels (their own, wider tube): at most one code from the strict result wherever they run
        eng.set_variant("auto" if ragged else "vec_lds")
        eng.set_precision("fast")
        c = eng.apply_yuv(dev, **kw)
        eng.set_precision("strict")
        if "fast" in eng.last_kernel:
            fast_runs += 1
            for i, (x, y) in enumerate(zip(c, b)):
                d = (x[:, rows_of(i)].to(torch.int32) - y.to(torch.int32)).abs()
                # one code at the LUT's depth; behind the 8-bit prologue that is up to 4 codes of a 10-bit output
                tol = 1 if kw.get("lut_depth", depth) == depth else 4
                if int(d.max()) > tol:
                    raise SystemExit(f"FAST MISMATCH run {runs}: n={n} {fmt} {kw} {eng.last_kernel} plane {i}: max {int(d.max())}")
        runs += 1
        px += 4 * w * h
        tube_tiles += st["tube_tiles"]
        tiles += st["tiles"]
    print(f"soak ok: {runs} runs ({fast_runs} also with the fast kernels, {pre_runs} with a shared prelut), {px / 1e6:.0f} Mpx compared, {tiles} tiles "
          f"({tube_tiles} through the tube), {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
