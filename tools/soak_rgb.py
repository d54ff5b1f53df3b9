#!/usr/bin/env python3
"""tools/soak_rgb.py [seconds] -- GPU soak of the RGB tube kernels (lutr_rgb2.hip): tube kernel vs the scalar generic kernel, bit for bit.

The tube kernels vote on the cells their pixels touch and send the lanes that left the tube to the gather body; blue-first orders
run with permuted strides and swapped node channels; packed samples are picked out of dwords by SDWA selects.  This drives them
with frames whose G-R and B-G differences sweep from zero to far outside the tube in every direction (plus per-sample noise, so
tiles are mixed), luma over the whole code range, random lattice sizes (whole-lattice mode, 33^3, 65^3), DOMAIN scales equal and
per channel, lattices inside and outside [0, 1], every packed order and planar depth, three modes, batches and row shards, codes
above 2^depth - 1 in 16-bit containers -- and compares every sample with the generic kernel (GPU against GPU; the generic kernel
is pinned against the oracle by tests/)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
os.environ.setdefault("LUTR_SMALL_JOB_MPX", "0")
os.environ["LUTR_RGB2"] = "all"
from lut_renderer_amd import _native, cube  # noqa: E402
from lut_renderer_amd.engine import LutEngine  # noqa: E402


def rgb_frame(rng, w, h, depth, wild):
    m = (1 << depth) - 1
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    kind = rng.integers(0, 3)
    grey = rng.integers(0, m + 1, size=(h, w)).astype(np.float32) if kind == 0 else (
        np.linspace(0, m, w, dtype=np.float32)[None, :] + np.zeros((h, 1), np.float32) if kind == 1 else np.full((h, w), float(rng.integers(0, m + 1)), np.float32))
    theta = rng.uniform(0, 2 * np.pi) + 2 * np.pi * xx / w * rng.integers(1, 5)
    rad = (yy / h) * rng.uniform(0.05, 0.7) * m
    noise = rng.choice([0.0, 0.5, 2.0, 8.0]) * (m / 255.0)
    r = grey + rad * np.cos(theta) + rng.normal(0, noise, size=(h, w))
    g = grey + rng.normal(0, noise, size=(h, w))
    b = grey + rad * np.sin(theta) + rng.normal(0, noise, size=(h, w))
    dt = np.uint16 if depth > 8 else np.uint8
    out = [np.clip(np.rint(c), 0, m).astype(dt) for c in (r, g, b)]
    if wild and depth in (10, 12):          # container values the coordinate table does not cover
        for p in out:
            p[rng.integers(0, h, 16), rng.integers(0, w, 16)] = rng.integers(1 << depth, 65536, 16)
    return out


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    eng = LutEngine(0)
    rng = np.random.default_rng(20261005)
    t0, runs, px, tube, tiles = time.time(), 0, 0, 0, 0
    packed = list(_native.PACKED_FORMATS)
    while time.time() - t0 < budget:
        n = int(rng.choice([2, 9, 17, 22, 26, 33, 33, 33, 41, 65]))
        lat = rng.uniform(-0.2, 1.2, size=(n, n, n, 3)).astype(np.float32) if rng.random() < 0.25 else (
            rng.uniform(0.0, 1.0, size=(n, n, n, 3)).astype(np.float32) if rng.random() < 0.3 else cube.log709_lattice(n))
        u = rng.random()
        scale = np.ones(3, np.float32) if u < 0.6 else (np.full(3, rng.uniform(0.5, 1.0), np.float32) if u < 0.8
                                                         else rng.uniform(0.5, 1.0, 3).astype(np.float32))
        eng.set_lut(cube.CubeLut(n, scale, lat))
        mode = str(rng.choice(["tetrahedral", "trilinear", "nearest"]))
        h = int(rng.choice([24, 67, 136]))
        nf = int(rng.choice([1, 3]))
        kw = {}
        if rng.random() < 0.3:
            r0 = int(rng.integers(0, h // 2)); kw = dict(row0=r0, rows=int(rng.integers(1, h - r0 + 1)))
        if rng.random() < 0.5:
            depth = int(rng.choice([8, 10, 12, 16]))
            w = int(rng.choice([256, 512, 1920]))
            r, g, b = rgb_frame(rng, w, h, depth, wild=rng.random() < 0.3)
            dev = [torch.from_numpy(p.view(np.int16) if depth > 8 else p).to(eng.device).unsqueeze(0).repeat(nf, 1, 1).contiguous() for p in (g, b, r)]
            what = f"gbrp{depth}"
            run = lambda: [t.clone() for t in eng.apply_rgb(dev, depth=depth, interp=mode, **kw)]
        else:
            fmt = str(rng.choice(packed))
            bits, nc, ro, go, bo = _native.PACKED_FORMATS[fmt]
            w = int(rng.choice([256, 512, 1920]))
            r, g, b = rgb_frame(rng, w, h, bits, wild=False)
            img = rng.integers(0, 1 << bits, size=(h, w, nc)).astype(r.dtype)
            img[..., ro], img[..., go], img[..., bo] = r, g, b
            dev1 = torch.from_numpy(img.view(np.int16) if bits == 16 else img).to(eng.device).unsqueeze(0).repeat(nf, 1, 1, 1).contiguous()
            what = fmt
            run = lambda: [eng.apply_packed(dev1, pix_fmt=fmt, interp=mode, **kw).clone()]
        eng.set_variant("vec_lds")
        eng.tile_stats(True)
        a = run()
        st = eng.tile_stats(False)
        name = eng.last_kernel
        equal = bool(scale[0] == scale[1] == scale[2])
        deep = what in ("gbrp12", "gbrp16") or _native.PACKED_FORMATS.get(what, (8,))[0] == 16
        if not name.startswith("k_rgb_tube") and (equal or not deep):      # per-channel tables exist for 8- and 10-bit data only
            raise SystemExit(f"run {runs}: {what} {mode} n={n} took {name}, not a tube kernel")
        eng.set_variant("generic")
        b2 = run()
        r0, rn = kw.get("row0", 0), kw.get("rows", h)
        for i, (x, y) in enumerate(zip(a, b2)):
            x, y = x[:, r0:r0 + rn], y[:, r0:r0 + rn]
            if not torch.equal(x, y):
                d = (x.to(torch.int32) - y.to(torch.int32)).abs()
                raise SystemExit(f"MISMATCH run {runs}: {what} {mode} n={n} scale={scale} {kw} {name} plane {i}: {int((d > 0).sum())} samples, max {int(d.max())}")
        runs += 1
        px += nf * w * rn
        tube += st["tube_tiles"]
        tiles += st["tiles"]
    print(f"rgb soak ok: {runs} runs, {px / 1e6:.0f} Mpx compared, {tiles} tiles ({tube} wholly inside the tube), {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
