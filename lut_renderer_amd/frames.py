"""Synthetic frames for tests and bench.py (the reference ships no media).

Two distributions, both seeded with `numpy.random.default_rng(0xC0BE + k)` for frame k:

* ``uniform``  -- i.i.d. codes over the legal range (Y 16..235, C 16..240 scaled to the bit
  depth; 0..max for full range / RGB).  Worst case for lattice locality: neighbouring
  pixels land in unrelated LUT cells.
* ``natural``  -- what graded camera footage looks like to the LUT: smooth luma / chroma
  gradients, eight soft-edged colour patches, and sensor-like noise (sigma = 2 codes at
  10 bit).  Neighbouring pixels share LUT cells, as they do in real video.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

SEED_BASE = 0xC0BE


def _ranges(depth: int, full_range: bool) -> Tuple[Tuple[int, int], Tuple[int, int]]:
    s = 1 << (depth - 8)
    m = (1 << depth) - 1
    if full_range:
        return (0, m), (0, m)
    return (16 * s, 235 * s), (16 * s, 240 * s)


def _dtype(depth: int):
    return np.uint8 if depth <= 8 else np.uint16


def chroma_shape(w: int, h: int, csx: int, csy: int) -> Tuple[int, int]:
    return (h + (1 << csy) - 1) >> csy, (w + (1 << csx) - 1) >> csx


def uniform_yuv(w: int, h: int, depth: int = 10, csx: int = 1, csy: int = 1, k: int = 0,
                full_range: bool = False) -> List[np.ndarray]:
    rng = np.random.default_rng(SEED_BASE + k)
    (ylo, yhi), (clo, chi) = _ranges(depth, full_range)
    ch, cw = chroma_shape(w, h, csx, csy)
    dt = _dtype(depth)
    return [rng.integers(ylo, yhi + 1, size=(h, w), dtype=np.int64).astype(dt),
            rng.integers(clo, chi + 1, size=(ch, cw), dtype=np.int64).astype(dt),
            rng.integers(clo, chi + 1, size=(ch, cw), dtype=np.int64).astype(dt)]


def _bilinear_field(rng, h: int, w: int, lo: float, hi: float) -> np.ndarray:
    c = rng.uniform(lo, hi, size=4)
    v = np.linspace(0.0, 1.0, h, dtype=np.float32)[:, None]
    u = np.linspace(0.0, 1.0, w, dtype=np.float32)[None, :]
    return ((1 - v) * ((1 - u) * c[0] + u * c[1]) + v * ((1 - u) * c[2] + u * c[3])).astype(np.float32)


def natural_yuv(w: int, h: int, depth: int = 10, csx: int = 1, csy: int = 1, k: int = 0,
                full_range: bool = False, noise_sigma_10bit: float = 2.0, chroma_gain: float = 1.0) -> List[np.ndarray]:
    rng = np.random.default_rng(SEED_BASE + k)
    (ylo, yhi), (clo, chi) = _ranges(depth, full_range)
    ch, cw = chroma_shape(w, h, csx, csy)
    s = float(1 << depth) / 1024.0                     # code scale relative to 10 bit
    mid = float(1 << (depth - 1))
    # normalised fields, luma in [0,1], chroma offsets in [-0.5, 0.5]
    y = _bilinear_field(rng, h, w, 0.12, 0.80)
    cb = _bilinear_field(rng, ch, cw, -0.06, 0.06)
    cr = _bilinear_field(rng, ch, cw, -0.06, 0.06)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    cyy, cxx = np.mgrid[0:ch, 0:cw].astype(np.float32)
    for _ in range(8):                                 # soft-edged colour patches
        px, py = rng.uniform(0.1, 0.9) * w, rng.uniform(0.1, 0.9) * h
        rx, ry = rng.uniform(0.04, 0.16) * w, rng.uniform(0.04, 0.16) * h
        dy, dcb, dcr = rng.uniform(-0.25, 0.25), rng.uniform(-0.12, 0.12), rng.uniform(-0.12, 0.12)
        edge = 0.08
        m = np.clip((1.0 - np.maximum(np.abs(xx - px) / rx, np.abs(yy - py) / ry)) / edge, 0.0, 1.0)
        y += dy * m
        mc = np.clip((1.0 - np.maximum(np.abs(cxx * (1 << csx) - px) / rx,
                                       np.abs(cyy * (1 << csy) - py) / ry)) / edge, 0.0, 1.0)
        cb += dcb * mc
        cr += dcr * mc
    cb *= chroma_gain
    cr *= chroma_gain
    sig = noise_sigma_10bit * s
    yc = ylo + np.clip(y, 0.0, 1.0) * (yhi - ylo) + rng.normal(0.0, sig, size=(h, w))
    cbc = mid + cb * (chi - clo) + rng.normal(0.0, sig, size=(ch, cw))
    crc = mid + cr * (chi - clo) + rng.normal(0.0, sig, size=(ch, cw))
    dt = _dtype(depth)
    return [np.clip(np.rint(yc), ylo, yhi).astype(dt),
            np.clip(np.rint(cbc), clo, chi).astype(dt),
            np.clip(np.rint(crc), clo, chi).astype(dt)]


def uniform_rgb(w: int, h: int, depth: int = 10, k: int = 0) -> List[np.ndarray]:
    """gbrp order (G, B, R), i.i.d. over the full code range."""
    rng = np.random.default_rng(SEED_BASE + k)
    m = (1 << depth) - 1
    return [rng.integers(0, m + 1, size=(h, w), dtype=np.int64).astype(_dtype(depth)) for _ in range(3)]


def natural_rgb(w: int, h: int, depth: int = 10, k: int = 0, noise_sigma_10bit: float = 2.0,
                chroma_gain: float = 1.0) -> List[np.ndarray]:
    """gbrp order (G, B, R): a grey-ish gradient with per-channel tint, patches and noise.  `chroma_gain` scales every
    pixel's distance from its own grey value (3 = the "vivid" frames)."""
    rng = np.random.default_rng(SEED_BASE + k)
    m = (1 << depth) - 1
    base = _bilinear_field(rng, h, w, 0.10, 0.85)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    chans = [base + _bilinear_field(rng, h, w, -0.05, 0.05) for _ in range(3)]
    for _ in range(8):
        px, py = rng.uniform(0.1, 0.9) * w, rng.uniform(0.1, 0.9) * h
        rx, ry = rng.uniform(0.04, 0.16) * w, rng.uniform(0.04, 0.16) * h
        mask = np.clip((1.0 - np.maximum(np.abs(xx - px) / rx, np.abs(yy - py) / ry)) / 0.08, 0.0, 1.0)
        for c in chans:
            c += rng.uniform(-0.2, 0.2) * mask
    if chroma_gain != 1.0:
        grey = (chans[0] + chans[1] + chans[2]) / 3.0
        chans = [grey + chroma_gain * (c - grey) for c in chans]
    sig = noise_sigma_10bit * float(1 << depth) / 1024.0
    return [np.clip(np.rint(np.clip(c, 0, 1) * m + rng.normal(0.0, sig, size=(h, w))), 0, m).astype(_dtype(depth))
            for c in chans]


def _noise_sigma(dist: str) -> float:
    """"noise<S>": the natural frame with luma AND per-sample chroma noise of sigma S codes (10-bit scale) instead of 2
    -- the range between clean footage and the uniform worst case (S = 16 is half a 33^3 lattice cell, 64 two cells)."""
    try:
        s = float(dist[5:])
    except ValueError:
        raise ValueError(f"unknown distribution '{dist}'") from None
    if not 0.0 <= s <= 1024.0:
        raise ValueError(f"unknown distribution '{dist}'")
    return s


def make_yuv(dist: str, w: int, h: int, depth: int, csx: int, csy: int, k: int = 0,
             full_range: bool = False) -> List[np.ndarray]:
    if dist == "uniform":
        return uniform_yuv(w, h, depth, csx, csy, k, full_range)
    if dist == "natural":
        return natural_yuv(w, h, depth, csx, csy, k, full_range)
    if dist == "vivid":        # the natural frame with three times the chroma: saturated fields and patches, far from the grey axis
        return natural_yuv(w, h, depth, csx, csy, k, full_range, chroma_gain=3.0)
    if dist.startswith("noise"):
        return natural_yuv(w, h, depth, csx, csy, k, full_range, noise_sigma_10bit=_noise_sigma(dist))
    raise ValueError(f"unknown distribution '{dist}'")


def make_rgb(dist: str, w: int, h: int, depth: int, k: int = 0) -> List[np.ndarray]:
    """gbrp order (G, B, R)."""
    if dist == "uniform":
        return uniform_rgb(w, h, depth, k)
    if dist == "natural":
        return natural_rgb(w, h, depth, k)
    if dist == "vivid":
        return natural_rgb(w, h, depth, k, chroma_gain=3.0)
    if dist.startswith("noise"):
        return natural_rgb(w, h, depth, k, noise_sigma_10bit=_noise_sigma(dist))
    raise ValueError(f"unknown distribution '{dist}'")
