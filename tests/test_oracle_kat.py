"""Analytic known-answer tests that pin the CPU oracle.

The reference holds no pixel-level golden vectors and no ffmpeg binary exists in this
image, so these closed-form cases -- plus the NumPy twin, written separately in FFmpeg's
branchy form -- are what the oracle is pinned by ("parity unpinned" against a live ffmpeg;
tests/test_ffmpeg_live.py activates if one ever appears).
Follows FFmpeg vf_lut3d.c semantics for the filter emitted at
/root/reference/src/lut_renderer/ffmpeg.py:246.
"""
import itertools

import numpy as np
import pytest

from lut_renderer_amd import cube, frames
from oracle import lut3d_numpy as npo

ONE = np.ones(3, dtype=np.float32)


def _grid_codes(depth, step):
    m = (1 << depth) - 1
    v = np.arange(0, m + 1, step, dtype=np.int64)
    if v[-1] != m:
        v = np.append(v, m)
    return v


@pytest.mark.parametrize("depth", [8, 10, 12])
@pytest.mark.parametrize("n", [2, 17, 33, 65])
def test_identity_returns_src_or_src_minus_one(orc, depth, n):
    """A.6 #1: truncation (not rounding) means identity may drop a code by one, never more."""
    lat = cube.identity_lattice(n)
    m = (1 << depth) - 1
    codes = _grid_codes(depth, max(1, m // 257))
    dt = np.uint8 if depth == 8 else np.uint16
    plane = np.tile(codes.astype(dt), (4, 1))
    for mode in ("trilinear", "tetrahedral"):
        out = orc.apply_rgb(lat, ONE, depth, mode, [plane, plane, plane])
        for o in out:
            d = plane.astype(np.int64) - o.astype(np.int64)
            assert d.min() >= 0 and d.max() <= 1, (mode, d.min(), d.max())
    assert orc.apply_pixel(lat, ONE, depth, "tetrahedral", (m, m, m)) == (m, m, m)   # A.6 #6: last node exactly
    assert orc.apply_pixel(lat, ONE, depth, "trilinear", (0, 0, 0)) == (0, 0, 0)


def test_constant_lattice_truncates(orc):
    """A.6 #2: every pixel -> (int)(a*M), both modes, truncation toward zero."""
    for a, b, c in ((0.25, 0.5, 0.999), (0.1, 0.7, 0.3333333)):
        lat = np.empty((5, 5, 5, 3), dtype=np.float32)
        lat[...] = (a, b, c)
        for depth in (8, 10, 16):
            m = (1 << depth) - 1
            want = tuple(int(np.float32(v) * np.float32(m)) for v in (a, b, c))
            for mode in ("nearest", "trilinear", "tetrahedral", "pyramid", "prism"):
                for px in ((0, 0, 0), (m, m, m), (m // 3, m // 2, m // 5), (1, m - 1, 7)):
                    got = orc.apply_pixel(lat, ONE, depth, mode, px)
                    # the blend of equal values can differ from the value by an ulp before truncation
                    assert all(abs(g - w) <= 1 for g, w in zip(got, want)), (mode, px, got, want)
            assert orc.apply_pixel(lat, ONE, depth, "nearest", (3, 4, 5)) == want


def test_affine_lattice_tri_equals_tetra(orc):
    """A.6 #3: a lattice affine in its coordinates is reproduced by both modes (+-1 LSB)."""
    n, depth = 9, 10
    m = (1 << depth) - 1
    A = np.array([[0.5, 0.2, 0.1], [0.1, 0.6, 0.2], [0.05, 0.15, 0.7]])
    t = np.array([0.02, 0.03, 0.01])
    lat = (cube.identity_lattice(n).astype(np.float64) @ A.T + t).astype(np.float32)
    rng = np.random.default_rng(3)
    for _ in range(300):
        px = tuple(int(v) for v in rng.integers(0, m + 1, 3))
        want = (A @ (np.array(px) / m) + t) * m
        for mode in ("trilinear", "tetrahedral", "pyramid", "prism"):
            got = orc.apply_pixel(lat, ONE, depth, mode, px)
            assert np.all(np.abs(np.array(got) - np.floor(want)) <= 1), (mode, px, got, want)


def test_six_tetrahedra_and_ties(orc):
    """A.6 #4: N=2 lattice with distinct corners; each ordering of (d.r,d.g,d.b) picks its branch,
    weights sum to 1, ties fall to the `else` side (same value either way)."""
    rng = np.random.default_rng(11)
    lat = rng.uniform(0, 1, size=(2, 2, 2, 3)).astype(np.float32)
    depth, m = 16, 65535
    vals = [0.15, 0.45, 0.8]
    f32 = np.float32
    for perm in itertools.permutations(range(3)):
        d = [0.0] * 3
        for rank, ch in enumerate(perm):
            d[ch] = vals[rank]
        px = tuple(int(round(v * m)) for v in d)
        got = orc.apply_pixel(lat, ONE, depth, "tetrahedral", px)
        # independent evaluation: walk from c000 to c111 along axes in descending-d order
        dd = [f32(f32(f32(p) * (f32(1.0) / f32(m))) * f32(1.0)) for p in px]
        order = sorted(range(3), key=lambda c: -dd[c])
        corner = [0, 0, 0]
        pts = [tuple(corner)]
        for c in order:
            corner[c] = 1
            pts.append(tuple(corner))
        ws = [f32(1) - dd[order[0]], dd[order[0]] - dd[order[1]], dd[order[1]] - dd[order[2]], dd[order[2]]]
        assert abs(float(sum(ws)) - 1.0) < 1e-6
        acc = np.zeros(3, dtype=np.float32)
        for w, p in zip(ws, pts):
            acc = (acc + f32(w) * lat[p]).astype(np.float32)
        want = tuple(int(v) for v in (acc * f32(m)))
        assert got == want, (perm, got, want)
    # ties: all three equal -> on the main diagonal, lerp(c000, c111, d)  (A.6 #5)
    for code in (0, 1000, 32768, 65535):
        got = orc.apply_pixel(lat, ONE, depth, "tetrahedral", (code, code, code))
        dd = f32(f32(code) * (f32(1.0) / f32(m)))
        acc = (((f32(1) - dd) * lat[0, 0, 0] + f32(0) * lat[0, 1, 0]).astype(np.float32)
               + f32(0) * lat[1, 1, 0]).astype(np.float32)
        acc = (acc + dd * lat[1, 1, 1]).astype(np.float32)
        assert got == tuple(int(v) for v in (acc * f32(m)))


def test_trilinear_cell_centre_is_corner_mean(orc):
    """A.6 #5: at the centre of the single cell of an N=2 lattice."""
    rng = np.random.default_rng(5)
    lat = rng.uniform(0, 1, size=(2, 2, 2, 3)).astype(np.float32)
    depth, m = 16, 65535
    # code m/2 is not exactly 0.5; use the oracle's own coordinate to form the expectation
    px = (32768, 32768, 32768)
    got = np.array(orc.apply_pixel(lat, ONE, depth, "trilinear", px))
    mean = lat.reshape(8, 3).astype(np.float64).mean(axis=0) * m
    assert np.all(np.abs(got - mean) <= 2)


def test_clamp_and_domain_scale(orc, tmp_path):
    """A.6 #6: DOMAIN_MAX 2 -> scale .5 addresses the lower half only; DOMAIN_MAX .5 -> 1/.5 clipped to 1."""
    lat = cube.log709_lattice(17)
    p2 = cube.write_cube(tmp_path / "d2.cube", lat, domain_min=(0, 0, 0), domain_max=(2, 2, 2))
    n, sc, tab = orc.parse_cube(p2)
    assert np.allclose(sc, 0.5)
    m = 1023
    top = orc.apply_pixel(tab, sc, 10, "trilinear", (m, m, m))
    # s = clip(1.0 * 0.5 * 16) = 8 -> exactly node (8,8,8)
    assert top == tuple(int(np.float32(v) * np.float32(m)) for v in tab[8, 8, 8])
    ph = cube.write_cube(tmp_path / "dh.cube", lat, domain_min=(0, 0, 0), domain_max=(0.5, 0.5, 0.5))
    _, sc2, _ = orc.parse_cube(ph)
    assert np.all(sc2 == 1.0)


def test_out_of_range_lattice_values_clip(orc):
    """A.6 #8: values < 0 -> 0, > 1 -> max, after truncation."""
    lat = np.empty((3, 3, 3, 3), dtype=np.float32)
    lat[...] = (-0.25, 1.5, 0.5)
    for depth in (8, 10):
        m = (1 << depth) - 1
        for mode in ("nearest", "trilinear", "tetrahedral"):
            assert orc.apply_pixel(lat, ONE, depth, mode, (5, 6, 7)) == (0, m, int(np.float32(0.5) * np.float32(m)))


@pytest.mark.parametrize("depth", [8, 10, 16])
def test_numpy_twin_agrees_with_c_oracle(orc, depth):
    """Two separately written restatements (C sorted-free branchy form vs NumPy) agree bit for bit."""
    rng = np.random.default_rng(depth)
    lat = rng.uniform(-0.1, 1.1, size=(7, 7, 7, 3)).astype(np.float32)
    sc = np.array([1.0, 0.75, 0.5], dtype=np.float32)
    planes = frames.uniform_rgb(96, 33, depth, k=depth)
    for mode in ("nearest", "trilinear", "tetrahedral"):
        a = orc.apply_rgb(lat, sc, depth, mode, planes)
        b = npo.apply_rgb(lat, sc, depth, mode, planes)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), mode
    # threaded slices == scalar
    c = orc.apply_rgb(lat, sc, depth, "tetrahedral", planes, nthreads=5)
    for x, y in zip(orc.apply_rgb(lat, sc, depth, "tetrahedral", planes), c):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("fmt", [(8, 1, 1), (10, 1, 1), (10, 1, 0), (10, 0, 0), (12, 1, 1)])
def test_yuv_contract_numpy_twin(orc, fmt):
    depth, csx, csy = fmt
    lat = cube.log709_lattice(17)
    for matrix, rin, rout in (("bt709", "tv", "tv"), ("bt2020nc", "pc", "tv"), ("smpte170m", "tv", "pc")):
        k = orc.yuv_constants(matrix, rin, matrix, rout, depth, None, None, 1 << (csx + csy))
        for w, h in ((64, 36), (33, 17)):
            src = frames.uniform_yuv(w, h, depth, csx, csy, k=1, full_range=(rin == "pc"))
            for mode in ("trilinear", "tetrahedral"):
                a = orc.apply_yuv(lat, ONE, mode, k, depth, depth, depth, csx, csy, src)
                b = npo.apply_yuv(lat, ONE, mode, k, depth, depth, depth, csx, csy, src)
                for x, y in zip(a, b):
                    assert np.array_equal(x, y), (fmt, matrix, mode, w, h)


def test_yuv_identity_round_trip(orc):
    """Identity lattice through YUV->RGB->lut3d->RGB->YUV stays within 1 code (10 bit) of the input."""
    lat = cube.identity_lattice(33)
    rng = np.random.default_rng(21)
    for depth, tol in ((8, 1), (10, 2)):
        k = orc.yuv_constants("bt709", "tv", "bt709", "tv", depth, None, None, 4)
        s = 1 << (depth - 8)
        dt = np.uint8 if depth == 8 else np.uint16
        # low-saturation content: stays inside the RGB gamut, so no clip breaks the round trip;
        # chroma is constant per 2x2 block by construction (4:2:0), luma varies freely
        src = [rng.integers(40 * s, 200 * s, size=(72, 128)).astype(dt),
               rng.integers(118 * s, 138 * s, size=(36, 64)).astype(dt),
               rng.integers(118 * s, 138 * s, size=(36, 64)).astype(dt)]
        out = orc.apply_yuv(lat, ONE, "tetrahedral", k, depth, depth, depth, 1, 1, src)
        for a, b in zip(out, src):
            assert np.abs(a.astype(np.int64) - b.astype(np.int64)).max() <= tol


def test_prologue_full_to_limited(orc):
    """scale=in_range=pc:out_range=tv,format=yuv420p (ffmpeg.py:225-233): 0->16, 255->235, 128->128."""
    k = orc.yuv_constants("bt709", "tv", "bt709", "tv", 8, 8, 8, 4, prologue=True)
    f = np.float32
    for y, want in ((0, 16), (255, 235), (128, 126)):
        assert int(np.floor(f(k.py) * f(y) + f(k.pyb))) == want
    for c, want in ((128, 128), (0, 16), (255, 240)):
        assert int(np.floor(f(k.pc) * f(c) + f(k.pcb))) == want
    k10 = orc.yuv_constants("bt709", "tv", "bt709", "tv", 10, 8, 8, 4, prologue=True)
    assert int(np.floor(f(k10.py) * f(1023) + f(k10.pyb))) == 235
    assert int(np.floor(f(k10.pc) * f(512) + f(k10.pcb))) == 128


@pytest.mark.parametrize("pix_fmt", ["rgb24", "bgra", "argb", "0bgr", "bgr48le", "rgba64le"])
def test_packed_oracle_is_the_pixel_function_through_the_rgba_map(orc, pix_fmt):
    """Packed formats: same arithmetic as planar at depth 8 / 16, components picked by the
    format's rgba map, fourth component copied."""
    bits, nc, ro, go, bo = orc.PACKED[pix_fmt]
    rng = np.random.default_rng(11)
    table = rng.random((5, 5, 5, 3), dtype=np.float32)
    scale = np.ones(3, np.float32)
    img = rng.integers(0, 1 << bits, size=(3, 7, nc), dtype=np.uint16 if bits == 16 else np.uint8)
    for mode in ("nearest", "trilinear", "tetrahedral", "pyramid", "prism"):
        out = orc.apply_packed(table, scale, pix_fmt, mode, img)
        for y, x in itertools.product(range(3), range(7)):
            px = img[y, x]
            r, g, b = orc.apply_pixel(table, scale, bits, mode, (px[ro], px[go], px[bo]))
            assert (out[y, x, ro], out[y, x, go], out[y, x, bo]) == (r, g, b)
            if nc == 4:
                ao = 6 - ro - go - bo
                assert out[y, x, ao] == px[ao]


def test_near_adds_a_double_half(orc, tmp_path):
    """vf_lut3d.c: NEAR(x) = (int)(x + .5) with a DOUBLE .5.  At 16 bit, N = 2, DOMAIN_MAX 1.243 the code 40730 lands on
    s = 0.49999997 (the float just below 1/2): the exact double sum truncates to node 0; a float sum would round up to
    1.0 and pick node 1.  The oracle, its NumPy twin and (on the GPU box) the kernels must say node 0."""
    lat = np.zeros((2, 2, 2, 3), dtype=np.float32)
    lat[1, :, :, 0] = 1.0                                  # red output = index of the red node chosen
    p = cube.write_cube(tmp_path / "near.cube", lat, domain_min=(0, 0, 0), domain_max=(1.243, 1.243, 1.243))
    lut = cube.read_cube(p)
    s = np.float32(np.float32(40730) * np.float32(np.float32(1.0) / np.float32(65535))) * np.float32(lut.scale[0] * np.float32(1))
    assert s == np.nextafter(np.float32(0.5), np.float32(0))
    assert orc.apply_pixel(lut.table, lut.scale, 16, "nearest", (40730, 0, 0))[0] == 0
    assert orc.apply_pixel(lut.table, lut.scale, 16, "nearest", (40731, 0, 0))[0] == 65535
    got = npo.apply_rgb(lut.table, lut.scale, 16, "nearest", [np.array([[0]], np.uint16), np.array([[0]], np.uint16),
                                                              np.array([[40730]], np.uint16)])
    assert int(got[2][0, 0]) == 0


def test_domain_scale_subtracts_in_float(orc, tmp_path):
    """vf_lut3d.c parse_cube: scale = av_clipf(1. / (max - min), 0, 1) with max and min FLOATS, so the subtraction
    rounds to float before the double division.  DOMAIN 0.3..1.9 tells the two apart: 0.62500006 (float
    subtraction) vs 0.625 (double subtraction)."""
    p = tmp_path / "dom.cube"
    p.write_text("LUT_3D_SIZE 2\nDOMAIN_MIN 0.3 0.2 0.1\nDOMAIN_MAX 1.9 2.3 1.3\n" +
                 "".join("%d %d %d\n" % (i & 1, (i >> 1) & 1, (i >> 2) & 1) for i in range(8)))
    want = np.array([np.float32(1.0 / float(np.float32(hi) - np.float32(lo))) for lo, hi in ((0.3, 1.9), (0.2, 2.3), (0.1, 1.3))],
                    dtype=np.float32)
    assert want[0] == np.float32(0.62500006) and want[0] != np.float32(0.625)
    lut = cube.read_cube(p)                               # liblutr's parser
    n, sc, _ = orc.parse_cube(p)
    n3, sc3, _ = npo.parse_cube_text(p.read_text())
    for got in (lut.scale, sc, sc3):
        assert np.array_equal(np.asarray(got, dtype=np.float32), want), (got, want)
