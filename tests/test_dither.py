"""Error-diffusion dither: the oracle's contract, on the CPU.

The reference only names the option (`zscale=dither=error_diffusion`, ffmpeg.py:305-307); zimg is not
vendored, so the arithmetic is this repo's own contract -- Floyd-Steinberg in zimg's order of
operations -- and is pinned here by known answers and by a second, pure-NumPy statement of the same loop.
"""
import numpy as np
import pytest

from lut_renderer_amd import cube, frames

F = np.float32


def _twin(x, maxv):
    """The same loop as oracle/lut3d_oracle.c orc_dither_plane, written independently with float32 scalars."""
    h, w = x.shape
    out = np.zeros((h, w), np.int64)
    top = np.zeros(w + 2, F)
    for y in range(h):
        cur = np.zeros(w + 2, F)
        left = F(0)
        for j in range(w):
            err = F(0)
            err = F(err + F(left * F(7.0 / 16.0)))
            err = F(err + F(top[j + 2] * F(3.0 / 16.0)))
            err = F(err + F(top[j + 1] * F(5.0 / 16.0)))
            err = F(err + F(top[j] * F(1.0 / 16.0)))
            v = F(x[y, j] + err)
            v = min(max(v, F(0)), F(maxv))
            q = F(np.rint(v))
            left = F(v - q)
            cur[j + 1] = left
            out[y, j] = int(q)
        top = cur
    return out


def test_integer_planes_pass_through_unchanged(orc):
    rng = np.random.default_rng(1)
    x = rng.integers(0, 256, size=(9, 31)).astype(F)
    assert np.array_equal(orc.dither_plane(x, 255.0, False), x.astype(np.uint8))


def test_constant_fraction_keeps_its_mean(orc):
    for val in (100.25, 17.5, 0.125, 254.9):
        q = orc.dither_plane(np.full((64, 128), val, F), 255.0, False).astype(np.float64)
        assert abs(q.mean() - val) < 0.02
        assert set(np.unique(q)) <= {np.floor(val), np.ceil(val)}


def test_first_pixels_known_answer(orc):
    # x = 0.5 everywhere: rint(0.5) = 0 (half to even), e = 0.5; next: 0.5 + 0.5*7/16 = 0.71875 -> 1, e = -0.28125; ...
    q = orc.dither_plane(np.full((1, 4), 0.5, F), 255.0, False)[0]
    assert q.tolist() == [0, 1, 0, 1]


def test_clips_to_the_code_range(orc):
    x = np.array([[-3.0, 300.0, 255.4, 0.4]], F)
    assert orc.dither_plane(x, 255.0, False)[0].tolist() == [0, 255, 255, 0]
    assert orc.dither_plane(np.array([[1023.7, 2000.0]], F), 1023.0, True)[0].tolist() == [1023, 1023]


@pytest.mark.parametrize("shape,maxv", [((7, 13), 255.0), ((20, 5), 1023.0), ((1, 40), 255.0), ((33, 1), 255.0)])
def test_numpy_twin_agrees(orc, shape, maxv):
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    x = (rng.random(shape) * (maxv + 8) - 4).astype(F)
    assert np.array_equal(orc.dither_plane(x, maxv, maxv > 255).astype(np.int64), _twin(x, maxv))


def test_dithered_yuv_path_differs_from_plain_only_by_one_code_and_keeps_the_mean(orc):
    lut = cube.log709_lattice(17)
    scale = np.ones(3, F)
    src = frames.natural_yuv(64, 36, 10, 1, 1, k=3)
    k = orc.yuv_constants(din=10, dl=10, dout=8)
    plain = orc.apply_yuv(lut, scale, "tetrahedral", k, 10, 10, 8, 1, 1, src)
    dith = orc.apply_yuv(lut, scale, "tetrahedral", k, 10, 10, 8, 1, 1, src, dither="error_diffusion")
    for a, b in zip(plain, dith):
        assert a.dtype == b.dtype == np.uint8
        assert np.abs(a.astype(np.int32) - b.astype(np.int32)).max() <= 1
    assert any(not np.array_equal(a, b) for a, b in zip(plain, dith))
    assert abs(plain[0].astype(np.float64).mean() - dith[0].astype(np.float64).mean()) < 0.6
