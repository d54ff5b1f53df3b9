"""Live comparison with a real `ffmpeg` binary -- the only way the FFmpeg-recall in SURVEY.md
Appendix A (and so the oracle) ever gets verified.  Skipped unless `ffmpeg` is on PATH; no
ffmpeg exists in the build image or on the GPU box of round 1, hence "parity unpinned".

When it runs: ffmpeg -f rawvideo -pix_fmt gbrp10le ... -vf lut3d=file=X.cube:interp=M must match
the oracle within the north star's tolerance (<=1 LSB at 8 bit, <=2 LSB at 10 bit), for
tetrahedral and trilinear separately (x86 builds use AVX2/FMA for tetrahedral planar, so exact
equality with the scalar C order is not expected).
"""
import shutil
import subprocess

import numpy as np
import pytest

from lut_renderer_amd import cube, frames

FFMPEG = shutil.which("ffmpeg")
pytestmark = pytest.mark.skipif(FFMPEG is None, reason="no ffmpeg binary on PATH (parity unpinned)")


@pytest.mark.parametrize("pix_fmt,depth,tol", [("gbrp", 8, 1), ("gbrp10le", 10, 2)])
@pytest.mark.parametrize("mode", ["trilinear", "tetrahedral", "nearest"])
def test_lut3d_rgb_matches_live_ffmpeg(orc, tmp_path, pix_fmt, depth, tol, mode):
    w, h = 128, 72
    lat = cube.log709_lattice(33)
    path = cube.write_cube(tmp_path / "look.cube", lat)
    n, sc, tab = orc.parse_cube(path)
    src = frames.uniform_rgb(w, h, depth, k=1)
    raw = b"".join(p.tobytes() for p in src)          # gbrp plane order: G, B, R
    out = subprocess.run([FFMPEG, "-hide_banner", "-loglevel", "error", "-f", "rawvideo", "-pix_fmt", pix_fmt,
                          "-s", f"{w}x{h}", "-i", "-", "-vf", f"lut3d=file='{path}':interp={mode}",
                          "-f", "rawvideo", "-pix_fmt", pix_fmt, "-"], input=raw, capture_output=True, check=True).stdout
    dt = src[0].dtype
    got = np.frombuffer(out, dtype=dt).reshape(3, h, w)
    want = orc.apply_rgb(tab, sc, depth, mode, src)
    for g, wv in zip(got, want):
        assert np.abs(g.astype(np.int64) - wv.astype(np.int64)).max() <= tol
