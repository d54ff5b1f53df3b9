"""Live comparison with a real `ffmpeg` binary -- the only way the FFmpeg-recall in SURVEY.md Appendix A (and so the
oracle) ever gets verified.  Skipped unless `ffmpeg` is on PATH; no ffmpeg exists in the build image or on the GPU boxes of
rounds 1-3 (`which ffmpeg ffprobe` is probed by this file's collection every run), hence "parity unpinned".

ONE run with a binary pins everything the engine restates:

  * `lut3d` on planar RGB (`gbrp`, `gbrp10le`, `gbrp12le`, `gbrp16le`) for nearest / trilinear / tetrahedral / pyramid / prism,
    ASSERTED at north_star's tolerance (<= 1 code at 8 bit, <= 2 at 10 bit and above; x86 builds run AVX2 / FMA for planar
    tetrahedral and trilinear, so equality with the scalar C order is not expected -- the number of differing samples and the
    maximum are printed),
  * `DOMAIN_MIN / DOMAIN_MAX` cubes, and the other file formats lut3d reads: `.3dl`, `.dat`, `.m3d`, `.csp` with ranges and
    `.csp` WITH a pre-LUT (asserted, same tolerance),
  * the whole reference chain on `yuv420p10le` / `yuv420p` / `yuv422p10le` sources as `build_command` emits it -- REPORT ONLY
    for the swscale-dependent part (the engine's YUV contract is its own): max |delta| and the share of samples
    within 1 / 2 / 4 codes are printed and written to gpurun_out/ffmpeg_live_report.json, with a loose sanity bound,
  * with a GPU as well: the STRICT and the FAST kernels of liblutr against the live output (RGB asserted at the tolerance;
    for the fused YUV path FAST is asserted against live lut3d applied between the engine's own YUV stages).
"""
import json
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from lut_renderer_amd import cube, frames

FFMPEG = shutil.which("ffmpeg")
FFPROBE = shutil.which("ffprobe")
pytestmark = pytest.mark.skipif(FFMPEG is None, reason="no ffmpeg binary on PATH (parity unpinned)")
ROOT = Path(__file__).resolve().parent.parent
REPORT = {}


def _tol(depth):
    return 1 if depth <= 8 else 2


def _ffmpeg_filter(raw: bytes, pix_in: str, w: int, h: int, vf: str, pix_out: str) -> bytes:
    cmd = [FFMPEG, "-hide_banner", "-loglevel", "error", "-f", "rawvideo", "-pix_fmt", pix_in, "-s", f"{w}x{h}", "-i", "-",
           "-vf", vf, "-f", "rawvideo", "-pix_fmt", pix_out, "-"]
    return subprocess.run(cmd, input=raw, capture_output=True, check=True).stdout


def _ffmpeg_lut3d_rgb(path, mode, planes, pix_fmt):
    """planes: gbrp order (G, B, R) arrays -> ffmpeg's lut3d output in the same layout."""
    h, w = planes[0].shape
    raw = b"".join(np.ascontiguousarray(p).tobytes() for p in planes)
    out = _ffmpeg_filter(raw, pix_fmt, w, h, f"lut3d=file='{path}':interp={mode}", pix_fmt)
    return list(np.frombuffer(out, dtype=planes[0].dtype).reshape(3, h, w))


def _compare(got, want, tol, what, assert_it=True):
    worst, ndiff, total = 0, 0, 0
    for g, wv in zip(got, want):
        d = np.abs(g.astype(np.int64) - wv.astype(np.int64))
        worst, ndiff, total = max(worst, int(d.max())), ndiff + int((d > 0).sum()), total + d.size
    REPORT[what] = {"max_abs": worst, "differing": ndiff, "samples": total, "tolerance": tol, "asserted": assert_it}
    print(f"[ffmpeg live] {what}: max |d| = {worst}, {ndiff} of {total} samples differ (tolerance {tol})")
    if assert_it:
        assert worst <= tol, what
    return worst


@pytest.fixture(scope="module", autouse=True)
def _write_report():
    yield
    out = ROOT / "gpurun_out"
    if REPORT and out.is_dir():
        ver = subprocess.run([FFMPEG, "-version"], capture_output=True, text=True).stdout.splitlines()[:1]
        (out / "ffmpeg_live_report.json").write_text(json.dumps({"ffmpeg": ver, "ffprobe": FFPROBE, "cases": REPORT}, indent=1))


# ------------------------------------------------------------------ lut3d on planar RGB: every mode, every depth
@pytest.mark.parametrize("pix_fmt,depth", [("gbrp", 8), ("gbrp10le", 10), ("gbrp12le", 12), ("gbrp16le", 16)])
@pytest.mark.parametrize("mode", ["nearest", "trilinear", "tetrahedral", "pyramid", "prism"])
def test_lut3d_rgb_matches_live_ffmpeg(orc, tmp_path, pix_fmt, depth, mode):
    w, h = 128, 72
    path = cube.write_cube(tmp_path / "look.cube", cube.log709_lattice(33))
    n, sc, tab = orc.parse_cube(path)
    for name, src in (("uniform", frames.uniform_rgb(w, h, depth, k=1)), ("natural", frames.natural_rgb(w, h, depth, k=2))):
        got = _ffmpeg_lut3d_rgb(path, mode, src, pix_fmt)
        want = orc.apply_rgb(tab, sc, depth, mode, src)
        _compare(got, want, _tol(depth), f"oracle vs ffmpeg lut3d {pix_fmt} {mode} {name}")


@pytest.mark.parametrize("mode", ["trilinear", "tetrahedral"])
def test_lut3d_lattice_sizes_and_domains(orc, tmp_path, mode):
    """2^3 ... 65^3 lattices, values outside [0, 1], DOMAIN_MIN / DOMAIN_MAX on every channel."""
    w, h, depth = 128, 72, 10
    rng = np.random.default_rng(3)
    src = frames.uniform_rgb(w, h, depth, k=3)
    cases = [("n2", rng.uniform(0, 1, (2, 2, 2, 3)), {}), ("n9_wide", rng.uniform(-0.2, 1.2, (9, 9, 9, 3)), {}),
             ("n17", cube.log709_lattice(17), {}), ("n65", cube.log709_lattice(65), {}),
             ("domain_0_2", cube.log709_lattice(17), dict(domain_min=(0, 0, 0), domain_max=(2, 2, 2))),
             ("domain_mixed", cube.log709_lattice(9), dict(domain_min=(0.1, 0, 0.25), domain_max=(1.0, 1.9, 1.25)))]
    for name, lat, kw in cases:
        path = cube.write_cube(tmp_path / f"{name}.cube", np.asarray(lat, dtype=np.float32), **kw)
        n, sc, tab = orc.parse_cube(path)
        got = _ffmpeg_lut3d_rgb(path, mode, src, "gbrp10le")
        _compare(got, orc.apply_rgb(tab, sc, depth, mode, src), 2, f"oracle vs ffmpeg lut3d {name} {mode}")


# ------------------------------------------------------------------ the other LUT file formats, prelut included
def _rows_blue_fastest(tab):
    n = tab.shape[0]
    return [tab[r, g, b] for r in range(n) for g in range(n) for b in range(n)]


def _rows_red_fastest(tab):
    n = tab.shape[0]
    return [tab[r, g, b] for b in range(n) for g in range(n) for r in range(n)]


def _write_formats(tmp_path):
    rng = np.random.default_rng(8)
    out = {}
    codes = rng.integers(0, 4096, size=(17, 17, 17, 3))
    p = tmp_path / "lustre.3dl"
    p.write_text(" ".join(str(min(64 * i, 1023)) for i in range(17)) + "\n" +
                 "".join("%d %d %d\n" % tuple(v) for v in _rows_blue_fastest(codes)))
    out["3dl"] = p
    tab = rng.random((9, 9, 9, 3)).astype(np.float32)
    p = tmp_path / "resolve.dat"
    p.write_text("3DLUTSIZE 9\n" + "".join("%.6f %.6f %.6f\n" % tuple(v) for v in _rows_red_fastest(tab)))
    out["dat"] = p
    vals = rng.integers(0, 1024, size=(8, 8, 8, 3))
    p = tmp_path / "pandora.m3d"
    p.write_text("name x\nin 512\nout 1024\nformat lut\nvalues\tred\tgreen\tblue\n" +
                 "".join("%d %d %d\n" % tuple(v) for v in _rows_blue_fastest(vals)))
    out["m3d"] = p
    p = tmp_path / "ranges.csp"
    p.write_text("CSPLUTV100\n3D\n\n2\n0.0 2.0\n0.0 1.0\n\n2\n0.0 1.0\n0.0 0.5\n\n2\n0.0 4.0\n0.25 1.0\n\n9 9 9\n" +
                 "".join("%.6f %.6f %.6f\n" % tuple(v) for v in _rows_red_fastest(tab)))
    out["csp_ranges"] = p
    p = tmp_path / "shaper.csp"
    xs = [np.array([0.0, 0.1, 0.25, 0.5, 0.75, 1.0]), np.linspace(0.0, 1.0, 11), np.array([0.0, 0.3, 0.6, 1.0])]
    ys = [np.array([0.0, 0.3, 0.5, 0.7, 0.85, 1.0]), np.linspace(0.0, 1.0, 11) ** 0.5, np.array([0.0, 0.2, 0.6, 1.0])]
    with open(p, "w") as f:
        f.write("CSPLUTV100\n3D\n\n")
        for x, y in zip(xs, ys):
            f.write("%d\n%s\n%s\n" % (len(x), " ".join("%.6f" % v for v in x), " ".join("%.6f" % v for v in y)))
        f.write("\n9 9 9\n" + "".join("%.6f %.6f %.6f\n" % tuple(v) for v in _rows_red_fastest(tab)))
    out["csp_prelut"] = p
    return out


@pytest.mark.parametrize("mode", ["trilinear", "tetrahedral"])
def test_other_lut_formats_match_live_ffmpeg(orc, tmp_path, mode):
    w, h, depth = 128, 72, 10
    src = frames.uniform_rgb(w, h, depth, k=4)
    for name, path in _write_formats(tmp_path).items():
        n, sc, tab, pre = orc.parse_lut_file_ex(path)
        got = _ffmpeg_lut3d_rgb(path, mode, src, "gbrp10le")
        want = orc.apply_rgb(tab, sc, depth, mode, src, prelut=pre)
        _compare(got, want, 2, f"oracle vs ffmpeg lut3d file format {name} {mode}")


# ------------------------------------------------------------------ the reference's whole chain on YUV sources (report only)
@pytest.mark.parametrize("pix_fmt,depth,cs,matrix", [("yuv420p10le", 10, (1, 1), "bt709"), ("yuv420p", 8, (1, 1), "bt709"),
                                                     ("yuv422p10le", 10, (1, 0), "bt2020nc"), ("yuv444p10le", 10, (0, 0), "bt709")])
@pytest.mark.parametrize("mode", ["trilinear", "tetrahedral"])
def test_full_chain_on_yuv_sources_report_only(orc, tmp_path, pix_fmt, depth, cs, matrix, mode):
    """`scale=in_color_matrix=M:out_color_matrix=M , lut3d , format=<pix_fmt>` as build_command emits it for a tagged source
    (ffmpeg.py:195-247): swscale's fixed-point YUV<->RGB and chroma filters are NOT what the engine's contract restates, so this
    only reports how far apart the two chains are (and bounds it loosely: a matrix or range mistake shows up as tens of codes)."""
    w, h = 128, 72
    path = cube.write_cube(tmp_path / "look.cube", cube.log709_lattice(33))
    n, sc, tab = orc.parse_cube(path)
    src = frames.natural_yuv(w, h, depth, cs[0], cs[1], k=5)
    raw = b"".join(np.ascontiguousarray(p).tobytes() for p in src)
    vf = f"scale=in_color_matrix={matrix}:out_color_matrix={matrix},lut3d=file='{path}':interp={mode},format={pix_fmt}"
    out = _ffmpeg_filter(raw, pix_fmt, w, h, vf, pix_fmt)
    dt = src[0].dtype
    sizes = [p.size for p in src]
    flat = np.frombuffer(out, dtype=dt)
    got = [flat[sum(sizes[:i]):sum(sizes[:i + 1])].reshape(src[i].shape) for i in range(3)]
    k = orc.yuv_constants(matrix, "tv", matrix, "tv", depth, depth, depth, 1 << sum(cs))
    want = orc.apply_yuv(tab, sc, mode, k, depth, depth, depth, cs[0], cs[1], src)
    worst = _compare(got, want, None, f"engine chain vs ffmpeg chain {pix_fmt} {matrix} {mode}", assert_it=False)
    d = np.concatenate([np.abs(g.astype(np.int64) - x.astype(np.int64)).reshape(-1) for g, x in zip(got, want)])
    REPORT[f"engine chain vs ffmpeg chain {pix_fmt} {matrix} {mode}"].update(
        {f"within_{t}": round(float((d <= t).mean()), 4) for t in (1, 2, 4)})
    assert worst <= (16 if depth > 8 else 6), "the two chains disagree by more than any rounding scheme explains"


# ------------------------------------------------------------------ the HIP kernels against the live binary
@pytest.mark.gpu
@pytest.mark.parametrize("pix_fmt,depth", [("gbrp", 8), ("gbrp10le", 10)])
@pytest.mark.parametrize("mode", ["nearest", "trilinear", "tetrahedral", "pyramid", "prism"])
def test_gpu_rgb_kernels_against_live_ffmpeg(engine, tmp_path, pix_fmt, depth, mode):
    import torch
    w, h = 512, 128
    path = cube.write_cube(tmp_path / "look.cube", cube.log709_lattice(33))
    engine.set_lut(cube.read_cube(path))
    src = frames.natural_rgb(w, h, depth, k=6)
    want = _ffmpeg_lut3d_rgb(path, mode, src, pix_fmt)
    dev = [torch.from_numpy(p.view(np.int16) if p.dtype == np.uint16 else p).to(engine.device) for p in src]
    for variant in ("auto", "generic"):
        engine.set_variant(variant)
        out = engine.apply_rgb(dev, depth=depth, interp=mode)
        got = [t.cpu().numpy().view(np.uint16) if depth > 8 else t.cpu().numpy() for t in out]
        _compare(got, want, _tol(depth), f"liblutr {engine.last_kernel} vs ffmpeg lut3d {pix_fmt} {mode}")
    engine.set_variant("auto")


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["strict", "fast"])
@pytest.mark.parametrize("mode", ["trilinear", "tetrahedral"])
def test_gpu_fused_yuv_kernels_against_live_lut3d_between_the_engines_own_yuv_stages(engine, orc, tmp_path, precision, mode):
    """The fused YUV kernels (strict and FAST) against LIVE lut3d: the engine's YUV -> RGB stage (identity LUT through the oracle
    gives the integer RGB it hands to lut3d), ffmpeg's lut3d on those RGB planes, the engine's RGB -> YUV stage.  Asserted at
    north_star's tolerance propagated through the output matrix (a 2-code RGB difference moves Y by at most 2 codes)."""
    import torch
    w, h, depth = 512, 128, 10
    path = cube.write_cube(tmp_path / "look.cube", cube.log709_lattice(33))
    lut = cube.read_cube(path)
    engine.set_lut(lut)
    engine.set_variant("vec_lds")
    engine.set_precision(precision)
    try:
        src = frames.natural_yuv(w, h, depth, 0, 0, k=7)                        # 4:4:4: no chroma resampling between the stages
        k = orc.yuv_constants("bt709", "tv", "bt709", "tv", depth, depth, depth, 1)
        # stages 1 and 3 on the CPU (the NumPy twin of the oracle, stage by stage), stage 2 by the live binary
        from oracle import lut3d_numpy as twin
        rgb = [a.astype(np.uint16) for a in twin.yuv_to_rgb_codes(k, 0, 0, src)]
        live = _ffmpeg_lut3d_rgb(path, mode, [rgb[1], rgb[2], rgb[0]], "gbrp10le")       # gbrp order
        want = twin.rgb_codes_to_yuv(k, depth, 0, 0, (live[2], live[0], live[1]))
        dev = [torch.from_numpy(p.view(np.int16)).to(engine.device) for p in src]
        out = engine.apply_yuv(dev, pix_fmt="yuv444p10le", interp=mode)
        got = [t.cpu().numpy().view(np.uint16) for t in out]
        _compare(got, want, 2, f"liblutr {engine.last_kernel} vs live lut3d inside the engine's YUV stages {mode}")
    finally:
        engine.set_precision("strict")
        engine.set_variant("auto")
