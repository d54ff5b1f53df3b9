"""The C-ABI boundary without a GPU: the library loads, exports every declared symbol, its
host-only entry points work, and compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

from lut_renderer_amd import _native
from lut_renderer_amd.engine import parse_pix_fmt, yuv_constants

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    header = (ROOT / "include" / "lutr.h").read_text()
    declared = set(re.findall(r"\b(lutr_[a-z_0-9]+)\s*\(", header))
    declared -= {"lutr_yuv_params", "lutr_planes", "lutr_ctx"}
    assert declared == set(_native.SYMBOLS), declared ^ set(_native.SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    nm = subprocess.run(["nm", "-D", "--defined-only", str(_native.LIB_PATH)], capture_output=True, text=True).stdout
    for sym in declared:
        assert re.search(rf"\bT {sym}\b", nm), f"{sym} is not an exported text symbol"
    assert lib.lutr_version().decode() == "0.1.0"


def test_product_library_does_not_link_the_oracle():
    out = subprocess.run(["ldd", str(_native.LIB_PATH)], capture_output=True, text=True).stdout
    assert "oracle" not in out and "liblut3d" not in out
    nm = subprocess.run(["nm", "-D", str(_native.LIB_PATH)], capture_output=True, text=True).stdout
    assert "orc_" not in nm
    for py in (ROOT / "lut_renderer_amd").glob("*.py"):
        text = py.read_text()
        assert "import oracle" not in text and "from oracle" not in text, py


def test_struct_layouts_match_the_header():
    assert C.sizeof(_native.YuvParams) == 8 * 4
    assert C.sizeof(_native.Planes) == 3 * 8 + 3 * 8 + 3 * 8
    assert _native.load().lutr_lattice_bytes(33) == 34 ** 3 * 16
    assert _native.load().lutr_lattice_bytes(65) == 66 ** 3 * 16
    assert _native.load().lutr_lattice_bytes(1) == 0 and _native.load().lutr_lattice_bytes(257) == 0


def test_format_codes_and_names():
    assert _native.fmt_code(10, 1, 1) == 10 | 0x100 | 0x200
    pf = parse_pix_fmt("yuv420p10le")
    assert (pf.family, pf.depth, pf.csx, pf.csy, pf.full_range) == ("yuv", 10, 1, 1, False)
    assert parse_pix_fmt("yuvj420p").full_range and parse_pix_fmt("yuvj420p").depth == 8
    assert parse_pix_fmt("yuv422p10le").plane_shape(1, 1920, 1080) == (1080, 960)
    assert parse_pix_fmt("yuv420p").plane_shape(2, 1921, 1081) == (541, 961)
    assert parse_pix_fmt("gbrp12le").depth == 12 and parse_pix_fmt("gbrp").family == "gbr"
    for bad in ("rgb24", "yuv420p7", "nv12", "", "gbr420p"):
        with pytest.raises(ValueError):
            parse_pix_fmt(bad)


@pytest.mark.parametrize("depth,csx,csy", [(8, 1, 1), (10, 1, 1), (10, 1, 0), (10, 0, 0), (12, 1, 1), (16, 0, 0)])
def test_yuv_constant_block_equals_the_oracles(orc, depth, csx, csy):
    """liblutr (csrc/lutr_api.cpp) and the oracle derive the same 32 floats, bit for bit."""
    fmt = _native.fmt_code(depth, csx, csy)
    for mi, mname in enumerate(("bt709", "smpte170m", "bt2020nc")):
        for mo, moname in enumerate(("bt709", "smpte170m", "bt2020nc")):
            for rin in (0, 1):
                for rout in (0, 1):
                    blk = yuv_constants(fmt_in=fmt, fmt_out=fmt, lut_depth=depth, matrix_in=mi, matrix_out=mo,
                                        range_src=rin, range_in=rin, range_out=rout)
                    k = orc.yuv_constants(mname, ("tv", "pc")[rin], moname, ("tv", "pc")[rout], depth, depth, depth,
                                          1 << (csx + csy))
                    assert np.array_equal(blk, k.as_block()), (mname, moname, rin, rout)


def test_yuv_constant_block_prologue_and_mixed_depth(orc):
    for din, dl, dout, rin in ((10, 8, 10, 0), (10, 8, 8, 0), (8, 8, 8, 0), (10, 8, 8, 1), (12, 8, 10, 0)):
        blk = yuv_constants(fmt_in=_native.fmt_code(din, 1, 1), fmt_out=_native.fmt_code(dout, 1, 1), lut_depth=dl,
                            matrix_in=0, matrix_out=2, range_src=1, range_in=rin, range_out=0)
        k = orc.yuv_constants("bt709", ("tv", "pc")[rin], "bt2020nc", "tv", din, dl, dout, 4, prologue=True)
        assert np.array_equal(blk, k.as_block()), (din, dl, dout, rin)
    blk = yuv_constants(fmt_in=_native.fmt_code(10, 1, 1), fmt_out=_native.fmt_code(8, 1, 1), lut_depth=10,
                        matrix_in=2, matrix_out=2, range_src=0, range_in=0, range_out=0)
    assert np.array_equal(blk, orc.yuv_constants("bt2020nc", "tv", "bt2020nc", "tv", 10, 10, 8, 4).as_block())


def test_yuv_constant_errors():
    from lut_renderer_amd._native import LutrError
    ok = dict(fmt_in=_native.fmt_code(10, 1, 1), fmt_out=_native.fmt_code(10, 1, 1), lut_depth=10, matrix_in=0,
              matrix_out=0, range_src=0, range_in=0, range_out=0)
    yuv_constants(**ok)
    for bad in (dict(matrix_in=7), dict(range_out=2), dict(lut_depth=7), dict(fmt_out=_native.fmt_code(10, 0, 0)),
                dict(fmt_in=_native.fmt_code(10, 0, 1), fmt_out=_native.fmt_code(10, 0, 1)),
                dict(range_src=0, range_in=1),            # a prologue from a limited-range source is undefined
                dict(lut_depth=8)):                        # depth change without a full-range source
        with pytest.raises(LutrError) as e:
            yuv_constants(**{**ok, **bad})
        assert e.value.code == _native.EINVAL and e.value.message


def test_no_gpu_means_loud_failure_not_a_fallback():
    """In the build container there is no GPU: creating a context must fail with LUTR_EIO."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _native.load()
    handle = C.c_void_p()
    rc = lib.lutr_ctx_create(0, C.byref(handle))
    assert rc == _native.EIO and not handle.value
    assert b"no CPU fallback" in lib.lutr_last_error()
    from lut_renderer_amd.engine import LutEngine
    with pytest.raises(RuntimeError):
        LutEngine(0)
    # null-argument checks do not need a device either
    assert lib.lutr_ctx_sync(None) == _native.EINVAL
    assert lib.lutr_apply_yuv(None, None, 2, 16, 16, 1, None, None, 0, 16) == _native.EINVAL


def test_missing_library_is_an_import_error(monkeypatch, tmp_path):
    monkeypatch.setenv("LUTR_LIBRARY", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_native, "_lib", None)
    with pytest.raises(ImportError):
        _native.load()
