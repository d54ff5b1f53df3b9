"""Committed pixel vectors (tests/golden/pixels.npz, made by tests/golden/make_pixel_fixtures.py).

CPU: the oracle still produces them (guards the oracle against drift between rounds).
GPU: liblutr's kernels produce them through the C-ABI -- generic kernel and, where the layout
allows, the vector and LDS-window kernels.  All integer codes: exact equality.
"""
import json
from pathlib import Path

import numpy as np
import pytest

GOLD = np.load(Path(__file__).parent / "golden" / "pixels.npz")
META = json.loads(str(GOLD["meta"]))


def _planes(i, what):
    return [GOLD[f"{i}/{what}{p}"] for p in range(3)]


@pytest.mark.parametrize("i", range(len(META)))
def test_oracle_reproduces_golden(orc, i):
    c = META[i]
    tab, sc = GOLD[f"lat/{c['lut']}"], GOLD[f"scale/{c['lut']}"]
    src = _planes(i, "src")
    if c["kind"] == "rgb":
        got = orc.apply_rgb(tab, sc, c["depth"], c["mode"], src)
    else:
        k = orc.yuv_constants(c["matrix"], c["rin"], c["matrix"], c["rout"], c["din"], c["dl"], c["dout"],
                              1 << (c["csx"] + c["csy"]), prologue=c["prologue"])
        got = orc.apply_yuv(tab, sc, c["mode"], k, c["din"], c["dl"], c["dout"], c["csx"], c["csy"], src)
    for a, b in zip(got, _planes(i, "dst")):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["generic", "vec_global", "vec_lds"])
def test_kernels_reproduce_golden(engine, variant):
    import torch
    from lut_renderer_amd.cube import CubeLut
    ran = 0
    for i, c in enumerate(META):
        fast_ok = c["mode"] in ("nearest", "trilinear", "tetrahedral") and \
            (c["kind"] == "rgb" or (c["din"] > 8) == (c["dout"] > 8))
        if variant != "generic" and not fast_ok:
            continue
        tab, sc = GOLD[f"lat/{c['lut']}"], GOLD[f"scale/{c['lut']}"]
        engine.set_lut(CubeLut(tab.shape[0], sc, tab))
        engine.set_variant(variant)
        src = [torch.from_numpy(p.view(np.int16) if p.dtype == np.uint16 else p).to(engine.device)
               for p in _planes(i, "src")]
        if c["kind"] == "rgb":
            out = engine.apply_rgb(src, depth=c["depth"], interp=c["mode"])
        else:
            odepth = c["dout"]
            out_fmt = c["fmt"] if odepth == c["din"] else c["fmt"].replace("10le", "") if odepth == 8 else c["fmt"]
            out = engine.apply_yuv(src, pix_fmt=c["fmt"], out_pix_fmt=out_fmt, interp=c["mode"], matrix_in=c["matrix"],
                                   range_src="pc" if (c["prologue"] or c["rin"] == "pc") else "tv", range_in=c["rin"],
                                   range_out=c["rout"], lut_depth=c["dl"])
        for a, b in zip(out, _planes(i, "dst")):
            g = a.cpu().numpy()
            g = g.view(np.uint16) if b.dtype == np.uint16 else g
            assert np.array_equal(g, b), (variant, c, engine.last_kernel)
        ran += 1
    engine.set_variant("auto")
    assert ran >= 20
