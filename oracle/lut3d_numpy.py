"""NumPy twin of the C oracle -- TEST INFRASTRUCTURE ONLY.

A second, independently written restatement (vectorised float32, FFmpeg's *branchy*
tetrahedral form rather than the sorted form the kernels use) used to cross-check
oracle/lut3d_oracle.c.  Follows SURVEY.md Appendix A (FFmpeg lut3d, the filter the
reference emits at /root/reference/src/lut_renderer/ffmpeg.py:246) and DESIGN.md's
YUV contract.  PARITY UNPINNED against a live ffmpeg, like the C oracle.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def parse_cube_text(text: str):
    """Pure-Python parse_cube (A.2) on file text; returns (n, scale, table[r,g,b,3])."""
    lines = text.split("\n")
    it = iter(lines)
    n = None
    for line in it:
        if line.startswith("LUT_3D_SIZE"):
            tail = line[12:].split()
            n = int(tail[0], 0) if tail else 0
            if n < 2 or n > 256:
                raise ValueError("EINVAL")
            break
    if n is None:
        raise ValueError("EILSEQ")
    dmin, dmax = [0.0] * 3, [1.0] * 3
    vals = []
    while len(vals) < n ** 3:
        try:
            line = next(it)
        except StopIteration:
            raise ValueError("EILSEQ")
        if line.startswith("DOMAIN_"):
            tgt = dmin if line.startswith("DOMAIN_MIN ") else dmax if line.startswith("DOMAIN_MAX ") else None
            if tgt is None:
                raise ValueError("EILSEQ")
            for i, tok in enumerate(line[11:].split()[:3]):
                tgt[i] = float(tok)
            continue
        if line.startswith("TITLE"):
            continue
        s = line.strip()
        if not s or s.startswith("#"):
            continue
        tok = s.split()
        try:
            vals.append((float(tok[0]), float(tok[1]), float(tok[2])))
        except (ValueError, IndexError):
            raise ValueError("EILSEQ")
    arr = np.array(vals, dtype=F).reshape(n, n, n, 3)        # file index [b][g][r]
    table = np.ascontiguousarray(np.transpose(arr, (2, 1, 0, 3)))
    # float subtraction, double division, float clip (vf_lut3d.c parse_cube)
    scale = np.array([np.clip(F(1.0 / float(F(dmax[c]) - F(dmin[c]))), 0, 1) for c in range(3)], dtype=F)
    return n, scale, table


def _fma(a, b, c):
    # float32 fma via 80-bit long double: the product is exact, the sum keeps 64 bits
    r = np.asarray(a, dtype=np.longdouble) * np.asarray(b, dtype=np.longdouble) + np.asarray(c, dtype=np.longdouble)
    return r.astype(F)


def _interp(table, mode, s):
    n = table.shape[0]
    sr, sg, sb = s
    if mode == "nearest":
        # NEAR(x) = (int)(x + .5) with a double .5: the sum is exact in double
        i = [(v.astype(np.float64) + 0.5).astype(np.int32) for v in (sr, sg, sb)]
        return table[i[0], i[1], i[2]]
    p = [v.astype(np.int32) for v in (sr, sg, sb)]
    x = [np.minimum(v + 1, n - 1) for v in p]
    d = [(v - q.astype(F)).astype(F) for v, q in zip((sr, sg, sb), p)]

    def c(i, j, k):
        return table[(x[0] if i else p[0]), (x[1] if j else p[1]), (x[2] if k else p[2])]

    if mode == "trilinear":
        def lerp(a, b, f):
            return (a + (b - a) * f[..., None]).astype(F)
        c00, c10 = lerp(c(0, 0, 0), c(1, 0, 0), d[0]), lerp(c(0, 1, 0), c(1, 1, 0), d[0])
        c01, c11 = lerp(c(0, 0, 1), c(1, 0, 1), d[0]), lerp(c(0, 1, 1), c(1, 1, 1), d[0])
        return lerp(lerp(c00, c10, d[1]), lerp(c01, c11, d[1]), d[2])
    if mode != "tetrahedral":
        raise ValueError(mode)
    dr, dg, db = d
    one = F(1)

    def blend(w0, w1, v1, w2, v2, w3):
        w = [q[..., None].astype(F) for q in (w0, w1, w2, w3)]
        return (((w[0] * c(0, 0, 0) + w[1] * v1).astype(F) + w[2] * v2).astype(F) + w[3] * c(1, 1, 1)).astype(F)

    cases = [
        ((dr > dg) & (dg > db), blend(one - dr, dr - dg, c(1, 0, 0), dg - db, c(1, 1, 0), db)),
        ((dr > dg) & ~(dg > db) & (dr > db), blend(one - dr, dr - db, c(1, 0, 0), db - dg, c(1, 0, 1), dg)),
        ((dr > dg) & ~(dg > db) & ~(dr > db), blend(one - db, db - dr, c(0, 0, 1), dr - dg, c(1, 0, 1), dg)),
        (~(dr > dg) & (db > dg), blend(one - db, db - dg, c(0, 0, 1), dg - dr, c(0, 1, 1), dr)),
        (~(dr > dg) & ~(db > dg) & (db > dr), blend(one - dg, dg - db, c(0, 1, 0), db - dr, c(0, 1, 1), dr)),
        (~(dr > dg) & ~(db > dg) & ~(db > dr), blend(one - dg, dg - dr, c(0, 1, 0), dr - db, c(1, 1, 0), db)),
    ]
    out = np.zeros(dr.shape + (3,), dtype=F)
    for m, v in cases:
        out[m] = v[m]
    return out


def lut3d_codes(table, scale, depth, mode, r, g, b):
    """A.3 on integer code arrays; returns integer code arrays (r, g, b)."""
    n = table.shape[0]
    m = (1 << depth) - 1
    scale_f = F(1.0) / F(m)
    lut_max = F(n - 1)
    s = []
    for v, sc in zip((r, g, b), scale):
        x = (v.astype(F) * scale_f).astype(F)
        s.append(np.clip((x * (F(sc) * lut_max)).astype(F), F(0), lut_max).astype(F))
    v = _interp(table, mode, s)
    t = (v * F(m)).astype(F)
    q = np.clip(np.trunc(t.astype(np.float64)), 0, m).astype(np.int64)
    return q[..., 0], q[..., 1], q[..., 2]


def apply_rgb(table, scale, depth, mode, planes):
    g, b, r = planes
    ro, go, bo = lut3d_codes(table, scale, depth, mode, r, g, b)
    dt = planes[0].dtype
    return [go.astype(dt), bo.astype(dt), ro.astype(dt)]


def _clip_floor(v, hi):
    return np.clip(np.floor(v), F(0), F(hi)).astype(F)


def yuv_to_rgb_codes(k, csx, csy, planes):
    """Stage 1 of the YUV contract: optional range/depth prologue, then YUV -> integer RGB at the LUT depth --
    what FFmpeg hands lut3d.  Returns (R, G, B) as int64 arrays at luma resolution."""
    y, cb, cr = [p.astype(F) for p in planes]
    h, w = y.shape
    if k.pre:
        y = _clip_floor(_fma(F(k.py), y, F(k.pyb)), k.pre_max)
        cb = _clip_floor(_fma(F(k.pc), cb, F(k.pcb)), k.pre_max)
        cr = _clip_floor(_fma(F(k.pc), cr, F(k.pcb)), k.pre_max)
    cbd = (cb - F(k.coff)).astype(F)
    crd = (cr - F(k.coff)).astype(F)
    rv = (F(k.krv) * crd).astype(F)
    gv = _fma(F(k.kgu), cbd, (F(k.kgv) * crd).astype(F))
    bu = (F(k.kbu) * cbd).astype(F)

    def up(a):
        return np.repeat(np.repeat(a, 1 << csy, axis=0), 1 << csx, axis=1)[:h, :w]

    yy = _fma(F(k.ky), y, F(k.yb))
    rq = _clip_floor((yy + up(rv)).astype(F), k.max_l)
    gq = _clip_floor((yy + up(gv)).astype(F), k.max_l)
    bq = _clip_floor((yy + up(bu)).astype(F), k.max_l)
    return rq.astype(np.int64), gq.astype(np.int64), bq.astype(np.int64)


def rgb_codes_to_yuv(k, dout, csx, csy, rgb):
    """Stage 3: integer RGB (lut3d's output) -> YUV at the output depth, chroma = block mean (1/n folded into the constants)."""
    ro, go, bo = [np.asarray(a).astype(F) for a in rgb]
    h, w = ro.shape
    yo = _clip_floor(_fma(F(k.cyr), ro, _fma(F(k.cyg), go, _fma(F(k.cyb), bo, F(k.yob)))), k.max_o)
    bh, bw = 1 << csy, 1 << csx
    ch, cw = (h + bh - 1) >> csy, (w + bw - 1) >> csx

    def block_sum(a):
        pad = np.pad(a, ((0, ch * bh - h), (0, cw * bw - w)), mode="edge")
        return pad.reshape(ch, bh, cw, bw).sum(axis=(1, 3)).astype(F)

    rs, gs, bs = block_sum(ro), block_sum(go), block_sum(bo)
    cbo = _clip_floor(_fma(F(k.cbr), rs, _fma(F(k.cbg), gs, _fma(F(k.cbb), bs, F(k.cob)))), k.max_o)
    cro = _clip_floor(_fma(F(k.crr), rs, _fma(F(k.crg), gs, _fma(F(k.crb), bs, F(k.cob)))), k.max_o)
    odt = np.uint8 if dout <= 8 else np.uint16
    return [yo.astype(odt), cbo.astype(odt), cro.astype(odt)]


def apply_yuv(table, scale, mode, k, din, dl, dout, csx, csy, planes):
    """k: oracle.binding.YuvConsts (constants are shared data, the pixel math is separate)."""
    rq, gq, bq = yuv_to_rgb_codes(k, csx, csy, planes)
    return rgb_codes_to_yuv(k, dout, csx, csy, lut3d_codes(table, scale, dl, mode, rq, gq, bq))
