// lutr_device.h -- device-side pieces shared by the gfx950 kernels (lutr_kernels.hip,
// lutr_packed.hip): the lut3d per-pixel restatement, the
// YUV contract (DESIGN.md "YUV contract") and little sample/word accessors.
//
// Everything here rounds exactly like FFmpeg's scalar C (-ffp-contract=off; fused
// multiply-adds only where written as __builtin_fmaf).
#pragma once

#include "lutr_internal.h"

namespace lutr {

// ---------------------------------------------------------------- small math
__device__ __forceinline__ float med3(float a, float lo, float hi) { return __builtin_amdgcn_fmed3f(a, lo, hi); }
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float min3(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---------------------------------------------------------------- lattice access
// Global-memory gather (served by L1 / the XCD's L2: a 33^3 padded lattice is 629 KB).
struct GFetch {
    const float4 *__restrict__ lat;
    float fr, fg;   // node strides as floats: n1*n1, n1 (blue stride is 1)
    int   sr, sg;   // the same as ints
    __device__ __forceinline__ explicit GFetch(const LutConsts &L)
        : lat(L.lat), fr((float)(L.n1 * L.n1)), fg((float)L.n1), sr(L.n1 * L.n1), sg(L.n1) {}
    // integer math: n1^3 reaches 2^24 at N = 256, one step past what fp32 holds exactly
    __device__ __forceinline__ int index(float pr, float pg, float pb) const
    {
        return ((int)pr * sg + (int)pg) * sg + (int)pb;
    }
    __device__ __forceinline__ float4 ld(int i) const { return lat[i]; }
};

// ---------------------------------------------------------------- lut3d core
struct Rgb { float r, g, b; };

__device__ __forceinline__ float lerpf(float v0, float v1, float f) { return v0 + (v1 - v0) * f; }

// FFmpeg's NEAR(x) = (int)(x + .5) adds a DOUBLE .5 to the float coordinate, so the sum is exact and the result is
// floor(x) + (frac(x) >= 1/2).  frac(x) = x - floor(x) is exact in fp32, which makes this form bit-identical without
// double arithmetic (floorf(x + .5f) is not: for x = k + 0.49999997 the float sum rounds up to k + 1).
__device__ __forceinline__ float near_f(float s)
{
    const float p = floorf(s);
    return (s - p >= .5f) ? p + 1.0f : p;
}

template <class F>
__device__ __forceinline__ Rgb interp_nearest(const F &f, float sr, float sg, float sb)
{
    const float4 c = f.ld(f.index(near_f(sr), near_f(sg), near_f(sb)));
    return Rgb{c.x, c.y, c.z};
}

template <class F>
__device__ __forceinline__ Rgb interp_trilinear(const F &f, float sr, float sg, float sb)
{
    const float pr = floorf(sr), pg = floorf(sg), pb = floorf(sb);
    const float dr = sr - pr, dg = sg - pg, db = sb - pb;
    const int i = f.index(pr, pg, pb);
    const float4 c000 = f.ld(i), c001 = f.ld(i + 1);
    const float4 c010 = f.ld(i + f.sg), c011 = f.ld(i + f.sg + 1);
    const float4 c100 = f.ld(i + f.sr), c101 = f.ld(i + f.sr + 1);
    const float4 c110 = f.ld(i + f.sr + f.sg), c111 = f.ld(i + f.sr + f.sg + 1);
    Rgb o;
#define TRI(ch) \
    { \
        const float c00 = lerpf(c000.ch, c100.ch, dr), c10 = lerpf(c010.ch, c110.ch, dr); \
        const float c01 = lerpf(c001.ch, c101.ch, dr), c11 = lerpf(c011.ch, c111.ch, dr); \
        const float c0 = lerpf(c00, c10, dg), c1 = lerpf(c01, c11, dg); \
        o_ = lerpf(c0, c1, db); \
    }
    float o_;
    TRI(x) o.r = o_;
    TRI(y) o.g = o_;
    TRI(z) o.b = o_;
#undef TRI
    return o;
}

// FFmpeg's six branches all evaluate (1-x)*c000 + (x-y)*cA + (y-z)*cB + z*c111 with
// (x,y,z) = (d.r,d.g,d.b) sorted descending, cA one step along x's axis and cB one more
// along y's.  Ties only change a tap whose weight is exactly 0, so for a finite lattice
// the max3/med3/min3 form below is bit-identical to the branchy original.
template <class F>
__device__ __forceinline__ Rgb interp_tetrahedral(const F &f, float sr, float sg, float sb)
{
    const float pr = floorf(sr), pg = floorf(sg), pb = floorf(sb);
    const float dr = sr - pr, dg = sg - pg, db = sb - pb;
    const int i = f.index(pr, pg, pb);
    const float x = max3(dr, dg, db), y = med3(dr, dg, db), z = min3(dr, dg, db);
    const int oa = (dr == x) ? f.sr : ((dg == x) ? f.sg : 1);
    const int oz = (db == z) ? 1 : ((dg == z) ? f.sg : f.sr);
    const int o111 = f.sr + f.sg + 1;
    const float4 c0 = f.ld(i), c1 = f.ld(i + oa), c2 = f.ld(i + o111 - oz), c3 = f.ld(i + o111);
    const float w0 = 1.0f - x, w1 = x - y, w2 = y - z, w3 = z;
    Rgb o;
    o.r = w0 * c0.x + w1 * c1.x + w2 * c2.x + w3 * c3.x;
    o.g = w0 * c0.y + w1 * c1.y + w2 * c2.y + w3 * c3.y;
    o.b = w0 * c0.z + w1 * c1.z + w2 * c2.z + w3 * c3.z;
    return o;
}

// pyramid / prism (generic kernel only)
template <class F>
__device__ Rgb interp_pyramid(const F &f, float sr, float sg, float sb)
{
    const float pr = floorf(sr), pg = floorf(sg), pb = floorf(sb);
    const float dr = sr - pr, dg = sg - pg, db = sb - pb;
    const int i = f.index(pr, pg, pb);
    const float4 c000 = f.ld(i), c001 = f.ld(i + 1);
    const float4 c010 = f.ld(i + f.sg), c011 = f.ld(i + f.sg + 1);
    const float4 c100 = f.ld(i + f.sr), c101 = f.ld(i + f.sr + 1);
    const float4 c110 = f.ld(i + f.sr + f.sg), c111 = f.ld(i + f.sr + f.sg + 1);
    Rgb o;
#define PYR(ch, out) \
    if (dg > dr && db > dr) { \
        out = c000.ch + (c111.ch - c011.ch) * dr + (c010.ch - c000.ch) * dg + (c001.ch - c000.ch) * db + \
              (c011.ch - c001.ch - c010.ch + c000.ch) * dg * db; \
    } else if (dr > dg && db > dg) { \
        out = c000.ch + (c100.ch - c000.ch) * dr + (c111.ch - c101.ch) * dg + (c001.ch - c000.ch) * db + \
              (c101.ch - c001.ch - c100.ch + c000.ch) * dr * db; \
    } else { \
        out = c000.ch + (c100.ch - c000.ch) * dr + (c010.ch - c000.ch) * dg + (c111.ch - c110.ch) * db + \
              (c110.ch - c100.ch - c010.ch + c000.ch) * dr * dg; \
    }
    PYR(x, o.r) PYR(y, o.g) PYR(z, o.b)
#undef PYR
    return o;
}

template <class F>
__device__ Rgb interp_prism(const F &f, float sr, float sg, float sb)
{
    const float pr = floorf(sr), pg = floorf(sg), pb = floorf(sb);
    const float dr = sr - pr, dg = sg - pg, db = sb - pb;
    const int i = f.index(pr, pg, pb);
    const float4 c000 = f.ld(i), c001 = f.ld(i + 1);
    const float4 c010 = f.ld(i + f.sg), c011 = f.ld(i + f.sg + 1);
    const float4 c100 = f.ld(i + f.sr), c101 = f.ld(i + f.sr + 1);
    const float4 c110 = f.ld(i + f.sr + f.sg), c111 = f.ld(i + f.sr + f.sg + 1);
    Rgb o;
#define PRI(ch, out) \
    if (db > dr) { \
        out = c000.ch + (c001.ch - c000.ch) * db + (c101.ch - c001.ch) * dr + (c010.ch - c000.ch) * dg + \
              (c000.ch - c010.ch - c001.ch + c011.ch) * db * dg + \
              (c001.ch - c011.ch - c101.ch + c111.ch) * dr * dg; \
    } else { \
        out = c000.ch + (c101.ch - c100.ch) * db + (c100.ch - c000.ch) * dr + (c010.ch - c000.ch) * dg + \
              (c100.ch - c110.ch - c101.ch + c111.ch) * db * dg + \
              (c000.ch - c010.ch - c100.ch + c110.ch) * dr * dg; \
    }
    PRI(x, o.r) PRI(y, o.g) PRI(z, o.b)
#undef PRI
    return o;
}

template <int INTERP, class F>
__device__ __forceinline__ Rgb interp(const F &f, float sr, float sg, float sb)
{
    if constexpr (INTERP == LUTR_INTERP_NEAREST) return interp_nearest(f, sr, sg, sb);
    else if constexpr (INTERP == LUTR_INTERP_TRILINEAR) return interp_trilinear(f, sr, sg, sb);
    else if constexpr (INTERP == LUTR_INTERP_PYRAMID) return interp_pyramid(f, sr, sg, sb);
    else if constexpr (INTERP == LUTR_INTERP_PRISM) return interp_prism(f, sr, sg, sb);
    else return interp_tetrahedral(f, sr, sg, sb);
}

// One pixel of A.3.  In: integer codes held as floats.  Out: integer codes held as
// floats (truncation toward zero, then clip to [0, M], exactly av_clip_uintp2((int)(v*M))).
template <int INTERP, class F>
__device__ __forceinline__ Rgb lut3d_px(const LutConsts &L, const F &f, float rc, float gc, float bc)
{
    float sr, sg, sb;
    if (L.pre) {         // a prelut (cineSpace shaper): the host folded shaper, scale and clip into one coordinate per integer code
        sr = L.pre[(int)rc]; sg = L.pre[L.pre_stride + (int)gc]; sb = L.pre[2 * L.pre_stride + (int)bc];
    } else {
        const float xr = rc * L.scale_f, xg = gc * L.scale_f, xb = bc * L.scale_f;
        sr = med3(xr * L.sc[0], 0.0f, L.lut_max);
        sg = med3(xg * L.sc[1], 0.0f, L.lut_max);
        sb = med3(xb * L.sc[2], 0.0f, L.lut_max);
    }
    const Rgb v = interp<INTERP>(f, sr, sg, sb);
    Rgb o;
    o.r = med3(truncf(v.r * L.maxf), 0.0f, L.maxf);
    o.g = med3(truncf(v.g * L.maxf), 0.0f, L.maxf);
    o.b = med3(truncf(v.b * L.maxf), 0.0f, L.maxf);
    return o;
}

template <class F>
__device__ __forceinline__ Rgb lut3d_px_rt(int mode, const LutConsts &L, const F &f, float r, float g, float b)
{
    switch (mode) {
    case LUTR_INTERP_NEAREST:   return lut3d_px<LUTR_INTERP_NEAREST>(L, f, r, g, b);
    case LUTR_INTERP_TRILINEAR: return lut3d_px<LUTR_INTERP_TRILINEAR>(L, f, r, g, b);
    case LUTR_INTERP_PYRAMID:   return lut3d_px<LUTR_INTERP_PYRAMID>(L, f, r, g, b);
    case LUTR_INTERP_PRISM:     return lut3d_px<LUTR_INTERP_PRISM>(L, f, r, g, b);
    default:                    return lut3d_px<LUTR_INTERP_TETRAHEDRAL>(L, f, r, g, b);
    }
}

// ---------------------------------------------------------------- YUV contract pieces
struct Chroma { float rv, gv, bu; };

__device__ __forceinline__ float clip_floor(float v, float hi) { return med3(floorf(v), 0.0f, hi); }

__device__ __forceinline__ Chroma chroma_terms(const YuvConsts &K, float cbv, float crv)
{
    if (K.pre != 0.0f) {
        cbv = clip_floor(fma_(K.pc, cbv, K.pcb), K.pre_max);
        crv = clip_floor(fma_(K.pc, crv, K.pcb), K.pre_max);
    }
    const float cb = cbv - K.coff, cr = crv - K.coff;
    Chroma c;
    c.rv = K.krv * cr;
    c.gv = fma_(K.kgu, cb, K.kgv * cr);
    c.bu = K.kbu * cb;
    return c;
}

__device__ __forceinline__ Rgb yuv_to_rgb(const YuvConsts &K, float yv, const Chroma &c)
{
    if (K.pre != 0.0f)
        yv = clip_floor(fma_(K.py, yv, K.pyb), K.pre_max);
    const float yy = fma_(K.ky, yv, K.yb);
    Rgb o;
    o.r = clip_floor(yy + c.rv, K.max_l);
    o.g = clip_floor(yy + c.gv, K.max_l);
    o.b = clip_floor(yy + c.bu, K.max_l);
    return o;
}

__device__ __forceinline__ float rgb_to_y(const YuvConsts &K, const Rgb &q)
{
    return clip_floor(fma_(K.cyr, q.r, fma_(K.cyg, q.g, fma_(K.cyb, q.b, K.yob))), K.max_o);
}

__device__ __forceinline__ float rgb_to_cb(const YuvConsts &K, float rs, float gs, float bs)
{
    return clip_floor(fma_(K.cbr, rs, fma_(K.cbg, gs, fma_(K.cbb, bs, K.cob))), K.max_o);
}

__device__ __forceinline__ float rgb_to_cr(const YuvConsts &K, float rs, float gs, float bs)
{
    return clip_floor(fma_(K.crr, rs, fma_(K.crg, gs, fma_(K.crb, bs, K.cob))), K.max_o);
}

// ---------------------------------------------------------------- sample access
__device__ __forceinline__ float ld_sample(const uint8_t *row, int x, int wide)
{
    return wide ? (float)((const uint16_t *)row)[x] : (float)row[x];
}

__device__ __forceinline__ void st_sample(uint8_t *row, int x, int wide, float v)
{
    const unsigned u = (unsigned)v;
    if (wide) ((uint16_t *)row)[x] = (uint16_t)u;
    else row[x] = (uint8_t)u;
}

// sample i of a little-endian word vector (i is a compile-time constant after unrolling)
template <int WIDE>
__device__ __forceinline__ float word_sample(const uint32_t *w, int i)
{
    if constexpr (WIDE) return (float)((w[i >> 1] >> ((i & 1) * 16)) & 0xffffu);
    else return (float)((w[i >> 2] >> ((i & 3) * 8)) & 0xffu);
}

template <int WIDE>
__device__ __forceinline__ void word_put(uint32_t *w, int i, float v)
{
    const uint32_t u = (uint32_t)v;
    if constexpr (WIDE) w[i >> 1] |= u << ((i & 1) * 16);
    else w[i >> 2] |= u << ((i & 3) * 8);
}

template <int NW>
__device__ __forceinline__ void ld_words(uint32_t *w, const uint8_t *p)
{
    if constexpr (NW == 4) {
        const uint4 v = *(const uint4 *)p;
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else if constexpr (NW == 2) {
        const uint2 v = *(const uint2 *)p;
        w[0] = v.x; w[1] = v.y;
    } else {
        w[0] = *(const uint32_t *)p;
    }
}

// NT: non-temporal store.  Pays where the stores would otherwise push the lattice out of the caches: k_rgb_vec on small gbrp10le
// launches 253 -> 360 Gpx/s; the YUV and packed vector kernels do not gain (+-1 %) and keep plain stores.
template <int NW, bool NT = false>
__device__ __forceinline__ void st_words(uint8_t *p, const uint32_t *w)
{
    if constexpr (NT) {
        typedef unsigned nt4 __attribute__((ext_vector_type(4)));
        typedef unsigned nt2 __attribute__((ext_vector_type(2)));
        if constexpr (NW == 4) __builtin_nontemporal_store(nt4{w[0], w[1], w[2], w[3]}, (nt4 *)p);
        else if constexpr (NW == 2) __builtin_nontemporal_store(nt2{w[0], w[1]}, (nt2 *)p);
        else __builtin_nontemporal_store(w[0], (uint32_t *)p);
    } else {
        if constexpr (NW == 4) *(uint4 *)p = make_uint4(w[0], w[1], w[2], w[3]);
        else if constexpr (NW == 2) *(uint2 *)p = make_uint2(w[0], w[1]);
        else *(uint32_t *)p = w[0];
    }
}

}  // namespace lutr
