// lutr_internal.h -- structures shared by the C-ABI layer (lutr_api.cpp) and the
// gfx950 kernels (lutr_kernels.hip).  Not part of the public boundary.
#pragma once

#ifdef LUTR_HOST_ONLY
// Sanitizer build of the two host parsers (oracle/Makefile `asan`: gcc -fsanitize=address,undefined, no HIP):
// they need the public header and set_error only.
#include <stdint.h>

#include "lutr.h"

namespace lutr {
void set_error(const char *fmt, ...);
}
#else

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lutr.h"

namespace lutr {

// Device lattice: (n+1)^3 nodes of float4 {r,g,b,0}, index ((r*n1+g)*n1+b), blue
// fastest like FFmpeg's lut[r*N*N + g*N + b].  Index n on every axis
// is a replica of node n-1, so NEXT(x) = min(prev+1, n-1) (A.4) becomes prev+1 with
// no clamp in the kernels.
struct LutConsts {
    const float4 *lat;
    const uint2  *lat16;  // fast variant: the same nodes as fp16 {r, g, b, 0} of (value * (2^depth - 1)), or nullptr
    int   n1;            // n + 1
    float scale_f;       // 1.0f / (2^depth - 1)
    float sc[3];         // scale.{r,g,b} * (n-1)
    float lut_max;       // (float)(n-1)
    float maxf;          // (float)(2^depth - 1)
    int   unit;          // 1 when every lattice node is known to lie in [0, 1] (lets the tile kernels drop the output clip)
    const float *pre;    // lut3d's prelut folded into a per-code table of lattice coordinates (3 x pre_stride floats: the coordinate of
    int   pre_stride;    // integer code i of channel c is pre[c * pre_stride + i]), or nullptr.  The generic / vector kernels and the RGB
                         // tube kernels read it per channel; the fused YUV tile kernels take it when it is `pre_shared`:
    int   pre_shared;    // 1: the three channels' tables are identical and non-decreasing over the codes 0 .. 2^depth - 1 (the usual
    float pre_kappa;     // cineSpace shaper): one coordinate table serves R, G and B; pre_kappa = the largest step between two codes (cells)
    const float *pre_host; // HOST copy of that shared table (2^depth entries; launcher only: the tube's bound is read off the curve itself)
};

// Constant block of the YUV contract; same fields, same order as the
// oracle's orc_yuv_consts so tests can compare them float for float.
struct YuvConsts {
    float ky, yb, coff, krv, kgu, kgv, kbu, max_l;
    float cyr, cyg, cyb, yob;
    float cbr, cbg, cbb;
    float crr, crg, crb, cob;
    float max_o;
    float pre;                       // 0 or 1
    float py, pyb, pc, pcb, pre_max;
    float pad[6];                    // -> 32 floats
};
static_assert(sizeof(YuvConsts) == 32 * sizeof(float), "YuvConsts is the 32-float block of lutr_yuv_constants");

struct PlaneSet {
    const uint8_t *s[3];
    uint8_t       *d[3];
    long long ss[3], ds[3];          // row strides, bytes
    long long sfs[3], dfs[3];        // frame strides, bytes
};

// one interleaved image (or batch): component offsets in units of one component
struct PackedSet {
    const uint8_t *s;
    uint8_t       *d;
    long long ss, ds, sfs, dfs;      // row / frame strides, bytes
    int ro, go, bo, ao;              // ao = the fourth slot of 4-component formats (alpha or padding)
};

// unquantised output planes of the dither path (device scratch, densely packed per frame)
struct FloatPlanes {
    float *y, *cb, *cr;
};

struct FrameGeom {
    int w, h, row0, rows, nframes;
};

enum Variant { VAR_AUTO = 0, VAR_GENERIC = 1, VAR_VEC_GLOBAL = 2, VAR_VEC_LDS = 3 };

// launchers (lutr_kernels.hip); return the kernel's name, or nullptr when the variant
// cannot take this layout (caller then falls back to the generic kernel)
const char *launch_rgb(hipStream_t st, int variant, const LutConsts &L, const PlaneSet &P,
                       const FrameGeom &G, int depth, int interp, unsigned *stats, unsigned *queue);
const char *launch_yuv(hipStream_t st, int variant, const LutConsts &L, const YuvConsts &K,
                       const PlaneSet &P, const FrameGeom &G, int din, int dout, int lut_depth, int csx, int csy,
                       int interp, bool fast, unsigned *stats, unsigned *queue);

// packed RGB (lutr_packed.hip)
const char *launch_packed(hipStream_t st, int variant, const LutConsts &L, const PackedSet &P, const FrameGeom &G,
                          int wide, int ncomp, int interp, unsigned *stats, unsigned *queue);

// round-3 RGB tube kernels (lutr_rgb2.hip, one translation unit per layout): planes in SLOT order -- planar callers pass
// (R, G, B), packed callers the one buffer in [0] and rev = 1 for B, G, R memory order; nullptr = cannot take the call
#define LUTR_R2_DECL(ly) \
    const char *launch_rgb_tube_ly##ly(hipStream_t st, const LutConsts &L, const PlaneSet &P, const FrameGeom &G, int depth, \
                                       int mode, int rev, unsigned *stats, unsigned *queue);
LUTR_R2_DECL(0) LUTR_R2_DECL(1) LUTR_R2_DECL(2) LUTR_R2_DECL(3) LUTR_R2_DECL(4) LUTR_R2_DECL(5) LUTR_R2_DECL(6) LUTR_R2_DECL(7)
#undef LUTR_R2_DECL

// error-diffusion dither path (lutr_dither.hip): whole frames, float planes in F
const char *launch_yuv_dither(hipStream_t st, const LutConsts &L, const YuvConsts &K, const PlaneSet &P,
                              const FrameGeom &G, const FloatPlanes &F, int din, int dout, int csx, int csy, int interp);

// round-2 tile kernels (lutr_tile2.hip, one translation unit per format: w<in wide><out wide>_c<csx><csy>); nullptr =
// this combination is not built / cannot take the call, the caller falls back
#define LUTR_T2_DECL(tag) \
    const char *launch_yuv_tile2_##tag(hipStream_t st, const LutConsts &L, const YuvConsts &K, const PlaneSet &P, \
                                       const FrameGeom &G, int din, int dout, int lut_depth, int csx, int csy, int mode, \
                                       bool fast, unsigned *stats, unsigned *queue);
LUTR_T2_DECL(w00_c11) LUTR_T2_DECL(w00_c10) LUTR_T2_DECL(w00_c00)
LUTR_T2_DECL(w11_c11) LUTR_T2_DECL(w11_c10) LUTR_T2_DECL(w11_c00)
LUTR_T2_DECL(w10_c11) LUTR_T2_DECL(w10_c10) LUTR_T2_DECL(w10_c00)
#undef LUTR_T2_DECL
// fp16 lattice of the fast variant (lutr_lat16.hip)
void launch_make_lat16(hipStream_t st, const float4 *lat, uint2 *out, size_t nodes, float m);

// persistent LDS-window kernels (lutr_tile.hip); layout already checked by launch_rgb/launch_yuv
const char *launch_rgb_tile(hipStream_t st, const LutConsts &L, const PlaneSet &P, const FrameGeom &G,
                            int depth, int interp, unsigned *stats, unsigned *queue);

// host helpers (yuv_consts.cpp / cube_parse.cpp)
int make_yuv_consts(const lutr_yuv_params &p, YuvConsts *out);
void set_error(const char *fmt, ...);

}  // namespace lutr
#endif  // LUTR_HOST_ONLY
