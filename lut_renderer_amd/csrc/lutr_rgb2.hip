// lutr_rgb2.hip -- round-3 RGB kernels: lut3d on planar (gbrp*) and packed (rgb24 ... rgba64le) frames with the lattice's
// GREY TUBE in LDS, persistent waves and 16-byte accesses.
//
// Replaces the slice-threaded per-row loops of FFmpeg's lut3d for its native formats
// (the filter /root/reference/src/lut_renderer/ffmpeg.py:246 emits; lut3d only takes RGB).
//
// Design:
//   * ONE LDS structure: the tube of lutr_tile2.hip in (r, g - r, b - g) coordinates -- every cell with |pg - pr| <= H and
//     |pb - pg| <= H, all of r -- staged once per workgroup.  RGB input needs no YUV matrix and a small per-code coordinate table
//     (256 entries at 8 bit, 1024 at 10), so the tube gets the LDS the fused YUV kernels spend on per-wave windows: H = 8 at
//     33^3 with 12-byte fp32 nodes (+-64 8-bit codes of G-R and B-G).  Lattices up to 22^3 are staged whole.
//     (Measured and dropped: H = 7 plus a 190-220-node window per wave for the tiles outside the tube, second pass from the
//     window -- the six extra accumulators and the restage code cost 24 VGPRs, the trilinear bodies spilled, and every format came
//     out 4-15 % slower than tube + gather: profiles/r03_exp6c_rgb2_windows.txt.)
//   * OPTIMISTIC BODY, ONE VOTE.  A tile is computed against the tube while every lane tracks the extremes of the two cell
//     differences of its pixels (4 VALU per pixel, in the units the body has in registers anyway: the table's `prev` values);
//     one vote per tile.  A tile that left the tube (or holds codes above 2^depth - 1) is computed again by the gather body
//     (taps from L1/L2).  LDS reads of an address outside the allocation return without a fault, so the optimistic pass is safe.
//   * PACKED FORMATS are runs of whole dwords per lane (48 bytes = 16 rgb24 pixels, 32 bytes = 8 rgba pixels, ...): three or two
//     16-byte accesses, samples picked out of the words by SDWA selects (code << 3 = table offset in one instruction, float ->
//     byte / half-word insertion in one instruction).  BGR orders run the same code: the lattice strides are permuted and the
//     tube is staged with R and B swapped, so "component k of memory" is axis k everywhere.  The fourth component is carried over.
//   * Arithmetic: the strict restatement (-ffp-contract=off, FFmpeg's scalar C order), bit-identical to the oracle.
//
// One translation unit per layout (Makefile: -DLUTR_R2_LAYOUT=0..7).
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <utility>

#include "lutr_internal.h"

#ifndef LUTR_R2_LAYOUT
#define LUTR_R2_LAYOUT 0
#endif
#ifndef LUTR_R2_WPB
#define LUTR_R2_WPB 16
#endif
#ifndef LUTR_R2_INPLACE
#define LUTR_R2_INPLACE 0         // 1: results overwrite the unit's input words and the rare gather path re-reads its unit (12 words
                                  // back, 4-pixel coordinate groups everywhere).  Parity green, but measured 2-9 % SLOWER than separate
                                  // output words (gbrp 605 vs 637, gbrp16le 383 vs 422 Gpx/s relative to the same reference kernel):
                                  // every write becomes a read-modify-write of a word whose other samples are still live input
#endif
#ifndef LUTR_R2_TB0
#define LUTR_R2_TB0 1             // tap batch of the computed-coordinate instances: 1 pixel (no spill) or 2
#endif
#ifndef LUTR_R2_NT
#define LUTR_R2_NT 1              // 1: non-temporal stores, 2: and loads
#endif

namespace lutr {
namespace r2 {

extern __shared__ __attribute__((aligned(16))) char smem[];

#define DEV __device__ __forceinline__

enum { LY_P8 = 0, LY_P16 = 1, LY_C3B = 2, LY_C3W = 3, LY_C4B0 = 4, LY_C4B1 = 5, LY_C4W0 = 6, LY_C4W1 = 7 };

// PX pixels of one row per lane and tile; NPL planes of NW dwords each; sample e of slot k of pixel i (slots = the three colour
// components in the order the body sees them: R, G, B for planar frames -- the launcher passes the planes in that order --
// and memory order for packed ones)
template <int LY> struct Lay;
template <> struct Lay<LY_P8>   { static constexpr int PX = 16, NPL = 3, NW = 4,  WIDE = 0, NC = 1, P0 = 0; };
template <> struct Lay<LY_P16>  { static constexpr int PX = 8,  NPL = 3, NW = 4,  WIDE = 1, NC = 1, P0 = 0; };
template <> struct Lay<LY_C3B>  { static constexpr int PX = 16, NPL = 1, NW = 12, WIDE = 0, NC = 3, P0 = 0; };
template <> struct Lay<LY_C3W>  { static constexpr int PX = 8,  NPL = 1, NW = 12, WIDE = 1, NC = 3, P0 = 0; };
template <> struct Lay<LY_C4B0> { static constexpr int PX = 8,  NPL = 1, NW = 8,  WIDE = 0, NC = 4, P0 = 0; };
template <> struct Lay<LY_C4B1> { static constexpr int PX = 8,  NPL = 1, NW = 8,  WIDE = 0, NC = 4, P0 = 1; };
template <> struct Lay<LY_C4W0> { static constexpr int PX = 4,  NPL = 1, NW = 8,  WIDE = 1, NC = 4, P0 = 0; };
template <> struct Lay<LY_C4W1> { static constexpr int PX = 4,  NPL = 1, NW = 8,  WIDE = 1, NC = 4, P0 = 1; };

DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DEV float med3(float a, float lo, float hi) { return __builtin_amdgcn_fmed3f(a, lo, hi); }
DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
DEV float vmin3(float a, float b, float c) { float o; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c)); return o; }
DEV float vmax3(float a, float b, float c) { float o; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c)); return o; }
DEV int lds_base() { return (int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem; }

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

struct Geom {
    int lw_log2;          // lanes across x per tile row (the other lanes go down)
    int uw, urows;        // units per row, rows
    int nsx, nry, ch, nrc, nchunks;
    int tab_entries;      // per-code coordinate table at LDS offset 0 (0: coordinates are computed)
    int max_code;         // codes above it are illegal for the table (10- / 12-bit data in 16-bit containers)
    int whole;            // the whole lattice is staged ((n+1)^3 nodes): no validity test;
    int whole_a, whole_b; //   node (r, g, b) at index r * whole_a + g * whole_b + b (strides padded against bank conflicts)
    int tube_h, tube_plane;
    int rev;              // memory order of the components is B, G, R: strides permuted, nodes staged with R and B swapped
    int three;            // three coordinate tables (one per channel: a prelut, or DOMAIN scales that differ), else one
    unsigned *queue, *stats;      // queue: device words {claims, waves done}, both 0 between launches (queue_leave, as lutr_tile2.hip)
    unsigned qbase;               // waves in the grid = the first chunk the counter hands out
};

struct Planes {
    const uint8_t *s[3];
    uint8_t       *d[3];
    unsigned ss[3], ds[3];
    unsigned long long sfs[3], dfs[3];
};

// ---------------------------------------------------------------- sample access inside the word vectors
// table byte offset (code << 3) of sample e: one SDWA shift
template <int WIDE> DEV unsigned code8(const uint32_t *w, int e, unsigned three)
{
    unsigned d;
    if constexpr (WIDE) {
        if ((e & 1) == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(d) : "v"(three), "v"(w[e >> 1]));
        else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(d) : "v"(three), "v"(w[e >> 1]));
    } else {
        switch (e & 3) {
        case 0: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(d) : "v"(three), "v"(w[e >> 2])); break;
        case 1: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(d) : "v"(three), "v"(w[e >> 2])); break;
        case 2: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(d) : "v"(three), "v"(w[e >> 2])); break;
        default: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(d) : "v"(three), "v"(w[e >> 2])); break;
        }
    }
    return d;
}
// the code itself as a float
template <int WIDE> DEV float codef(const uint32_t *w, int e)
{
    if constexpr (WIDE) return (float)((w[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
    else return (float)((w[e >> 2] >> ((e & 3) * 8)) & 0xffu);
}
// floor(v) (v >= 0) into sample e, the other samples of the word preserved; FIRST: the word's other samples are not live yet
template <int WIDE, bool KEEP> DEV void put(uint32_t *w, int e, float v)
{
    if constexpr (WIDE) {
        uint32_t &d = w[e >> 1];
        if ((e & 1) == 0) {
            if constexpr (KEEP) asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(v));
            else d = (uint32_t)v;
        } else asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(v));
    } else {
        uint32_t &d = w[e >> 2];
        switch (e & 3) {
        case 0:
            if constexpr (KEEP) asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(v));
            else d = (uint32_t)v;
            break;
        case 1: asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(v)); break;
        case 2: asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(v)); break;
        default: asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(v)); break;
        }
    }
}

template <int LY> struct Unit { uint32_t w[Lay<LY>::NPL][Lay<LY>::NW]; };

// word vector and sample index of slot k of pixel i
template <int LY> DEV int samp(int i, int k)
{
    using Y = Lay<LY>;
    if constexpr (Y::NC == 1) return i;
    else if constexpr (Y::NC == 3) return 3 * i + k;
    else return 4 * i + Y::P0 + k;
}
template <int LY> DEV int plane_of(int k) { return Lay<LY>::NC == 1 ? k : 0; }

template <int NW> DEV void ldw(uint32_t *w, const uint8_t *p)
{
    typedef unsigned nt4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < NW / 4; j++) {
#if LUTR_R2_NT >= 2
        const nt4 v = __builtin_nontemporal_load((const nt4 *)(p + 16 * j));
#else
        const nt4 v = *(const nt4 *)(p + 16 * j);
#endif
        w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
    }
}
template <int NW> DEV void stw(uint8_t *p, const uint32_t *w)
{
    typedef unsigned nt4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < NW / 4; j++) {
#if LUTR_R2_NT >= 1
        __builtin_nontemporal_store(nt4{w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]}, (nt4 *)(p + 16 * j));
#else
        *(nt4 *)(p + 16 * j) = nt4{w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]};
#endif
    }
}

// ---------------------------------------------------------------- coordinates
struct Crd { float p, d; };

template <int INTERP>
DEV Crd crd_compute(const LutConsts &L, float code)
{
    const float x = code * L.scale_f;
    const float s = fminf(x * L.sc[0], L.lut_max);       // codes and scales are >= 0: only the upper clip can bind
    Crd c;
    if constexpr (INTERP == LUTR_INTERP_NEAREST) {        // NEAR(x) = (int)(x + .5) with a double .5 (lutr_device.h near_f)
        const float fl = floorf(s);
        c.p = (s - fl >= .5f) ? fl + 1.0f : fl;
        c.d = 0.0f;
    } else { c.p = floorf(s); c.d = s - c.p; }
    return c;
}

// coordinates of channel `ch` in the general case: a prelut (lut3d's 1D shaper, folded by the host into one lattice coordinate per
// integer code and channel, LutConsts::pre) or per-channel DOMAIN scales
template <int INTERP>
DEV Crd crd_general(const LutConsts &L, int ch, float code)
{
    float s;
    if (L.pre) s = L.pre[ch * L.pre_stride + min((int)code, L.pre_stride - 1)];
    else s = fminf((code * L.scale_f) * L.sc[ch], L.lut_max);
    Crd c;
    if constexpr (INTERP == LUTR_INTERP_NEAREST) {
        const float fl = floorf(s);
        c.p = (s - fl >= .5f) ? fl + 1.0f : fl;
        c.d = 0.0f;
    } else { c.p = floorf(s); c.d = s - c.p; }
    return c;
}

typedef __attribute__((address_space(3))) const f2v *lds_f2p;
DEV Crd crd_table8(unsigned off)
{
    const f2v e = *(lds_f2p)(uintptr_t)off;
    return Crd{e.x, e.y};
}

// lattice addressing of one body: byte strides of the three slots and of a node, base
struct Addr {
    float f0, f1, f2, fc;         // LDS: byte address of c000 = (int) fma(p0, f0, fma(p1, f1, fma(p2, f2, fc)))
    int o0, o1, o2;               // byte steps of +1 along the slots
};

#ifndef LUTR_R2_TRI12
#define LUTR_R2_TRI12 0           // 1: trilinear stages 12-byte nodes too (wider tube, 16 LDS reads per pixel instead of 8)
#endif
#ifndef LUTR_R2_NODE16
#define LUTR_R2_NODE16 0          // 1: the 4-tap modes stage float4 nodes too (one ds_read_b128 per tap, 4 LDS cycles instead of 6; tube H = 7
                                  // instead of 8 at 33^3): rgb24 608 vs 618, gbrp 646 vs 649, sigma-16 frames -5 % (profiles/r03_exp25): the
                                  // LDS is not what these kernels wait for
#endif
// INTERP as a template argument: lut3d's 0 / 1 / 2, and R2_TET16 = tetrahedral on float4 nodes, for the whole-lattice mode (one ds_read_b128
// per tap, far fewer bank collisions between scattered taps; lutr_tile2.hip T2_TET16)
constexpr int R2_TET16 = 3;
template <int INTERP> struct NodeB {
    static constexpr int lds = ((INTERP == LUTR_INTERP_TRILINEAR && !LUTR_R2_TRI12) || INTERP == R2_TET16 || (INTERP != LUTR_INTERP_TRILINEAR && LUTR_R2_NODE16)) ? 16 : 12;
};

template <bool LDS, int NB> DEV f4 tap(const LutConsts &L, int a)
{
    if constexpr (LDS && NB == 16) return *(const __attribute__((address_space(3))) f4 *)(uintptr_t)(unsigned)a;
    else if constexpr (LDS) {
        const __attribute__((address_space(3))) float *p = (const __attribute__((address_space(3))) float *)(uintptr_t)(unsigned)a;
        f4 v; v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = 0.0f;
        return v;
    } else return *(const f4 *)((const char *)L.lat + a);
}

DEV float tlerp(float v0, float v1, float f) { return v0 + (v1 - v0) * f; }

struct Rgb3 { float c0, c1, c2; };

// One pixel in three phases, so that the LDS round trips of several pixels overlap (the machine scheduler is kept from
// sinking the reads back next to their uses by sched_barrier in tile_body): prepare (address, weights, tap offsets),
// load (4 / 8 / 1 taps), blend.
struct Prep { int a, oa, oz; float w0, w1, w2, w3; };     // trilinear: w0..w2 = the fractions of slots 0..2
template <int INTERP> struct Taps { f4 t[INTERP == LUTR_INTERP_TRILINEAR ? 8 : (INTERP == LUTR_INTERP_NEAREST ? 1 : 4)]; };

template <bool LDS, int INTERP>
DEV Prep px_prep(const Addr &A, const Crd &q0, const Crd &q1, const Crd &q2)
{
    Prep c;
    if constexpr (LDS) c.a = (int)fma_(q0.p, A.f0, fma_(q1.p, A.f1, fma_(q2.p, A.f2, A.fc)));
    else c.a = (int)q0.p * A.o0 + (int)q1.p * A.o1 + (int)q2.p * A.o2;
    c.oa = c.oz = 0;
    c.w0 = q0.d; c.w1 = q1.d; c.w2 = q2.d; c.w3 = 0.0f;
    if constexpr (INTERP == LUTR_INTERP_TETRAHEDRAL || INTERP == R2_TET16) {
        // the six branches of FFmpeg's tetrahedral form as (1-x) c000 + (x-y) cA + (y-z) cB + z c111 with the fractions sorted:
        // symmetric in the axes, ties only choose between taps of weight 0
        const float d0 = q0.d, d1 = q1.d, d2 = q2.d;
        const float x = fmaxf(fmaxf(d0, d1), d2), y = med3(d0, d1, d2), z = fminf(fminf(d0, d1), d2);
        const bool g01 = d0 > d1, g12 = d1 > d2, g02 = d0 > d2;
        const int z0 = A.o1 + A.o2, z1 = A.o0 + A.o2, z2 = A.o0 + A.o1;
        c.oa = (g01 && g02) ? A.o0 : (g12 ? A.o1 : A.o2);
        c.oz = (g12 && g02) ? z2 : (g01 ? z1 : z0);
        c.w0 = 1.0f - x; c.w1 = x - y; c.w2 = y - z; c.w3 = z;
    }
    return c;
}

template <bool LDS, int INTERP>
DEV Taps<INTERP> px_load(const LutConsts &L, const Addr &A, const Prep &c)
{
    constexpr int NB = LDS ? NodeB<INTERP>::lds : 16;
    Taps<INTERP> T;
    const int a = c.a;
    if constexpr (INTERP == LUTR_INTERP_NEAREST) T.t[0] = tap<LDS, NB>(L, a);
    else if constexpr (INTERP == LUTR_INTERP_TRILINEAR) {
        // t[4 i0 + 2 i1 + i2]: corner (i0, i1, i2) along the slots
        const int a0 = a + A.o0, a1 = a + A.o1, a01 = a0 + A.o1;
        T.t[0] = tap<LDS, NB>(L, a); T.t[1] = tap<LDS, NB>(L, a + A.o2); T.t[2] = tap<LDS, NB>(L, a1); T.t[3] = tap<LDS, NB>(L, a1 + A.o2);
        T.t[4] = tap<LDS, NB>(L, a0); T.t[5] = tap<LDS, NB>(L, a0 + A.o2); T.t[6] = tap<LDS, NB>(L, a01); T.t[7] = tap<LDS, NB>(L, a01 + A.o2);
    } else {
        T.t[0] = tap<LDS, NB>(L, a); T.t[1] = tap<LDS, NB>(L, a + c.oa); T.t[2] = tap<LDS, NB>(L, a + c.oz);
        T.t[3] = tap<LDS, NB>(L, a + A.o0 + A.o1 + A.o2);
    }
    return T;
}

// blended lattice value times M, truncated (and clipped unless UNIT).  `swap` (blue-first memory order): trilinear's lerp order
// follows r, g, b, and the gather body's global nodes are {r, g, b} while slot 0 is blue.
template <bool LDS, int INTERP, bool UNIT>
DEV Rgb3 px_blend(const LutConsts &L, const Prep &c, const Taps<INTERP> &T, bool swap)
{
    Rgb3 v;
    if constexpr (INTERP == LUTR_INTERP_NEAREST) { v.c0 = T.t[0].x; v.c1 = T.t[0].y; v.c2 = T.t[0].z; }
    else if constexpr (INTERP == LUTR_INTERP_TRILINEAR) {
        // FFmpeg's interp_trilinear lerps along r first, then g, then b (each lerp rounds, so the order is part of the contract)
#define R2_TRI(ch, out, A00, A01, B00, B01, C00, C01, D00, D01, F0, F2) \
        { \
            const float c00 = tlerp(T.t[A00].ch, T.t[A01].ch, F0), c10 = tlerp(T.t[B00].ch, T.t[B01].ch, F0); \
            const float c01 = tlerp(T.t[C00].ch, T.t[C01].ch, F0), c11 = tlerp(T.t[D00].ch, T.t[D01].ch, F0); \
            const float c0 = tlerp(c00, c10, c.w1), c1 = tlerp(c01, c11, c.w1); \
            out = tlerp(c0, c1, F2); \
        }
        if (!swap) {          // slot 0 is red: c000 c100 | c010 c110 | c001 c101 | c011 c111
            R2_TRI(x, v.c0, 0, 4, 2, 6, 1, 5, 3, 7, c.w0, c.w2) R2_TRI(y, v.c1, 0, 4, 2, 6, 1, 5, 3, 7, c.w0, c.w2)
            R2_TRI(z, v.c2, 0, 4, 2, 6, 1, 5, 3, 7, c.w0, c.w2)
        } else {              // slot 2 is red
            R2_TRI(x, v.c0, 0, 1, 2, 3, 4, 5, 6, 7, c.w2, c.w0) R2_TRI(y, v.c1, 0, 1, 2, 3, 4, 5, 6, 7, c.w2, c.w0)
            R2_TRI(z, v.c2, 0, 1, 2, 3, 4, 5, 6, 7, c.w2, c.w0)
        }
#undef R2_TRI
    } else {
        v.c0 = c.w0 * T.t[0].x + c.w1 * T.t[1].x + c.w2 * T.t[2].x + c.w3 * T.t[3].x;
        v.c1 = c.w0 * T.t[0].y + c.w1 * T.t[1].y + c.w2 * T.t[2].y + c.w3 * T.t[3].y;
        v.c2 = c.w0 * T.t[0].z + c.w1 * T.t[1].z + c.w2 * T.t[2].z + c.w3 * T.t[3].z;
    }
    if constexpr (!LDS) {        // global nodes are {r, g, b}: blue-first memory order takes them the other way round
        const float r = v.c0, b = v.c2;
        v.c0 = swap ? b : r; v.c2 = swap ? r : b;
    }
    v.c0 *= L.maxf; v.c1 *= L.maxf; v.c2 *= L.maxf;
    Rgb3 o;
    if constexpr (UNIT) { o.c0 = truncf(v.c0); o.c1 = truncf(v.c1); o.c2 = truncf(v.c2); }
    else { o.c0 = med3(truncf(v.c0), 0.0f, L.maxf); o.c1 = med3(truncf(v.c1), 0.0f, L.maxf); o.c2 = med3(truncf(v.c2), 0.0f, L.maxf); }
    return o;
}

// ---------------------------------------------------------------- one tile
// Per-lane extremes of the two cell differences of neighbouring slots -- (g - r, b - g), or blue first (g - b, r - g): the tube
// is symmetric.  They come out of the table's `prev` values, whatever the taps read.
struct Acc { float amin, amax, bmin, bmax; };

// `out` may be `in` itself (LUTR_R2_INPLACE): a pixel's three samples are read before they are written, every write preserves the
// other samples of its word, and groups run in program order.
template <bool LDS, int LY, int INTERP, int TAB, bool UNIT>
DEV Acc tile_body(const LutConsts &L, const Addr &A, const Geom &TG, Unit<LY> &in, Unit<LY> &out)
{
    using Y = Lay<LY>;
    unsigned three;
    asm volatile("v_mov_b32 %0, 3" : "=v"(three));
    Acc acc;
    acc.amin = acc.bmin = 1e9f; acc.amax = acc.bmax = -1e9f;
    const bool swap = TG.rev != 0;
    // pixels whose coordinate reads are issued together: 4 where the registers allow (units of 8 words), else 2 -- a unit of 12
    // words in, 12 prefetched and 12 out leaves the tetrahedral body no room for 24 coordinates (one spilled register = scratch)
    constexpr int GP = ((!LUTR_R2_INPLACE && Y::NPL * Y::NW >= 12) || (INTERP == LUTR_INTERP_TRILINEAR && Y::NW >= 12) || Y::PX < 4) ? 2 : 4;
    // pixels whose taps are in flight together (the instances that compute their coordinates -- 14- and 16-bit data -- spilled 3-6 registers
    // to scratch with two)
    constexpr int TB = INTERP == LUTR_INTERP_TRILINEAR ? 1 : (INTERP == LUTR_INTERP_NEAREST ? GP : ((TAB == 0 && LUTR_R2_TB0 == 1 && Y::NPL * Y::NW >= 12) ? 1 : 2));
#pragma unroll
    for (int g = 0; g < Y::PX / GP; g++) {
        Crd q[GP][3];
#pragma unroll
        for (int t = 0; t < GP; t++)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int i = g * GP + t, e = samp<LY>(i, k);
                // the gather body also takes codes above 2^depth - 1 (16-bit containers), which the table does not cover and which
                // lut3d does not clip before scaling: it computes its coordinates
                if constexpr (TAB == 1 && LDS) q[t][k] = crd_table8(code8<Y::WIDE>(in.w[plane_of<LY>(k)], e, three));
                else if constexpr (TAB == 3 && LDS) {
                    // slot k's channel: planar frames arrive as (R, G, B), packed ones in memory order
                    const unsigned tb = (unsigned)(swap ? 2 - k : k) * (unsigned)TG.tab_entries * 8u;
                    q[t][k] = crd_table8(code8<Y::WIDE>(in.w[plane_of<LY>(k)], e, three) + tb);
                } else if constexpr (TAB == 3) q[t][k] = crd_general<INTERP>(L, swap ? 2 - k : k, codef<Y::WIDE>(in.w[plane_of<LY>(k)], e));
                else q[t][k] = crd_compute<INTERP>(L, codef<Y::WIDE>(in.w[plane_of<LY>(k)], e));
            }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LDS) {
#pragma unroll
            for (int t = 0; t < GP; t += 2) {
                const float ha0 = q[t][1].p - q[t][0].p, hb0 = q[t][2].p - q[t][1].p;
                const float ha1 = q[t + 1][1].p - q[t + 1][0].p, hb1 = q[t + 1][2].p - q[t + 1][1].p;
                acc.amin = vmin3(acc.amin, ha0, ha1); acc.amax = vmax3(acc.amax, ha0, ha1);
                acc.bmin = vmin3(acc.bmin, hb0, hb1); acc.bmax = vmax3(acc.bmax, hb0, hb1);
            }
        }
#pragma unroll
        for (int tb = 0; tb < GP; tb += TB) {
            Prep pc[TB];
            Taps<INTERP> tp[TB];
#pragma unroll
            for (int t = 0; t < TB; t++) {
                pc[t] = px_prep<LDS, INTERP>(A, q[tb + t][0], q[tb + t][1], q[tb + t][2]);
                tp[t] = px_load<LDS, INTERP>(L, A, pc[t]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TB; t++) {
                const int i = g * GP + tb + t;
                const Rgb3 o = px_blend<LDS, INTERP, UNIT>(L, pc[t], tp[t], swap);
                constexpr bool keep = Y::NC == 4 || LUTR_R2_INPLACE;   // the word holds alpha, or input samples still to be read
                put<Y::WIDE, keep>(out.w[plane_of<LY>(0)], samp<LY>(i, 0), o.c0);
                put<Y::WIDE, keep>(out.w[plane_of<LY>(1)], samp<LY>(i, 1), o.c1);
                put<Y::WIDE, keep>(out.w[plane_of<LY>(2)], samp<LY>(i, 2), o.c2);
            }
        }
        // keep the groups in program order (instruction selection would hoist every group's reads to the top)
        asm volatile("" : "+v"(acc.amin), "+v"(acc.amax), "+v"(acc.bmin), "+v"(acc.bmax));
        {
            constexpr int SH = Y::WIDE ? 1 : 2;
            const int e_lo = samp<LY>(g * GP, 0), e_hi = samp<LY>(g * GP + GP - 1, 2);
#pragma unroll
            for (int p = 0; p < Y::NPL; p++)
#pragma unroll
                for (int k = 0; k < Y::NW; k++)
                    if (k >= (e_lo >> SH) && k <= (e_hi >> SH)) asm volatile("" : "+v"(out.w[p][k]));
        }
    }
    return acc;
}

// ---------------------------------------------------------------- work distribution (as lutr_tile2.hip)
DEV bool chunk_at(const Geom &TG, unsigned c, int &fr, int &sx, int &ry, int &rem)
{
    if (c >= (unsigned)TG.nchunks) return false;
    const int per_frame = TG.nrc * TG.nsx;
    fr = (int)c / per_frame;
    const int r = (int)c - fr * per_frame;
    const int rc = r / TG.nsx;
    sx = r - rc * TG.nsx;
    ry = rc * TG.ch;
    rem = min(TG.ch, TG.nry - ry);
    return true;
}
// The two-level chunk queue of lutr_tile2.hip (claim_chunk there): a wave's first chunk is its id; after that it draws a ticket
// from the workgroup's LDS counter, ticket 16 j + slot is chunk base[j] + slot, and the wave that draws slot 0 fetches
// base[j] = atomicAdd(queue, 16) and publishes it.  LDS words at `wgq_off`: ticket at +0, base[8] at +32, ready[8] at +64.
typedef __attribute__((address_space(3))) volatile unsigned *lds_vup;
constexpr int kWgq = 128;
DEV bool claim_chunk(const Geom &TG, int lane, int wgq_off, int &fr, int &sx, int &ry, int &rem, bool &first)
{
    unsigned c = 0;
    if (first) {          // (a wave whose id is not a chunk has no work: the counter starts behind the ids; runs before the LDS words are initialised)
        first = false;
        c = (unsigned)((int)(blockIdx.x * LUTR_R2_WPB) + uni((int)(threadIdx.x >> 6)));
        return chunk_at(TG, c, fr, sx, ry, rem);
    }
    const lds_vup q = (lds_vup)(uintptr_t)(unsigned)(lds_base() + wgq_off);
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned *)q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    t = (unsigned)uni((int)t);
    const unsigned j = t >> 4, slot = t & 15u, r = j & 7u;
    if (slot == 0) {
        if (lane == 0) {
            c = atomicAdd(TG.queue, 16u) + TG.qbase;
            q[8 + r] = c;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            q[16 + r] = j + 1u;
        }
        c = (unsigned)uni((int)c);
    } else {
        while ((unsigned)uni((int)q[16 + r]) != j + 1u) __builtin_amdgcn_s_sleep(2);
        c = (unsigned)uni((int)q[8 + r]) + slot;
    }
    return chunk_at(TG, c, fr, sx, ry, rem);
}

// a wave that will claim no more; the last one zeroes the two words for the next launch (no memset node per launch)
DEV void queue_leave(const Geom &TG, int lane, int wgq_off)
{
    // two levels, like the claims: the waves of a workgroup count themselves out in LDS (word 24 of the allocator's block), the last one
    // reports the workgroup -- 4096 atomics on one address at the end of a short launch cost it 15 us
    if (lane == 0) {
        const lds_vup q = (lds_vup)(uintptr_t)(unsigned)(lds_base() + wgq_off);
        const unsigned left = __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned *)(q + 24), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (left == (unsigned)LUTR_R2_WPB - 1u) {
            const unsigned done = atomicAdd(TG.queue + 1, 1u);
            if (done == gridDim.x - 1u) {
                __hip_atomic_store(TG.queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(TG.queue + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int LY, int INTERP, int TAB, bool UNIT>
__global__ __launch_bounds__(64 * LUTR_R2_WPB, 4)
void k_rgb_tube(LutConsts L, Planes P, FrameGeom G, Geom TG)
{
    using Y = Lay<LY>;
    constexpr int NBL = NodeB<INTERP>::lds;
    const int lane = threadIdx.x & 63;
    // ---- LDS: [coordinate table][tube or whole lattice]
    if constexpr (TAB == 1) {
        for (int q = threadIdx.x; q < TG.tab_entries; q += 64 * LUTR_R2_WPB) {
            const Crd c = crd_compute<INTERP>(L, fminf((float)q, L.maxf));
            *(float2 *)(smem + q * 8) = make_float2(c.p, c.d);
        }
    } else if constexpr (TAB == 3) {
        for (int q = threadIdx.x; q < 3 * TG.tab_entries; q += 64 * LUTR_R2_WPB) {
            const int ch = q / TG.tab_entries, code = q - ch * TG.tab_entries;
            const Crd c = crd_general<INTERP>(L, ch, (float)code);
            *(float2 *)(smem + q * 8) = make_float2(c.p, c.d);
        }
    }
    const int wgq_off = (TAB == 3 ? 3 : 1) * TG.tab_entries * 8;                // the workgroup's chunk allocator (claim_chunk)
    if (threadIdx.x < 24) ((unsigned *)(smem + wgq_off))[threadIdx.x + (threadIdx.x ? 7 : 0)] = 0u;
    const int lat_off = wgq_off + kWgq;
    const int nb = 2 * TG.tube_h + 3;
    {
        char *dst = smem + lat_off;
        const int n1 = L.n1, nmax = L.n1 - 1;
        const int nodes = TG.whole ? n1 * n1 * n1 : n1 * TG.tube_plane;
        for (int i = threadIdx.x; i < nodes; i += 64 * LUTR_R2_WPB) {
            int src, o = i;
            if (TG.whole) {
                src = i;
                const int r = i / (n1 * n1), rem = i - r * n1 * n1, g = rem / n1;
                o = r * TG.whole_a + g * TG.whole_b + (rem - g * n1);
            } else {
                // node (ir, ig, ib) = lattice (r, g = r + ig - H - 1, b = g + ib - H - 1), clamped (a clamped node is never read by a valid pixel)
                const int ir = i / TG.tube_plane, rem = i - ir * TG.tube_plane, ig = min(rem / nb, nb - 1), ib = rem - ig * nb;
                const int gq = ir + ig - TG.tube_h - 1;
                const int g = min(max(gq, 0), nmax), b = min(max(gq + ib - TG.tube_h - 1, 0), nmax);
                src = (ir * n1 + g) * n1 + b;
            }
            float4 v = L.lat[src];
            if (TG.rev) { const float t = v.x; v.x = v.z; v.z = t; }
            if constexpr (NBL == 16) ((float4 *)dst)[o] = v;
            else { float *q = (float *)(dst + 12 * o); q[0] = v.x; q[1] = v.y; q[2] = v.z; }
        }
    }
    __syncthreads();
    // ---- addressing
    Addr AL, AG;
    {
        int sr, sg, sb, base;       // node steps of r, g, b
        if (TG.whole) { sr = TG.whole_a; sg = TG.whole_b; sb = 1; base = 0; }
        else { sr = TG.tube_plane - nb; sg = nb - 1; sb = 1; base = (TG.tube_h + 1) * nb + TG.tube_h + 1; }
        AL.o0 = NBL * (TG.rev ? sb : sr); AL.o1 = NBL * sg; AL.o2 = NBL * (TG.rev ? sr : sb);
        AL.f0 = (float)AL.o0; AL.f1 = (float)AL.o1; AL.f2 = (float)AL.o2;
        AL.fc = (float)(lds_base() + lat_off + NBL * base);
        // (the slots' steps in the tube: with blue first the tube's r axis is slot 2 -- its coordinates are (r, g - r, b - g) either way)
        const int gr = 16 * L.n1 * L.n1, gg = 16 * L.n1, gb = 16;
        AG.o0 = TG.rev ? gb : gr; AG.o1 = gg; AG.o2 = TG.rev ? gr : gb;
        AG.f0 = AG.f1 = AG.f2 = AG.fc = 0.0f;
    }
    int fr, sx, ry, rem;
    bool first = true;
    if (!claim_chunk(TG, lane, wgq_off, fr, sx, ry, rem, first)) { queue_leave(TG, lane, wgq_off); return; }
    const int lw = 1 << TG.lw_log2, lh_log2 = 6 - TG.lw_log2;
    const int lx = lane & (lw - 1), ly = lane >> TG.lw_log2;
    constexpr int UB = Y::NW * 4;                       // bytes of a unit per plane

    struct TilePos { const uint8_t *s[Y::NPL]; uint8_t *d[Y::NPL]; int xlim, ylim; };
    auto pos_at = [&](int f, int tsx, int try_) {
        TilePos q;
        const long long row0 = G.row0 + (try_ << lh_log2);
#pragma unroll
        for (int p = 0; p < Y::NPL; p++) {
            q.s[p] = P.s[p] + f * P.sfs[p] + row0 * (long long)P.ss[p] + (long long)tsx * lw * UB;
            q.d[p] = P.d[p] + f * P.dfs[p] + row0 * (long long)P.ds[p] + (long long)tsx * lw * UB;
        }
        q.xlim = TG.uw - 1 - tsx * lw;
        q.ylim = TG.urows - 1 - (try_ << lh_log2);
        return q;
    };
    auto pos_down = [&](TilePos &q) {
#pragma unroll
        for (int p = 0; p < Y::NPL; p++) { q.s[p] += (unsigned)((int)P.ss[p] << lh_log2); q.d[p] += (unsigned)((int)P.ds[p] << lh_log2); }
        q.ylim -= 1 << lh_log2;
    };
    auto load_tile = [&](Unit<LY> &dst, const TilePos &q) {
        const unsigned lxc = (unsigned)min(lx, q.xlim), lyc = (unsigned)min(ly, q.ylim);
#pragma unroll
        for (int p = 0; p < Y::NPL; p++) ldw<Y::NW>(dst.w[p], q.s[p] + (__umul24(lyc, P.ss[p]) + lxc * UB));
    };

    unsigned st_tiles = 0, st_tube = 0, st_gather = 0;
    TilePos np = pos_at(fr, sx, ry);
    Unit<LY> nxt;
    load_tile(nxt, np);
    for (bool more = true; more;) {
        Unit<LY> in = nxt;
        const TilePos cp = np;
        if (--rem > 0) { ry++; pos_down(np); }
        else {
            more = claim_chunk(TG, lane, wgq_off, fr, sx, ry, rem, first);
            if (more) np = pos_at(fr, sx, ry);
        }
        load_tile(nxt, np);

        // codes the table does not cover (10- / 12-bit data in 16-bit containers): the gather body clamps
        bool lane_ok = true;
        if constexpr (TAB != 0 && Y::WIDE) {
            uint32_t acc = 0;
#pragma unroll
            for (int p = 0; p < Y::NPL; p++)
#pragma unroll
                for (int k = 0; k < Y::NW; k++) acc |= in.w[p][k];
            const uint32_t hi = ~(uint32_t)TG.max_code & 0xffffu;
            lane_ok = (acc & (hi | (hi << 16))) == 0u;
        }
        st_tiles++;
#if LUTR_R2_INPLACE
        Unit<LY> &out = in;
#else
        Unit<LY> out;
        if constexpr (Y::NC == 4) out = in;
#endif
        {
            // optimistic pass for every lane (a lane with illegal codes or colours outside the tube reads wherever its numbers
            // point -- LDS reads cannot fault -- and its result is thrown away)
            const Acc acc = tile_body<true, LY, INTERP, TAB, UNIT>(L, AL, TG, in, out);
            const float lim = (float)TG.tube_h;
            if (!TG.whole) lane_ok = lane_ok && acc.amin >= -lim && acc.amax <= lim && acc.bmin >= -lim && acc.bmax <= lim;
        }
        // Only the lanes that need it run the gather body (a divergent branch: the other lanes keep their result).  Its
        // instructions still issue once for the wave, but its memory requests -- what a gather costs -- shrink to those lanes:
        // a tile crossed by a saturated edge costs about two bodies instead of five.
        if (__all(lane_ok)) st_tube++;
        else {
            st_gather++;
            if (!lane_ok) {
#if LUTR_R2_INPLACE
                // the optimistic pass has overwritten the unit: read it again (its own bytes: nothing has been stored over them yet,
                // also when the caller works in place)
                {
                    const unsigned lxc = (unsigned)min(lx, cp.xlim), lyc = (unsigned)min(ly, cp.ylim);
#pragma unroll
                    for (int p = 0; p < Y::NPL; p++) ldw<Y::NW>(in.w[p], cp.s[p] + (__umul24(lyc, P.ss[p]) + lxc * UB));
                }
#else
                if constexpr (Y::NC == 4) out = in;
#endif
                (void)tile_body<false, LY, INTERP, TAB, UNIT>(L, AG, TG, in, out);
            }
        }
        {
            const unsigned lxc = (unsigned)min(lx, cp.xlim), lyc = (unsigned)min(ly, cp.ylim);
#pragma unroll
            for (int p = 0; p < Y::NPL; p++) stw<Y::NW>(cp.d[p] + (__umul24(lyc, P.ds[p]) + lxc * UB), out.w[p]);
        }
    }
    queue_leave(TG, lane, wgq_off);
    if (TG.stats && lane == 0) {
        atomicAdd(&TG.stats[0], st_tiles); atomicAdd(&TG.stats[2], st_gather); atomicAdd(&TG.stats[12], st_tube);
    }
}

}  // namespace r2

// ================================================================= launcher
namespace {

int device_cus()
{
    static const int cus = [] {
        int dev = 0, n = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        return n > 0 ? n : 256;
    }();
    return cus;
}

bool allow_lds(const void *kernel, size_t bytes)
{
    static std::set<std::pair<int, const void *>> done;      // the attribute is per device
    static std::mutex mu;
    if (bytes <= 65536) return true;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    done.insert({dev, kernel});
    return true;
}

// see tube_plane_stride in lutr_tile2.hip: node index = pr * A + pg * B + pb, no collision mod 32 (mod 16 for the 16-byte nodes of
// trilinear, read with ds_read_b128) for steps of +-1 (+-2 if possible)
int tube_plane_stride(int nb, int node)
{
    const int mod = node == 16 ? 16 : 32;
    int best = nb * nb, best_bad = 1 << 30;
    for (int pad = 0; pad < 12; pad++) {
        const int plane = nb * nb + pad, A = plane - nb, B = nb - 1;
        int bad = 0;
        for (int dr = -2; dr <= 2; dr++)
            for (int dg = -2; dg <= 2; dg++)
                for (int db = -2; db <= 2; db++) {
                    if (!dr && !dg && !db) continue;
                    if (((dr * A + dg * B + db) % mod + mod) % mod == 0) bad += (abs(dr) <= 1 && abs(dg) <= 1 && abs(db) <= 1) ? 100 : 1;
                }
        if (bad < best_bad) { best_bad = bad; best = plane; }
        if (!bad) break;
    }
    return best;
}
// whole-lattice mode: row and plane strides of the (n+1)^3 copy, padded the same way (lutr_tile2.hip whole_strides); bytes, or 0 if none fits
long long whole_strides(int n1, int node, long long room, int *A, int *B)
{
    const int mod = node == 16 ? 16 : 32;
    long long best_bytes = 0;
    int best_bad = 1 << 30;
    for (int pb = 0; pb < 4; pb++)
        for (int pa = 0; pa < 16; pa++) {
            const int b = n1 + pb, a = n1 * b + pa;
            const long long bytes = (long long)n1 * a * node;
            if (bytes > room) continue;
            int bad = 0;
            for (int dr = -2; dr <= 2; dr++)
                for (int dg = -2; dg <= 2; dg++)
                    for (int db = -2; db <= 2; db++) {
                        if (!dr && !dg && !db) continue;
                        if (((dr * a + dg * b + db) % mod + mod) % mod == 0) bad += (abs(dr) <= 1 && abs(dg) <= 1 && abs(db) <= 1) ? 100 : 1;
                    }
            if (bad < best_bad || (bad == best_bad && bytes < best_bytes)) { best_bad = bad; best_bytes = bytes; *A = a; *B = b; }
        }
    return best_bytes;
}
// (17-node rows of 16-byte nodes collide on every g step whatever the plane stride; skipping H = 7 for H = 6 there measured WORSE --
// gbrp trilinear 410 -> 379, rgb24 407 -> 369 Gpx/s: the wider tube is worth more than the conflicts cost)

}  // namespace

#define R2_CAT_(a) launch_rgb_tube_ly##a
#define R2_CAT(a) R2_CAT_(a)
#define R2_ENTRY R2_CAT(LUTR_R2_LAYOUT)

// Planes in SLOT order: planar callers pass (R, G, B) = gbrp planes (2, 0, 1); packed callers pass the one buffer in [0].
// `rev`: packed memory order is B, G, R.  Returns nullptr when this kernel cannot take the call (the caller falls back).
const char *R2_ENTRY(hipStream_t st, const LutConsts &L, const PlaneSet &P, const FrameGeom &G, int depth, int mode, int rev,
                     unsigned *stats, unsigned *queue)
{
    using namespace r2;
    constexpr int LY = LUTR_R2_LAYOUT;
    using Y = Lay<LY>;
    // one coordinate table serves the three channels unless the file has a prelut (.csp shaper) or DOMAIN scales that differ:
    // then each channel gets its own (8- and 10-bit data; deeper containers compute their coordinates and take neither)
    const bool three = !(L.sc[0] == L.sc[1] && L.sc[1] == L.sc[2]) || L.pre != nullptr;
    if (mode != LUTR_INTERP_NEAREST && mode != LUTR_INTERP_TRILINEAR && mode != LUTR_INTERP_TETRAHEDRAL) return nullptr;
    if ((depth > 8) != (Y::WIDE != 0) || G.w % Y::PX) return nullptr;
    for (int p = 0; p < Y::NPL; p++)
        if (P.sfs[p] < 0 || P.dfs[p] < 0 || P.ss[p] <= 0 || P.ds[p] <= 0 || P.ss[p] >= (1 << 24) || P.ds[p] >= (1 << 24) ||
            ((uintptr_t)P.s[p] | (uintptr_t)P.d[p] | (uintptr_t)P.ss[p] | (uintptr_t)P.ds[p] | (uintptr_t)P.sfs[p] | (uintptr_t)P.dfs[p]) & 15)
            return nullptr;
    const bool tab = three ? depth <= 10 : depth <= 12;
    if (three && !tab) return nullptr;
    if (!Y::WIDE && !tab) return nullptr;
    Geom tg;
    const int uw = G.w / Y::PX;
    // lanes across x: 32 (x 2 rows) for the 16-bit containers (memory side: long runs), 8 (x 8 rows) for the 8-bit ones (VALU side:
    // compact tiles see fewer colours and pass the vote more often -- rgb24 617 / 609 / 569, rgba 562 / 561 / 550 Gpx/s at 8 / 16 / 32
    // lanes, gbrp10le 425 / 442 / 462, profiles/r03_exp27) -- unless another shape wastes 2 % fewer lanes at the frame's edges
    int best = Y::WIDE ? 5 : 3;
    double best_eff = -1.0;
    const int order[2][4] = {{3, 4, 5, 6}, {5, 4, 6, 3}};
    for (int l : order[Y::WIDE ? 1 : 0]) {
        const int lw = 1 << l, lh = 64 >> l;
        const double eff = ((double)uw / (((uw + lw - 1) / lw) * lw)) * ((double)G.rows / (((G.rows + lh - 1) / lh) * lh));
        if (eff > best_eff + 0.02) { best_eff = eff; best = l; }
    }
    if (const char *e = getenv("LUTR_LW_LOG2")) { const int c = atoi(e); if (c >= 2 && c <= 6) best = c; }
    tg.lw_log2 = best; tg.uw = uw; tg.urows = G.rows;
    tg.nsx = (uw + (1 << best) - 1) >> best;
    tg.nry = (G.rows + (64 >> best) - 1) / (64 >> best);
    const int max_waves = device_cus() * LUTR_R2_WPB;
    const int tile_px = Y::PX * 64;
    // 4096 pixels per claim, 2048 when a wave gets fewer than 64 tiles (the two-level queue makes small chunks free, lutr_tile2.hip)
    int ch = (4096 + tile_px - 1) / tile_px;
    if ((long long)G.nframes * tg.nsx * tg.nry < 64ll * max_waves) ch = (2048 + tile_px - 1) / tile_px;
    if (ch < 1) ch = 1;
    if (const char *e = getenv("LUTR_CHUNK")) { const int c = atoi(e); if (c >= 1 && c <= 256) ch = c; }
    while (ch > 1 && (long long)G.nframes * tg.nsx * ((tg.nry + ch - 1) / ch) < max_waves / 4) ch >>= 1;
    tg.ch = ch; tg.nrc = (tg.nry + ch - 1) / ch; tg.nchunks = G.nframes * tg.nrc * tg.nsx;
    tg.tab_entries = tab ? (1 << depth) : 0;
    tg.max_code = (1 << depth) - 1;
    tg.rev = rev;
    tg.three = three ? 1 : 0;
    int node = ((mode == LUTR_INTERP_TRILINEAR && !LUTR_R2_TRI12) || (mode != LUTR_INTERP_TRILINEAR && LUTR_R2_NODE16)) ? 16 : 12;
    const long long room = 163840 - (long long)(three ? 3 : 1) * tg.tab_entries * 8 - r2::kWgq;
    // whole-lattice mode, strides padded; tetrahedral on float4 nodes when those fit too (R2_TET16)
    long long whole_bytes = 0;
    bool whole16 = false;
    tg.whole = 0; tg.whole_a = L.n1 * L.n1; tg.whole_b = L.n1;
    if (!getenv("LUTR_NO_WHOLE")) {
        int a, b;
        // (table variants only: the instances that compute their coordinates -- 14- and 16-bit data -- have no registers for float4 taps)
        if (node == 12 && mode == LUTR_INTERP_TETRAHEDRAL && tab && !getenv("LUTR_NO_WHOLE16")) {
            const long long bytes = whole_strides(L.n1, 16, room, &a, &b);
            if (bytes) { tg.whole = 1; tg.whole_a = a; tg.whole_b = b; whole_bytes = bytes; whole16 = true; node = 16; }
        }
        if (!tg.whole) {
            const long long bytes = whole_strides(L.n1, node, room, &a, &b);
            if (bytes) { tg.whole = 1; tg.whole_a = a; tg.whole_b = b; whole_bytes = bytes; }
        }
    }
    tg.tube_h = 0; tg.tube_plane = 0;
    long long lat_bytes = whole_bytes;
    if (!tg.whole) {
        int h = L.n1 - 2;                               // |differences| never exceed n - 1
        if (const char *e = getenv("LUTR_TUBE_H")) h = atoi(e);
        for (; h >= 2; h--) {
            const int nb = 2 * h + 3, plane = tube_plane_stride(nb, node);
            const long long bytes = (long long)L.n1 * plane * node;
            if (bytes <= room) { tg.tube_h = h; tg.tube_plane = plane; lat_bytes = bytes; break; }
        }
        if (tg.tube_h < 2) return nullptr;
    }
    tg.queue = queue; tg.stats = stats;
    const int waves = tg.nchunks < max_waves ? tg.nchunks : max_waves;
    const dim3 grid((waves + LUTR_R2_WPB - 1) / LUTR_R2_WPB), block(64 * LUTR_R2_WPB);
    const size_t lds = (size_t)(three ? 3 : 1) * tg.tab_entries * 8 + kWgq + (size_t)lat_bytes;
    Planes TP;
    for (int i = 0; i < 3; i++) {
        const int p = i < Y::NPL ? i : 0;
        TP.s[i] = P.s[p]; TP.d[i] = P.d[p];
        TP.ss[i] = (unsigned)P.ss[p]; TP.ds[i] = (unsigned)P.ds[p];
        TP.sfs[i] = (unsigned long long)P.sfs[p]; TP.dfs[i] = (unsigned long long)P.dfs[p];
    }
    if (getenv("LUTR_DEBUG"))
        fprintf(stderr, "[lutr r2] layout %d nsx %d nry %d chunk %d chunks %d blocks %u lds %zu tab %d whole %d tube h %d plane %d rev %d\n",
                LY, tg.nsx, tg.nry, tg.ch, tg.nchunks, grid.x, lds, tg.tab_entries, tg.whole, tg.tube_h, tg.tube_plane, rev);
    tg.qbase = grid.x * LUTR_R2_WPB;      // (the counter is at zero: the previous launch left it so)
    const bool unit = L.unit != 0;

#define R2_LAUNCH(I, T, U, NAME) \
    do { \
        auto kern = k_rgb_tube<LY, I, T, U>; \
        if (!allow_lds((const void *)kern, lds)) return nullptr; \
        hipLaunchKernelGGL(kern, grid, block, lds, st, L, TP, G, tg); \
        return tg.whole ? NAME "+whole-lattice" : NAME "+tube"; \
    } while (0)
#define R2_STR_(x) #x
#define R2_STR(x) R2_STR_(x)
#define R2_NAME(I, SUF) "k_rgb_tube<ly" R2_STR(LUTR_R2_LAYOUT) "," #I SUF ">"
#define R2_MODE_(I, K, SUF) \
        if (tab) { \
            if constexpr (LY != LY_C3W && LY != LY_C4W0 && LY != LY_C4W1) { \
                if (three) R2_LAUNCH(K, 3, false, R2_NAME(I, ",tab3" SUF)); \
                if (unit) R2_LAUNCH(K, 1, true, R2_NAME(I, ",tab,unit" SUF)); \
                R2_LAUNCH(K, 1, false, R2_NAME(I, ",tab" SUF)); \
            } \
        } else { \
            if constexpr (Y::WIDE) { \
                if (unit) R2_LAUNCH(K, 0, true, R2_NAME(I, ",unit" SUF)); \
                R2_LAUNCH(K, 0, false, R2_NAME(I, "" SUF)); \
            } \
        }
#define R2_MODE(I) if (mode == I) { R2_MODE_(I, I, "") }
    if (mode == LUTR_INTERP_TETRAHEDRAL && whole16) {
        if constexpr (LY != LY_C3W && LY != LY_C4W0 && LY != LY_C4W1) {
            if (three) R2_LAUNCH(R2_TET16, 3, false, R2_NAME(2, ",tab3,n16"));
            if (unit) R2_LAUNCH(R2_TET16, 1, true, R2_NAME(2, ",tab,unit,n16"));
            R2_LAUNCH(R2_TET16, 1, false, R2_NAME(2, ",tab,n16"));
        }
    }
    R2_MODE(0) R2_MODE(1) R2_MODE(2)
    return nullptr;
}

}  // namespace lutr
