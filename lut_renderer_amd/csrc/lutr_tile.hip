// lutr_tile.hip -- planar RGB (gbrp*) on persistent waves with a per-wave LDS lattice window: the round-1 design, kept for the
// planar-RGB formats.  The fused YUV kernels moved to lutr_tile2.hip in round 2 (raw-code-space validity instead of the
// optimistic pass described below).
//
// Replaces the slice-threaded per-row loops of FFmpeg's lut3d
// (the filter /root/reference/src/lut_renderer/ffmpeg.py:246 emits) when the frames are planar RGB.
//
// Design (DESIGN.md "Kernels"):
//   * A 33^3 fp32 lattice (431 KB; 629 KB as padded float4) does not fit the 160 KB LDS,
//     and gathering 4-8 taps of 16 B per pixel from L1/L2 caps at ~1 px/clk/CU.  Video
//     pixels that are close on screen are close in colour, so each WAVE keeps a small
//     WINDOW of the lattice in LDS and reads its taps with ds_read_b128 (broadcast when
//     neighbouring lanes share a cell).
//   * The window is a box in SHEARED lattice coordinates (r, g-r, b-r): luma moves a
//     colour along the cube's diagonal (one long axis), chroma across it (two short
//     axes).  A box that is long in r and 4..16 nodes wide in the other two covers a
//     tile with a dark-to-bright edge where an axis-aligned RGB box of the same size
//     could not.
//   * Lookups are OPTIMISTIC: a tile is computed against whatever window the wave holds
//     while per-lane min/max of the cell coordinates are accumulated; one wave vote at
//     the end of the tile says whether every tap was inside.  Only on a miss are the
//     bounds reduced across the wave, the window re-staged from L2 (or, if the tile's
//     colours do not fit, the tile routed to the global-gather body) and the tile redone.
//     Waves walk DOWN a column strip, so consecutive tiles mostly hit.
//   * No workgroup barrier in the tile loop (one at kernel start, after the coordinate table is
//     filled): waves are independent; a wave's only shared state is its own LDS slice.
//   * Persistent grid (16 waves per CU in 8-wave workgroups that share one per-code coordinate
//     table); work is handed out in chunks of 16 tile rows through one atomic counter, every
//     wave's first chunk by its id (no burst of atomics at start-up).
//   * LDS nodes are 12 bytes {r,g,b} for the 4-tap modes, 16 for trilinear (DESIGN.md "Kernels").
//   * Launches too small to fill this machinery go to the plain vector kernels instead
//     (lutr_kernels.hip small_job()).
//
// Arithmetic is the strict restatement (see lutr_kernels.hip): -ffp-contract=off, FFmpeg's
// scalar C order, bit-identical to the oracle.
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <utility>

#include "lutr_internal.h"
#ifndef LUTR_NT
#define LUTR_NT 1        // non-temporal stores: every output byte is written once (+0.3 % here; non-temporal LOADS cost 2.5 % on gbrp10le)
#endif


// Waves per workgroup.  The waves of a block share one coordinate table, so bigger blocks leave more of the
// CU's 160 KB to the windows (4 waves -> 512 nodes per wave at 10 bit, 8 -> 576, 16 -> 608; 8 measured best).
#ifndef LUTR_WPB
#define LUTR_WPB 8
#endif

namespace lutr {

extern __shared__ __attribute__((aligned(16))) char lutr_smem[];

// ---------------------------------------------------------------- small math
__device__ __forceinline__ float tmed3(float a, float lo, float hi) { return __builtin_amdgcn_fmed3f(a, lo, hi); }
__device__ __forceinline__ float tfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

// A wave-uniform constant used by many full-rate VOP2 ops per pixel is worth a VGPR: an SGPR source
// operand drops v_mul/v_add/v_fmac to the 0.62x issue class on gfx950 (tools/ubench/op_rates.hip).
// The asm makes the copy opaque so hipcc cannot fold it back into the SGPR.
__device__ __forceinline__ float in_vgpr(float s)
{
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}

// ---------------------------------------------------------------- window state (wave-uniform)
struct Win {
    // byte address of the tap at cell (pr,pg,pb): (int) fma(pr, fr, fma(pg, fg, fma(pb, fb, fc)))
    float fr, fg, fb, fc;
    int   o_r, o_g;                               // byte steps of +1 along r and g (blue is always kOB)
    unsigned a_max;                               // LDS only: highest base address whose 8 corners stay inside
                                                  // the workgroup's allocation (optimistic reads never leave it)
    float r_lo, r_hi, g_lo, g_hi, b_lo, b_hi;     // cells (pr, pg-pr, pb-pr) whose 8 corners are staged
};

// Blue is the fastest axis of the global lattice and of every LDS window: its step is one 16-byte node.
// A compile-time constant lets the +1-blue taps use the immediate offset field of ds_read / global_load.
constexpr int kOB = 16;
// Bytes per node in an LDS window: 16 (the global float4 node, one ds_read_b128 per tap) or 12 ({r,g,b} only:
// a third more nodes per wave, a tap = ds_read2_b32 + ds_read_b32, 3 registers instead of 4).  The 4-tap
// modes take 12: a third fewer tiles fall back to the gather body (20,416 -> 13,728 of 518,400 on the headline
// workload) and the 6 VGPRs it frees hold three more per-pixel constants (LUTR_PIN_MORE), together +3.5 %;
// the 8-tap trilinear body keeps 16 (16 LDS instructions per pixel would cost more than the window gains).
// LUTR_LDS_NODE forces one size for every mode (experiments).
template <int INTERP> constexpr int lds_node()
{
#ifdef LUTR_LDS_NODE
    return LUTR_LDS_NODE;
#else
    return INTERP == LUTR_INTERP_TRILINEAR ? 16 : 12;
#endif
}
static inline int lds_node_rt(int mode)
{
#ifdef LUTR_LDS_NODE
    return LUTR_LDS_NODE;
#else
    return mode == LUTR_INTERP_TRILINEAR ? 16 : 12;
#endif
}
template <bool LDS, int INTERP> constexpr int node_b() { return LDS ? lds_node<INTERP>() : kOB; }

struct Bnd { float rmin, rmax, gmin, gmax, bmin, bmax; };

__device__ __forceinline__ void bnd_reset(Bnd &b)
{
    b.rmin = b.gmin = b.bmin = 1e9f;
    b.rmax = b.gmax = b.bmax = -1e9f;
}

// A node is read as one 16-byte access.  Loading through the 4-wide ext_vector type keeps it a
// ds_read_b128 / global_load_dwordx4 even though .w is padding (a float4 struct load gets narrowed
// to b96, which the LDS serves at 8 cycles per wave-instruction instead of 4; MI355X_MICROARCH.md).
typedef float f4 __attribute__((ext_vector_type(4)));

// bounds of the touched cells in sheared coordinates (r, g-r, b-r)
__device__ __forceinline__ void bnd_update(Bnd &bn, float pr, float pg, float pb)
{
    const float hg = pg - pr, hb = pb - pr;
    bn.rmin = fminf(bn.rmin, pr); bn.rmax = fmaxf(bn.rmax, pr);
    bn.gmin = fminf(bn.gmin, hg); bn.gmax = fmaxf(bn.gmax, hg);
    bn.bmin = fminf(bn.bmin, hb); bn.bmax = fmaxf(bn.bmax, hb);
}

// Sheared cell coordinates of one pixel, and the bounds update for a PAIR of pixels: v_min3 / v_max3
// fold two pixels per instruction (3 slow-class VALU per pixel instead of 6).  Written as asm because
// hipcc only forms min3/max3 from some of the equivalent fminf/fmaxf chains.
struct Cell { float r, hg, hb; };

__device__ __forceinline__ float vmin3(float a, float b, float c)
{
    float o;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}
__device__ __forceinline__ float vmax3(float a, float b, float c)
{
    float o;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}

__device__ __forceinline__ void bnd_update2(Bnd &bn, const Cell &a, const Cell &b)
{
    bn.rmin = vmin3(bn.rmin, a.r, b.r);    bn.rmax = vmax3(bn.rmax, a.r, b.r);
    bn.gmin = vmin3(bn.gmin, a.hg, b.hg);  bn.gmax = vmax3(bn.gmax, a.hg, b.hg);
    bn.bmin = vmin3(bn.bmin, a.hb, b.hb);  bn.bmax = vmax3(bn.bmax, a.hb, b.hb);
}

// LDS taps take an ABSOLUTE LDS byte address (the window constants include lds_base()): going
// through `lutr_smem + a` costs a v_add_u32 with the relocated symbol per pixel.
typedef const __attribute__((address_space(3))) f4 lds_f4;
__device__ __forceinline__ int lds_base()
{
    return (int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lutr_smem;
}

template <bool LDS, int NODE = 16>
__device__ __forceinline__ f4 tap(const float4 *__restrict__ lat, int a)
{
    if constexpr (LDS && NODE == 16) {
        return *(lds_f4 *)(uintptr_t)(unsigned)a;
    } else if constexpr (LDS) {
        typedef const __attribute__((address_space(3))) float lds_f;
        lds_f *p = (lds_f *)(uintptr_t)(unsigned)a;
        f4 v;
        v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = 0.0f;
        return v;
    } else {
        return *(const f4 *)((const char *)lat + a);
    }
}

struct Rgb3 { float r, g, b; };

__device__ __forceinline__ float tlerp(float v0, float v1, float f) { return v0 + (v1 - v0) * f; }

// One pixel of SURVEY A.3-A.5 against window W, in three stages so that a tile body can run each
// stage for several pixels back to back (independent instructions; a lone dependent chain issues
// one VALU per ~6 cycles on gfx950).  Integer codes in (as floats), integer codes out (as floats).
struct PxC {
    int a;                  // byte address of the c000 tap
    int oa, oz;             // tetrahedral: byte offsets of the 2nd and 3rd taps
    float w0, w1, w2, w3;   // tetrahedral weights; trilinear keeps d.r, d.g, d.b in w0..w2
};

// (prev, frac) of one channel.  Computed form: FFmpeg's float ops on the code (A.3).  Table form:
// the same ops were applied once per code when the kernel started (coord_table_fill), one
// ds_read_b64 fetches the pair.  Both give identical bits.
struct Crd { float p, d; };

template <int INTERP>
__device__ __forceinline__ Crd crd_compute(const LutConsts &L, float code, float sc)
{
    const float x = code * L.scale_f;
    // clipf(x*scale, 0, lut_max): codes and scales are >= 0, so only the upper bound can bind and a
    // plain v_min_f32 (full rate) replaces v_med3_f32 (0.62x rate on gfx950, tools/ubench/op_rates.hip)
    const float s = fminf(x * sc, L.lut_max);
    Crd c;
    if constexpr (INTERP == LUTR_INTERP_NEAREST) {     // NEAR(x) with FFmpeg's double .5 (near_f, lutr_device.h)
        const float fl = floorf(s);
        c.p = (s - fl >= .5f) ? fl + 1.0f : fl;
        c.d = 0.0f;
    }
    else { c.p = floorf(s); c.d = s - c.p; }
    return c;
}

__device__ __forceinline__ Crd crd_table(unsigned code)
{
    const float2 e = *(const float2 *)(lutr_smem + code * 8u);
    return Crd{e.x, e.y};
}

// All 256 threads of the block fill the per-code table at LDS offset 0 (before any wave claims work).
template <int INTERP>
__device__ __forceinline__ void coord_table_fill(const LutConsts &L, int entries)
{
    for (int q = threadIdx.x; q < entries; q += 64 * LUTR_WPB) {
        const Crd c = crd_compute<INTERP>(L, (float)q, L.sc[0]);
        *(float2 *)(lutr_smem + q * 8) = make_float2(c.p, c.d);
    }
    __syncthreads();
}

template <bool LDS, int INTERP>
__device__ __forceinline__ PxC px_finish(const LutConsts &L, const Win &W, const Crd &cr, const Crd &cg, const Crd &cb, Cell &cell)
{
    const float pr = cr.p, pg = cg.p, pb = cb.p;
    cell.r = pr; cell.hg = pg - pr; cell.hb = pb - pr;
    PxC c;
    if constexpr (LDS) {
        // exact in fp32: every term is an integer well below 2^24 for a window of <= 4096 nodes
        c.a = (int)tfma(pr, W.fr, tfma(pg, W.fg, tfma(pb, W.fb, W.fc)));
        c.a = (int)min((unsigned)c.a, W.a_max);   // a stale window may not contain this cell: stay in LDS
    } else {
        c.a = (((int)pr * L.n1 + (int)pg) * L.n1 + (int)pb) * 16;
    }
    c.oa = c.oz = 0;
    c.w0 = c.w1 = c.w2 = c.w3 = 0.0f;
    if constexpr (INTERP == LUTR_INTERP_TRILINEAR) {
        c.w0 = cr.d; c.w1 = cg.d; c.w2 = cb.d;
    } else if constexpr (INTERP == LUTR_INTERP_TETRAHEDRAL) {
        // sorted form (see lutr_kernels.hip interp_tetrahedral for why this is bit-identical to
        // FFmpeg's six branches on a finite lattice)
        const float dr = cr.d, dg = cg.d, db = cb.d;
        const float x = fmaxf(fmaxf(dr, dg), db), y = tmed3(dr, dg, db), z = fminf(fminf(dr, dg), db);
        const bool rg = dr > dg, gb = dg > db, rb = dr > db;
        // by-value copies: a ?: over struct members is an lvalue select, which pins W in scratch
        const int o_r = W.o_r, o_g = W.o_g, o_b = node_b<LDS, INTERP>();
        const int o111 = o_r + o_g + o_b;
        const int z_r = o111 - o_r, z_g = o111 - o_g, z_b = o111 - o_b;
        // first step along the axis of the largest fraction, last step along the smallest
        // Written without negations (each would become one more v_cmp): when r is not the strict maximum,
        // g > b already makes g a maximum (r <= g or r <= b < g); when b is not the strict minimum, r > g
        // already makes g a minimum.  Ties only ever pick between taps whose weight is exactly 0.
        c.oa = (rg && rb) ? o_r : (gb ? o_g : o_b);
        c.oz = (gb && rb) ? z_b : (rg ? z_g : z_r);
        c.w0 = 1.0f - x; c.w1 = x - y; c.w2 = y - z; c.w3 = z;
    }
    return c;
}

// float codes (already clipped to [0, M]) -> PxC, computed coordinates
template <bool LDS, int INTERP>
__device__ __forceinline__ PxC px_coords(const LutConsts &L, const Win &W, float rc, float gc, float bc, Cell &cell)
{
    return px_finish<LDS, INTERP>(L, W, crd_compute<INTERP>(L, rc, L.sc[0]), crd_compute<INTERP>(L, gc, L.sc[1]),
                                  crd_compute<INTERP>(L, bc, L.sc[2]), cell);
}

// integer codes in [0, M] -> PxC, table coordinates
template <bool LDS, int INTERP>
__device__ __forceinline__ PxC px_coords_tab(const LutConsts &L, const Win &W, unsigned ri, unsigned gi, unsigned bi, Cell &cell)
{
    return px_finish<LDS, INTERP>(L, W, crd_table(ri), crd_table(gi), crd_table(bi), cell);
}

template <bool LDS, int INTERP>
__device__ __forceinline__ Rgb3 px_blend(const LutConsts &L, const Win &W, const PxC &c)
{
    Rgb3 v;
    const int a = c.a;
    if constexpr (INTERP == LUTR_INTERP_NEAREST) {
        const f4 t = tap<LDS, lds_node<INTERP>()>(L.lat, a);
        v.r = t.x; v.g = t.y; v.b = t.z;
    } else if constexpr (INTERP == LUTR_INTERP_TRILINEAR) {
        const float dr = c.w0, dg = c.w1, db = c.w2;
        const int ag = a + W.o_g, ar = a + W.o_r, arg = ar + W.o_g;       // 3 address adds; +blue is an immediate
        constexpr int ob = node_b<LDS, INTERP>();
        const f4 c000 = tap<LDS, lds_node<INTERP>()>(L.lat, a), c001 = tap<LDS, lds_node<INTERP>()>(L.lat, a + ob);
        const f4 c010 = tap<LDS, lds_node<INTERP>()>(L.lat, ag), c011 = tap<LDS, lds_node<INTERP>()>(L.lat, ag + ob);
        const f4 c100 = tap<LDS, lds_node<INTERP>()>(L.lat, ar), c101 = tap<LDS, lds_node<INTERP>()>(L.lat, ar + ob);
        const f4 c110 = tap<LDS, lds_node<INTERP>()>(L.lat, arg), c111 = tap<LDS, lds_node<INTERP>()>(L.lat, arg + ob);
#define TRI(ch, out) \
        { \
            const float c00 = tlerp(c000.ch, c100.ch, dr), c10 = tlerp(c010.ch, c110.ch, dr); \
            const float c01 = tlerp(c001.ch, c101.ch, dr), c11 = tlerp(c011.ch, c111.ch, dr); \
            const float c0 = tlerp(c00, c10, dg), c1 = tlerp(c01, c11, dg); \
            out = tlerp(c0, c1, db); \
        }
        TRI(x, v.r) TRI(y, v.g) TRI(z, v.b)
#undef TRI
    } else {
        const int o111 = W.o_r + W.o_g + node_b<LDS, INTERP>();
        const f4 c0 = tap<LDS, lds_node<INTERP>()>(L.lat, a), c1 = tap<LDS, lds_node<INTERP>()>(L.lat, a + c.oa);
        const f4 c2 = tap<LDS, lds_node<INTERP>()>(L.lat, a + c.oz), c3 = tap<LDS, lds_node<INTERP>()>(L.lat, a + o111);
        v.r = c.w0 * c0.x + c.w1 * c1.x + c.w2 * c2.x + c.w3 * c3.x;
        v.g = c.w0 * c0.y + c.w1 * c1.y + c.w2 * c2.y + c.w3 * c3.y;
        v.b = c.w0 * c0.z + c.w1 * c1.z + c.w2 * c2.z + c.w3 * c3.z;
    }
    return v;
}

// (int)(v * M) clipped to [0, M].  UNIT: every lattice node lies in [0, 1] (checked on the host when
// the lattice is set), so the nearest / trilinear / tetrahedral blends stay in [0, 1 + a few ulp]:
// all weights and nodes are >= 0 and the rounding of each product and sum is monotone, hence
// 0 <= v * M < M + 1 and the truncation alone already lands in [0, M]; the clip is dead code.
template <bool UNIT>
__device__ __forceinline__ Rgb3 px_quant(const LutConsts &L, const Rgb3 &v)
{
    Rgb3 o;
    if constexpr (UNIT) {
        o.r = truncf(v.r * L.maxf);
        o.g = truncf(v.g * L.maxf);
        o.b = truncf(v.b * L.maxf);
    } else {
        o.r = tmed3(truncf(v.r * L.maxf), 0.0f, L.maxf);
        o.g = tmed3(truncf(v.g * L.maxf), 0.0f, L.maxf);
        o.b = tmed3(truncf(v.b * L.maxf), 0.0f, L.maxf);
    }
    return o;
}

// ---------------------------------------------------------------- window management
__device__ __forceinline__ void win_global(Win &W, const LutConsts &L)
{
    const int n1 = L.n1;
    W.fr = W.fg = W.fb = W.fc = 0.0f;             // unused: the global body addresses with integers
    W.o_r = 16 * n1 * n1; W.o_g = 16 * n1;
    W.a_max = 0;
    W.r_lo = W.g_lo = W.b_lo = 1.0f;
    W.r_hi = W.g_hi = W.b_hi = 0.0f;
}

// No window staged yet: every tap of an optimistic pass reads the first node of the slice
// (harmless), and the admissible set is empty so the pass is always declared a miss.
__device__ __forceinline__ void win_empty(Win &W, int slice_off)
{
    W.fr = W.fg = W.fb = 0.0f; W.fc = (float)(lds_base() + slice_off);
    W.o_r = W.o_g = 0;
    W.a_max = (unsigned)(lds_base() + slice_off);
    W.r_lo = W.g_lo = W.b_lo = 1.0f;
    W.r_hi = W.g_hi = W.b_hi = 0.0f;
}

__device__ __forceinline__ bool win_holds(const Win &W, const Bnd &b)
{
    const bool ok = b.rmin >= W.r_lo && b.rmax <= W.r_hi && b.gmin >= W.g_lo && b.gmax <= W.g_hi &&
                    b.bmin >= W.b_lo && b.bmax <= W.b_hi;
    return __all(ok);
}

// Reduce the tile's bounds over the wave and, if a window of `cap` nodes can hold every
// corner the tile touches PLUS one spare cell on each side of the two chroma-like axes (sensor
// noise moves the extremes by a cell from tile to tile; without the spare every other tile
// misses), stage it into this wave's LDS slice and describe it in W.  The r axis takes the
// rest of the capacity, centred.  Returns false (W untouched) when the tile's colours are too
// spread out for the slice.
template <int kLN>
__device__ __forceinline__ bool win_restage(Win &W, const LutConsts &L, const Bnd &bn_, int cap, int slice_off,
                                            int lds_bytes, int lane)
{
    // The reductions below are 36 dependent ds_bpermute round trips.  They are side-effect free, so
    // hipcc hoists them out of the (rare) miss branch into every tile unless their inputs pass
    // through a volatile asm that can only execute inside the branch.
    Bnd bn = bn_;
    asm volatile("" : "+v"(bn.rmin), "+v"(bn.rmax), "+v"(bn.gmin), "+v"(bn.gmax), "+v"(bn.bmin), "+v"(bn.bmax));
    const int rmin = uni((int)wave_min(bn.rmin)), rmax = uni((int)wave_max(bn.rmax));
    const int gmin = uni((int)wave_min(bn.gmin)), gmax = uni((int)wave_max(bn.gmax));
    const int bmin = uni((int)wave_min(bn.bmin)), bmax = uni((int)wave_max(bn.bmax));
    // corners: r..r+1, (g-r)-1..(g-r)+1, (b-r)-1..(b-r)+1
    const int need_r = rmax - rmin + 2, need_g = gmax - gmin + 3, need_b = bmax - bmin + 3;
    int ng = need_g + 2, nb = need_b + 2;                    // one spare cell each side
    int sr = (ng * nb) | 1;                                   // odd plane stride: r-neighbours hit different banks
    if (need_r + 2 > cap / sr) { ng = need_g; nb = need_b; sr = (ng * nb) | 1; }   // no room for spares
    const int nr = cap / sr;
    if (need_r > nr || ng > 128 || nb > 128) return false;
    const int r0 = rmin - ((nr - need_r) >> 1);
    const int g0 = gmin - 1 - ((ng - need_g) >> 1);
    const int b0 = bmin - 1 - ((nb - need_b) >> 1);
    const int n1 = L.n1, nmax = L.n1 - 1;
    const int plane = ng * nb, total = nr * plane;
    // i -> (ir, ig, ib): floor(i/d) = umulhi(i, ceil(2^32/d)), exact while i*d < 2^32 (i < 8192, d <= 16641)
    const unsigned inv_plane = (unsigned)((0x100000000ull + plane - 1) / plane);
    const unsigned inv_nb = (unsigned)((0x100000000ull + nb - 1) / nb);
    for (int i = lane; i < total; i += 64) {
        const int ir = (int)__umulhi((unsigned)i, inv_plane), rem = i - ir * plane;
        const int ig = (int)__umulhi((unsigned)rem, inv_nb), ib = rem - ig * nb;
        int r = r0 + ir, g = r + g0 + ig, b = r + b0 + ib;
        // nodes outside the cube are never referenced by a valid pixel: clamp to stay in bounds
        r = min(max(r, 0), nmax); g = min(max(g, 0), nmax); b = min(max(b, 0), nmax);
        const float4 v = L.lat[(r * n1 + g) * n1 + b];
        if constexpr (kLN == 16) {
            *(float4 *)(lutr_smem + slice_off + 16 * (ir * sr + ig * nb + ib)) = v;
        } else {
            float *q = (float *)(lutr_smem + slice_off + kLN * (ir * sr + ig * nb + ib));
            q[0] = v.x; q[1] = v.y; q[2] = v.z;
        }
    }
    // node index = (pr-r0)*sr + (pg-pr-g0)*nb + (pb-pr-b0)
    W.o_r = kLN * (sr - nb - 1); W.o_g = kLN * nb;
    W.fr = (float)W.o_r; W.fg = (float)W.o_g; W.fb = (float)kLN;
    W.fc = (float)(lds_base() + slice_off - kLN * (r0 * sr + g0 * nb + b0));
    W.a_max = (unsigned)(lds_base() + lds_bytes - (W.o_r + W.o_g + kLN) - kLN);
    W.r_lo = (float)r0;       W.r_hi = (float)(r0 + nr - 2);
    W.g_lo = (float)(g0 + 1); W.g_hi = (float)(g0 + ng - 2);
    W.b_lo = (float)(b0 + 1); W.b_hi = (float)(b0 + nb - 2);
    return true;
}

// optional per-launch statistics (tests, tuning): [0] tiles, [1] LDS passes that missed,
// [2] tiles computed by the global-gather body, [3] windows staged
struct WaveStats {
    unsigned n[4];
    __device__ __forceinline__ WaveStats() : n{0, 0, 0, 0} {}
    __device__ __forceinline__ void flush(unsigned *stats, int lane) const
    {
        if (stats && lane == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) atomicAdd(&stats[i], n[i]);
        }
    }
};

// ---------------------------------------------------------------- sample helpers
template <int WIDE>
__device__ __forceinline__ float wsample(const uint32_t *w, int i)
{
    if constexpr (WIDE) return (float)((w[i >> 1] >> ((i & 1) * 16)) & 0xffffu);
    else return (float)((w[i >> 2] >> ((i & 3) * 8)) & 0xffu);
}

template <int WIDE>
__device__ __forceinline__ void wput(uint32_t *w, int i, float v)
{
    const uint32_t u = (uint32_t)v;
    if constexpr (WIDE) w[i >> 1] |= u << ((i & 1) * 16);
    else w[i >> 2] |= u << ((i & 3) * 8);
}

template <int NW>
__device__ __forceinline__ void ldw(uint32_t *w, const uint8_t *p)
{
#if LUTR_NT >= 2
    typedef unsigned nt4 __attribute__((ext_vector_type(4)));
    typedef unsigned nt2 __attribute__((ext_vector_type(2)));
    if constexpr (NW == 4) { const nt4 v = __builtin_nontemporal_load((const nt4 *)p); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
    else if constexpr (NW == 2) { const nt2 v = __builtin_nontemporal_load((const nt2 *)p); w[0] = v.x; w[1] = v.y; }
    else w[0] = __builtin_nontemporal_load((const uint32_t *)p);
#else
    if constexpr (NW == 4) { const uint4 v = *(const uint4 *)p; w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
    else if constexpr (NW == 2) { const uint2 v = *(const uint2 *)p; w[0] = v.x; w[1] = v.y; }
    else w[0] = *(const uint32_t *)p;
#endif
}

template <int NW>
__device__ __forceinline__ void stw(uint8_t *p, const uint32_t *w)
{
#if LUTR_NT
    typedef unsigned nt4 __attribute__((ext_vector_type(4)));
    typedef unsigned nt2 __attribute__((ext_vector_type(2)));
    if constexpr (NW == 4) __builtin_nontemporal_store(nt4{w[0], w[1], w[2], w[3]}, (nt4 *)p);
    else if constexpr (NW == 2) __builtin_nontemporal_store(nt2{w[0], w[1]}, (nt2 *)p);
    else __builtin_nontemporal_store(w[0], (uint32_t *)p);
#else
    if constexpr (NW == 4) *(uint4 *)p = make_uint4(w[0], w[1], w[2], w[3]);
    else if constexpr (NW == 2) *(uint2 *)p = make_uint2(w[0], w[1]);
    else *(uint32_t *)p = w[0];
#endif
}

// Zero-instruction ordering fence.  hipcc's instruction selection linearises an unrolled block
// with every pure computation first, which keeps the coordinates and weights of all 16 pixels
// of a unit live at once (>1000 spilled registers).  Passing the not-yet-consumed input words
// and the partly built output words through volatile asm statements makes group j+1's inputs
// unknowable until group j's outputs exist, so pixel groups are emitted one after the other.
template <int N>
__device__ __forceinline__ void fence_words(uint32_t *w)
{
    if constexpr (N == 1) asm volatile("" : "+v"(w[0]));
    else if constexpr (N == 2) asm volatile("" : "+v"(w[0]), "+v"(w[1]));
    else if constexpr (N == 4) asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    else if constexpr (N == 8) asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
}

// ---------------------------------------------------------------- tile geometry
struct TileGeom {
    int lw_log2;          // lanes along x per tile row = 1 << lw_log2 (the other lanes go down)
    int uw, urows;        // units per row, unit rows (a unit = PXT px x BH rows)
    int nsx, nry;         // tiles across, tiles down
    int tiles;            // nframes * nsx * nry
    // Work is handed out in CHUNKS of `ch` consecutive tile rows of one strip.  Chunk id =
    // (frame * nrc + row_chunk) * nsx + strip, so chunks claimed at about the same time are
    // neighbouring strips of the same rows (shared DRAM pages).  Waves claim chunks from `queue`
    // (one atomicAdd per chunk): tiles on colour edges cost 2-4x a plain tile and a static split
    // left the slowest wave at 1.76x the mean.
    int ch, nrc, nchunks;
    unsigned *queue;      // device counter, zeroed by the launcher before every launch
    int win_nodes;        // LDS window capacity per wave, in 16-byte nodes
    int tab_bytes;        // per-code coordinate table at LDS offset 0 (0 = coordinates are computed)
    unsigned *stats;      // optional device counters; nullptr = off
};

// The tile kernels' view of the planes: 32-bit row strides (launch_rgb only sends layouts with positive
// strides below 2^30 here), which halves the scalar registers they occupy -- the kernels run out of SGPRs and
// every spilled one costs a v_readlane per tile.
struct TilePlanes {
    const uint8_t *s[3];
    uint8_t       *d[3];
    int ss[3], ds[3];
    long long sfs[3], dfs[3];
};
static inline TilePlanes tile_planes(const PlaneSet &P)
{
    TilePlanes T;
    for (int i = 0; i < 3; i++) {
        T.s[i] = P.s[i]; T.d[i] = P.d[i];
        T.ss[i] = (int)P.ss[i]; T.ds[i] = (int)P.ds[i];
        T.sfs[i] = P.sfs[i]; T.dfs[i] = P.dfs[i];
    }
    return T;
}

// ================================================================= work distribution
// 4 waves per SIMD: the kernel needs 109 VGPRs once the input tile is consumed in place (keeping
// a pristine copy for the rare miss pass cost 41 registers); 5 waves (96 VGPRs) spills into the hot loop.
#ifndef LUTR_TILE_WAVES_PER_EU
#define LUTR_TILE_WAVES_PER_EU 4
#endif

// Position of chunk c on the batch; false when c is past the end.
__device__ __forceinline__ bool chunk_at(const TileGeom &TG, unsigned c, int &fr, int &sx, int &ry, int &rem)
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

// Claim the next chunk for this wave; returns false when the queue is drained.  The queue counter starts at
// LUTR_STATIC_ROUNDS x the number of waves in the grid: wave w takes chunks w, w + waves, ... as its first ones
// without an atomic (4096 atomics on one address at kernel start serialise in the L2 for ~50 us before the
// last wave has work, and the waves are still in step when they finish their first chunk).
#ifndef LUTR_STATIC_ROUNDS
#define LUTR_STATIC_ROUNDS 1
#endif
__device__ __forceinline__ bool claim_chunk(const TileGeom &TG, int lane, int &fr, int &sx, int &ry, int &rem, int &round)
{
    unsigned c = 0;
    if (round < LUTR_STATIC_ROUNDS) {
        c = (unsigned)(round * (int)(gridDim.x * LUTR_WPB) + (int)(blockIdx.x * LUTR_WPB) + uni((int)(threadIdx.x >> 6)));
        round++;
        if (c < (unsigned)TG.nchunks) return chunk_at(TG, c, fr, sx, ry, rem);
        round = LUTR_STATIC_ROUNDS;          // the static share is used up: continue with the queue
    }
    if (lane == 0) c = atomicAdd(TG.queue, 1u);
    c = (unsigned)uni((int)c);
    return chunk_at(TG, c, fr, sx, ry, rem);
}

// ================================================================= planar RGB tile kernel
template <int WIDE>
struct RgbTile {
    static constexpr int PXT = WIDE ? 8 : 16;
    uint32_t g[4], b[4], r[4];
};

template <int WIDE>
__device__ __forceinline__ unsigned wcode(const uint32_t *w, int i)
{
    if constexpr (WIDE) return (w[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
    else return (w[i >> 2] >> ((i & 3) * 8)) & 0xffu;
}

template <bool LDS, int WIDE, int INTERP, int TAB>
__device__ __forceinline__ void rgb_tile_body(const LutConsts &L, const Win &W, RgbTile<WIDE> &in,
                                              RgbTile<WIDE> &out, Bnd &bn)
{
    // `in` is consumed: the ordering fences run its words through volatile asm, so the caller re-loads the tile before a second pass.  Groups of 4 pixels, staged: coordinates for all four,
    // then taps + blend, then packing, so neighbouring instructions are independent.
#pragma unroll
    for (int k = 0; k < 4; k++) { out.g[k] = 0; out.b[k] = 0; out.r[k] = 0; }
#pragma unroll
    for (int g = 0; g < RgbTile<WIDE>::PXT / 4; g++) {
        PxC pc[4];
        Cell cell[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int i = g * 4 + p;
            if constexpr (TAB) {
                // codes above 2^depth-1 cannot occur in a valid plane; clamp so a stray one cannot index past the table
                const unsigned mi = (unsigned)L.maxf;
                pc[p] = px_coords_tab<LDS, INTERP>(L, W, min(wcode<WIDE>(in.r, i), mi), min(wcode<WIDE>(in.g, i), mi),
                                                   min(wcode<WIDE>(in.b, i), mi), cell[p]);
            } else {
                pc[p] = px_coords<LDS, INTERP>(L, W, wsample<WIDE>(in.r, i), wsample<WIDE>(in.g, i), wsample<WIDE>(in.b, i), cell[p]);
            }
            if (p & 1) bnd_update2(bn, cell[p - 1], cell[p]);
        }
        Rgb3 o[4];
#pragma unroll
        for (int p = 0; p < 4; p++) o[p] = px_quant<TAB == 2>(L, px_blend<LDS, INTERP>(L, W, pc[p]));
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int i = g * 4 + p;
            wput<WIDE>(out.g, i, o[p].g);
            wput<WIDE>(out.b, i, o[p].b);
            wput<WIDE>(out.r, i, o[p].r);
        }
        fence_words<4>(in.g); fence_words<4>(in.b); fence_words<4>(in.r);
        fence_words<4>(out.g); fence_words<4>(out.b); fence_words<4>(out.r);
    }
}

template <int WIDE, int INTERP, int TAB>
__device__ __forceinline__ void rgb_tile_bounds(const LutConsts &L, RgbTile<WIDE> &in, Bnd &bn)
{
#pragma unroll
    for (int i = 0; i < RgbTile<WIDE>::PXT; i++) {
        if constexpr (TAB) {
            const unsigned mi = (unsigned)L.maxf;
            bnd_update(bn, crd_table(min(wcode<WIDE>(in.r, i), mi)).p, crd_table(min(wcode<WIDE>(in.g, i), mi)).p,
                       crd_table(min(wcode<WIDE>(in.b, i), mi)).p);
        } else {
            bnd_update(bn, crd_compute<INTERP>(L, wsample<WIDE>(in.r, i), L.sc[0]).p,
                       crd_compute<INTERP>(L, wsample<WIDE>(in.g, i), L.sc[1]).p,
                       crd_compute<INTERP>(L, wsample<WIDE>(in.b, i), L.sc[2]).p);
        }
        if ((i & 3) == 3) {
            fence_words<4>(in.g); fence_words<4>(in.b); fence_words<4>(in.r);
            asm volatile("" : "+v"(bn.rmin), "+v"(bn.rmax), "+v"(bn.gmin), "+v"(bn.gmax), "+v"(bn.bmin), "+v"(bn.bmax));
        }
    }
}

template <int WIDE, int INTERP, int TAB>
__global__ __launch_bounds__(64 * LUTR_WPB, LUTR_TILE_WAVES_PER_EU)
void k_rgb_tile(LutConsts L_, TilePlanes P, FrameGeom G, TileGeom TG)
{
    LutConsts L = L_;
    // used 3 times per pixel by full-rate ops: worth a VGPR (an SGPR source drops them to the slow issue class)
    if constexpr (INTERP != LUTR_INTERP_TRILINEAR) L.maxf = in_vgpr(L_.maxf);
    if constexpr (TAB) coord_table_fill<INTERP>(L, TG.tab_bytes / 8);
    using T = RgbTile<WIDE>;
    const int lane = threadIdx.x & 63;
    const int wib = uni(threadIdx.x >> 6);
    const int slice_off = TG.tab_bytes + wib * TG.win_nodes * lds_node<INTERP>();
    int fr, sx, ry, rem;                                      // the tile being fetched next
    int round = 0;
    if (!claim_chunk(TG, lane, fr, sx, ry, rem, round)) return;   // wave-uniform; the first chunks go by wave id
    const int lw = 1 << TG.lw_log2, lh_log2 = 6 - TG.lw_log2;
    const int lx = lane & (lw - 1), ly = lane >> TG.lw_log2;

    const int lds_bytes = TG.tab_bytes + LUTR_WPB * TG.win_nodes * lds_node<INTERP>();
    Win W, WG;
    win_empty(W, slice_off);
    win_global(WG, L);
    bool lds_mode = true;

    auto load_tile = [&](T &dst, int f, int tsx, int try_) {
        const int lxc = min(lx, TG.uw - 1 - tsx * lw), lyc = min(ly, TG.urows - 1 - (try_ << lh_log2));
        const long long row0 = G.row0 + (try_ << lh_log2);                // wave-uniform
        const long long xb = (long long)tsx * lw * 16;
        ldw<4>(dst.g, P.s[0] + f * P.sfs[0] + row0 * (long long)P.ss[0] + xb + (unsigned)(lyc * (int)P.ss[0] + lxc * 16));
        ldw<4>(dst.b, P.s[1] + f * P.sfs[1] + row0 * (long long)P.ss[1] + xb + (unsigned)(lyc * (int)P.ss[1] + lxc * 16));
        ldw<4>(dst.r, P.s[2] + f * P.sfs[2] + row0 * (long long)P.ss[2] + xb + (unsigned)(lyc * (int)P.ss[2] + lxc * 16));
    };

    WaveStats ws;
    T nxt;
    load_tile(nxt, fr, sx, ry);
    bool nxt_fresh = true;
    for (bool more = true; more;) {
        T in = nxt;
        const bool fresh = nxt_fresh;
        nxt_fresh = false;
        const int cfr = fr, csx = sx, cry = ry;
        if (--rem > 0) ry++;                    // next tile of the chunk, or a new chunk 
        else { more = claim_chunk(TG, lane, fr, sx, ry, rem, round); nxt_fresh = true; }
        load_tile(nxt, fr, sx, ry);

        T out;
        Bnd bn;
        if (fresh) {
            bnd_reset(bn);
            rgb_tile_bounds<WIDE, INTERP, TAB>(L, in, bn);
            if (!lds_mode || !win_holds(W, bn)) {
                lds_mode = win_restage<lds_node<INTERP>()>(W, L, bn, TG.win_nodes, slice_off, lds_bytes, lane);
                if (lds_mode) ws.n[3]++;
            }
            load_tile(in, cfr, csx, cry);
            __builtin_amdgcn_s_waitcnt(0x0f70);
        }
        for (;;) {
            bnd_reset(bn);
            if (lds_mode) {
                rgb_tile_body<true, WIDE, INTERP, TAB>(L, W, in, out, bn);
                if (win_holds(W, bn)) break;
                ws.n[1]++;
                if (!win_restage<lds_node<INTERP>()>(W, L, bn, TG.win_nodes, slice_off, lds_bytes, lane)) lds_mode = false;
                else ws.n[3]++;
                load_tile(in, cfr, csx, cry);
                __builtin_amdgcn_s_waitcnt(0x0f70);
            } else {
                rgb_tile_body<false, WIDE, INTERP, TAB>(L, WG, in, out, bn);
                ws.n[2]++;
                if (win_restage<lds_node<INTERP>()>(W, L, bn, TG.win_nodes, slice_off, lds_bytes, lane)) { lds_mode = true; ws.n[3]++; }
                break;
            }
        }
        {
            const int lxc = min(lx, TG.uw - 1 - csx * lw), lyc = min(ly, TG.urows - 1 - (cry << lh_log2));
            const long long row0 = G.row0 + (cry << lh_log2);
            const long long xb = (long long)csx * lw * 16;
            stw<4>(P.d[0] + cfr * P.dfs[0] + row0 * (long long)P.ds[0] + xb + (unsigned)(lyc * (int)P.ds[0] + lxc * 16), out.g);
            stw<4>(P.d[1] + cfr * P.dfs[1] + row0 * (long long)P.ds[1] + xb + (unsigned)(lyc * (int)P.ds[1] + lxc * 16), out.b);
            stw<4>(P.d[2] + cfr * P.dfs[2] + row0 * (long long)P.ds[2] + xb + (unsigned)(lyc * (int)P.ds[2] + lxc * 16), out.r);
        }
        ws.n[0]++;
    }
    ws.flush(TG.stats, lane);
}

// ================================================================= launchers
// Launchers may run on several threads at once (one context per thread): process-wide
// state is initialised exactly once (function-local statics, call_once) or guarded by a mutex.
static int device_cus()
{
    static const int cus = [] {
        int dev = 0, n = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        return n > 0 ? n : 256;
    }();
    return cus;
}

// Split the 64 lanes of a wave between x (units of one row) and y (unit rows) so that a
// row of `uw` units wastes as few lanes as possible; prefer wide tiles (longer bursts).
static void plan_tiles(TileGeom *tg, int uw, int urows, int nframes, int win_nodes, int waves_per_cu, unsigned *stats,
                       unsigned *queue)
{
    tg->stats = stats;
    tg->queue = queue;
    int best = 6;
    double best_eff = -1.0;
    for (int l = 6; l >= 2; l--) {
        const int lw = 1 << l, lh = 64 >> l;
        const double eff = ((double)uw / (((uw + lw - 1) / lw) * lw)) * ((double)urows / (((urows + lh - 1) / lh) * lh));
        if (eff > best_eff + 0.02) { best_eff = eff; best = l; }
    }
    if (const char *e = getenv("LUTR_LW_LOG2")) { const int v = atoi(e); if (v >= 2 && v <= 6) best = v; }
    tg->lw_log2 = best;
    tg->uw = uw;
    tg->urows = urows;
    tg->nsx = (uw + (1 << best) - 1) >> best;
    tg->nry = (urows + (64 >> best) - 1) / (64 >> best);
    tg->tiles = nframes * tg->nsx * tg->nry;
    // chunk height: 16 tile rows (a staged window is reused ~15 times); only jobs too small to give
    // half the resident waves a chunk get shorter chunks (each chunk start costs a window miss)
    const int max_waves = device_cus() * waves_per_cu;
    int ch = 16;
    if (const char *e = getenv("LUTR_CHUNK")) { const int v = atoi(e); if (v >= 2 && v <= 256) ch = v; }
    int shrink_div = 4;
    if (const char *e = getenv("LUTR_SHRINK_DIV")) { const int v = atoi(e); if (v >= 1 && v <= 64) shrink_div = v; }
    while (ch > 1 && (long long)nframes * tg->nsx * ((tg->nry + ch - 1) / ch) < max_waves / shrink_div) ch >>= 1;
    tg->ch = ch;
    tg->nrc = (tg->nry + ch - 1) / ch;
    tg->nchunks = nframes * tg->nrc * tg->nsx;
    tg->win_nodes = win_nodes;
}

// persistent grid: as many waves as fit on the chip, or one per chunk if there are fewer chunks
static int tile_blocks(const TileGeom &tg, int waves_per_cu)
{
    const int max_waves = device_cus() * waves_per_cu;
    const int waves = tg.nchunks < max_waves ? tg.nchunks : max_waves;
    return (waves + LUTR_WPB - 1) / LUTR_WPB;
}

static int g_win_nodes = 1024;       // 10 KB per wave, 40 KB per 256-thread block, 4 blocks per CU = all 160 KB
static int g_waves_per_cu = 16;
static std::once_flag g_env_once;

static void read_env_tuning()
{
    std::call_once(g_env_once, [] {
        if (const char *e = getenv("LUTR_WIN_NODES")) { const int v = atoi(e); if (v >= 64 && v <= 2048) g_win_nodes = v; }
        if (const char *e = getenv("LUTR_WAVES_PER_CU")) { const int v = atoi(e); if (v >= 4 && v <= 32) g_waves_per_cu = v; }
    });
}

// The per-code coordinate table needs equal per-channel scales (one table serves R, G and B) and a
// depth of at most 10 bits (8 KB); the window shrinks so that 4 blocks still share the CU's 160 KB.
static int plan_table(TileGeom *tg, const LutConsts &L, int kLN)
{
    const int entries = (int)L.maxf + 1;
    const bool ok = L.sc[0] == L.sc[1] && L.sc[1] == L.sc[2] && entries <= 1024 && !getenv("LUTR_NO_TAB");
    tg->tab_bytes = ok ? entries * 8 : 0;
    const int blocks_per_cu = g_waves_per_cu / LUTR_WPB > 0 ? g_waves_per_cu / LUTR_WPB : 1;
    const int cap = (163840 / blocks_per_cu - tg->tab_bytes) / (kLN * LUTR_WPB);   // nodes per wave: the CU's 160 KB over its resident blocks
    if (tg->win_nodes > cap) tg->win_nodes = cap;
    return tg->tab_bytes;
}

// Blocks of more than 4 waves need more than the default 64 KB of dynamic LDS: allow it once per kernel.
static void allow_lds(const void *kernel, size_t bytes)
{
    static std::set<std::pair<int, const void *>> done;      // the attribute is per device
    static std::mutex mu;
    if (bytes <= 65536) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done.insert({dev, kernel});
}
#define LUTR_LAUNCH_TILE(kernel, ...) \
    do { allow_lds((const void *)(kernel), lds); hipLaunchKernelGGL((kernel), grid, block, lds, st, __VA_ARGS__); } while (0)

const char *launch_rgb_tile(hipStream_t st, const LutConsts &L, const PlaneSet &P, const FrameGeom &G, int depth, int mode,
                            unsigned *stats, unsigned *queue)
{
    const int wide = depth > 8, pxt = wide ? 8 : 16;
    TileGeom tg;
    read_env_tuning();
    plan_tiles(&tg, G.w / pxt, G.rows, G.nframes, g_win_nodes, g_waves_per_cu, stats, queue);
    const dim3 grid(tile_blocks(tg, g_waves_per_cu)), block(64 * LUTR_WPB);
    // the queue starts behind the chunks the waves take by their id (claim_chunk)
    // (round 1's single counter keeps its memset, on a word of its own: words 0 and 1 of the block belong to the round-3 kernels, which
    // leave them at zero themselves)
    queue += 2; tg.queue = queue;
    if (hipMemsetD32Async((hipDeviceptr_t)queue, (int)(grid.x * LUTR_WPB * LUTR_STATIC_ROUNDS), 1, st) != hipSuccess) return nullptr;
    const bool tab = plan_table(&tg, L, lds_node_rt(mode)) != 0;
    const TilePlanes TP = tile_planes(P);
    const size_t lds = (size_t)tg.tab_bytes + (size_t)LUTR_WPB * tg.win_nodes * lds_node_rt(mode);
#define RGB_CASE(W, I) \
    if (wide == W && mode == I) { \
        if (tab && L.unit) LUTR_LAUNCH_TILE((k_rgb_tile<W, I, 2>), L, TP, G, tg); \
        else if (tab) LUTR_LAUNCH_TILE((k_rgb_tile<W, I, 1>), L, TP, G, tg); \
        else LUTR_LAUNCH_TILE((k_rgb_tile<W, I, 0>), L, TP, G, tg); \
        return tab ? (L.unit ? "k_rgb_tile<" #W "," #I ",tab,unit>" : "k_rgb_tile<" #W "," #I ",tab>") : "k_rgb_tile<" #W "," #I ">"; \
    }
    RGB_CASE(0, 0) RGB_CASE(0, 1) RGB_CASE(0, 2)
    RGB_CASE(1, 0) RGB_CASE(1, 1) RGB_CASE(1, 2)
#undef RGB_CASE
    return nullptr;
}

}  // namespace lutr
