// lutr_tile2.hip -- round-2 fused YUV tile kernels: persistent waves, a workgroup-shared GREY TUBE of the lattice in LDS for
// near-neutral content, per-wave LDS lattice windows for the saturated rest, and validity decided in RAW CODE SPACE before
// any pixel is computed.
//
// Replaces the slice-threaded per-row loops of FFmpeg's lut3d + the scalers around it
// (the filters /root/reference/src/lut_renderer/ffmpeg.py:212-247,:304-310 emits).
//
// What changed against round 1's lutr_tile.hip (DESIGN.md "Kernels" has the measurements):
//   * VALIDITY WITHOUT PER-PIXEL WORK.  Round 1 computed a tile optimistically, accumulated min/max of every
//     pixel's lattice cell (6 VALU per pixel), voted, and redid the tile on a miss.  Here a window carries a BOX
//     [y]x[cb]x[cr] of raw input codes with the guarantee that every pixel inside the box only touches staged
//     nodes.  A tile's raw extremes come from v_pk_min/max_u16 on the packed input words (~1.7 VALU per pixel),
//     one saturating-subtract test per plane says whether the lane's unit is inside the box, one vote per tile.
//     No tile is ever computed twice, no address clamp is needed, and the body has no bounds code at all.
//     box -> cells is a conservative map (map_box below) that exploits what the sheared window exploits: in
//     (r, g-r, b-r) coordinates luma cancels out of the two chroma-like axes.
//   * THE GREY TUBE (level 0, tube_lane / tube_holds).  In the sheared coordinates the two difference axes depend on chroma alone, so
//     "all of r, |g - r| and |b - r| up to H cells" can be staged ONCE per workgroup (98 KB as fp16 at 33^3, H = 8) and serves
//     every tile whose chroma stays within +-64 8-bit codes of G-R and B-R whatever its luma does: 95 % of the tiles of the
//     natural test frames, with no window, no raw box, no second level and no restage.  The windows keep the rest.
//   * PADDED PER-CODE TABLE.  The {prev, frac} table covers every index the YUV->RGB sum can reach (negative sums
//     saturate to 0 in v_cvt_u32_f32), so the per-channel v_min_u32 is gone.
//   * OUTPUT PACKING BY SDWA.  v_cvt_u32_f32_sdwa writes a sample straight into its half-word / byte.
//   * FAST VARIANT (V = 3).  LDS nodes are 8-byte {r,g,b,-} fp16 pre-multiplied by 2^depth-1; the blend is an fma chain
//     of v_fma_mix_f32 (fp16 tap x fp32 weight + fp32 accumulator, exact products, one rounding per step): 12 VALU
//     instead of 24 per pixel, a tap is one ds_read_b64, the window holds 2-3x the nodes.  (Measured alternative: fp32
//     float4 nodes pre-multiplied by M with plain v_fma_f32 -- full-rate VALU, but ds_read_b128 taps run into LDS bank
//     conflicts: 403 vs 560 Gpx/s.)  |fast - strict| <= 1 code at 8 and 10 bit (tests/test_fast_variant.py, DESIGN.md
//     3.4); the strict kernels stay bit-identical to the oracle.
//   * Input and output depth are independent (10-bit in, 8-bit out is the reference's libx264 default).
//
// Arithmetic of the strict variants: -ffp-contract=off, FFmpeg's scalar C order, bit-identical to the oracle.
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "lutr_internal.h"
#ifndef LUTR_NT
#define LUTR_NT 2        // 1: non-temporal stores, 2: and loads -- frames are read once and written once (+0.9 % / +0.5 % on 4:2:0, +1 % each on 4:4:4)
#endif

// One translation unit per (input width, output width, chroma layout): the Makefile compiles this file nine times
// (-DLUTR_T2_WI=.. -DLUTR_T2_WO=.. -DLUTR_T2_X=.. -DLUTR_T2_Y=..), in parallel, each defining launch_yuv_tile2_w<WI><WO>_c<X><Y>.
#ifndef LUTR_T2_WI
#define LUTR_T2_WI 1
#define LUTR_T2_WO 1
#define LUTR_T2_X 1
#define LUTR_T2_Y 1
#endif
#define T2_CAT_(a, b, c, d) launch_yuv_tile2_w##a##b##_c##c##d
#define T2_CAT(a, b, c, d) T2_CAT_(a, b, c, d)
#define T2_ENTRY T2_CAT(LUTR_T2_WI, LUTR_T2_WO, LUTR_T2_X, LUTR_T2_Y)

#ifndef LUTR_T2_WPB
#define LUTR_T2_WPB 16            // waves per workgroup: one workgroup per CU shares one coordinate table
#endif
#ifndef LUTR_T2_WAVES_PER_EU
#define LUTR_T2_WAVES_PER_EU 4
#endif
#ifndef LUTR_T2_PIN
#define LUTR_T2_PIN 2             // wave-uniform constants copied to VGPRs: 1-2 kernel-wide (Y rows), 3-4 per tile (window, chroma rows)
#endif
#ifndef LUTR_T2_PK
#define LUTR_T2_PK 0              // bit mask (1: luma add of R,G; 2: blends; 4: * M; 8: chroma sums; 15 = round 2's "1"): packed fp32 (v_pk_add/mul_f32) for the R,G pair of the strict blends, the luma add and the chroma
                                  // sums -- bit-identical, 14 % fewer VALU instructions in the strict body (68.2 -> 58.7 per pixel), and
                                  // 1-2 % SLOWER: a packed op holds the fp32 pipe as long as the two scalar ops it replaces (4.6 vs 2 x 2.5 cycles)
#endif
#ifndef LUTR_T2_TUBE_BG
#define LUTR_T2_TUBE_BG 1         // the tube's second difference axis: 1 = (b - g), 0 = (b - r) as in round 2.  With BT.709 / 601 / 2020
                                  // g - r is almost -Cr and b - g almost +Cb (dG = -0.21 cb - 2.33 cr, dBG = 2.33 cb + 0.53 cr at 10 bit),
                                  // so the tube's cross-section is a near-square in the chroma plane; b - r = 2.12 cb - 1.80 cr makes it a
                                  // parallelogram stretched along the magenta-green diagonal and thin along orange-blue, where video lives
#endif
#ifndef LUTR_T2_MIXED
#define LUTR_T2_MIXED 1           // mixed tiles (see the vote in k_yuv_tile2)
#endif
#ifndef LUTR_T2_WIN_BG
#define LUTR_T2_WIN_BG LUTR_T2_TUBE_BG      // the per-wave windows' second difference axis (independent of the tube's)
#endif
#ifndef LUTR_T2_WIN_PAD
#define LUTR_T2_WIN_PAD 1         // 1: window plane strides padded against LDS bank collisions (win_plane_stride), 0: round 2's `| 1`
#endif
#ifndef LUTR_T2_PRIO
#define LUTR_T2_PRIO 1            // s_setprio around the phases of a tile.  1: a wave between its body and the next one -- stores, chunk
                                  // claim, the next tile's loads -- goes first, so memory operations leave as early as they can
                                  // (strict 587-589 -> 593 Gpx/s, fast +0.1 %); 2: the body goes first (-1.7 %); 0: off
#endif
#ifndef LUTR_T2_MIXED_FAST
#define LUTR_T2_MIXED_FAST 0      // 1: mixed tiles for the fast kernels too: sigma-16 frames 512 -> 537 Gpx/s, natural 662 -> 655, saturated 530 -> 522
                                  // (10 more VGPRs in a kernel that has them to lose; profiles/r03_exp34_mixed_tiles_fast_kernels.txt): off
#endif
#ifndef LUTR_T2_TRIREC
#define LUTR_T2_TRIREC 1          // 1: fast trilinear stages node + r-difference records (Node::rec)
#endif
#ifndef LUTR_T2_NODE16
#define LUTR_T2_NODE16 0          // 1: strict 4-tap kernels stage float4 nodes (one ds_read_b128 per tap, 4 LDS cycles) instead of 12-byte ones (ds_read2_b32 + ds_read_b32, 6 cycles)
#endif
#ifndef LUTR_T2_PHASES
#define LUTR_T2_PHASES 1          // scheduling barriers between the load and use phases of a pixel group (tile_body)
#endif

namespace lutr {
namespace t2 {

extern __shared__ __attribute__((aligned(16))) char smem[];

// LDS behind the coordinate table: 64 bytes of scratch per wave (window cell ranges, raw box, counters), then the WORKGROUP's
// chunk allocator (claim_chunk): {ticket} at +0, base[8] at +32, ready[8] at +64.
constexpr int kWaveScratch = LUTR_T2_WPB * 64;
constexpr int kScratch = kWaveScratch + 128;

enum { V_GEN = 0, V_TAB = 1, V_UNIT = 2, V_FAST = 3 };

#define DEV __device__ __forceinline__

// ---------------------------------------------------------------- small machine helpers
DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DEV float med3(float a, float lo, float hi) { return __builtin_amdgcn_fmed3f(a, lo, hi); }
DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
DEV float unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
DEV float in_vgpr(float s)
{
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}
DEV uint32_t pk_min(uint32_t a, uint32_t b) { uint32_t o; asm("v_pk_min_u16 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b)); return o; }
DEV uint32_t pk_max(uint32_t a, uint32_t b) { uint32_t o; asm("v_pk_max_u16 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b)); return o; }
// per 16-bit half: max(a - b, 0)
DEV uint32_t pk_subsat_sv(uint32_t a, uint32_t b) { uint32_t o; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(o) : "s"(a), "v"(b)); return o; }
DEV uint32_t pk_subsat_vs(uint32_t a, uint32_t b) { uint32_t o; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(o) : "v"(a), "s"(b)); return o; }

// fp32 weight x fp16 tap (+ fp32 accumulator): v_fma_mix_f32, products exact, one rounding (== fmaf(w, (float)h, c))
DEV float mix0_lo(float w, uint32_t h) { float d; asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(d) : "v"(w), "v"(h)); return d; }
DEV float mix0_hi(float w, uint32_t h) { float d; asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(w), "v"(h)); return d; }
DEV float mix_lo(float w, uint32_t h, float c) { float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(d) : "v"(w), "v"(h), "v"(c)); return d; }
DEV float mix_hi(float w, uint32_t h, float c) { float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(w), "v"(h), "v"(c)); return d; }
// fp16 - fp16 -> fp32 (one rounding), and f * t + fp16: the two halves of a lerp whose end points are fp16 taps
DEV float hsub_lo(uint32_t a, uint32_t b) { float d; asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
DEV float hsub_hi(uint32_t a, uint32_t b) { float d; asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
DEV float hlerp_lo(float t, float f, uint32_t v0) { float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(t), "v"(f), "v"(v0)); return d; }
DEV float hlerp_hi(float t, float f, uint32_t v0) { float d; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(t), "v"(f), "v"(v0)); return d; }
// fp16 difference * fp32 fraction + fp16 node: D in the low (D_HI = 0) or high half of `dw`, v0 in the low or high half of `vw`
template <int D_HI, int V_HI> DEV float reclerp(uint32_t dw, float f, uint32_t vw)
{
    float d;
    if constexpr (!D_HI && !V_HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(dw), "v"(f), "v"(vw));
    else if constexpr (!D_HI && V_HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(dw), "v"(f), "v"(vw));
    else if constexpr (D_HI && !V_HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(dw), "v"(f), "v"(vw));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(dw), "v"(f), "v"(vw));
    return d;
}
DEV uint32_t f2h_bits(float v) { return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)v); }     // v_cvt_f16_f32, round to nearest even
// the record of a node `a` whose r + 1 neighbour is `b` (both {r | g << 16, b} of fp16): differences exact in fp32, then rounded to fp16
struct u3 { uint32_t x, y, z; };
DEV u3 make_rec(uint2 a, uint2 b)
{
    const uint32_t dr = f2h_bits(hsub_lo(b.x, a.x)), dg = f2h_bits(hsub_hi(b.x, a.x)), db = f2h_bits(hsub_lo(b.y, a.y));
    return u3{a.x, (a.y & 0xffffu) | (dr << 16), dg | (db << 16)};
}

// {a.x + s, a.y + s} in one v_pk_add_f32: op_sel_hi makes the high half read the LOW dword of the second operand too (the
// compiler scalarises a splat add).  The second operand is a register pair whose high half is never read.
typedef float f2v __attribute__((ext_vector_type(2)));
DEV f2v pk_add_lo(f2v a, float s)
{
    f2v b, d;
    b.x = s;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// a * w.lo / a * w.hi on both halves (the weight pair is shared by two such products, no register is wasted), a * s with s
// broadcast from a scalar pair, and a + b / a - b as the compiler's own v_pk_add_f32.  Each half rounds exactly like the scalar op.
template <int HI> DEV f2v pk_mul_w(f2v a, f2v w)
{
    f2v d;
    if constexpr (HI) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(w));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(w));
    return d;
}
DEV f2v pk_mul_lo(f2v a, float s)
{
    f2v b, d;
    b.x = s;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

DEV int lds_base() { return (int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem; }

typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------- geometry shared with the launcher
struct Geom {
    int lw_log2;          // lanes along x per tile row = 1 << lw_log2 (the other lanes go down)
    int uw, urows;        // units per row, unit rows (a unit = PXT px x BH rows)
    int nsx, nry;         // tiles across, tiles down
    int ch, nrc, nchunks; // chunk = `ch` consecutive tile rows of one strip; chunk id = (frame*nrc + rc)*nsx + strip
    int win_nodes;        // LDS window capacity per wave, in nodes
    int tab_entries;      // per-code coordinate table at LDS offset 0 (0 = coordinates are computed)
    int max_raw;          // 2^din - 1: a raw code above it cannot be trusted to stay inside the table
    int whole;            // 1: the WHOLE lattice fits the workgroup's LDS (N <= 21 strict / 25 fast at 10 bit): staged once,
    int whole_a, whole_b; //    node (r, g, b) at index r * whole_a + g * whole_b + b (strides padded against LDS bank conflicts)
                          // per workgroup, shared by its waves, no windows, no validity tests -- content cannot matter
    int tube_h;           // > 0: a workgroup-shared "grey tube" of the lattice is staged once, behind the scratch area: every cell with
                          // |pg - pr| <= tube_h and |pb - pr| <= tube_h, all of r.  Tiles whose chroma keeps them inside it (tube_holds)
                          // need no window at all; the per-wave windows serve the saturated rest.
    int tube_plane;       // nodes from one r plane of the tube to the next: (2 tube_h + 3)^2 plus a few nodes of padding chosen by the
                          // launcher so that cells one step apart on any two axes never share an LDS bank (tube_plane_stride)
    int mix_max;          // a tile with at most this many lanes outside the tube runs as a MIXED tile (tube body + gather for the outliers)
    float tube_t;         // the test: |gv - rv| and |bu - rv| (RGB codes, chroma only) must stay <= tube_t over the lane's unit
    unsigned tube_rlo, tube_rhi;   // a raw chroma interval [lo, hi] (packed like the window boxes) that implies the test for both planes:
                                   // four saturating subtractions instead of ~35 VALU for the tiles that fit it (tube_rlo > tube_rhi: none)
    unsigned *queue;      // device words {claims, waves done}: both 0 between launches -- the last wave to leave resets them (queue_leave),
                          // so a launch needs no memset node in front of it
    unsigned qbase;       // waves in the grid = the first chunk the counter hands out (the waves' ids come before it)
    unsigned *stats;      // optional device counters; nullptr = off
};

struct Planes2 {          // 32-bit strides: the launcher only sends layouts that fit (planes_aligned)
    const uint8_t *s[3];
    uint8_t       *d[3];
    unsigned ss[3], ds[3];
    unsigned long long sfs[3], dfs[3];
};

// ---------------------------------------------------------------- variant traits
// INTERP as a template argument: lut3d's modes 0 / 1 / 2, and T2_TET16 = tetrahedral on float4 nodes -- the strict kernels' whole-lattice
// mode.  A tap is then one ds_read_b128 (4 LDS-array cycles) instead of ds_read2_b32 + ds_read_b32 (6), and scattered taps collide far
// less: with the whole 17^3 lattice in LDS, uniform-random frames 351 -> 515 Gpx/s, sigma-64 noise 390 -> 524, sigma-16 +3 %, natural
// frames the same (profiles/r03_exp35_whole_lattice_16byte_nodes.txt).  Everywhere else a third more nodes per KB is worth more (5.6).
constexpr int T2_TET16 = 3;
template <int INTERP, int V> struct Node {
    static constexpr bool fast = V == V_FAST;
    // bytes per node in a window: fast = four fp16 {r,g,b,-} (one ds_read_b64 per tap); strict 4-tap modes pack fp32 {r,g,b}
    // (a third more nodes per wave than float4; a tap is ds_read2_b32 + ds_read_b32); strict trilinear reads float4 nodes
    // fast trilinear stages RECORDS: the node plus the fp16 difference to its r + 1 neighbour, {r | g << 16, b | Dr << 16, Dg | Db << 16}
    // -- FFmpeg's first four lerps (along r) then take ONE v_fma_mix_f32 each instead of two, and a pixel reads 4 records instead of
    // 8 nodes.  (The strict kernels cannot afford it: fp32 records are 24 bytes, the tube would shrink to 4 cells and the windows to
    // nothing.)  The gather body rounds its differences to fp16 as well, so both paths compute the same number.
    static constexpr bool rec = LUTR_T2_TRIREC && fast && INTERP == LUTR_INTERP_TRILINEAR;
    static constexpr int lds = fast ? (rec ? 12 : 8) : ((INTERP == LUTR_INTERP_TRILINEAR || INTERP == T2_TET16 || LUTR_T2_NODE16) ? 16 : 12);
    static constexpr int glb = fast ? 8 : 16;                                              // bytes per node in HBM/L2
};

// ---------------------------------------------------------------- window (wave-uniform)
struct Win {
    float fr, fg, fb, fc;        // LDS byte address of the c000 tap: (int) fma(pr, fr, fma(pg, fg, fma(pb, fb, fc)))
    int   o_r, o_g;              // byte steps of +1 along r and g (blue is one node)
};
// The raw-code box a window is good for lives in the wave's LDS scratch, not in registers: {ylo, yhi, cblo, cbhi} at +32,
// {crlo, crhi} at +48, each bound replicated in both 16-bit halves (8-bit planes: code << 8, upper bounds | 0xff -- see
// extremes()).  Registers are what this kernel runs out of, and a value spilled to scratch is worse than it looks: a
// scratch_load in the per-tile path makes the wave wait with vmcnt(0), i.e. for the next tile's prefetch it has just issued.
struct Box { uint32_t ylo, yhi, cblo, cbhi, crlo, crhi; };
DEV void box_store(int scratch_off, const Box &b)
{
    *(uint4 *)(smem + scratch_off + 32) = make_uint4(b.ylo, b.yhi, b.cblo, b.cbhi);
    *(uint2 *)(smem + scratch_off + 48) = make_uint2(b.crlo, b.crhi);
}

struct Ext { uint32_t ymin, ymax, cbmin, cbmax, crmin, crmax; };   // packed per-lane extremes of a unit

// (prev, frac) of one channel
struct Crd { float p, d; };

template <int INTERP>
DEV Crd crd_compute(const LutConsts &L, float code, float sc)
{
    const float x = code * L.scale_f;
    const float s = fminf(x * sc, L.lut_max);       // codes and scales are >= 0: only the upper clip can bind
    Crd c;
    if constexpr (INTERP == LUTR_INTERP_NEAREST) {  // NEAR(x) = (int)(x + .5) with a double .5 (lutr_device.h near_f)
        const float fl = floorf(s);
        c.p = (s - fl >= .5f) ? fl + 1.0f : fl;
        c.d = 0.0f;
    } else { c.p = floorf(s); c.d = s - c.p; }
    return c;
}

DEV Crd crd_table(unsigned idx)
{
    const float2 e = *(const float2 *)(smem + idx * 8u);
    return Crd{e.x, e.y};
}
// The per-pixel paths index the table by BYTE offset: their YUV -> RGB constants are pre-multiplied by 8 (KB in the kernel;
// a power of two commutes with every rounding), so (unsigned)(8 * v) & ~7 == 8 * floor(v): the `* 8` that was a
// quarter-rate v_lshl_add_u32 per channel is a v_and_b32 on the other pipe.
// The table sits at LDS address 0 (the kernel has no static LDS, the dynamic block starts at 0 -- checked at kernel start):
// the offset IS the address, no v_add of the block's base.
typedef __attribute__((address_space(3))) const f2v *lds_f2p;
typedef __attribute__((address_space(3))) const float *lds_fp;
DEV Crd crd_table8(unsigned off)
{
    const f2v e = *(lds_f2p)(uintptr_t)off;
    return Crd{e.x, e.y};
}

// Entry i of the table = coordinates of code min(i, M): indices past M are what an out-of-gamut YUV triple produces
// before FFmpeg's clip to the depth; the clip is folded into the table.
template <int INTERP>
DEV void coord_table_fill(const LutConsts &L, int entries)
{
    for (int q = threadIdx.x; q < entries; q += 64 * LUTR_T2_WPB) {
        Crd c;
        if (L.pre) {                 // a shared prelut (LutConsts::pre_shared): the folded coordinate of the code, then prev / frac as ever
            const float s = L.pre[min(q, (int)L.maxf)];
            if constexpr (INTERP == LUTR_INTERP_NEAREST) { const float fl = floorf(s); c.p = (s - fl >= .5f) ? fl + 1.0f : fl; c.d = 0.0f; }
            else { c.p = floorf(s); c.d = s - c.p; }
        } else c = crd_compute<INTERP>(L, fminf((float)q, L.maxf), L.sc[0]);
        *(float2 *)(smem + q * 8) = make_float2(c.p, c.d);
    }
    __syncthreads();
}

// ---------------------------------------------------------------- tiles
template <int WIN, int WOUT, int CSX, int CSY>
struct Tile {
    static constexpr int PXT = WIN ? 8 : 16;                         // luma samples per unit row (16 bytes in)
    static constexpr int BH = 1 << CSY, BW = 1 << CSX;
    static constexpr int NC = PXT >> CSX;                            // chroma samples per unit
    static constexpr int YWI = 4, YWO = PXT * (WOUT ? 2 : 1) / 4;    // luma words per row, in / out
    static constexpr int CWI = NC * (WIN ? 2 : 1) / 4, CWO = NC * (WOUT ? 2 : 1) / 4;
    static_assert(CWI >= 1 && CWO >= 1, "a unit must own whole chroma words");
};
template <int WIN, int WOUT, int CSX, int CSY> struct TileIn { using T = Tile<WIN, WOUT, CSX, CSY>; uint32_t y[T::BH][T::YWI], cb[T::CWI], cr[T::CWI]; };
template <int WIN, int WOUT, int CSX, int CSY> struct TileOut { using T = Tile<WIN, WOUT, CSX, CSY>; uint32_t y[T::BH][T::YWO], cb[T::CWO], cr[T::CWO]; };

template <int WIDE>
DEV float wsample(const uint32_t *w, int i)
{
    if constexpr (WIDE) return (float)((w[i >> 1] >> ((i & 1) * 16)) & 0xffffu);
    else return (float)((w[i >> 2] >> ((i & 3) * 8)) & 0xffu);
}

// floor(v) for v >= 0 (negatives saturate to 0) written into sample i of the word vector: one SDWA conversion
// (v_cvt_u32_f32 truncates) instead of convert + shift-or.
template <int WIDE>
DEV void wput(uint32_t *w, int i, float v)
{
    if constexpr (WIDE) {
        if ((i & 1) == 0) w[i >> 1] = (uint32_t)v;
        else asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w[i >> 1]) : "v"(v));
    } else {
        uint32_t &d = w[i >> 2];
        switch (i & 3) {
        case 0: d = (uint32_t)v; break;
        case 1: asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(v)); break;
        case 2: asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(v)); break;
        default: asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(d) : "v"(v)); break;
        }
    }
}

template <int NW> DEV void ldw(uint32_t *w, const uint8_t *p)
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
template <int NW> DEV void stw(uint8_t *p, const uint32_t *w)
{
#if LUTR_NT
    typedef unsigned nt4 __attribute__((ext_vector_type(4)));
    typedef unsigned nt2 __attribute__((ext_vector_type(2)));
    if constexpr (NW == 8) { __builtin_nontemporal_store(nt4{w[0], w[1], w[2], w[3]}, (nt4 *)p); __builtin_nontemporal_store(nt4{w[4], w[5], w[6], w[7]}, (nt4 *)(p + 16)); }
    else if constexpr (NW == 4) __builtin_nontemporal_store(nt4{w[0], w[1], w[2], w[3]}, (nt4 *)p);
    else if constexpr (NW == 2) __builtin_nontemporal_store(nt2{w[0], w[1]}, (nt2 *)p);
    else __builtin_nontemporal_store(w[0], (uint32_t *)p);
#else
    if constexpr (NW == 8) { *(uint4 *)p = make_uint4(w[0], w[1], w[2], w[3]); *(uint4 *)(p + 16) = make_uint4(w[4], w[5], w[6], w[7]); }
    else if constexpr (NW == 4) *(uint4 *)p = make_uint4(w[0], w[1], w[2], w[3]);
    else if constexpr (NW == 2) *(uint2 *)p = make_uint2(w[0], w[1]);
    else *(uint32_t *)p = w[0];
#endif
}

template <int N> DEV void fence_words(uint32_t *w)
{
    if constexpr (N == 1) asm volatile("" : "+v"(w[0]));
    else if constexpr (N == 2) asm volatile("" : "+v"(w[0]), "+v"(w[1]));
    else if constexpr (N == 4) asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    else if constexpr (N == 8) asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
    else if constexpr (N == 16) { fence_words<8>(w); fence_words<8>(w + 8); }
}

// ---------------------------------------------------------------- raw extremes of a unit
// 16-bit containers: the words ARE pairs of samples.  8-bit: a word and the word shifted left by 8 are two streams of
// u16 whose HIGH byte is a sample (the low byte only breaks ties), so the same packed min/max yields exact byte
// extremes in the high bytes; bounds are compared as (code << 8) and (code << 8 | 0xff).
template <int WIN, int N>
DEV void ext_words(const uint32_t *w, uint32_t &mn, uint32_t &mx, bool first)
{
#pragma unroll
    for (int k = 0; k < N; k++) {
        if (first && k == 0) { mn = mx = w[0]; }
        else { mn = pk_min(mn, w[k]); mx = pk_max(mx, w[k]); }
        if constexpr (!WIN) { const uint32_t t = w[k] << 8; mn = pk_min(mn, t); mx = pk_max(mx, t); }
    }
}

// max only (the luma minimum is wanted by the window tests alone: tiles the tube takes never compute it)
template <int WIN, int N>
DEV void max_words(const uint32_t *w, uint32_t &mx, bool first)
{
#pragma unroll
    for (int k = 0; k < N; k++) {
        if (first && k == 0) mx = w[0];
        else mx = pk_max(mx, w[k]);
        if constexpr (!WIN) mx = pk_max(mx, w[k] << 8);
    }
}
template <int WIN, int N>
DEV void min_words(const uint32_t *w, uint32_t &mn, bool first)
{
#pragma unroll
    for (int k = 0; k < N; k++) {
        if (first && k == 0) mn = w[0];
        else mn = pk_min(mn, w[k]);
        if constexpr (!WIN) mn = pk_min(mn, w[k] << 8);
    }
}

// everything but the luma minimum (luma_min below)
template <int WIN, int WOUT, int CSX, int CSY>
DEV Ext extremes(const TileIn<WIN, WOUT, CSX, CSY> &in)
{
    using T = Tile<WIN, WOUT, CSX, CSY>;
    Ext e;
    e.ymin = 0u;
#pragma unroll
    for (int dy = 0; dy < T::BH; dy++) max_words<WIN, T::YWI>(in.y[dy], e.ymax, dy == 0);
    ext_words<WIN, T::CWI>(in.cb, e.cbmin, e.cbmax, true);
    ext_words<WIN, T::CWI>(in.cr, e.crmin, e.crmax, true);
    return e;
}
template <int WIN, int WOUT, int CSX, int CSY>
DEV uint32_t luma_min(const TileIn<WIN, WOUT, CSX, CSY> &in)
{
    using T = Tile<WIN, WOUT, CSX, CSY>;
    uint32_t mn = 0u;
#pragma unroll
    for (int dy = 0; dy < T::BH; dy++) min_words<WIN, T::YWI>(in.y[dy], mn, dy == 0);
    return mn;
}

DEV uint32_t pk_subsat(uint32_t a, uint32_t b) { uint32_t o; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(o) : "v"(a), "v"(b)); return o; }

DEV bool box_holds(int scratch_off, const Ext &e)
{
    const uint4 p = *(const uint4 *)(smem + scratch_off + 32);      // ylo, yhi, cblo, cbhi
    const uint2 q = *(const uint2 *)(smem + scratch_off + 48);      // crlo, crhi
    // every half >= lo and <= hi  <=>  all six saturating differences are zero
    const uint32_t a = pk_subsat(p.x, e.ymin) | pk_subsat(e.ymax, p.y) | pk_subsat(p.z, e.cbmin);
    const uint32_t b = pk_subsat(e.cbmax, p.w) | pk_subsat(q.x, e.crmin) | pk_subsat(e.crmax, q.y);
    return __all((a | b) == 0u);
}

// ---------------------------------------------------------------- box -> cells (conservative, wave-uniform)
// Raw box (float codes, inclusive) -> inclusive ranges of the sheared cell coordinates (pr, pg-pr, pb-pr) that any
// pixel inside the box can produce.  Every float op of the pixel pipeline is monotone in its inputs, so per-channel
// cell ranges follow exactly from evaluating the SAME ops at the right corners.  For the two difference axes that would
// throw away the one thing that makes the sheared window small (luma cancels), so they are bounded analytically:
//   pg - pr in (sg - sr - 1, sg - sr + 1),  sg - sr = kappa (qG - qR) +- 2 delta_s,
//   qG - qR in (dG - 1 - eps, dG + 1 + eps) with dG = gv - rv a function of chroma only (monotone in cb and in cr),
// and where FFmpeg's clip to [0, M] can bind inside the box the clipped difference lies between 0 and the unclipped one.
// The result is intersected with plain interval arithmetic on the per-channel ranges (also valid).
struct Cells { int r0, r1, g0, g1, b0, b1; };

template <int INTERP, int PRE, int V>
DEV float cell_of(const LutConsts &L, float q, int ch, int tab_entries)
{
    if constexpr (V >= V_TAB) { (void)ch; (void)tab_entries; return crd_table((unsigned)q).p; }
    else { (void)tab_entries; return crd_compute<INTERP>(L, q, L.sc[ch]).p; }
}

DEV float qclip(float v, float m) { return med3(floorf(v), 0.0f, m); }
DEV float cfloor(float v, float hi) { return med3(floorf(v), 0.0f, hi); }

template <int INTERP, int PRE, int V>
DEV Cells map_box(const LutConsts &L, const YuvConsts &K, const Geom &TG, float y0, float y1, float cb0, float cb1, float cr0, float cr1)
{
    if constexpr (PRE) {       // the range/depth prologue is a monotone map of each plane
        y0 = qclip(fma_(K.py, y0, K.pyb), K.pre_max); y1 = qclip(fma_(K.py, y1, K.pyb), K.pre_max);
        cb0 = qclip(fma_(K.pc, cb0, K.pcb), K.pre_max); cb1 = qclip(fma_(K.pc, cb1, K.pcb), K.pre_max);
        cr0 = qclip(fma_(K.pc, cr0, K.pcb), K.pre_max); cr1 = qclip(fma_(K.pc, cr1, K.pcb), K.pre_max);
    }
    const float yy0 = fma_(K.ky, y0, K.yb), yy1 = fma_(K.ky, y1, K.yb);
    const float cbd0 = cb0 - K.coff, cbd1 = cb1 - K.coff, crd0 = cr0 - K.coff, crd1 = cr1 - K.coff;
    // krv, kbu > 0; kgu, kgv < 0 for every matrix (make_yuv_consts)
    const float rv0 = K.krv * crd0, rv1 = K.krv * crd1, bu0 = K.kbu * cbd0, bu1 = K.kbu * cbd1;
    const float gv0 = fma_(K.kgu, cbd1, K.kgv * crd1), gv1 = fma_(K.kgu, cbd0, K.kgv * crd0);      // min, max
    const float m = K.max_l;
    const float vr0 = yy0 + rv0, vr1 = yy1 + rv1, vg0 = yy0 + gv0, vg1 = yy1 + gv1, vb0 = yy0 + bu0, vb1 = yy1 + bu1;
    const int te = TG.tab_entries;
    const float pr0 = cell_of<INTERP, PRE, V>(L, qclip(vr0, m), 0, te), pr1 = cell_of<INTERP, PRE, V>(L, qclip(vr1, m), 0, te);
    const float pg0 = cell_of<INTERP, PRE, V>(L, qclip(vg0, m), 1, te), pg1 = cell_of<INTERP, PRE, V>(L, qclip(vg1, m), 1, te);
    const float pb0 = cell_of<INTERP, PRE, V>(L, qclip(vb0, m), 2, te), pb1 = cell_of<INTERP, PRE, V>(L, qclip(vb1, m), 2, te);
    Cells c;
    c.r0 = (int)pr0; c.r1 = (int)pr1;
    // the second difference axis is (b - g) [LUTR_T2_WIN_BG] or (b - r)
    float g_lo = pg0 - pr1, g_hi = pg1 - pr0;                                                   // interval arithmetic
    float b_lo = LUTR_T2_WIN_BG ? pb0 - pg1 : pb0 - pr1, b_hi = LUTR_T2_WIN_BG ? pb1 - pg0 : pb1 - pr0;
    if (L.pre || (L.sc[0] == L.sc[1] && L.sc[1] == L.sc[2])) {
        // chroma-only difference terms at the corners that extremise them: gv - rv falls in cb and in cr;
        // bu - rv rises in cb and falls in cr; bu - gv rises in both (gv1 = gv at (cb0, cr0), gv0 = gv at (cb1, cr1))
        float dg0 = gv0 - rv1, dg1 = gv1 - rv0;
        float db0 = LUTR_T2_WIN_BG ? bu0 - gv1 : bu0 - rv1, db1 = LUTR_T2_WIN_BG ? bu1 - gv0 : bu1 - rv0;
        // (a shared prelut is a monotone map whose slope lies between 0 and pre_kappa cells per code: a code difference d moves the
        // cell by at most pre_kappa |d| and possibly not at all -- the same one-sided form as where the clip can bind)
        const bool clips = L.pre || vr0 < 0.0f || vg0 < 0.0f || vb0 < 0.0f || vr1 >= m + 1.0f || vg1 >= m + 1.0f || vb1 >= m + 1.0f;
        if (clips) { dg0 = fminf(dg0, 0.0f); dg1 = fmaxf(dg1, 0.0f); db0 = fminf(db0, 0.0f); db1 = fmaxf(db1, 0.0f); }
        const float kappa = L.pre ? L.pre_kappa : L.sc[0] * L.scale_f;
        const float eps = m * (1.0f / 2097152.0f) + 1e-3f;                    // float rounding of the sums, and then some
        const float slack = 1.0f + 2.0f * L.lut_max * (1.0f / 2097152.0f) + 2e-3f;
        g_lo = fmaxf(g_lo, floorf(kappa * (dg0 - 1.0f - eps) - slack) + 1.0f);
        g_hi = fminf(g_hi, ceilf(kappa * (dg1 + 1.0f + eps) + slack) - 1.0f);
        b_lo = fmaxf(b_lo, floorf(kappa * (db0 - 1.0f - eps) - slack) + 1.0f);
        b_hi = fminf(b_hi, ceilf(kappa * (db1 + 1.0f + eps) + slack) - 1.0f);
    }
    c.g0 = (int)g_lo; c.g1 = (int)g_hi; c.b0 = (int)b_lo; c.b1 = (int)b_hi;
    return c;
}

// ---------------------------------------------------------------- exact cell bounds of a tile (second-level test)
// When the raw-box test fails the tile is not necessarily outside the window: the box is a conservative, axis-aligned
// description (uncorrelated sensor noise in Cb and Cr spans a box whose corners no pixel reaches).  The second level
// computes what round 1 computed for every tile: the exact cells each pixel touches, min/max per lane, one vote against
// the window's cell ranges.  ~18 VALU per pixel, paid only by the tiles that fail the cheap test.
struct Bnd { float rmin, rmax, gmin, gmax, bmin, bmax; };

DEV float vmin3(float a, float b, float c) { float o; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c)); return o; }
DEV float vmax3(float a, float b, float c) { float o; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c)); return o; }

template <int WIN, int WOUT, int CSX, int CSY, int INTERP, int PRE, int V>
DEV Bnd tile_bounds(const LutConsts &L, const YuvConsts &K, const Geom &TG, TileIn<WIN, WOUT, CSX, CSY> &in)
{
    using T = Tile<WIN, WOUT, CSX, CSY>;
    Bnd bn;
    bn.rmin = bn.gmin = bn.bmin = 1e9f;
    bn.rmax = bn.gmax = bn.bmax = -1e9f;
#pragma unroll
    for (int j = 0; j < T::NC; j++) {
        float cbv = wsample<WIN>(in.cb, j), crv = wsample<WIN>(in.cr, j);
        if constexpr (PRE) {
            cbv = cfloor(fma_(K.pc, cbv, K.pcb), K.pre_max);
            crv = cfloor(fma_(K.pc, crv, K.pcb), K.pre_max);
        }
        const float cbd = cbv - K.coff, crd = crv - K.coff;
        const float rv = K.krv * crd, gv = fma_(K.kgu, cbd, K.kgv * crd), bu = K.kbu * cbd;
        constexpr int NQ = T::BH * T::BW;
        float pr[NQ], pg_[NQ], pb_[NQ], hg[NQ], hb[NQ];
        // all table reads of the chroma block first, then their uses (one LDS round trip per block, not per pixel)
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int dy = q / T::BW, dx = q % T::BW;
            float yv = wsample<WIN>(in.y[dy], j * T::BW + dx);
            if constexpr (PRE) yv = cfloor(fma_(K.py, yv, K.pyb), K.pre_max);
            const float yy = fma_(K.ky, yv, K.yb);
            if constexpr (V >= V_TAB) {
                // K is the kernel's KB; the caller has checked that every raw code is legal, so the padded table covers the sums
                pr[q] = *(lds_fp)(uintptr_t)((unsigned)(yy + rv) & ~7u);
                pg_[q] = *(lds_fp)(uintptr_t)((unsigned)(yy + gv) & ~7u);
                pb_[q] = *(lds_fp)(uintptr_t)((unsigned)(yy + bu) & ~7u);
            } else {
                pr[q] = crd_compute<INTERP>(L, cfloor(yy + rv, K.max_l), L.sc[0]).p;
                pg_[q] = crd_compute<INTERP>(L, cfloor(yy + gv, K.max_l), L.sc[1]).p;
                pb_[q] = crd_compute<INTERP>(L, cfloor(yy + bu, K.max_l), L.sc[2]).p;
            }
        }
#if LUTR_T2_PHASES
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int q = 0; q < NQ; q++) { hg[q] = pg_[q] - pr[q]; hb[q] = pb_[q] - (LUTR_T2_WIN_BG ? pg_[q] : pr[q]); }
        if constexpr (T::BH * T::BW >= 2) {
#pragma unroll
            for (int q = 0; q + 1 < T::BH * T::BW; q += 2) {
                bn.rmin = vmin3(bn.rmin, pr[q], pr[q + 1]); bn.rmax = vmax3(bn.rmax, pr[q], pr[q + 1]);
                bn.gmin = vmin3(bn.gmin, hg[q], hg[q + 1]); bn.gmax = vmax3(bn.gmax, hg[q], hg[q + 1]);
                bn.bmin = vmin3(bn.bmin, hb[q], hb[q + 1]); bn.bmax = vmax3(bn.bmax, hb[q], hb[q + 1]);
            }
        } else {
            bn.rmin = fminf(bn.rmin, pr[0]); bn.rmax = fmaxf(bn.rmax, pr[0]);
            bn.gmin = fminf(bn.gmin, hg[0]); bn.gmax = fmaxf(bn.gmax, hg[0]);
            bn.bmin = fminf(bn.bmin, hb[0]); bn.bmax = fmaxf(bn.bmax, hb[0]);
        }
        // keep the blocks in program order (see tile_body): the fences do not change the words
        fence_words<T::YWI * T::BH>(&in.y[0][0]);
        fence_words<T::CWI>(in.cb);
        fence_words<T::CWI>(in.cr);
        asm volatile("" : "+v"(bn.rmin), "+v"(bn.rmax), "+v"(bn.gmin), "+v"(bn.gmax), "+v"(bn.bmin), "+v"(bn.bmax));
    }
    return bn;
}

// the window's cell ranges live in the wave's 64-byte LDS scratch (they are only read on the second-level path)
DEV bool cells_hold(int scratch_off, const Bnd &b)
{
    const float4 lo = *(const float4 *)(smem + scratch_off);        // r_lo, g_lo, b_lo, r_hi
    const float2 hi = *(const float2 *)(smem + scratch_off + 16);   // g_hi, b_hi
    const bool ok = b.rmin >= lo.x && b.rmax <= lo.w && b.gmin >= lo.y && b.gmax <= hi.x && b.bmin >= lo.z && b.bmax <= hi.y;
    return __all(ok);
}

template <int WIN> DEV int lo_of_(uint32_t mn) { const int a = (int)(mn & 0xffffu), b = (int)(mn >> 16); const int v = a < b ? a : b; return WIN ? v : v >> 8; }
template <int WIN> DEV int hi_of_(uint32_t mx) { const int a = (int)(mx & 0xffffu), b = (int)(mx >> 16); const int v = a > b ? a : b; return WIN ? v : v >> 8; }

// ---------------------------------------------------------------- the grey tube
// Is every pixel of the tile inside the tube?  In sheared coordinates luma cancels out of the two difference axes (map_box), so the
// answer depends on chroma alone: with dG = gv - rv and dB = bu - rv (monotone in cb and cr, evaluated at the corners of the lane's
// chroma extremes), every cell has |pg - pr| <= H and |pb - pr| <= H as soon as |dG|, |dB| <= tube_t =
// (H + 1 - slack) / kappa - 1 - eps -- map_box's own bound solved for d.  Where FFmpeg's clip to [0, M] binds, the clipped
// difference lies between 0 and the unclipped one, and 0 is inside.  One vote per tile, ~35 VALU per lane, no luma needed.
template <int WIN, int PRE>
DEV bool tube_holds(const YuvConsts &K, const Geom &TG, const Ext &e)
{
    float cb0 = (float)lo_of_<WIN>(e.cbmin), cb1 = (float)hi_of_<WIN>(e.cbmax);
    float cr0 = (float)lo_of_<WIN>(e.crmin), cr1 = (float)hi_of_<WIN>(e.crmax);
    if constexpr (PRE) {       // the range/depth prologue is a monotone map of each plane
        cb0 = qclip(fma_(K.pc, cb0, K.pcb), K.pre_max); cb1 = qclip(fma_(K.pc, cb1, K.pcb), K.pre_max);
        cr0 = qclip(fma_(K.pc, cr0, K.pcb), K.pre_max); cr1 = qclip(fma_(K.pc, cr1, K.pcb), K.pre_max);
    }
    const float cbd0 = cb0 - K.coff, cbd1 = cb1 - K.coff, crd0 = cr0 - K.coff, crd1 = cr1 - K.coff;
    // krv, kbu > 0; kgu, kgv < 0 for every matrix (make_yuv_consts)
    const float rv0 = K.krv * crd0, rv1 = K.krv * crd1, bu0 = K.kbu * cbd0, bu1 = K.kbu * cbd1;
    const float gv0 = fma_(K.kgu, cbd1, K.kgv * crd1), gv1 = fma_(K.kgu, cbd0, K.kgv * crd0);      // min, max
    // (b - g: bu - gv rises in cb and in cr, its extremes sit at the corners (cb0, cr0) and (cb1, cr1) where gv is gv1 and gv0)
    const float dg0 = gv0 - rv1, dg1 = gv1 - rv0;
    const float db0 = LUTR_T2_TUBE_BG ? bu0 - gv1 : bu0 - rv1, db1 = LUTR_T2_TUBE_BG ? bu1 - gv0 : bu1 - rv0;
    const float worst = vmax3(fmaxf(-dg0, dg1), -db0, db1);
    return worst <= TG.tube_t;          // per lane: the caller votes
}

// The same bound sample by sample, for units with at most four chroma samples (4:2:0 and 4:2:2 at 10 bit): a pixel's chroma IS one
// of them, so no pairing of one sample's Cb with another's Cr inflates the differences.  About the cost of the corner form; under
// per-sample chroma noise it keeps more tiles in the tube (sigma = 16, strict kernels' H = 6: 17 % -> 32 % of the tiles).
template <int WIN, int WOUT, int CSX, int CSY, int PRE>
DEV bool tube_holds_samples(const YuvConsts &K, const Geom &TG, const TileIn<WIN, WOUT, CSX, CSY> &in)
{
    using T = Tile<WIN, WOUT, CSX, CSY>;
    float worst = 0.0f;
#pragma unroll
    for (int j = 0; j < T::NC; j++) {
        float cbv = wsample<WIN>(in.cb, j), crv = wsample<WIN>(in.cr, j);
        if constexpr (PRE) {
            cbv = cfloor(fma_(K.pc, cbv, K.pcb), K.pre_max);
            crv = cfloor(fma_(K.pc, crv, K.pcb), K.pre_max);
        }
        const float cbd = cbv - K.coff, crd = crv - K.coff;
        const float rv = K.krv * crd, gv = fma_(K.kgu, cbd, K.kgv * crd), bu = K.kbu * cbd;
        worst = vmax3(worst, fabsf(gv - rv), fabsf(LUTR_T2_TUBE_BG ? bu - gv : bu - rv));
    }
    return worst <= TG.tube_t;          // per lane: the caller votes
}

// ---------------------------------------------------------------- restage
DEV uint32_t shx(uint32_t v, int m) { return (uint32_t)__shfl_xor((int)v, m, 64); }
DEV float wave_min(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
    return v;
}
DEV float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

template <int WIN> DEV int lo_of(uint32_t mn) { const int a = (int)(mn & 0xffffu), b = (int)(mn >> 16); const int v = a < b ? a : b; return WIN ? v : v >> 8; }
template <int WIN> DEV int hi_of(uint32_t mx) { const int a = (int)(mx & 0xffffu), b = (int)(mx >> 16); const int v = a > b ? a : b; return WIN ? v : v >> 8; }
template <int WIN> DEV uint32_t pack_lo(int v) { const uint32_t h = WIN ? (uint32_t)v : (uint32_t)v << 8; return h | (h << 16); }
template <int WIN> DEV uint32_t pack_hi(int v) { const uint32_t h = WIN ? (uint32_t)v : ((uint32_t)v << 8) | 0xffu; return h | (h << 16); }

// The window holds this tile (second-level test) but its raw box does not -- typically a window that was staged around the
// exact cells of an edge tile and has no box at all.  Try to give it one without staging anything: the largest of a few raw
// boxes around this tile's raw extremes whose conservative cell map lies inside the window's cell ranges.  Called with
// back-off (1st, 4th, 16th consecutive second-level pass), so a wave leaves the expensive mode once it is past the edge.
template <int WIN, int INTERP, int PRE, int V>
DEV bool rebox(const LutConsts &L, const YuvConsts &K, const Geom &TG, const Ext &e_, int scratch_off, int lane)
{
    Ext e = e_;
    asm volatile("" : "+v"(e.ymin), "+v"(e.ymax), "+v"(e.cbmin), "+v"(e.cbmax), "+v"(e.crmin), "+v"(e.crmax));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        e.ymin = pk_min(e.ymin, shx(e.ymin, m)); e.ymax = pk_max(e.ymax, shx(e.ymax, m));
        e.cbmin = pk_min(e.cbmin, shx(e.cbmin, m)); e.cbmax = pk_max(e.cbmax, shx(e.cbmax, m));
        e.crmin = pk_min(e.crmin, shx(e.crmin, m)); e.crmax = pk_max(e.crmax, shx(e.crmax, m));
    }
    const int y0 = uni(lo_of<WIN>(e.ymin)), y1 = uni(hi_of<WIN>(e.ymax));
    const int cb0 = uni(lo_of<WIN>(e.cbmin)), cb1 = uni(hi_of<WIN>(e.cbmax));
    const int cr0 = uni(lo_of<WIN>(e.crmin)), cr1 = uni(hi_of<WIN>(e.crmax));
    const int mr = TG.max_raw;
    const float4 lo = *(const float4 *)(smem + scratch_off);        // r_lo, g_lo, b_lo, r_hi
    const float2 hi = *(const float2 *)(smem + scratch_off + 16);   // g_hi, b_hi
    const int wr0 = uni((int)lo.x), wg0 = uni((int)lo.y), wb0 = uni((int)lo.z), wr1 = uni((int)lo.w), wg1 = uni((int)hi.x), wb1 = uni((int)hi.y);
    const float cell_y = L.maxf / (L.lut_max * K.ky * (PRE ? K.py : 1.0f));
    const int unit = (mr + 1) >> 8;
#pragma unroll 1
    for (int t = 0; t < 3; t++) {
        const int mc = t == 0 ? unit : (t == 1 ? (unit + 3) / 4 : 0);
        const int my = t == 2 ? 0 : (int)((t == 0 ? 1.5f : 0.5f) * cell_y);
        const int ty0 = max(y0 - my, 0), ty1 = min(y1 + my, mr);
        const int tcb0 = max(cb0 - mc, 0), tcb1 = min(cb1 + mc, mr), tcr0 = max(cr0 - mc, 0), tcr1 = min(cr1 + mc, mr);
        const Cells c = map_box<INTERP, PRE, V>(L, K, TG, (float)ty0, (float)ty1, (float)tcb0, (float)tcb1, (float)tcr0, (float)tcr1);
        const bool fits = uni((int)(c.r0 >= wr0 && c.r1 <= wr1 && c.g0 >= wg0 && c.g1 <= wg1 && c.b0 >= wb0 && c.b1 <= wb1)) != 0;
        if (fits) {
            if (lane == 0) {
                Box bx;
                bx.ylo = pack_lo<WIN>(ty0); bx.yhi = pack_hi<WIN>(ty1);
                bx.cblo = pack_lo<WIN>(tcb0); bx.cbhi = pack_hi<WIN>(tcb1);
                bx.crlo = pack_lo<WIN>(tcr0); bx.crhi = pack_hi<WIN>(tcr1);
                box_store(scratch_off, bx);
            }
            return true;
        }
    }
    return false;
}

// Neither test vouches for the tile: stage a new window around the tile's EXACT cells (spare cells on the two
// chroma-like axes as capacity allows, the luma-like axis gets the rest), then look for the largest raw box around the
// tile's raw extremes whose conservative cell map (map_box) lies inside what was staged: that box is what the cheap
// first-level test of the following tiles runs against (it may come out empty for very noisy content; those tiles then
// pay the second level).  Returns false (W untouched; the caller runs the global-gather body for this tile) when the
// tile's colours do not fit a window, or a raw code lies above 2^din - 1.
// Plane stride of a window: the smallest stride >= `nodes` for which cells one step apart on any axes never share an LDS bank
// (see tube_plane_stride in the launcher: node index = pr * A + pg * B + pb, collision when dr A + dg B + db = 0 mod 32, i.e.
// when A mod 32 lies within one of 0, +B or -B).  kGoodA[B mod 32] has bit (A mod 32) set for the strides that are fine; a
// rotate and a find-first-set pick the padding -- this runs in every restage, and every wave starts with one (a search loop
// here cost a short launch 30 us).
__constant__ unsigned kGoodA[32] = {
    0x00000000u, 0x00000000u, 0x1ffffff0u, 0x0fffffe0u, 0x47ffffc4u, 0x63ffff8cu, 0x71ffff1cu, 0x78fffe3cu, 0x7c7ffc7cu, 0x7e3ff8fcu,
    0x7f1ff1fcu, 0x7f8fe3fcu, 0x7fc7c7fcu, 0x7fe38ffcu, 0x7ff11ffcu, 0x7ff83ffcu, 0x7ffc7ffcu, 0x7ff83ffcu, 0x7ff11ffcu, 0x7fe38ffcu,
    0x7fc7c7fcu, 0x7f8fe3fcu, 0x7f1ff1fcu, 0x7e3ff8fcu, 0x7c7ffc7cu, 0x78fffe3cu, 0x71ffff1cu, 0x63ffff8cu, 0x47ffffc4u, 0x0fffffe0u,
    0x1ffffff0u, 0x00000000u};
__constant__ unsigned short kGoodA16[16] = {      // the same for 16-byte nodes (ds_read_b128: indices collide mod 16)
    0x0000u, 0x0000u, 0x1ff0u, 0x0fe0u, 0x47c4u, 0x638cu, 0x711cu, 0x783cu, 0x7c7cu, 0x783cu, 0x711cu, 0x638cu, 0x47c4u, 0x0fe0u, 0x1ff0u, 0x0000u};
template <int NODE>
DEV int win_plane_stride(int nodes, int nb)
{
    if (!LUTR_T2_WIN_PAD) return nodes | 1;
    constexpr int M = NODE == 16 ? 16 : 32;
    const int B = LUTR_T2_WIN_BG ? nb - 1 : nb, a0 = (LUTR_T2_WIN_BG ? nodes - nb : nodes - nb - 1) & (M - 1);
    const unsigned m = NODE == 16 ? (unsigned)kGoodA16[B & 15] : kGoodA[B & 31];
    const unsigned r = a0 ? ((m >> a0) | (m << (M - a0))) & (NODE == 16 ? 0xffffu : 0xffffffffu) : m;      // bit k: padding k is fine
    return r ? nodes + __builtin_ctz(r) : (nodes | 1);
}

template <int WIN, int INTERP, int PRE, int V>
DEV bool restage(Win &W, const LutConsts &L, const YuvConsts &K, const Geom &TG, const Ext &e_, const Bnd &bn_, int slice_off,
                 int scratch_off, int lane_)
{
    using N = Node<INTERP, V>;
    constexpr int kLN = N::lds;
    // the lane id, recomputed: kept live from the kernel's start it is the one register the strict trilinear instance spilled to scratch
    (void)lane_;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    Ext e = e_;
    Bnd bn = bn_;
    // side-effect free reductions would be hoisted out of the (rare) miss branch into every tile
    asm volatile("" : "+v"(e.ymin), "+v"(e.ymax), "+v"(e.cbmin), "+v"(e.cbmax), "+v"(e.crmin), "+v"(e.crmax));
    asm volatile("" : "+v"(bn.rmin), "+v"(bn.rmax), "+v"(bn.gmin), "+v"(bn.gmax), "+v"(bn.bmin), "+v"(bn.bmax));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        e.ymin = pk_min(e.ymin, shx(e.ymin, m)); e.ymax = pk_max(e.ymax, shx(e.ymax, m));
        e.cbmin = pk_min(e.cbmin, shx(e.cbmin, m)); e.cbmax = pk_max(e.cbmax, shx(e.cbmax, m));
        e.crmin = pk_min(e.crmin, shx(e.crmin, m)); e.crmax = pk_max(e.crmax, shx(e.crmax, m));
    }
    const int y0 = uni(lo_of<WIN>(e.ymin)), y1 = uni(hi_of<WIN>(e.ymax));
    const int cb0 = uni(lo_of<WIN>(e.cbmin)), cb1 = uni(hi_of<WIN>(e.cbmax));
    const int cr0 = uni(lo_of<WIN>(e.crmin)), cr1 = uni(hi_of<WIN>(e.crmax));
    const int mr = TG.max_raw;
    if (y1 > mr || cb1 > mr || cr1 > mr) return false;
    const int cap = TG.win_nodes;
    int ng = 0, nb = 0, sr = 1, nr = 0, r0 = 0, g0 = 0, b0 = 0;
    int by0 = 1, by1 = 0, bcb0 = 1, bcb1 = 0, bcr0 = 1, bcr1 = 0;             // raw box of the first-level test; empty so far
    bool placed = false;
    const float cell_y = L.maxf / (L.lut_max * K.ky * (PRE ? K.py : 1.0f));   // raw codes per lattice cell along luma
    const int unit = (mr + 1) >> 8;                                            // one 8-bit code in raw codes
    // (1) A window that covers the conservative image of the tile's raw box plus a margin: the tiles that follow then pass
    //     the cheap first-level test.  Margins from generous to none; the first whose cells fit the capacity wins.
    // (Measured: 95 % of the tiles that leave a raw box leave it through a chroma bound -- vertical colour edges that a
    // 256-px-wide tile straddles for a whole chunk.  Trading luma room for 4-6 codes of chroma margin did not pay: -0.5 %.)
#pragma unroll 1
    for (int t = 0; t < 4 && !placed; t++) {
        const int mc = t == 0 ? 2 * unit : (t == 1 ? unit : (t == 2 ? (unit + 3) / 4 : 0));
        const int my = t == 3 ? 0 : (int)((t == 0 ? 2.0f : (t == 1 ? 1.0f : 0.3f)) * cell_y);
        const int ty0 = max(y0 - my, 0), ty1 = min(y1 + my, mr);
        const int tcb0 = max(cb0 - mc, 0), tcb1 = min(cb1 + mc, mr), tcr0 = max(cr0 - mc, 0), tcr1 = min(cr1 + mc, mr);
        const Cells c = map_box<INTERP, PRE, V>(L, K, TG, (float)ty0, (float)ty1, (float)tcb0, (float)tcb1, (float)tcr0, (float)tcr1);
        // corners: r..r+1, (g-r)-1..(g-r)+1, (b-r)-1..(b-r)+1
        const int need_r = uni(c.r1 - c.r0 + 2), need_g = uni(c.g1 - c.g0 + 3), need_b = uni(c.b1 - c.b0 + 3);
        if (need_r < 2 || need_g < 3 || need_b < 3 || need_g > 128 || need_b > 128) continue;
        // odd row and plane strides: the nodes of neighbouring cells (the ones a wave reads together) differ by +-1, +-nb,
        // +-(sr - nb - 1) and small sums of those -- kept away from multiples of 16 nodes, i.e. from the same LDS banks
        const int nb_ = need_b | 1;
        const int sr_ = win_plane_stride<kLN>(need_g * nb_, nb_), nr_ = cap / sr_;
        if (nr_ < need_r) continue;
        ng = need_g; nb = nb_; sr = sr_; nr = nr_;
        r0 = uni(c.r0) - ((nr - need_r) >> 1); g0 = uni(c.g0) - 1; b0 = uni(c.b0) - 1;
        by0 = ty0; by1 = ty1; bcb0 = tcb0; bcb1 = tcb1; bcr0 = tcr0; bcr1 = tcr1;
        placed = true;
        // the luma-like axis got whatever capacity was left: widen the luma range of the box to match, if the map agrees
        const int room = (nr - need_r) >> 1;
        if (room > 0) {
            const int my2 = my + (int)(0.8f * (float)room * cell_y);
            const int uy0 = max(y0 - my2, 0), uy1 = min(y1 + my2, mr);
            const Cells c2 = map_box<INTERP, PRE, V>(L, K, TG, (float)uy0, (float)uy1, (float)tcb0, (float)tcb1, (float)tcr0, (float)tcr1);
            const bool fits = uni((int)(c2.r0 >= r0 && c2.r1 <= r0 + nr - 2 && c2.g0 >= g0 + 1 && c2.g1 <= g0 + ng - 2 &&
                                        c2.b0 >= b0 + 1 && c2.b1 <= b0 + nb - 2)) != 0;
            if (fits) { by0 = uy0; by1 = uy1; }
        }
    }
    if (!placed) {
        // (2) The raw box is too loose a description of this tile (an edge between two colours: Cb and Cr move together,
        //     the box spans every combination).  Stage around the tile's EXACT cells; the first-level box stays empty and
        //     the tiles that follow are vouched for by the second-level test.
        const int rmin = uni((int)wave_min(bn.rmin)), rmax = uni((int)wave_max(bn.rmax));
        const int gmin = uni((int)wave_min(bn.gmin)), gmax = uni((int)wave_max(bn.gmax));
        const int bmin = uni((int)wave_min(bn.bmin)), bmax = uni((int)wave_max(bn.bmax));
        const int need_r = rmax - rmin + 2, need_g = gmax - gmin + 3, need_b = bmax - bmin + 3;
        int spare = 1;
        for (;;) {                                   // one spare cell on each side of the chroma-like axes if it fits
            ng = need_g + 2 * spare; nb = (need_b + 2 * spare) | 1;
            sr = win_plane_stride<kLN>(ng * nb, nb);
            nr = cap / sr;
            if (nr >= need_r + 2 * spare || spare == 0) break;
            spare--;
        }
        if (need_r > nr || ng > 128 || nb > 128) return false;
        r0 = rmin - ((nr - need_r) >> 1);
        g0 = gmin - 1 - spare;
        b0 = bmin - 1 - spare;
    }
    // cells whose 8 corners are staged
    const int wr0 = r0, wr1 = r0 + nr - 2, wg0 = g0 + 1, wg1 = g0 + ng - 2, wb0 = b0 + 1, wb1 = b0 + nb - 2;

    const int n1 = L.n1, nmax = L.n1 - 1;
    const int plane = ng * nb, total = nr * plane;
    // i -> (ir, ig, ib) by float reciprocals: floor(i / d) == (int)((i + 0.5) * (1 / d)) while the quotient stays below 2^7 and
    // d below 2^12 (error of the product < 2^-16, distance of (i + 0.5) / d from an integer >= 2^-13; d <= window capacity); every product fits 24 bits
    const float rcp_plane = 1.0f / (float)plane, rcp_nb = 1.0f / (float)nb;
    // Batches of kB nodes per lane: all kB global reads are issued before the first LDS write, so a restage costs about
    // one L2 round trip per batch instead of one per node (the wave is stalled meanwhile; 4 waves per SIMD cannot hide it).
    constexpr int kB = 6;          // (9 and 18 for the 8-byte nodes of the fast windows were measured: no difference)
    for (int base = 0; base < total; base += 64 * kB) {
        int dst[kB];
        typename std::conditional<N::fast, uint2, float4>::type val[kB];
        uint2 val1[N::rec ? kB : 1];                                          // records: the r + 1 neighbour
#pragma unroll
        for (int k = 0; k < kB; k++) {
            const int i = min(base + k * 64 + lane, total - 1);              // the last batch re-reads the final node: harmless
            const int ir = (int)(((float)i + 0.5f) * rcp_plane), rem = i - __mul24(ir, plane);
            const int ig = (int)(((float)rem + 0.5f) * rcp_nb), ib = rem - __mul24(ig, nb);
            int r = r0 + ir, g = r + g0 + ig, b = (LUTR_T2_WIN_BG ? g : r) + b0 + ib;
            // nodes outside the cube are never referenced by a valid pixel: clamp to stay inside the lattice
            r = min(max(r, 0), nmax); g = min(max(g, 0), nmax); b = min(max(b, 0), nmax);
            const int src = __mul24(__mul24(r, n1) + g, n1) + b;
            dst[k] = slice_off + kLN * (__mul24(ir, sr) + __mul24(ig, nb) + ib);
            if constexpr (N::fast) val[k] = L.lat16[src];
            else val[k] = L.lat[src];
            if constexpr (N::rec) val1[k] = L.lat16[__mul24(__mul24(min(r + 1, nmax), n1) + g, n1) + b];
        }
#pragma unroll
        for (int k = 0; k < kB; k++) {
            char *q = smem + dst[k];
            if constexpr (N::rec) { const u3 rc = make_rec(val[k], val1[k]); ((uint32_t *)q)[0] = rc.x; ((uint32_t *)q)[1] = rc.y; ((uint32_t *)q)[2] = rc.z; }
            else if constexpr (N::fast) *(uint2 *)q = val[k];
            else if constexpr (kLN == 16) *(float4 *)q = val[k];
            else { ((float *)q)[0] = val[k].x; ((float *)q)[1] = val[k].y; ((float *)q)[2] = val[k].z; }
        }
    }
    if (lane == 0) {
        *(float4 *)(smem + scratch_off) = make_float4((float)wr0, (float)wg0, (float)wb0, (float)wr1);
        *(float2 *)(smem + scratch_off + 16) = make_float2((float)wg1, (float)wb1);
    }
    // node index = (pr-r0)*sr + (pg-pr-g0)*nb + (pb-pg-b0)     [(pb-pr-b0) without LUTR_T2_WIN_BG]
    if (LUTR_T2_WIN_BG) { W.o_r = kLN * (sr - nb); W.o_g = kLN * (nb - 1); }
    else { W.o_r = kLN * (sr - nb - 1); W.o_g = kLN * nb; }
    W.fr = (float)W.o_r; W.fg = (float)W.o_g; W.fb = (float)kLN;
    W.fc = (float)(lds_base() + slice_off - kLN * (r0 * sr + g0 * nb + b0));
    if (lane == 0) {
        Box bx;
        if (by1 >= by0) {
            bx.ylo = pack_lo<WIN>(by0); bx.yhi = pack_hi<WIN>(by1);
            bx.cblo = pack_lo<WIN>(bcb0); bx.cbhi = pack_hi<WIN>(bcb1);
            bx.crlo = pack_lo<WIN>(bcr0); bx.crhi = pack_hi<WIN>(bcr1);
        } else {
            bx.ylo = bx.cblo = bx.crlo = 0xffffffffu; bx.yhi = bx.cbhi = bx.crhi = 0u;      // empty: lo > hi
        }
        box_store(scratch_off, bx);
    }
    return true;
}

// ---------------------------------------------------------------- one pixel
struct PxC {
    int a;                  // byte address (LDS) or byte offset (global) of the c000 tap
    int oa, oz;             // tetrahedral: byte offsets of the 2nd and 3rd taps
    f2v w01, w23;           // tetrahedral weights {w0, w1}, {w2, w3} as register pairs; trilinear keeps {d.r, d.g}, {d.b, -}
};
struct Rgb3 { float r, g, b; };

template <bool LDS, int INTERP, int V>
DEV PxC px_finish(const LutConsts &L, const Win &W, const Crd &cr, const Crd &cg, const Crd &cb)
{
    using N = Node<INTERP, V>;
    constexpr int nb_ = LDS ? N::lds : N::glb;
    PxC c;
    if constexpr (LDS) {
        // exact in fp32: every term is an integer well below 2^24; the box test guarantees the address is inside the slice
        c.a = (int)fma_(cr.p, W.fr, fma_(cg.p, W.fg, fma_(cb.p, W.fb, W.fc)));
    } else {
        c.a = (((int)cr.p * L.n1 + (int)cg.p) * L.n1 + (int)cb.p) * nb_;
    }
    c.oa = c.oz = 0;
    c.w01 = f2v{0.0f, 0.0f}; c.w23 = f2v{0.0f, 0.0f};
    if constexpr (INTERP == LUTR_INTERP_TRILINEAR) {
        c.w01.x = cr.d; c.w01.y = cg.d; c.w23.x = cb.d;
    } else if constexpr (INTERP == LUTR_INTERP_TETRAHEDRAL || INTERP == T2_TET16) {
        // FFmpeg's six branches all evaluate (1-x) c000 + (x-y) cA + (y-z) cB + z c111 with (x,y,z) the fractions sorted
        // descending; ties only ever choose between taps whose weight is exactly 0 (finite lattice), so the sorted form is
        // bit-identical.
        const float dr = cr.d, dg = cg.d, db = cb.d;
        const float x = fmaxf(fmaxf(dr, dg), db), y = med3(dr, dg, db), z = fminf(fminf(dr, dg), db);
        const bool rg = dr > dg, gb = dg > db, rb = dr > db;
        const int o_r = LDS ? W.o_r : nb_ * L.n1 * L.n1, o_g = LDS ? W.o_g : nb_ * L.n1, o_b = nb_;
        const int z_r = o_g + o_b, z_g = o_r + o_b, z_b = o_r + o_g;         // o111 minus the step of the smallest fraction
        c.oa = (rg && rb) ? o_r : (gb ? o_g : o_b);
        c.oz = (gb && rb) ? z_b : (rg ? z_g : z_r);
        c.w01.x = 1.0f - x; c.w01.y = x - y; c.w23.x = y - z; c.w23.y = z;
    }
    return c;
}

// strict taps: one node as floats
template <bool LDS, int NB>
DEV f4 tap(const LutConsts &L, int a)
{
    if constexpr (LDS && NB == 16) {
        return *(const __attribute__((address_space(3))) f4 *)(uintptr_t)(unsigned)a;
    } else if constexpr (LDS) {
        const __attribute__((address_space(3))) float *p = (const __attribute__((address_space(3))) float *)(uintptr_t)(unsigned)a;
        f4 v; v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = 0.0f;
        return v;
    } else {
        return *(const f4 *)((const char *)L.lat + a);
    }
}
// fast taps: one node as two words {r | g << 16, b} of fp16
template <bool LDS>
DEV u2 tap16(const LutConsts &L, int a)
{
    if constexpr (LDS) return *(const __attribute__((address_space(3))) u2 *)(uintptr_t)(unsigned)a;
    else return *(const u2 *)((const char *)L.lat16 + a);
}

DEV float tlerp(float v0, float v1, float f) { return v0 + (v1 - v0) * f; }

// The taps of one pixel, loaded in one go so that several pixels' reads are in flight together (tile_body phases).
template <bool LDS, int INTERP, int V> struct Taps {
    static constexpr bool rec = LDS && Node<INTERP, V>::rec;          // four records instead of eight nodes
    static constexpr int n = rec ? 4 : (INTERP == LUTR_INTERP_TRILINEAR ? 8 : (INTERP == LUTR_INTERP_NEAREST ? 1 : 4));
    typename std::conditional<rec, u3, typename std::conditional<V == V_FAST, u2, f4>::type>::type t[n];
};
DEV u3 tap_rec(int a)
{
    const __attribute__((address_space(3))) uint32_t *p = (const __attribute__((address_space(3))) uint32_t *)(uintptr_t)(unsigned)a;
    return u3{p[0], p[1], p[2]};
}

template <bool LDS, int INTERP, int V>
DEV Taps<LDS, INTERP, V> px_taps(const LutConsts &L, const Win &W, const PxC &c)
{
    using N = Node<INTERP, V>;
    constexpr int nb_ = LDS ? N::lds : N::glb;
    const int o_r = LDS ? W.o_r : nb_ * L.n1 * L.n1, o_g = LDS ? W.o_g : nb_ * L.n1;
    const int a = c.a;
    Taps<LDS, INTERP, V> T;
    if constexpr (Taps<LDS, INTERP, V>::rec) {
        const int ag = a + o_g;
        T.t[0] = tap_rec(a); T.t[1] = tap_rec(a + nb_); T.t[2] = tap_rec(ag); T.t[3] = tap_rec(ag + nb_);      // c00x c01x ... : (g, b) corners
        return T;
    } else {
    auto ld = [&](int addr) {
        if constexpr (N::fast) return tap16<LDS>(L, addr);
        else return tap<LDS, nb_>(L, addr);
    };
    if constexpr (INTERP == LUTR_INTERP_NEAREST) {
        T.t[0] = ld(a);
    } else if constexpr (INTERP == LUTR_INTERP_TRILINEAR) {
        const int ag = a + o_g, ar = a + o_r, arg = ar + o_g;
        T.t[0] = ld(a); T.t[1] = ld(a + nb_); T.t[2] = ld(ag); T.t[3] = ld(ag + nb_);       // c000 c001 c010 c011
        T.t[4] = ld(ar); T.t[5] = ld(ar + nb_); T.t[6] = ld(arg); T.t[7] = ld(arg + nb_);   // c100 c101 c110 c111
    } else {
        T.t[0] = ld(a); T.t[1] = ld(a + c.oa); T.t[2] = ld(a + c.oz); T.t[3] = ld(a + o_r + o_g + nb_);
    }
    return T;
    }
}

// lattice value as the integer code it quantises to, held as float
template <bool LDS, int INTERP, int V>
DEV Rgb3 px_blend(const LutConsts &L, const PxC &c, const Taps<LDS, INTERP, V> &T)
{
    Rgb3 v;
    if constexpr (Taps<LDS, INTERP, V>::rec) {
        // records {r | g << 16, b | Dr << 16, Dg | Db << 16} of the (g, b) corners 00, 01 (b + 1), 10 (g + 1), 11
        const float dr = c.w01.x, dg = c.w01.y, db = c.w23.x;
        {
            const float c00 = reclerp<1, 0>(T.t[0].y, dr, T.t[0].x), c01 = reclerp<1, 0>(T.t[1].y, dr, T.t[1].x);
            const float c10 = reclerp<1, 0>(T.t[2].y, dr, T.t[2].x), c11 = reclerp<1, 0>(T.t[3].y, dr, T.t[3].x);
            const float c0 = fma_(c10 - c00, dg, c00), c1 = fma_(c11 - c01, dg, c01);
            v.r = fma_(c1 - c0, db, c0);
        }
        {
            const float c00 = reclerp<0, 1>(T.t[0].z, dr, T.t[0].x), c01 = reclerp<0, 1>(T.t[1].z, dr, T.t[1].x);
            const float c10 = reclerp<0, 1>(T.t[2].z, dr, T.t[2].x), c11 = reclerp<0, 1>(T.t[3].z, dr, T.t[3].x);
            const float c0 = fma_(c10 - c00, dg, c00), c1 = fma_(c11 - c01, dg, c01);
            v.g = fma_(c1 - c0, db, c0);
        }
        {
            const float c00 = reclerp<1, 0>(T.t[0].z, dr, T.t[0].y), c01 = reclerp<1, 0>(T.t[1].z, dr, T.t[1].y);
            const float c10 = reclerp<1, 0>(T.t[2].z, dr, T.t[2].y), c11 = reclerp<1, 0>(T.t[3].z, dr, T.t[3].y);
            const float c0 = fma_(c10 - c00, dg, c00), c1 = fma_(c11 - c01, dg, c01);
            v.b = fma_(c1 - c0, db, c0);
        }
        return v;
    } else
    if constexpr (V == V_FAST) {
        // nodes are fp16 of (lattice * M): the blend IS the code before truncation.  v_fma_mix_f32 takes the fp16 tap as it
        // is (exact conversion inside the instruction), fp32 weight, fp32 accumulator: one rounding per step.
        if constexpr (INTERP == LUTR_INTERP_NEAREST) {
            v.r = mix0_lo(1.0f, T.t[0].x); v.g = mix0_hi(1.0f, T.t[0].x); v.b = mix0_lo(1.0f, T.t[0].y);
        } else if constexpr (INTERP == LUTR_INTERP_TRILINEAR) {
            const float dr = c.w01.x, dg = c.w01.y, db = c.w23.x;
            // (with records in LDS this is the gather body: its differences are rounded to fp16 as the staged ones are)
#define TRI16(SUB_, LERP_, W_, out) \
            { \
                auto SUB = [](uint32_t a_, uint32_t b_) { \
                    const float d_ = SUB_(a_, b_); \
                    if constexpr (Node<INTERP, V>::rec) return (float)(_Float16)d_; else return d_; \
                }; \
                auto LERP = [](float t_, float f_, uint32_t v0_) { return LERP_(t_, f_, v0_); }; \
                const float c00 = LERP(SUB(T.t[4].W_, T.t[0].W_), dr, T.t[0].W_), c10 = LERP(SUB(T.t[6].W_, T.t[2].W_), dr, T.t[2].W_); \
                const float c01 = LERP(SUB(T.t[5].W_, T.t[1].W_), dr, T.t[1].W_), c11 = LERP(SUB(T.t[7].W_, T.t[3].W_), dr, T.t[3].W_); \
                const float c0 = fma_(c10 - c00, dg, c00), c1 = fma_(c11 - c01, dg, c01); \
                out = fma_(c1 - c0, db, c0); \
            }
            TRI16(hsub_lo, hlerp_lo, x, v.r) TRI16(hsub_hi, hlerp_hi, x, v.g) TRI16(hsub_lo, hlerp_lo, y, v.b)
#undef TRI16
        } else {
            const float w0 = c.w01.x, w1 = c.w01.y, w2 = c.w23.x, w3 = c.w23.y;
            v.r = mix_lo(w3, T.t[3].x, mix_lo(w2, T.t[2].x, mix_lo(w1, T.t[1].x, mix0_lo(w0, T.t[0].x))));
            v.g = mix_hi(w3, T.t[3].x, mix_hi(w2, T.t[2].x, mix_hi(w1, T.t[1].x, mix0_hi(w0, T.t[0].x))));
            v.b = mix_lo(w3, T.t[3].y, mix_lo(w2, T.t[2].y, mix_lo(w1, T.t[1].y, mix0_lo(w0, T.t[0].y))));
        }
        return v;
    } else {
        if constexpr (INTERP == LUTR_INTERP_NEAREST) {
            v.r = T.t[0].x; v.g = T.t[0].y; v.b = T.t[0].z;
        } else if constexpr (INTERP == LUTR_INTERP_TRILINEAR) {
            const float dr = c.w01.x, dg = c.w01.y, db = c.w23.x;
#if LUTR_T2_PK & 2
            // R and G ride in one register pair through FFmpeg's seven lerps (v0 + (v1 - v0) * f, each op rounded on its own, as
            // the scalar code below): 21 packed + 21 scalar instructions instead of 63
#define XY(k) f2v{T.t[k].x, T.t[k].y}
            auto lerp2 = [](f2v v0, f2v v1, f2v w, auto hi) {
                const f2v d = v1 - v0;
                return v0 + pk_mul_w<decltype(hi)::value>(d, w);
            };
            using LO = std::integral_constant<int, 0>; using HI_ = std::integral_constant<int, 1>;
            const f2v c00 = lerp2(XY(0), XY(4), c.w01, LO{}), c10 = lerp2(XY(2), XY(6), c.w01, LO{});
            const f2v c01 = lerp2(XY(1), XY(5), c.w01, LO{}), c11 = lerp2(XY(3), XY(7), c.w01, LO{});
            const f2v c0 = lerp2(c00, c10, c.w01, HI_{}), c1 = lerp2(c01, c11, c.w01, HI_{});
            const f2v rg = lerp2(c0, c1, c.w23, LO{});
#undef XY
            v.r = rg.x; v.g = rg.y;
            {
                const float c00 = tlerp(T.t[0].z, T.t[4].z, dr), c10 = tlerp(T.t[2].z, T.t[6].z, dr);
                const float c01 = tlerp(T.t[1].z, T.t[5].z, dr), c11 = tlerp(T.t[3].z, T.t[7].z, dr);
                const float c0 = tlerp(c00, c10, dg), c1 = tlerp(c01, c11, dg);
                v.b = tlerp(c0, c1, db);
            }
#else
#define TRI(ch, out) \
            { \
                const float c00 = tlerp(T.t[0].ch, T.t[4].ch, dr), c10 = tlerp(T.t[2].ch, T.t[6].ch, dr); \
                const float c01 = tlerp(T.t[1].ch, T.t[5].ch, dr), c11 = tlerp(T.t[3].ch, T.t[7].ch, dr); \
                const float c0 = tlerp(c00, c10, dg), c1 = tlerp(c01, c11, dg); \
                out = tlerp(c0, c1, db); \
            }
            TRI(x, v.r) TRI(y, v.g) TRI(z, v.b)
#undef TRI
#endif
        } else {
            const float w0 = c.w01.x, w1 = c.w01.y, w2 = c.w23.x, w3 = c.w23.y;
#if LUTR_T2_PK & 2
            // FFmpeg's w0 c000 + w1 cA + w2 cB + w3 c111, left to right, products and sums rounded one by one: R and G as a pair
            f2v rg = pk_mul_w<0>(f2v{T.t[0].x, T.t[0].y}, c.w01);
            rg = rg + pk_mul_w<1>(f2v{T.t[1].x, T.t[1].y}, c.w01);
            rg = rg + pk_mul_w<0>(f2v{T.t[2].x, T.t[2].y}, c.w23);
            rg = rg + pk_mul_w<1>(f2v{T.t[3].x, T.t[3].y}, c.w23);
            v.r = rg.x; v.g = rg.y;
#else
            v.r = w0 * T.t[0].x + w1 * T.t[1].x + w2 * T.t[2].x + w3 * T.t[3].x;
            v.g = w0 * T.t[0].y + w1 * T.t[1].y + w2 * T.t[2].y + w3 * T.t[3].y;
#endif
            v.b = w0 * T.t[0].z + w1 * T.t[1].z + w2 * T.t[2].z + w3 * T.t[3].z;
        }
#if LUTR_T2_PK & 4
        if constexpr (INTERP != LUTR_INTERP_NEAREST) {
            const f2v rgm = pk_mul_lo(f2v{v.r, v.g}, L.maxf);
            v.r = rgm.x; v.g = rgm.y; v.b *= L.maxf;
            return v;
        }
#endif
        v.r *= L.maxf; v.g *= L.maxf; v.b *= L.maxf;
        return v;
    }
}

// (int)(v * M) clipped to [0, M]; V >= V_UNIT: every lattice node lies in [0, 1], the truncation alone lands in [0, M]
// (all weights and nodes >= 0, rounding of products and sums is monotone): the clip is dead code.
template <int INTERP, int V>
DEV Rgb3 px_quant(const LutConsts &L, const Rgb3 &v)
{
    Rgb3 o;
    // (Measured for the fast tetrahedral kernel: rounding by two full-rate adds of 1.5 * 2^23 on a chain started at -0.5 + 2^-12
    // instead of the quarter-rate v_trunc_f32 -- three more instructions per pixel, 3 % slower: instruction issue binds first.)
    if constexpr (V >= V_UNIT) { o.r = truncf(v.r); o.g = truncf(v.g); o.b = truncf(v.b); }
    else { o.r = med3(truncf(v.r), 0.0f, L.maxf); o.g = med3(truncf(v.g), 0.0f, L.maxf); o.b = med3(truncf(v.b), 0.0f, L.maxf); }
    return o;
}

template <bool DEAD> DEV float ofloor(float v, float hi) { if constexpr (DEAD) return v; else return fminf(v, hi); }

// ---------------------------------------------------------------- tile body
// Coordinates of the four pixels of a group (phase A): twelve table reads (or their computed twins) issued back to back.
struct GroupCrd { Crd r[4], g[4], b[4]; };

template <bool LDS, int WIN, int WOUT, int CSX, int CSY, int INTERP, int PRE, int V>
DEV void group_coords(const LutConsts &L, const YuvConsts &K, const Geom &TG, const TileIn<WIN, WOUT, CSX, CSY> &in, int g, GroupCrd &q)
{
    using T = Tile<WIN, WOUT, CSX, CSY>;
    constexpr int GW = 4 / T::BH, NCG = (GW >> CSX) > 0 ? (GW >> CSX) : 1;
    float rv[NCG], gv[NCG], bu[NCG];
    f2v rgv[NCG];                   // {rv, gv} as a register pair: one v_pk_add_f32 adds luma to both (LUTR_T2_PK)
#pragma unroll
    for (int c = 0; c < NCG; c++) {
        const int j = g * NCG + c;
        float cbv = wsample<WIN>(in.cb, j), crv = wsample<WIN>(in.cr, j);
        if constexpr (PRE) {
            cbv = cfloor(fma_(K.pc, cbv, K.pcb), K.pre_max);
            crv = cfloor(fma_(K.pc, crv, K.pcb), K.pre_max);
        }
        const float cbd = cbv - K.coff, crd = crv - K.coff;
        rv[c] = K.krv * crd; gv[c] = fma_(K.kgu, cbd, K.kgv * crd); bu[c] = K.kbu * cbd;
        rgv[c].x = rv[c]; rgv[c].y = gv[c];
    }
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int dy = p / GW, i = g * GW + p % GW, c = (p % GW) >> CSX;
        float yv = wsample<WIN>(in.y[dy], i);
        if constexpr (PRE) yv = cfloor(fma_(K.py, yv, K.pyb), K.pre_max);
        const float yy = fma_(K.ky, yv, K.yb);
        if constexpr (V >= V_TAB) {
            // clip(floor(v), 0, M): v_cvt_u32_f32 floors and saturates negatives to 0; the table is padded past M
            // K is the kernel's KB here: sums are 8 x the code, the masked conversion is the table's byte offset
#if LUTR_T2_PK & 1
            const f2v rg = pk_add_lo(rgv[c], yy);  // {rv + yy, gv + yy}: the same two roundings as the scalar adds
            unsigned ri = (unsigned)rg.x & ~7u, gi = (unsigned)rg.y & ~7u, bi = (unsigned)(yy + bu[c]) & ~7u;
#else
            unsigned ri = (unsigned)(yy + rv[c]) & ~7u, gi = (unsigned)(yy + gv[c]) & ~7u, bi = (unsigned)(yy + bu[c]) & ~7u;
#endif
            if constexpr (!LDS) {           // the gather body also serves tiles with raw codes nobody vouched for
                const unsigned top = (unsigned)(TG.tab_entries - 1) * 8u;
                ri = min(ri, top); gi = min(gi, top); bi = min(bi, top);
            }
            q.r[p] = crd_table8(ri); q.g[p] = crd_table8(gi); q.b[p] = crd_table8(bi);
        } else {
            const float rq = cfloor(yy + rv[c], K.max_l), gq = cfloor(yy + gv[c], K.max_l), bq = cfloor(yy + bu[c], K.max_l);
            q.r[p] = crd_compute<INTERP>(L, rq, L.sc[0]); q.g[p] = crd_compute<INTERP>(L, gq, L.sc[1]);
            q.b[p] = crd_compute<INTERP>(L, bq, L.sc[2]);
        }
    }
}

#ifndef LUTR_T2_PIPE
#define LUTR_T2_PIPE 0            // 1: software pipeline across pixel groups (group g+1's coordinate reads before group g's blend): +4 % for the strict kernels while they had registers for it; with the tube state live it spills (-2.7 %)
#endif
#ifndef LUTR_T2_FENCE_ALL
#define LUTR_T2_FENCE_ALL 0
#endif
#ifndef LUTR_T2_PIPE_FAST
#define LUTR_T2_PIPE_FAST 0
#endif
#ifndef LUTR_T2_KARG
#define LUTR_T2_KARG 1
#endif
#ifndef LUTR_T2_TUBE_SAMPLES
#define LUTR_T2_TUBE_SAMPLES 1
#endif
#ifndef LUTR_T2_TB_FAST
#define LUTR_T2_TB_FAST 4
#endif

// One tile.  Every LDS result is requested well before it is used, inside the wave: with four waves per SIMD (the
// register budget) the hardware alone cannot hide an LDS round trip per pixel.  Order per group g:
//   A(g+1)  coordinate reads of the next group      (12 ds_read_b64, consumed one group later)
//   B(g)    tap addresses + weights, then tap reads  (TB pixels' taps in flight together)
//   C(g)    blends, truncation, RGB -> YUV, packing
// sched_barrier keeps the machine scheduler from sinking the reads back next to their uses; the zero-instruction
// fences at the end of a group keep instruction selection from hoisting every group to the top (> 1000 spilled registers).
template <bool LDS, int WIN, int WOUT, int CSX, int CSY, int INTERP, int PRE, int V>
DEV void tile_body(const LutConsts &L, const YuvConsts &K_, const Win &W_, const Geom &TG, TileIn<WIN, WOUT, CSX, CSY> &in,
                   TileOut<WIN, WOUT, CSX, CSY> &out)
{
    // the four address factors go to VGPRs for the length of the tile (three fma per pixel at full rate instead of the
    // SGPR-operand rate); kernel-wide they would be live across the restage code, which has no registers to spare
    Win W = W_;
    if constexpr (LDS && LUTR_T2_PIN >= 3 && V == V_FAST && INTERP != LUTR_INTERP_TRILINEAR) {
        W.fr = in_vgpr(W_.fr); W.fg = in_vgpr(W_.fg); W.fb = in_vgpr(W_.fb); W.fc = in_vgpr(W_.fc);
    }
    YuvConsts K = K_;
    if constexpr (LDS && LUTR_T2_PIN >= 4 && V == V_FAST && INTERP != LUTR_INTERP_TRILINEAR) {
        // the chroma terms and the RGB -> CbCr rows: 12 moves per tile buy 3 full-rate ops per pixel
        K.krv = in_vgpr(K_.krv); K.kgu = in_vgpr(K_.kgu); K.kgv = in_vgpr(K_.kgv); K.kbu = in_vgpr(K_.kbu); K.coff = in_vgpr(K_.coff);
        K.cbr = in_vgpr(K_.cbr); K.cbg = in_vgpr(K_.cbg); K.cbb = in_vgpr(K_.cbb); K.cob = in_vgpr(K_.cob);
        K.crr = in_vgpr(K_.crr); K.crg = in_vgpr(K_.crg); K.crb = in_vgpr(K_.crb);
    }
    using T = Tile<WIN, WOUT, CSX, CSY>;
    // a group = 4 pixels: all BH rows of GW columns; it owns NCG chroma samples
    constexpr int GW = 4 / T::BH, NG = T::PXT / GW, NCG = (GW >> CSX) > 0 ? (GW >> CSX) : 1;
    constexpr bool kDead = V >= V_UNIT;          // launcher checked the RGB->YUV maxima too (out_clip_dead)
    // measured (UHD yuv420p10le tetrahedral, Gpx/s): strict 503 with the pipeline, 484 without; fast 458 with (the extra live
    // coordinates push its 4-pixel tap batches into spills), 550-565 without
    constexpr bool kPipe = LUTR_T2_PIPE && LDS && V >= V_TAB && (V != V_FAST || LUTR_T2_PIPE_FAST);
    constexpr int TB = INTERP == LUTR_INTERP_TRILINEAR ? (V == V_FAST ? 2 : 1)
                                                       : (INTERP == LUTR_INTERP_NEAREST ? 4 : (V == V_FAST ? LUTR_T2_TB_FAST : 2));
    GroupCrd cq[2];
    if constexpr (kPipe) group_coords<LDS, WIN, WOUT, CSX, CSY, INTERP, PRE, V>(L, K, TG, in, 0, cq[0]);
#pragma unroll
    for (int g = 0; g < NG; g++) {
        GroupCrd &q = cq[kPipe ? (g & 1) : 0];
        if constexpr (kPipe) {
            if (g + 1 < NG) group_coords<LDS, WIN, WOUT, CSX, CSY, INTERP, PRE, V>(L, K, TG, in, g + 1, cq[(g + 1) & 1]);
        } else {
            group_coords<LDS, WIN, WOUT, CSX, CSY, INTERP, PRE, V>(L, K, TG, in, g, q);
        }
#if LUTR_T2_PHASES
        __builtin_amdgcn_sched_barrier(0);
#endif
        Rgb3 o[4];
#pragma unroll
        for (int qb = 0; qb < 4; qb += TB) {
            PxC pc[TB];
            Taps<LDS, INTERP, V> tp[TB];
#pragma unroll
            for (int t = 0; t < TB; t++) {
                pc[t] = px_finish<LDS, INTERP, V>(L, W, q.r[qb + t], q.g[qb + t], q.b[qb + t]);
                tp[t] = px_taps<LDS, INTERP, V>(L, W, pc[t]);
            }
#if LUTR_T2_PHASES
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int t = 0; t < TB; t++) o[qb + t] = px_quant<INTERP, V>(L, px_blend<LDS, INTERP, V>(L, pc[t], tp[t]));
        }
        float rs[NCG], gs[NCG], bs[NCG];
#if LUTR_T2_PK & 8
        f2v rgs[NCG];               // chroma-block sums of R and G as a pair: one v_pk_add_f32 per pixel, same order of additions
#endif
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int dy = p / GW, i = g * GW + p % GW, c = (p % GW) >> CSX;
#if LUTR_T2_PK & 8
            const f2v orgp = {o[p].r, o[p].g};
            if (dy == 0 && ((p % GW) & (T::BW - 1)) == 0) { rgs[c] = orgp; bs[c] = o[p].b; }
            else { rgs[c] = rgs[c] + orgp; bs[c] += o[p].b; }
            rs[c] = rgs[c].x; gs[c] = rgs[c].y;
#else
            if (dy == 0 && ((p % GW) & (T::BW - 1)) == 0) { rs[c] = o[p].r; gs[c] = o[p].g; bs[c] = o[p].b; }
            else { rs[c] += o[p].r; gs[c] += o[p].g; bs[c] += o[p].b; }
#endif
            wput<WOUT>(out.y[dy], i, ofloor<kDead>(fma_(K.cyr, o[p].r, fma_(K.cyg, o[p].g, fma_(K.cyb, o[p].b, K.yob))), K.max_o));
        }
#pragma unroll
        for (int c = 0; c < NCG; c++) {
            const int j = g * NCG + c;
            wput<WOUT>(out.cb, j, ofloor<kDead>(fma_(K.cbr, rs[c], fma_(K.cbg, gs[c], fma_(K.cbb, bs[c], K.cob))), K.max_o));
            wput<WOUT>(out.cr, j, ofloor<kDead>(fma_(K.crr, rs[c], fma_(K.crg, gs[c], fma_(K.crb, bs[c], K.cob))), K.max_o));
        }
#if LUTR_T2_FENCE_ALL
        fence_words<T::YWI * T::BH>(&in.y[0][0]);
        fence_words<T::CWI>(in.cb);
        fence_words<T::CWI>(in.cr);
        fence_words<T::YWO * T::BH>(&out.y[0][0]);
        fence_words<T::CWO>(out.cb);
        fence_words<T::CWO>(out.cr);
#else
        // Only what is live anyway: the input words later groups still read (they cannot start before this point) and the
        // output words this group wrote (it cannot finish after it).  Fencing consumed input words or unborn output words
        // would keep 2 x 12 registers allocated through the whole tile.
        if (g + 1 < NG) {
            constexpr int ISH = WIN ? 1 : 2;
            const int y0 = ((g + 1) * GW) >> ISH, c0 = ((g + 1) * NCG) >> ISH;
#pragma unroll
            for (int dy = 0; dy < T::BH; dy++)
#pragma unroll
                for (int k = 0; k < T::YWI; k++) if (k >= y0) fence_words<1>(&in.y[dy][k]);
#pragma unroll
            for (int k = 0; k < T::CWI; k++) if (k >= c0) { fence_words<1>(&in.cb[k]); fence_words<1>(&in.cr[k]); }
        }
        {
            constexpr int OSH = WOUT ? 1 : 2;
            const int y0 = (g * GW) >> OSH, y1 = ((g + 1) * GW - 1) >> OSH, c0 = (g * NCG) >> OSH, c1 = ((g + 1) * NCG - 1) >> OSH;
#pragma unroll
            for (int dy = 0; dy < T::BH; dy++)
#pragma unroll
                for (int k = 0; k < T::YWO; k++) if (k >= y0 && k <= y1) fence_words<1>(&out.y[dy][k]);
#pragma unroll
            for (int k = 0; k < T::CWO; k++) if (k >= c0 && k <= c1) { fence_words<1>(&out.cb[k]); fence_words<1>(&out.cr[k]); }
        }
#endif
    }
}

// ---------------------------------------------------------------- work distribution
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

// Every wave takes its first chunk by its id (a burst of atomics on one address at kernel start serialises in the L2).  After
// that chunks come from a TWO-LEVEL queue: a wave draws a ticket from its workgroup's LDS counter (a ds_add_rtn, ~100 cycles on the
// lgkm counter, the vector-memory pipeline keeps running); ticket 16 j + slot means chunk base[j] + slot, and the wave that draws
// slot 0 fetches base[j] = atomicAdd(queue, 16) for the block and publishes it in LDS (the others of that block, if they arrive
// before it has landed, spin on the ready tag -- all waves of a workgroup are resident, the publisher cannot be descheduled).  One
// claim in sixteen pays the L2 round trip that round 2 paid on every claim (its phase timers: 9 % of a wave's time, with the memory
// pipeline drained behind it), and the global counter sees a sixteenth of the traffic, so short launches can use smaller chunks.
#ifndef LUTR_T2_QUEUE2
#define LUTR_T2_QUEUE2 1
#endif
typedef __attribute__((address_space(3))) volatile unsigned *lds_vup;
DEV bool claim_chunk(const Geom &TG, int lane, int wgq_off, int &fr, int &sx, int &ry, int &rem, bool &first)
{
    unsigned c = 0;
    if (first) {
        // (a wave whose id is not a chunk has no work at all: the counter starts behind the ids.  It must not touch the
        // allocator's LDS words either -- this call runs before they are initialised.)
        first = false;
        c = (unsigned)((int)(blockIdx.x * LUTR_T2_WPB) + uni((int)(threadIdx.x >> 6)));
        return chunk_at(TG, c, fr, sx, ry, rem);
    }
#if LUTR_T2_QUEUE2
    const lds_vup q = (lds_vup)(uintptr_t)(unsigned)(lds_base() + wgq_off);
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned *)q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    t = (unsigned)uni((int)t);
    const unsigned j = t >> 4, slot = t & 15u, r = j & 7u;
    if (slot == 0) {
        if (lane == 0) {
            c = atomicAdd(TG.queue, 16u) + TG.qbase;
            q[8 + r] = c;                                  // base[r]
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            q[16 + r] = j + 1u;                            // ready[r]
        }
        c = (unsigned)uni((int)c);
    } else {
        while ((unsigned)uni((int)q[16 + r]) != j + 1u) __builtin_amdgcn_s_sleep(2);
        c = (unsigned)uni((int)q[8 + r]) + slot;
    }
#else
    if (lane == 0) c = atomicAdd(TG.queue, 1u) + TG.qbase;
    c = (unsigned)uni((int)c);
#endif
    return chunk_at(TG, c, fr, sx, ry, rem);
}

// Every wave calls this once, when it will claim no more: the last one puts the two words back to zero for the next launch.  (A wave's
// claims have returned before it gets here -- it needed their values -- so the plain stores cannot overtake anybody's atomic.)
DEV void queue_leave(const Geom &TG, int lane, int wgq_off)
{
    // two levels, like the claims: the waves of a workgroup count themselves out in LDS (word 24 of the allocator's block), the last one
    // reports the workgroup -- 4096 atomics on one address at the end of a short launch cost it 15 us
    if (lane == 0) {
        const lds_vup q = (lds_vup)(uintptr_t)(unsigned)(lds_base() + wgq_off);
        const unsigned left = __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned *)(q + 24), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (left == (unsigned)LUTR_T2_WPB - 1u) {
            const unsigned done = atomicAdd(TG.queue + 1, 1u);
            if (done == gridDim.x - 1u) {
                __hip_atomic_store(TG.queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(TG.queue + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int WIN, int WOUT, int CSX, int CSY, int INTERP, int PRE, int V>
__global__ __launch_bounds__(64 * LUTR_T2_WPB, LUTR_T2_WAVES_PER_EU)
void k_yuv_tile2(LutConsts L_, YuvConsts K_, Planes2 P, FrameGeom G, Geom TG)
{
    using T = Tile<WIN, WOUT, CSX, CSY>;
    using N = Node<INTERP, V>;
    // mixed tiles: the strict kernels (see the vote below); not trilinear with the prologue, whose body has no register left for the
    // lanes' verdict (it spilled three VGPRs to scratch inside the body)
    // (likewise 8-bit 4:4:4 trilinear: eight spilled registers)
    constexpr bool kMixed = LUTR_T2_MIXED && (V != V_FAST || LUTR_T2_MIXED_FAST) && !(INTERP == LUTR_INTERP_TRILINEAR && (PRE || (!WIN && !CSX)));
    LutConsts L = L_;
    YuvConsts K = K_;
    // a wave-uniform constant used by several VALU ops per pixel is worth a VGPR (an SGPR operand halves the issue rate
    // of plain fp32 ops on gfx950, tools/ubench); the trilinear bodies have no registers to spare
    if constexpr (INTERP != LUTR_INTERP_TRILINEAR && LUTR_T2_PIN >= 1) {
        K.cyr = in_vgpr(K_.cyr); K.cyg = in_vgpr(K_.cyg); K.cyb = in_vgpr(K_.cyb);
        if constexpr (LUTR_T2_PIN >= 2) { K.ky = in_vgpr(K_.ky); K.yb = in_vgpr(K_.yb); K.yob = in_vgpr(K_.yob); }
        if constexpr (!N::fast) L.maxf = in_vgpr(L_.maxf);
    }
    if (lds_base() != 0) __builtin_trap();        // crd_table8 addresses the table absolutely
#if LUTR_T2_KARG && defined(__HIP_DEVICE_COMPILE__)
    {   // pos_at reads the plane descriptors from the kernel-argument segment at the offset a struct of the parameters gives them:
        // make sure that IS where they are before any address is built from them
        struct KernArgsChk { LutConsts L; YuvConsts K; Planes2 P; FrameGeom G; Geom TG; };
        const __attribute__((address_space(4))) Planes2 *chk = (const __attribute__((address_space(4))) Planes2 *)(
            (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KernArgsChk, P));
        if (chk->s[0] != P.s[0] || chk->d[2] != P.d[2] || chk->ss[1] != P.ss[1] || chk->dfs[2] != P.dfs[2]) __builtin_trap();
    }
#endif
    // the per-pixel constants of the table variants, times 8 (crd_table8); map_box keeps the plain ones
    YuvConsts KB = K;
    if constexpr (V >= V_TAB) {
        KB.ky = K.ky * 8.0f; KB.yb = K.yb * 8.0f;
        KB.krv = K.krv * 8.0f; KB.kgu = K.kgu * 8.0f; KB.kgv = K.kgv * 8.0f; KB.kbu = K.kbu * 8.0f;
    }
    const int tube_nb = 2 * TG.tube_h + 3;                    // nodes across each difference axis: cells -H..H, corners -H-1..H+1
    const int tube_nodes = TG.tube_h > 0 ? L.n1 * TG.tube_plane : 0;
    const int lane = threadIdx.x & 63;
    const int wib = uni(threadIdx.x >> 6);
    const int tab_bytes = TG.tab_entries * 8;
    const int scratch_off = tab_bytes + wib * 64;             // the window's cell ranges (second-level test)
    const int tube_off = tab_bytes + kScratch;
    const int slice_off = tube_off + tube_nodes * N::lds + wib * TG.win_nodes * N::lds;
    Win Wt;                                                   // the tube as a window: same address form as a staged one
    // index = pr * plane + (pg - pr + H + 1) * nb + (pb - pg + H + 1)   [(pb - pr + H + 1) without LUTR_T2_TUBE_BG]
    if (LUTR_T2_TUBE_BG) { Wt.o_g = N::lds * (tube_nb - 1); Wt.o_r = N::lds * (TG.tube_plane - tube_nb); }
    else { Wt.o_g = N::lds * tube_nb; Wt.o_r = N::lds * (TG.tube_plane - tube_nb - 1); }
    Wt.fr = (float)Wt.o_r; Wt.fg = (float)Wt.o_g; Wt.fb = (float)N::lds;
    Wt.fc = (float)(lds_base() + tube_off + N::lds * ((TG.tube_h + 1) * tube_nb + TG.tube_h + 1));
    unsigned st_tube = 0, st_mixed = 0;
    int fr, sx, ry, rem;                                      // the tile being fetched next
    bool first = true;
    const int wgq_off = tab_bytes + kWaveScratch;             // the workgroup's chunk allocator (claim_chunk)
    if (threadIdx.x < 24) ((unsigned *)(smem + wgq_off))[threadIdx.x + (threadIdx.x ? 7 : 0)] = 0u;      // ticket, base[8], ready[8]
    const bool have_work = claim_chunk(TG, lane, wgq_off, fr, sx, ry, rem, first);      // (no atomic: a wave's first chunk is its id)
    const int lw = 1 << TG.lw_log2, lh_log2 = 6 - TG.lw_log2;
    const int lx = lane & (lw - 1), ly = lane >> TG.lw_log2;
    const int cr0 = G.row0 >> CSY;                            // first unit row of this call's row range

    Win W;
    W.fr = W.fg = W.fb = W.fc = 0.0f; W.o_r = W.o_g = 0;
    if (lane == 0) {     // empty box: lo > hi in every plane, so the first tile always restages
        Box bx;
        bx.ylo = bx.cblo = bx.crlo = 0xffffffffu; bx.yhi = bx.cbhi = bx.crhi = 0u;
        box_store(scratch_off, bx);
    }
    if (TG.whole) {
        // lattice layout of the global copy ((N+1)^3 nodes, blue fastest, index N replicates N-1): prev + 1 is always staged
        W.o_r = N::lds * TG.whole_a; W.o_g = N::lds * TG.whole_b;
        W.fr = (float)W.o_r; W.fg = (float)W.o_g; W.fb = (float)N::lds;
        W.fc = (float)(lds_base() + tab_bytes + kScratch);
    }
    constexpr int YIB = T::YWI * 4, YOB = T::YWO * 4, CIB = T::CWI * 4, COB = T::CWO * 4;

    // Where a tile lives: wave-uniform strip pointers (frame, strip and unit row folded in) plus the clamps for lanes past
    // the right / bottom edge.  Built from scratch when a chunk is claimed; inside a chunk the next tile is one step down,
    // i.e. six 64-bit scalar adds -- recomputing the 64-bit products per tile cost ~140 scalar instructions per tile and wave.
    struct TilePos { const uint8_t *s0, *s1, *s2; uint8_t *d0, *d1, *d2; int xlim, ylim; };
    const unsigned ystep_s0 = (unsigned)((T::BH << lh_log2) * (int)P.ss[0]), ystep_s1 = (unsigned)((int)P.ss[1] << lh_log2),
                   ystep_s2 = (unsigned)((int)P.ss[2] << lh_log2);
    const unsigned ystep_d0 = (unsigned)((T::BH << lh_log2) * (int)P.ds[0]), ystep_d1 = (unsigned)((int)P.ds[1] << lh_log2),
                   ystep_d2 = (unsigned)((int)P.ds[2] << lh_log2);
    // The plane bases and frame strides (24 SGPRs) are only needed here, once per chunk: read them from the kernel-argument
    // segment when they are, instead of keeping them resident (and spilled to VGPR lanes) through every tile
    // (SGPR spills of the headline kernel 101 -> 82; the layout assumption is checked at kernel start).
    auto pos_at = [&](int f, int tsx, int try_) {
        TilePos q;
#if LUTR_T2_KARG && defined(__HIP_DEVICE_COMPILE__)
        struct KernArgs { LutConsts L; YuvConsts K; Planes2 P; FrameGeom G; Geom TG; };   // the kernel's parameter list
        typedef const __attribute__((address_space(4))) Planes2 *PlanesK;
        PlanesK Pk = (PlanesK)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KernArgs, P));
        asm volatile("" : "+s"(Pk));                                      // opaque: the loads stay here, not hoisted out of the loop
        const Planes2 P = *Pk;
#endif
        const long long urow0 = cr0 + (try_ << lh_log2);                  // wave-uniform
        q.s0 = P.s[0] + f * P.sfs[0] + urow0 * T::BH * (long long)P.ss[0] + (long long)tsx * lw * YIB;
        q.s1 = P.s[1] + f * P.sfs[1] + urow0 * (long long)P.ss[1] + (long long)tsx * lw * CIB;
        q.s2 = P.s[2] + f * P.sfs[2] + urow0 * (long long)P.ss[2] + (long long)tsx * lw * CIB;
        q.d0 = P.d[0] + f * P.dfs[0] + urow0 * T::BH * (long long)P.ds[0] + (long long)tsx * lw * YOB;
        q.d1 = P.d[1] + f * P.dfs[1] + urow0 * (long long)P.ds[1] + (long long)tsx * lw * COB;
        q.d2 = P.d[2] + f * P.dfs[2] + urow0 * (long long)P.ds[2] + (long long)tsx * lw * COB;
        q.xlim = TG.uw - 1 - tsx * lw;
        q.ylim = TG.urows - 1 - (try_ << lh_log2);
        return q;
    };
    auto pos_down = [&](TilePos &q) {
        q.s0 += ystep_s0; q.s1 += ystep_s1; q.s2 += ystep_s2;
        q.d0 += ystep_d0; q.d1 += ystep_d1; q.d2 += ystep_d2;
        q.ylim -= 1 << lh_log2;
    };
    // Input words of a tile.  Idle lanes of edge tiles re-read a valid unit of the same tile.  32-bit lane offsets: the
    // launcher only sends strides below 2^24 (a tile spans at most 128 rows).
    auto load_tile = [&](TileIn<WIN, WOUT, CSX, CSY> &dst, const TilePos &q) {
        const unsigned lxc = (unsigned)min(lx, q.xlim), lyc = (unsigned)min(ly, q.ylim);
        const unsigned y0 = __umul24(lyc * T::BH, P.ss[0]) + lxc * YIB;
#pragma unroll
        for (int dy = 0; dy < T::BH; dy++) ldw<T::YWI>(dst.y[dy], q.s0 + (y0 + dy * P.ss[0]));
        ldw<T::CWI>(dst.cb, q.s1 + (__umul24(lyc, P.ss[1]) + lxc * CIB));
        ldw<T::CWI>(dst.cr, q.s2 + (__umul24(lyc, P.ss[2]) + lxc * CIB));
    };

    bool have_win = false;
    int l2run = 0;                 // consecutive tiles vouched for by the second-level test only
    // event counters (lutr_ctx_tile_stats): kept in the wave's LDS scratch, not in registers (see Box)
    const bool counting = TG.stats != nullptr;
    unsigned *cnt = (unsigned *)(smem + scratch_off + 24);      // [0] second-level tests, [1] restage attempts; +56: [8] gather, [9] staged
    if (lane == 0) { cnt[0] = cnt[1] = 0; cnt[8] = cnt[9] = 0; }
    unsigned st_tiles = 0;
#define T2_COUNT(i) do { if (counting && lane == 0) cnt[i] += 1u; } while (0)
#ifdef LUTR_T2_DEBUG_STATS
    unsigned tk_head = 0, tk_l2 = 0, tk_rest = 0, tk_body = 0, tk_gath = 0, tk_store = 0, tk_wait = 0;
    unsigned long long tk = __builtin_amdgcn_s_memrealtime();
#define TK(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); acc += (unsigned)(now_ - tk); tk = now_; }
#else
#define TK(acc)
#endif
    // The first tile's loads go out BEFORE the workgroup stages its table and tube: a launch of a few frames would otherwise
    // spend its first ~10 us with an idle memory system.
    TilePos np = pos_at(have_work ? fr : 0, have_work ? sx : 0, have_work ? ry : 0);          // the tile being fetched
    TileIn<WIN, WOUT, CSX, CSY> nxt;
    if (have_work) load_tile(nxt, np);
    if constexpr (V >= V_TAB) coord_table_fill<INTERP>(L, TG.tab_entries);     // the kernel's only barrier ...
    if (TG.whole) {                                                             // ... but for this one, in whole-lattice mode
        char *dst = smem + TG.tab_entries * 8 + kScratch;
        const int nodes = L.n1 * L.n1 * L.n1, plane = L.n1 * L.n1;
        for (int i = threadIdx.x; i < nodes; i += 64 * LUTR_T2_WPB) {
            const int r = i / plane, rem = i - r * plane, g = rem / L.n1, b = rem - g * L.n1;
            const int o = r * TG.whole_a + g * TG.whole_b + b;                  // (padded strides: see whole_strides)
            if constexpr (N::rec) {
                const u3 rc = make_rec(L.lat16[i], L.lat16[i + plane < nodes ? i + plane : i]);
                uint32_t *q = (uint32_t *)(dst + 12 * o); q[0] = rc.x; q[1] = rc.y; q[2] = rc.z;
            } else if constexpr (N::fast) ((uint2 *)dst)[o] = L.lat16[i];
            else {
                const float4 v = L.lat[i];
                if constexpr (N::lds == 16) ((float4 *)dst)[o] = v;
                else { float *q = (float *)(dst + 12 * o); q[0] = v.x; q[1] = v.y; q[2] = v.z; }
            }
        }
        __syncthreads();
    }
    if (TG.tube_h > 0) {
        // node (ir, ig, ib) = lattice (r, g = r + ig - H - 1, b = g + ib - H - 1) [b = r + ib - H - 1 without LUTR_T2_TUBE_BG], clamped
        // (a clamped node is never read by a valid pixel)
        char *dst = smem + TG.tab_entries * 8 + kScratch;
        const int plane = TG.tube_plane, nmax = L.n1 - 1;
        // four nodes per thread in flight: the staging is a chain of L2 round trips, and a short launch pays it in full
        constexpr int kSB = 4;
        for (int base = threadIdx.x; base < tube_nodes; base += kSB * 64 * LUTR_T2_WPB) {
            int src[kSB], src1[kSB];
#pragma unroll
            for (int k = 0; k < kSB; k++) {
                const int i = min(base + k * 64 * LUTR_T2_WPB, tube_nodes - 1);
                const int ir = i / plane, rem = i - ir * plane, ig = min(rem / tube_nb, tube_nb - 1), ib = rem - ig * tube_nb;   // (padding: any node)
                const int gq = ir + ig - TG.tube_h - 1;
                const int g = min(max(gq, 0), nmax), b = min(max((LUTR_T2_TUBE_BG ? gq : ir) + ib - TG.tube_h - 1, 0), nmax);
                src[k] = (ir * L.n1 + g) * L.n1 + b;
                src1[k] = (min(ir + 1, nmax) * L.n1 + g) * L.n1 + b;
            }
            typename std::conditional<N::fast, uint2, float4>::type val[kSB];
            uint2 val1[N::rec ? kSB : 1];
#pragma unroll
            for (int k = 0; k < kSB; k++) {
                if constexpr (N::fast) val[k] = L.lat16[src[k]];
                else val[k] = L.lat[src[k]];
                if constexpr (N::rec) val1[k] = L.lat16[src1[k]];
            }
#pragma unroll
            for (int k = 0; k < kSB; k++) {
                const int i = min(base + k * 64 * LUTR_T2_WPB, tube_nodes - 1);      // (the last batch re-writes the final node: harmless)
                if constexpr (N::rec) { const u3 rc = make_rec(val[k], val1[k]); uint32_t *q = (uint32_t *)(dst + 12 * i); q[0] = rc.x; q[1] = rc.y; q[2] = rc.z; }
                else if constexpr (N::fast) ((uint2 *)dst)[i] = val[k];
                else if constexpr (N::lds == 16) ((float4 *)dst)[i] = val[k];
                else { float *q = (float *)(dst + 12 * i); q[0] = val[k].x; q[1] = val[k].y; q[2] = val[k].z; }
            }
        }
        __syncthreads();
    }
    __syncthreads();                          // (the allocator's LDS words are initialised; the general variant has no other barrier)
    if (!have_work) { queue_leave(TG, lane, wgq_off); return; }      // (after the barriers above: every wave of the workgroup takes part in the staging)

    for (bool more = true; more;) {
        TileIn<WIN, WOUT, CSX, CSY> in = nxt;
        const TilePos cp = np;                // the tile being computed (its store pointers and clamps)
        // Next tile: the one below in this chunk, else the first tile of a newly claimed chunk.  Its loads are issued
        // NOW, before this tile's stores (vmcnt retires in order).  When the queue is drained the current tile is simply
        // fetched again, so every path has the same number of memory operations in flight.
        if (--rem > 0) { ry++; pos_down(np); }
        else {
            more = claim_chunk(TG, lane, wgq_off, fr, sx, ry, rem, first);
            if (more) np = pos_at(fr, sx, ry);
        }
        load_tile(nxt, np);
#if LUTR_T2_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#elif LUTR_T2_PRIO == 2
        __builtin_amdgcn_s_setprio(3);
#endif

#ifndef LUTR_T2_EXP
#define LUTR_T2_EXP 0             // timing experiments only (wrong pixels): 1 = trust the first window forever, 2 = + no stores, 3 = no body, 5.. = body repeated
#endif
#ifdef LUTR_T2_DEBUG_STATS
        TK(tk_store)                 // issuing the next tile's loads counts as store/loop time
        fence_words<T::YWI * T::BH>(&in.y[0][0]); fence_words<T::CWI>(in.cb); fence_words<T::CWI>(in.cr);
        TK(tk_wait)                  // ... and this is the wait for this tile's own loads
#endif
        Ext e = extremes<WIN, WOUT, CSX, CSY>(in);             // without the luma minimum: see below
        bool use_tube = false, mixed = false, lane_in = true;
        if (TG.tube_h > 0) {
            // level 0: the tile's chroma keeps it inside the workgroup's grey tube (and its raw codes are legal for the clamp-free body)
            const uint32_t top = pack_hi<WIN>(TG.max_raw);
            // one vote for the common case: chroma inside the raw interval (which lies inside the legal codes) and luma legal
            const uint32_t ybits = pk_subsat_vs(e.ymax, top);
            const uint32_t rbits = pk_subsat_sv(TG.tube_rlo, e.cbmin) | pk_subsat_vs(e.cbmax, TG.tube_rhi) |
                                   pk_subsat_sv(TG.tube_rlo, e.crmin) | pk_subsat_vs(e.crmax, TG.tube_rhi);
            if (__all((ybits | rbits) == 0u)) use_tube = true;
            else if (__all((ybits | pk_subsat_vs(e.cbmax, top) | pk_subsat_vs(e.crmax, top)) == 0u)) {
                // legal, but outside the interval: the bound itself.  Per sample for the strict kernels (their tube is 6 cells wide:
                // sigma = 16 frames +9 %, natural -0.8 %, 3 x chroma -1.5 %); the fast kernels' 8-cell tube gains 1.7 % there and loses
                // 1.4 % on 3 x chroma: corner form
                if constexpr (T::NC <= 4 && V != V_FAST && LUTR_T2_TUBE_SAMPLES)
                    lane_in = tube_holds_samples<WIN, WOUT, CSX, CSY, PRE>(K, TG, in);
                else lane_in = tube_holds<WIN, PRE>(K, TG, e);
                use_tube = __all(lane_in);
                // MIXED tile: a few lanes outside the tube (sensor noise: one outlier unit in 64 sends a whole tile away).  Every
                // lane runs the tube body -- an outlier reads wherever its numbers point, LDS reads cannot fault -- and the
                // outliers alone run the gather body afterwards, under a divergent branch: its instructions issue once for the
                // wave, its memory requests (what a gather costs) shrink to those lanes.
                // (strict kernels only: with a cheap restage -- win_plane_stride -- the fast kernels' H = 8 tube and 367-node windows do as
                // well or better without it: three times the chroma 540 vs 518 Gpx/s, sigma-16 526 vs 539, profiles/r03_exp16_*.txt)
                if constexpr (kMixed)
                    if (!use_tube) mixed = __popcll(__ballot(!lane_in)) <= TG.mix_max;
            }
        }
        bool use_lds = use_tube || mixed;
        if (!use_lds && !TG.whole) {
            e.ymin = luma_min<WIN, WOUT, CSX, CSY>(in);
            use_lds = box_holds(scratch_off, e);               // first level: raw extremes against the window's raw box
        }
        if (LUTR_T2_EXP >= 1 && have_win) use_lds = true;
        TK(tk_head)
        st_tiles++;
        if (TG.whole) {
            // every cell is in LDS; only raw codes above 2^din - 1 (which the padded table does not cover) need the clamping body
            const uint32_t top = pack_hi<WIN>(TG.max_raw);
            use_lds = __all((pk_subsat_vs(e.ymax, top) | pk_subsat_vs(e.cbmax, top) | pk_subsat_vs(e.crmax, top)) == 0u);
        } else if (use_tube) {
            st_tube++;
        } else if (mixed) {
            st_mixed++;
        } else if (use_lds) {
#ifdef LUTR_T2_DEBUG_STATS
            if (l2run > 0 && lane == 0) atomicAdd(&TG.stats[21 + (l2run <= 1 ? 0 : l2run <= 2 ? 1 : l2run <= 4 ? 2 : l2run <= 8 ? 3 : l2run <= 16 ? 4 : 5)], (unsigned)l2run);
#endif
            l2run = 0;
        } else {
            T2_COUNT(0);
#ifdef LUTR_T2_DEBUG_STATS
            {   // which bound of the raw box failed (wave-wide), and was there a box at all
                const uint4 p = *(const uint4 *)(smem + scratch_off + 32);
                const uint2 q = *(const uint2 *)(smem + scratch_off + 48);
                const bool f0 = !__all(pk_subsat(p.x, e.ymin) == 0u), f1 = !__all(pk_subsat(e.ymax, p.y) == 0u);
                const bool f2 = !__all(pk_subsat(p.z, e.cbmin) == 0u), f3 = !__all(pk_subsat(e.cbmax, p.w) == 0u);
                const bool f4 = !__all(pk_subsat(q.x, e.crmin) == 0u), f5 = !__all(pk_subsat(e.crmax, q.y) == 0u);
                if (lane == 0) {
                    if (p.x > p.y) atomicAdd(&TG.stats[18], 1u);          // empty box
                    else {
                        if (f0) atomicAdd(&TG.stats[12], 1u); if (f1) atomicAdd(&TG.stats[13], 1u);
                        if (f2) atomicAdd(&TG.stats[14], 1u); if (f3) atomicAdd(&TG.stats[15], 1u);
                        if (f4) atomicAdd(&TG.stats[16], 1u); if (f5) atomicAdd(&TG.stats[17], 1u);
                        if ((f0 || f1) && !(f2 || f3 || f4 || f5)) atomicAdd(&TG.stats[19], 1u);   // luma only
                        if (!(f0 || f1) && (f2 || f3 || f4 || f5)) atomicAdd(&TG.stats[20], 1u);   // chroma only
                    }
                }
            }
#endif

            // second level: exact cells against the window's cell ranges.  Raw codes must be legal (<= 2^din - 1) for the
            // clamp-free table reads of tile_bounds and of the LDS body; a tile with wild container values takes the gather body.
            const uint32_t top = pack_hi<WIN>(TG.max_raw);
            const bool legal = __all((pk_subsat_vs(e.ymax, top) | pk_subsat_vs(e.cbmax, top) | pk_subsat_vs(e.crmax, top)) == 0u);
            Bnd bn;
            bn.rmin = bn.gmin = bn.bmin = 1e9f; bn.rmax = bn.gmax = bn.bmax = -1e9f;
            if (legal) bn = tile_bounds<WIN, WOUT, CSX, CSY, INTERP, PRE, V>(L, KB, TG, in);
            use_lds = have_win && legal && cells_hold(scratch_off, bn);
            TK(tk_l2)
            if (use_lds) {
                l2run++;
                // back-off: the 1st, 4th, 16th consecutive pass tries to give the window a raw box again.  (Staging a boxed window
                // instead when that fails was measured: such streaks sit on vertical colour edges, where no box fits: -0.7 %.)
                if (l2run == 1 || l2run == 4 || l2run == 16) rebox<WIN, INTERP, PRE, V>(L, K, TG, e, scratch_off, lane);
            } else {
#ifdef LUTR_T2_DEBUG_STATS
                if (l2run > 0 && lane == 0) atomicAdd(&TG.stats[21 + (l2run <= 1 ? 0 : l2run <= 2 ? 1 : l2run <= 4 ? 2 : l2run <= 8 ? 3 : l2run <= 16 ? 4 : 5)], (unsigned)l2run);
#endif
                l2run = 0;
            }
            if (!use_lds) {
                T2_COUNT(1);
                use_lds = restage<WIN, INTERP, PRE, V>(W, L, K, TG, e, bn, slice_off, scratch_off, lane);
                if (use_lds) { T2_COUNT(9); have_win = true; }
                TK(tk_rest)
            }
        }
        TileOut<WIN, WOUT, CSX, CSY> out;
        if (LUTR_T2_EXP == 3) {
#pragma unroll
            for (int dy = 0; dy < T::BH; dy++)
#pragma unroll
                for (int k = 0; k < T::YWO; k++) out.y[dy][k] = in.y[dy][k % T::YWI];
#pragma unroll
            for (int k = 0; k < T::CWO; k++) { out.cb[k] = in.cb[k % T::CWI]; out.cr[k] = in.cr[k % T::CWI]; }
        } else
        if (use_lds) {
            tile_body<true, WIN, WOUT, CSX, CSY, INTERP, PRE, V>(L, KB, (use_tube || mixed) ? Wt : W, TG, in, out); TK(tk_body)
            if constexpr (LUTR_T2_EXP >= 5) {      // the body again, EXP - 4 times, on inputs the compiler cannot tell are the same
#pragma unroll 1
                for (int rep = 0; rep < LUTR_T2_EXP - 4; rep++) {
                    fence_words<T::YWI * T::BH>(&in.y[0][0]); fence_words<T::CWI>(in.cb); fence_words<T::CWI>(in.cr);
                    tile_body<true, WIN, WOUT, CSX, CSY, INTERP, PRE, V>(L, KB, W, TG, in, out);
                }
            }
        }
        bool gather = !use_lds;                      // whole tiles (wave-uniform) ...
        if constexpr (kMixed) gather = gather || (mixed && !lane_in);     // ... or the outliers of a mixed tile (divergent)
        if (gather) {
            tile_body<false, WIN, WOUT, CSX, CSY, INTERP, PRE, V>(L, KB, W, TG, in, out); TK(tk_gath)
        }
        if (!use_lds) T2_COUNT(8);
#if LUTR_T2_PRIO == 1
        __builtin_amdgcn_s_setprio(3);      // stores, the queue and the next tile's loads go first
#elif LUTR_T2_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
        {
            // Idle lanes of edge tiles processed a duplicate of a valid unit of this tile (load_tile clamps), so they
            // store the same bytes to the same place as its owner: no branch, fixed store count.
            const unsigned lxc = (unsigned)min(lx, cp.xlim), lyc = (unsigned)min(ly, cp.ylim);
            const unsigned y0 = __umul24(lyc * T::BH, P.ds[0]) + lxc * YOB;
            if (LUTR_T2_EXP != 2 || out.cb[0] == 0x12345u) {
#pragma unroll
            for (int dy = 0; dy < T::BH; dy++) stw<T::YWO>(cp.d0 + (y0 + dy * P.ds[0]), out.y[dy]);
            stw<T::CWO>(cp.d1 + (__umul24(lyc, P.ds[1]) + lxc * COB), out.cb);
            stw<T::CWO>(cp.d2 + (__umul24(lyc, P.ds[2]) + lxc * COB), out.cr);
            }
        }
        TK(tk_store)
    }
#ifdef LUTR_T2_DEBUG_STATS
    if (TG.stats && lane == 0) {
        atomicAdd(&TG.stats[4], tk_head >> 4); atomicAdd(&TG.stats[5], tk_l2 >> 4); atomicAdd(&TG.stats[7], tk_rest >> 4);
        atomicAdd(&TG.stats[11], tk_wait >> 4); atomicAdd(&TG.stats[8], tk_body >> 4); atomicAdd(&TG.stats[9], tk_gath >> 4); atomicAdd(&TG.stats[10], tk_store >> 4);
    }
#endif
    queue_leave(TG, lane, wgq_off);
    if (TG.stats && lane == 0) {
        atomicAdd(&TG.stats[0], st_tiles); atomicAdd(&TG.stats[1], cnt[1]);
        atomicAdd(&TG.stats[2], cnt[8]); atomicAdd(&TG.stats[3], cnt[9]); atomicAdd(&TG.stats[6], cnt[0]);
        atomicAdd(&TG.stats[12], st_tube); atomicAdd(&TG.stats[30], st_mixed);
    }
}

}  // namespace t2

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

// false: the runtime refused 160 KB of dynamic LDS for this kernel (the launch would fail): the caller declines the call
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

// Largest index (unsigned)(yy + {rv, gv, bu}) can take for raw codes in [0, 2^din - 1]: each sum is monotone in its inputs.
int table_entries(const YuvConsts &K, int din)
{
    const float mr = (float)((1 << din) - 1);
    auto pre = [&](float v, float a, float b) { return K.pre != 0.0f ? fminf(fmaxf(floorf(fmaf(a, v, b)), 0.0f), K.pre_max) : v; };
    const float y1 = pre(mr, K.py, K.pyb), c0 = pre(0.0f, K.pc, K.pcb), c1 = pre(mr, K.pc, K.pcb);
    const float yy = fmaf(K.ky, y1, K.yb);
    const float r = yy + K.krv * (c1 - K.coff), g = yy + fmaf(K.kgu, c0 - K.coff, K.kgv * (c0 - K.coff)), b = yy + K.kbu * (c1 - K.coff);
    const float top = fmaxf(fmaxf(r, g), fmaxf(b, K.max_l));
    return ((int)top + 3) & ~1;           // even: the window slices behind the table stay 16-byte aligned (ds_read_b128)
}

// Nodes between two r planes of the tube.  The lanes of a wave read cells that are mostly one step apart (neighbouring pixels):
// with node index = pr * A + pg * B + pb two of them collide in the LDS banks when dr * A + dg * B + db is a multiple of 32 (a tap
// read is 32 lanes per pass, the bank is the dword address mod 32 or 64, node strides of 2 or 3 dwords are invertible mod 32).
// The unpadded 15 x 15 and 17 x 17 planes of the strict kernels' tubes have exactly that for (dr, dg) = +-(1, 1): every luma step
// that moves r and g but not b costs a second LDS pass.  A few nodes of padding per plane remove it.
int tube_plane_stride(int nb, int node)
{
    // 16-byte nodes are read with ds_read_b128: 16 lanes per pass, bank = dword address mod 64, a node is four dwords -- two lanes
    // collide when their node indices agree mod 16 (not 32)
    const int mod = node == 16 ? 16 : 32;
    int best = nb * nb, best_bad = 1 << 30;
    for (int pad = 0; pad < 12; pad++) {
        const int plane = nb * nb + pad;
        const int A = LUTR_T2_TUBE_BG ? plane - nb : plane - nb - 1, B = LUTR_T2_TUBE_BG ? nb - 1 : nb;
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
bool out_clip_dead(const YuvConsts &K, int chroma_n)
{
    const float m = K.max_l, mn = K.max_l * (float)chroma_n;
    auto hi = [](float c, float v) { return c > 0.0f ? v : 0.0f; };
    const float y = fmaf(K.cyr, hi(K.cyr, m), fmaf(K.cyg, hi(K.cyg, m), fmaf(K.cyb, hi(K.cyb, m), K.yob)));
    const float cb = fmaf(K.cbr, hi(K.cbr, mn), fmaf(K.cbg, hi(K.cbg, mn), fmaf(K.cbb, hi(K.cbb, mn), K.cob)));
    const float cr = fmaf(K.crr, hi(K.crr, mn), fmaf(K.crg, hi(K.crg, mn), fmaf(K.crb, hi(K.crb, mn), K.cob)));
    const float lim = K.max_o + 1.0f;
    return y < lim && cb < lim && cr < lim;
}

}  // namespace

#ifndef LUTR_T2_ONLY
#define LUTR_T2_ONLY 0      // 1: build the headline instance alone (development)
#endif

// The LUTR_* tuning knobs of this launcher (tools/ only) are read ONCE per process: round 2 called getenv a dozen times per
// launch, which a stream of single-frame applies pays every time (bench.py host_us_per_apply).
namespace {
struct Knob {
    const char *v;
    explicit Knob(const char *name) : v(getenv(name)) {}
    explicit operator bool() const { return v != nullptr; }
    int num() const { return atoi(v); }
};
#define T2_KNOB(NAME) ([]() -> const Knob & { static const Knob k(NAME); return k; }())
}  // namespace

// Whole-lattice mode: node (r, g, b) sits at index r * A + g * B + b.  With A = n1^2, B = n1 neighbouring cells collide in the LDS
// banks for unlucky sizes (n1 = 20: A = 400 = 0 mod 16 -- every step along r lands in the same bank group of a ds_read_b128).  A few
// nodes of padding per row and per plane remove that, as tube_plane_stride does for the tube.  Returns the bytes, or 0 if no layout fits.
static long long whole_strides(int n1, int node, long long room, int *A, int *B)
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

// The tube's chroma bound under a shared prelut (LutConsts::pre_shared).  FFmpeg resamples a cineSpace shaper WITHOUT normalising the
// interpolation weight by the segment width (parse_cinespace: `mix = x - in_prelut[idx]`), so the curve lut3d applies is a staircase:
// nearly flat inside a segment, a jump at every input point -- its largest step between two codes says nothing about the rise over
// 100.  What the tube needs is D(d) = max over x of s(x + d) - s(x), the most the coordinate can rise over d codes: two channel codes at
// most d apart then sit at most floor(D(d)) + 1 cells apart.  Returns the largest d with D(d) <= h - 0.002 (0 if none).
static int prelut_tube_bound(const float *s, int maxi, int h)
{
    struct Memo { const float *s; int maxi, h, d; float first, last; };
    static thread_local Memo memo = {nullptr, 0, 0, 0, 0.0f, 0.0f};
    if (!s || maxi < 1) return 0;
    if (memo.s == s && memo.maxi == maxi && memo.h == h && memo.first == s[1] && memo.last == s[maxi]) return memo.d;
    const float lim = (float)h - 2e-3f;
    int d = 0;
    for (int cand = 1; cand <= maxi; cand++) {
        float rise = 0.0f;
        for (int x = 0; x + cand <= maxi; x++) rise = fmaxf(rise, s[x + cand] - s[x]);
        if (rise > lim) break;
        d = cand;
    }
    memo = Memo{s, maxi, h, d, s[1], s[maxi]};
    return d;
}

// Which variant serves this call?
static int tile2_variant(const LutConsts &L, const YuvConsts &K, int lut_depth, int csx, int csy, bool fast)
{
    const bool eq = L.pre ? L.pre_shared != 0 : (L.sc[0] == L.sc[1] && L.sc[1] == L.sc[2]);      // one coordinate table for R, G and B
    const bool tab = eq && lut_depth <= 10 && (L.pre || !T2_KNOB("LUTR_NO_TAB"));
    const bool unit = L.unit && out_clip_dead(K, 1 << (csx + csy));
    if (fast && tab && unit && L.lat16 && !L.pre) return t2::V_FAST;      // (the fast variant is defined without a prelut: its CPU twin has none)
    if (tab && unit) return t2::V_UNIT;
    if (tab) return t2::V_TAB;
    return t2::V_GEN;
}

const char *T2_ENTRY(hipStream_t st, const LutConsts &L, const YuvConsts &K, const PlaneSet &P, const FrameGeom &G,
                             int din, int dout, int lut_depth, int csx, int csy, int mode, bool fast, unsigned *stats, unsigned *queue)
{
    using namespace t2;
    constexpr int win = LUTR_T2_WI, wout = LUTR_T2_WO;
    if ((din > 8) != win || (dout > 8) != wout || csx != LUTR_T2_X || csy != LUTR_T2_Y) return nullptr;
    const int pxt = win ? 8 : 16;
    const bool pre = K.pre != 0.0f;
    const int v = tile2_variant(L, K, lut_depth, csx, csy, fast);
    if (pre && v != V_UNIT && v != V_FAST) return nullptr;
    if (L.pre && v < V_TAB) return nullptr;                   // a prelut lives in the coordinate table: the general variant computes
    // nearest has no blend to speed up and no prologue instances: strict clip-free kernel, or the round-1 path
    if (mode == LUTR_INTERP_NEAREST && pre) return nullptr;
    const int vv = (mode == LUTR_INTERP_NEAREST && v == V_FAST) ? V_UNIT : v;
    for (int i = 0; i < 3; i++)
        if (P.sfs[i] < 0 || P.dfs[i] < 0 || P.ss[i] < 0 || P.ds[i] < 0 || P.ss[i] >= (1 << 24) || P.ds[i] >= (1 << 24)) return nullptr;

    Geom tg;
    const int uw = G.w / pxt, urows = G.rows >> csy;
    // lanes across x: 16 (x 4 down: 128 x 8 px at 10-bit 4:2:0) unless another shape wastes 2 % fewer lanes at the frame's edges.
    // Compact tiles see fewer colours: with the tube, 16 x 4 measures 625 / 564 Gpx/s fast / strict against 607 / 542 for 32 x 2
    // and 615 / 562 for 8 x 8 (UHD).
    // 4:2:2 / 4:4:4 (one row per unit) are on the memory side and want 512-byte runs per tile row: 32 x 2 lanes there
    // (yuv422p10le 489 vs 341 Gpx/s, yuv444p10le 407 vs 332).
    int best = csy ? 4 : 5;
    double best_eff = -1.0;
    const int order[2][5] = {{5, 4, 6, 3, 2}, {4, 5, 3, 6, 2}};
    for (int l : order[csy ? 1 : 0]) {
        const int lw = 1 << l, lh = 64 >> l;
        const double eff = ((double)uw / (((uw + lw - 1) / lw) * lw)) * ((double)urows / (((urows + lh - 1) / lh) * lh));
        if (eff > best_eff + 0.02) { best_eff = eff; best = l; }
    }
    if (const Knob &e = T2_KNOB("LUTR_LW_LOG2")) { const int c = e.num(); if (c >= 2 && c <= 6) best = c; }
    tg.lw_log2 = best; tg.uw = uw; tg.urows = urows;
    tg.nsx = (uw + (1 << best) - 1) >> best;
    tg.nry = (urows + (64 >> best) - 1) / (64 >> best);
    int waves_per_cu = 16;
    if (const Knob &e = T2_KNOB("LUTR_WAVES_PER_CU")) { const int c = e.num(); if (c >= LUTR_T2_WPB && c <= 32 && c % LUTR_T2_WPB == 0) waves_per_cu = c; }
    const int max_waves = device_cus() * waves_per_cu;
    // Chunk = `ch` consecutive tile rows of a strip.  Round 2's single counter capped the claim rate (4096 waves got ~85 M atomic adds
    // per second out of it: UHD 10-bit throughput was proportional to the chunk height up to 7 tiles -- 348 / 434 / 515 / 585 Gpx/s at
    // 4 / 5 / 6 / 7 -- and flat from 8), so chunks were at least 8192 pixels.  With the two-level queue (claim_chunk: one global atomic
    // per 16 claims) small chunks are free -- 256 UHD frames at 8 / 4 / 2 tiles: 581 / 582 / 576 Gpx/s strict -- and they are what a
    // short launch needs for its tail: 8 frames 347 / 361 / 436, 16 frames 438 / 490 / 501, 32 frames 518 / 535 / 536
    // (profiles/r03_exp18_two_level_queue.txt).  4096 pixels per claim, 2048 when a wave gets fewer than 64 tiles.
    const int tile_px = pxt * (1 << csy) * 64;
    int ch = (4096 + tile_px - 1) / tile_px;
    if ((long long)G.nframes * tg.nsx * tg.nry < 64ll * max_waves) ch = (2048 + tile_px - 1) / tile_px;
    if (!LUTR_T2_QUEUE2) { ch = 32 / (64 >> best); if (ch * tile_px < 8192) ch = (8192 + tile_px - 1) / tile_px; }
    if (ch < 1) ch = 1;
    if (const Knob &e = T2_KNOB("LUTR_CHUNK")) { const int c = e.num(); if (c >= 1 && c <= 256) ch = c; }
    while (ch > 1 && (long long)G.nframes * tg.nsx * ((tg.nry + ch - 1) / ch) < max_waves / 4) ch >>= 1;
    tg.ch = ch; tg.nrc = (tg.nry + ch - 1) / ch; tg.nchunks = G.nframes * tg.nrc * tg.nsx;
    tg.tab_entries = vv >= V_TAB ? table_entries(K, din) : 0;
    tg.max_raw = (1 << din) - 1;
    const int node_ = vv == V_FAST ? ((LUTR_T2_TRIREC && mode == LUTR_INTERP_TRILINEAR) ? 12 : 8)
                                   : ((mode == LUTR_INTERP_TRILINEAR || LUTR_T2_NODE16) ? 16 : 12);
    const int blocks_per_cu = waves_per_cu / LUTR_T2_WPB > 0 ? waves_per_cu / LUTR_T2_WPB : 1;
    const int lds_block = 163840 / blocks_per_cu - tg.tab_entries * 8 - t2::kScratch;
    // whole-lattice mode: (N+1)^3 nodes behind the table in one workgroup's LDS, row and plane strides padded against bank conflicts.
    // The strict tetrahedral kernels stage float4 nodes there when those fit too (T2_TET16; N <= 19 at 10 bit).
    int node = node_;
    long long whole_bytes = 0;
    bool whole16 = false;
    tg.whole = 0; tg.whole_a = L.n1 * L.n1; tg.whole_b = L.n1;
    if (blocks_per_cu == 1 && !T2_KNOB("LUTR_NO_WHOLE")) {
        if (node_ == 12 && mode == LUTR_INTERP_TETRAHEDRAL && vv != V_GEN && !T2_KNOB("LUTR_NO_WHOLE16")) {
            int a, b;
            const long long bytes = whole_strides(L.n1, 16, lds_block, &a, &b);
            if (bytes) { tg.whole = 1; tg.whole_a = a; tg.whole_b = b; whole_bytes = bytes; whole16 = true; node = 16; }
        }
        if (!tg.whole) {
            int a, b;
            const long long bytes = whole_strides(L.n1, node_, lds_block, &a, &b);
            if (bytes) { tg.whole = 1; tg.whole_a = a; tg.whole_b = b; whole_bytes = bytes; }
        }
    }
    // The grey tube (Geom::tube_h): all of r, |g - r| and |b - r| up to H cells.  Needs the table variants (equal channel scales: the
    // chroma-only bound of map_box), a blend (nearest rounds to a node, the bound is for floor), and room left for windows.
    // H: as wide as 70 % of the block's LDS allows while every wave keeps a window of 256 nodes, at most 8 cells of a 33^3 lattice
    // (+-64 8-bit codes of G-R and B-R; 34 x 19 x 19 fp16 nodes = 98 KB, windows of 367 nodes).  Measured, UHD yuv420p10le fast,
    // H = 5 / 7 / 8 / 9: natural frames 611 / 620 / 628 / 636 Gpx/s, three times the chroma 516 / 521 / 530 / 519, sigma = 8 noise
    // 503 / 543 / 550 / 569, sigma = 16 250 / 388 / 481 / 500: the tube, not the windows, is what carries natural content; 9 leaves
    // windows of 197 nodes and starts to cost saturated frames.  The strict kernels (12-byte nodes) get H = 7 (below).
    tg.tube_h = 0; tg.tube_t = 0.0f; tg.tube_plane = 0;
    long long tube_bytes = 0;
    // (65^3: a tube that fits is 5 of ITS cells wide, +-20 8-bit codes -- 497 / 427 Gpx/s with it, 530 / 492 without: off above 40^3)
    if (!tg.whole && vv >= V_TAB && mode != LUTR_INTERP_NEAREST && (L.n1 <= 41 || T2_KNOB("LUTR_TUBE_H"))) {
        int h = (8 * (L.n1 - 2) + 16) / 32;                       // 8 at 33^3
        // a short launch (under ~100 tiles per wave) does not earn back the ~8 us it takes to stage the widest tube: 5 cells there
        // (UHD, 8 / 16 / 32 / 64 frames per launch, Gpx/s at H = 8 | 5: 351 | 369, 445 | 456, 533 | 541, 598 | 595)
        if ((long long)G.nframes * tg.nsx * tg.nry < 100ll * max_waves) h = (5 * (L.n1 - 2) + 16) / 32;
        if (const Knob &e = T2_KNOB("LUTR_TUBE_H")) h = e.num();
        const float kappa = L.pre ? fmaxf(L.pre_kappa, 1e-6f) : L.sc[0] * L.scale_f;   // cells per code, at most (a shared prelut: its largest step)
        const float eps = K.max_l * (1.0f / 2097152.0f) + 1e-3f;                 // as map_box
        const float slack = 1.0f + 2.0f * L.lut_max * (1.0f / 2097152.0f) + 2e-3f;
        // fast (8-byte nodes): at most 70 % of the block's LDS and windows of >= 256 nodes (H = 8 / 367 at 33^3).  The strict 4-tap
        // kernels' 12-byte nodes do not fit that at H = 7; they trade window size for tube width (H = 7, 16 windows of 139 nodes
        // instead of H = 6 / 277): natural +2 %, sigma-8 +0 %, sigma-16 +43 %, three times the chroma -7 % (profiles/r03_exp4.txt)
        int min_win = node == 12 ? 128 : 256, tube_pct = node == 12 ? 85 : 70;
        if (const Knob &e = T2_KNOB("LUTR_MIN_WIN")) { const int c = e.num(); if (c >= 128 && c <= 4096) min_win = c; }
        if (const Knob &e = T2_KNOB("LUTR_TUBE_PCT")) { const int c = e.num(); if (c >= 10 && c <= 95) tube_pct = c; }
        while (h >= 3) {
            const long long nb = 2 * h + 3, plane = T2_KNOB("LUTR_TUBE_NOPAD") ? nb * nb : tube_plane_stride((int)nb, node);
            const long long bytes = (long long)L.n1 * plane * node;
            const bool fits = bytes <= (long long)lds_block * tube_pct / 100 && (lds_block - bytes) / (node * LUTR_T2_WPB) >= min_win;
            float t = ((float)(h + 1) - slack) / kappa - 1.0f - eps;
            if (L.pre) t = fits ? prelut_tube_bound(L.pre_host, (int)L.maxf, h) - 0.5f - eps : 0.0f;      // (a scan of the curve: memoised)
            if (fits && t > 0.0f) {
                tg.tube_h = h; tg.tube_t = t; tg.tube_plane = (int)plane; tube_bytes = bytes;
                break;
            }
            h--;
        }
    }
    // 63: whenever at least one lane is inside the tube; a tile with all 64 lanes outside goes to the wave's window as before.
    // Measured for the strict kernels (64 UHD frames, Gpx/s, profiles/r03_exp16_mixed_tiles_cheap_restage.txt): sigma-16 frames
    // 409 without mixed tiles, 423 / 426 / 428 / 435 at 8 / 16 / 32 / 63 lanes; sigma-8 498 -> 510; natural and saturated frames +-1 %.
    // (profiles/r03_exp7_mixed_tiles.txt shows 286 -> 402: that build still paid 30 us of stride search per restage.)
    tg.mix_max = 63;
    if (const Knob &e = T2_KNOB("LUTR_MIX_MAX")) { const int c = e.num(); if (c >= 0 && c <= 63) tg.mix_max = c; }
    tg.tube_rlo = 0xffffffffu; tg.tube_rhi = 0u;
    if (tg.tube_h > 0) {
        // the square |cb' - coff|, |cr' - coff| <= R (after the prologue) inside the tube's chroma region: both differences are linear in
        // the two offsets, so their worst corners bound them; 1 % and one code of margin absorb the float rounding of the per-lane form
        const float s1 = fabsf(K.kgu) + fabsf(K.kgv - K.krv);
        const float s2 = LUTR_T2_TUBE_BG ? fabsf(K.kbu - K.kgu) + fabsf(K.kgv) : fabsf(K.kbu) + fabsf(K.krv);
        const float R = floorf(0.99f * tg.tube_t / fmaxf(s1, s2)) - 1.0f;
        // (one entry of memory: a stream of applies repeats the same constants, and the scan is up to 65,536 codes long)
        struct Memo { float pre, pc, pcb, pre_max, coff, R; int max_raw, lo, hi; };
        static thread_local Memo memo = {0, 0, 0, 0, 0, -1.0f, -1, -1, -1};
        int lo = -1, hi = -1;
        if (memo.pre == K.pre && memo.pc == K.pc && memo.pcb == K.pcb && memo.pre_max == K.pre_max && memo.coff == K.coff &&
            memo.R == R && memo.max_raw == tg.max_raw) { lo = memo.lo; hi = memo.hi; }
        else {
            for (int raw = 0; raw <= tg.max_raw; raw++) {           // the prologue is a monotone map of raw codes
                float c = (float)raw;
                if (K.pre != 0.0f) c = fminf(fmaxf(floorf(fmaf(K.pc, c, K.pcb)), 0.0f), K.pre_max);
                if (fabsf(c - K.coff) <= R) { if (lo < 0) lo = raw; hi = raw; }
            }
            memo = Memo{K.pre, K.pc, K.pcb, K.pre_max, K.coff, R, tg.max_raw, lo, hi};
        }
        if (R >= 1.0f && lo >= 0) {
            auto rep = [&](unsigned v16) { return v16 | (v16 << 16); };
            tg.tube_rlo = win ? rep((unsigned)lo) : rep((unsigned)lo << 8);
            tg.tube_rhi = win ? rep((unsigned)hi) : rep(((unsigned)hi << 8) | 0xffu);
        }
    }
    int cap = (int)((lds_block - tube_bytes) / (node * LUTR_T2_WPB));
    if (const Knob &e = T2_KNOB("LUTR_WIN_NODES")) { const int c = e.num(); if (c >= 64 && c < cap) cap = c; }
    if (!tg.whole && cap < 128) return nullptr;
    tg.win_nodes = tg.whole ? 0 : cap;
    tg.queue = queue; tg.stats = stats;
    const int waves = tg.nchunks < max_waves ? tg.nchunks : max_waves;
    const dim3 grid((waves + LUTR_T2_WPB - 1) / LUTR_T2_WPB), block(64 * LUTR_T2_WPB);
    tg.qbase = grid.x * LUTR_T2_WPB;      // (the counter is at zero: the previous launch left it so)
    const size_t lds = (size_t)tg.tab_entries * 8 + kScratch +
                       (tg.whole ? (size_t)whole_bytes : (size_t)tube_bytes + (size_t)LUTR_T2_WPB * tg.win_nodes * node);
    Planes2 TP;
    for (int i = 0; i < 3; i++) {
        TP.s[i] = P.s[i]; TP.d[i] = P.d[i];
        TP.ss[i] = (unsigned)P.ss[i]; TP.ds[i] = (unsigned)P.ds[i];
        TP.sfs[i] = (unsigned long long)P.sfs[i]; TP.dfs[i] = (unsigned long long)P.dfs[i];
    }
    if (T2_KNOB("LUTR_DEBUG"))
        fprintf(stderr, "[lutr t2] nsx %d nry %d chunk %d chunks %d blocks %u lds/block %zu win_nodes %d tab %d variant %d tube h %d plane %d t %.1f\n",
                tg.nsx, tg.nry, tg.ch, tg.nchunks, grid.x, lds, tg.win_nodes, tg.tab_entries, vv, tg.tube_h, tg.tube_plane, tg.tube_t);

#define T2_LAUNCH(WI, WO, X, Y, I, PR, VV, NAME) \
    do { \
        auto kern = k_yuv_tile2<WI, WO, X, Y, I, PR, VV>; \
        if (!allow_lds((const void *)kern, lds)) return nullptr; \
        hipLaunchKernelGGL(kern, grid, block, lds, st, L, K, TP, G, tg); \
        return tg.whole ? NAME "+whole-lattice" : (tg.tube_h ? NAME "+tube" : NAME); \
    } while (0)
#define T2_NAME(WI, WO, X, Y, I, SUF) "k_yuv_tile2<" #WI #WO "," #X #Y "," #I SUF ">"
#define T2_CASE(WI, WO, X, Y, I) \
    if (mode == I) { \
        if (I == LUTR_INTERP_TETRAHEDRAL && whole16) {      /* (other I: the template argument falls back to an instance that exists anyway) */ \
            if (pre) T2_LAUNCH(WI, WO, X, Y, (I == 2 ? T2_TET16 : 2), 1, V_UNIT, T2_NAME(WI, WO, X, Y, I, ",pre,tab,unit,n16")); \
            if (vv == V_UNIT) T2_LAUNCH(WI, WO, X, Y, (I == 2 ? T2_TET16 : 2), 0, V_UNIT, T2_NAME(WI, WO, X, Y, I, ",tab,unit,n16")); \
            if (vv == V_TAB) T2_LAUNCH(WI, WO, X, Y, (I == 2 ? T2_TET16 : 2), 0, V_TAB, T2_NAME(WI, WO, X, Y, I, ",tab,n16")); \
        } \
        if (I != LUTR_INTERP_NEAREST) { \
            if (pre && vv == V_FAST) T2_LAUNCH(WI, WO, X, Y, (I == 0 ? 2 : I), 1, V_FAST, T2_NAME(WI, WO, X, Y, I, ",pre,tab,unit,fast")); \
            if (pre) T2_LAUNCH(WI, WO, X, Y, (I == 0 ? 2 : I), 1, V_UNIT, T2_NAME(WI, WO, X, Y, I, ",pre,tab,unit")); \
            if (vv == V_FAST) T2_LAUNCH(WI, WO, X, Y, (I == 0 ? 2 : I), 0, V_FAST, T2_NAME(WI, WO, X, Y, I, ",tab,unit,fast")); \
        } \
        if (vv == V_UNIT) T2_LAUNCH(WI, WO, X, Y, I, 0, V_UNIT, T2_NAME(WI, WO, X, Y, I, ",tab,unit")); \
        if (vv == V_TAB) T2_LAUNCH(WI, WO, X, Y, I, 0, V_TAB, T2_NAME(WI, WO, X, Y, I, ",tab")); \
        T2_LAUNCH(WI, WO, X, Y, I, 0, V_GEN, T2_NAME(WI, WO, X, Y, I, "")); \
    }
#if LUTR_T2_ONLY
    T2_CASE(LUTR_T2_WI, LUTR_T2_WO, LUTR_T2_X, LUTR_T2_Y, 2)
#else
    T2_CASE(LUTR_T2_WI, LUTR_T2_WO, LUTR_T2_X, LUTR_T2_Y, 0)
    T2_CASE(LUTR_T2_WI, LUTR_T2_WO, LUTR_T2_X, LUTR_T2_Y, 1)
    T2_CASE(LUTR_T2_WI, LUTR_T2_WO, LUTR_T2_X, LUTR_T2_Y, 2)
#endif
    return nullptr;
}

}  // namespace lutr
