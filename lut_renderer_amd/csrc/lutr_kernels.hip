// lutr_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the 3D-LUT apply engine.
//
// What they replace: the per-pixel loops of FFmpeg's lut3d filter (and the scalers
// around it) that the reference invokes through
//   /root/reference/src/lut_renderer/ffmpeg.py:246   lut3d=file=...:interp=...
//   /root/reference/src/lut_renderer/ffmpeg.py:212-236, :304-310  scale= / format=
// Semantics: SURVEY.md Appendix A (lut3d) and DESIGN.md "YUV contract".
//
// Compiled with -ffp-contract=off: every product and sum of the lut3d restatement
// rounds separately, in FFmpeg's scalar C order, so results are bit-identical to the
// CPU oracle.  Fused multiply-adds appear only where written as __builtin_fmaf.
//
// This is gather + lerp: no MFMA.  The bound is HBM (6 B/px for yuv420p10le in+out)
// and, before that, VALU issue; see DESIGN.md "Kernels".
#include <cstdlib>

#include "lutr_device.h"

namespace lutr {

// ================================================================= generic kernels
// Any depth 8..16, any strides/alignment, odd sizes, all five modes.  One thread per
// pixel (RGB) or per chroma block (YUV).  This is the ragged-input path, not the fast one.

__global__ __launch_bounds__(256) void k_rgb_generic(LutConsts L, PlaneSet P, FrameGeom G, int wide, int mode)
{
    const GFetch f(L);
    const long long total = (long long)G.w * G.rows * G.nframes;
    for (long long u = blockIdx.x * 256ll + threadIdx.x; u < total; u += (long long)gridDim.x * 256ll) {
        const int x = (int)(u % G.w);
        const long long t = u / G.w;
        const int y = G.row0 + (int)(t % G.rows);
        const long long fr = t / G.rows;
        const float g = ld_sample(P.s[0] + fr * P.sfs[0] + y * P.ss[0], x, wide);
        const float b = ld_sample(P.s[1] + fr * P.sfs[1] + y * P.ss[1], x, wide);
        const float r = ld_sample(P.s[2] + fr * P.sfs[2] + y * P.ss[2], x, wide);
        const Rgb o = lut3d_px_rt(mode, L, f, r, g, b);
        st_sample(P.d[0] + fr * P.dfs[0] + y * P.ds[0], x, wide, o.g);
        st_sample(P.d[1] + fr * P.dfs[1] + y * P.ds[1], x, wide, o.b);
        st_sample(P.d[2] + fr * P.dfs[2] + y * P.ds[2], x, wide, o.r);
    }
}

__global__ __launch_bounds__(256) void k_yuv_generic(LutConsts L, YuvConsts K, PlaneSet P, FrameGeom G,
                                                     int win, int wout, int csx, int csy, int mode)
{
    const GFetch f(L);
    const int bw = 1 << csx, bh = 1 << csy;
    const int cw = (G.w + bw - 1) >> csx;             // chroma blocks per row
    const int cr0 = G.row0 >> csy;
    const int crows = ((G.row0 + G.rows + bh - 1) >> csy) - cr0;
    const long long total = (long long)cw * crows * G.nframes;
    for (long long u = blockIdx.x * 256ll + threadIdx.x; u < total; u += (long long)gridDim.x * 256ll) {
        const int cx = (int)(u % cw);
        const long long t = u / cw;
        const int cy = cr0 + (int)(t % crows);
        const long long fr = t / crows;
        const float cbv = ld_sample(P.s[1] + fr * P.sfs[1] + cy * P.ss[1], cx, win);
        const float crv = ld_sample(P.s[2] + fr * P.sfs[2] + cy * P.ss[2], cx, win);
        const Chroma c = chroma_terms(K, cbv, crv);
        float rs = 0.f, gs = 0.f, bs = 0.f;
        for (int dy = 0; dy < bh; dy++) {
            const int yy = cy * bh + dy;
            const int y = yy < G.h ? yy : G.h - 1;    // odd height: replicate the edge row into the block
            for (int dx = 0; dx < bw; dx++) {
                const int xx = cx * bw + dx;
                const int x = xx < G.w ? xx : G.w - 1;
                const float yv = ld_sample(P.s[0] + fr * P.sfs[0] + y * P.ss[0], x, win);
                const Rgb q = yuv_to_rgb(K, yv, c);
                const Rgb o = lut3d_px_rt(mode, L, f, q.r, q.g, q.b);
                rs += o.r; gs += o.g; bs += o.b;
                if (yy < G.h && xx < G.w)
                    st_sample(P.d[0] + fr * P.dfs[0] + y * P.ds[0], x, wout, rgb_to_y(K, o));
            }
        }
        st_sample(P.d[1] + fr * P.dfs[1] + cy * P.ds[1], cx, wout, rgb_to_cb(K, rs, gs, bs));
        st_sample(P.d[2] + fr * P.dfs[2] + cy * P.ds[2], cx, wout, rgb_to_cr(K, rs, gs, bs));
    }
}

// ================================================================= vector kernels, global gather
// The low-latency path of small launches (and the A/B reference of the LDS window).  kVecBytes bytes per
// plane row and thread: 4 px (16-bit containers) or 8 px (8-bit).  With 16-byte accesses these kernels needed
// 256 VGPRs (one wave per SIMD); at 8 bytes they keep several waves per SIMD, which is what a gather wants.
constexpr int kVecBytes = 8;

template <int WIDE, int INTERP>
__global__ __launch_bounds__(256) void k_rgb_vec(LutConsts L, PlaneSet P, FrameGeom G)
{
    constexpr int PXT = kVecBytes / (WIDE ? 2 : 1), NW = kVecBytes / 4;
    const GFetch f(L);
    const unsigned uw = (unsigned)G.w / PXT;
    const unsigned total = uw * (unsigned)G.rows * (unsigned)G.nframes;
    const unsigned u = blockIdx.x * 256u + threadIdx.x;
    if (u >= total) return;
    const unsigned xu = u % uw, t = u / uw;
    const int y = G.row0 + (int)(t % (unsigned)G.rows);
    const long long fr = t / (unsigned)G.rows;
    const long long xo = (long long)xu * kVecBytes;
    uint32_t gw[NW], bw[NW], rw[NW], go[NW], bo[NW], ro[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) { go[k] = 0; bo[k] = 0; ro[k] = 0; }
    ld_words<NW>(gw, P.s[0] + fr * P.sfs[0] + y * P.ss[0] + xo);
    ld_words<NW>(bw, P.s[1] + fr * P.sfs[1] + y * P.ss[1] + xo);
    ld_words<NW>(rw, P.s[2] + fr * P.sfs[2] + y * P.ss[2] + xo);
#pragma unroll
    for (int i = 0; i < PXT; i++) {
        const Rgb o = lut3d_px<INTERP>(L, f, word_sample<WIDE>(rw, i), word_sample<WIDE>(gw, i),
                                       word_sample<WIDE>(bw, i));
        word_put<WIDE>(go, i, o.g);
        word_put<WIDE>(bo, i, o.b);
        word_put<WIDE>(ro, i, o.r);
    }
    st_words<NW, true>(P.d[0] + fr * P.dfs[0] + y * P.ds[0] + xo, go);
    st_words<NW, true>(P.d[1] + fr * P.dfs[1] + y * P.ds[1] + xo, bo);
    st_words<NW, true>(P.d[2] + fr * P.dfs[2] + y * P.ds[2] + xo, ro);
}

// WIN / WOUT: 16-bit containers in / out.  A 10-bit source written as 8 bit (the reference's libx264 default,
// ffmpeg.py:287-302) takes 16 bytes of luma per thread and row so that its 8-bit chroma output is still a whole word.
template <int WIN, int WOUT> constexpr int vec_bytes() { return (WIN && !WOUT) ? 16 : kVecBytes; }

template <int WIN, int WOUT, int CSX, int CSY, int INTERP>
__global__ __launch_bounds__(256) void k_yuv_vec(LutConsts L, YuvConsts K, PlaneSet P, FrameGeom G)
{
    constexpr int VB = vec_bytes<WIN, WOUT>();
    constexpr int PXT = VB / (WIN ? 2 : 1);               // luma samples per thread per row
    constexpr int YWI = VB / 4, YWO = PXT * (WOUT ? 2 : 1) / 4;   // luma words per thread per row, in / out
    constexpr int BH = 1 << CSY, BW = 1 << CSX;
    constexpr int NC = PXT >> CSX;                        // chroma samples per thread
    constexpr int CWI = NC * (WIN ? 2 : 1) / 4, CWO = NC * (WOUT ? 2 : 1) / 4;
    static_assert(CWI >= 1 && CWO >= 1 && YWO >= 1, "a thread must own whole words");
    const GFetch f(L);
    const unsigned uw = (unsigned)G.w / PXT;
    const unsigned ub = (unsigned)G.rows >> CSY;
    const unsigned total = uw * ub * (unsigned)G.nframes;
    const unsigned u = blockIdx.x * 256u + threadIdx.x;
    if (u >= total) return;
    const unsigned xu = u % uw, t = u / uw;
    const int cy = (G.row0 >> CSY) + (int)(t % ub);
    const long long fr = t / ub;
    const long long xi = (long long)xu * VB, xo = (long long)xu * (YWO * 4), cxi = (long long)xu * (CWI * 4), cxo = (long long)xu * (CWO * 4);

    uint32_t yw[BH][YWI], cbw[CWI], crw[CWI];
    uint32_t yo[BH][YWO], cbo[CWO], cro[CWO];
#pragma unroll
    for (int dy = 0; dy < BH; dy++) {
        ld_words<YWI>(yw[dy], P.s[0] + fr * P.sfs[0] + (long long)(cy * BH + dy) * P.ss[0] + xi);
#pragma unroll
        for (int k = 0; k < YWO; k++) yo[dy][k] = 0;
    }
    ld_words<CWI>(cbw, P.s[1] + fr * P.sfs[1] + (long long)cy * P.ss[1] + cxi);
    ld_words<CWI>(crw, P.s[2] + fr * P.sfs[2] + (long long)cy * P.ss[2] + cxi);
#pragma unroll
    for (int k = 0; k < CWO; k++) { cbo[k] = 0; cro[k] = 0; }

#pragma unroll
    for (int j = 0; j < NC; j++) {
        const Chroma c = chroma_terms(K, word_sample<WIN>(cbw, j), word_sample<WIN>(crw, j));
        float rs = 0.f, gs = 0.f, bs = 0.f;
#pragma unroll
        for (int dy = 0; dy < BH; dy++) {
#pragma unroll
            for (int dx = 0; dx < BW; dx++) {
                const int i = j * BW + dx;
                const Rgb q = yuv_to_rgb(K, word_sample<WIN>(yw[dy], i), c);
                const Rgb o = lut3d_px<INTERP>(L, f, q.r, q.g, q.b);
                rs += o.r; gs += o.g; bs += o.b;
                word_put<WOUT>(yo[dy], i, rgb_to_y(K, o));
            }
        }
        word_put<WOUT>(cbo, j, rgb_to_cb(K, rs, gs, bs));
        word_put<WOUT>(cro, j, rgb_to_cr(K, rs, gs, bs));
        // Zero-instruction fence: without it hipcc hoists the coordinates and taps of every chroma block of the
        // thread to the top (256 VGPRs, one wave per SIMD); with it blocks are emitted one after the other.
#pragma unroll
        for (int dy = 0; dy < BH; dy++) {
#pragma unroll
            for (int k = 0; k < YWI; k++) asm volatile("" : "+v"(yw[dy][k]));
#pragma unroll
            for (int k = 0; k < YWO; k++) asm volatile("" : "+v"(yo[dy][k]));
        }
#pragma unroll
        for (int k = 0; k < CWI; k++) asm volatile("" : "+v"(cbw[k]), "+v"(crw[k]));
#pragma unroll
        for (int k = 0; k < CWO; k++) asm volatile("" : "+v"(cbo[k]), "+v"(cro[k]));
    }
#pragma unroll
    for (int dy = 0; dy < BH; dy++)
        st_words<YWO>(P.d[0] + fr * P.dfs[0] + (long long)(cy * BH + dy) * P.ds[0] + xo, yo[dy]);
    st_words<CWO>(P.d[1] + fr * P.dfs[1] + (long long)cy * P.ds[1] + cxo, cbo);
    st_words<CWO>(P.d[2] + fr * P.dfs[2] + (long long)cy * P.ds[2] + cxo, cro);
}

// ================================================================= launchers
static inline bool aligned_to(const void *p, long long a) { return ((uintptr_t)p % (uintptr_t)a) == 0; }

static bool planes_aligned(const PlaneSet &P, int plane, long long a, bool batch)
{
    // the fast kernels address rows with 32-bit positive offsets; bottom-up (negative linesize) or
    // huge strides go to the generic kernel, which does 64-bit signed arithmetic.  A tile kernel forms
    // (row in tile) * stride + 16 * (unit in row) as one unsigned 32-bit offset with up to 31 rows and 63 units.
    constexpr long long kMaxStride = (0xffffffffll - 64 * 16) / 32;
    if (P.ss[plane] <= 0 || P.ds[plane] <= 0 || P.ss[plane] > kMaxStride || P.ds[plane] > kMaxStride) return false;
    if (!aligned_to(P.s[plane], a) || !aligned_to(P.d[plane], a)) return false;
    if (P.ss[plane] % a || P.ds[plane] % a) return false;
    if (batch && (P.sfs[plane] % a || P.dfs[plane] % a)) return false;
    return true;
}

// The persistent tile kernels pay a fixed start-up (coordinate table, tube staging, a wave's first tile at a quarter of the issue
// rate) and end with a tail of partly idle CUs, so they only win on big launches.  Round 3, Gpx/s tile / plain vector kernels (taps
// gathered from L1/L2, 5-8 waves per SIMD; profiles/r03_exp19_small_launches.txt), fused yuv420p10le strict: UHD 1 frame 181 / 272,
// 2 frames 253 / 322, 4 frames 316 / 357, 8 frames 442 / 326, 16 frames 507 / 333, 64 frames 560 / 338; 1080p 8 frames 253 / 326,
// 16 frames 337 / 300, 32 frames 424 / 324, 64 frames 478 / 329 -- the two-level chunk queue and its small chunks moved the
// crossover of the fused kernels from 70 Mpx (round 2) to ~33 Mpx.  The RGB tile kernels keep 70 (their queue is round 1's).
// LUTR_SMALL_JOB_MPX moves both boundaries (0 = never).
static bool small_job(long long px, long long mpx = 70)
{
    if (const char *e = getenv("LUTR_SMALL_JOB_MPX")) { const long long v = atoll(e); if (v >= 0 && v <= 100000) mpx = v; }
    return px < mpx * 1000000ll;
}
constexpr long long kSmallYuvMpx = 33;

static unsigned grid_for(long long units, unsigned cap = 0x7fffffffu)
{
    long long b = (units + 255) / 256;
    if (b < 1) b = 1;
    if (b > (long long)cap) b = cap;
    return (unsigned)b;
}

const char *launch_rgb(hipStream_t st, int variant, const LutConsts &L, const PlaneSet &P,
                       const FrameGeom &G, int depth, int mode, unsigned *stats, unsigned *queue)
{
    // a prelut (.csp shaper): the round-3 tube kernels read one (three coordinate tables, 8- and 10-bit data); the round-1 tile kernel does not
    const bool tiles = L.pre == nullptr;
    const int wide = depth > 8;
    const int pxt = wide ? 8 : 16;
    const long long px = (long long)G.w * G.rows * G.nframes;
    bool vec_ok = (mode == LUTR_INTERP_NEAREST || mode == LUTR_INTERP_TRILINEAR || mode == LUTR_INTERP_TETRAHEDRAL) &&
                  G.w % pxt == 0 && px / pxt < 0x7fffffffll;
    for (int c = 0; c < 3 && vec_ok; c++)
        vec_ok = planes_aligned(P, c, 16, G.nframes > 1);
    if (variant == VAR_GENERIC) vec_ok = false;
    // which LDS kernel would take the launch decides where "small" ends: the tube kernels have the two-level queue (33 Mpx), round 1's kernel not (70)
    const char *pol0 = getenv("LUTR_RGB2");
    const bool tube_first = !((pol0 && pol0[0] == '0') || getenv("LUTR_NO_RGB2")) &&
                            ((pol0 && pol0[0] == 'a') || !tiles || mode != LUTR_INTERP_TRILINEAR);
    if ((tiles || depth <= 10) && vec_ok && ((variant == VAR_AUTO && !small_job(px, tube_first ? kSmallYuvMpx : 70)) || variant == VAR_VEC_LDS)) {
        // round 3: the tube kernels (lutr_rgb2.hip) take the planes in (R, G, B) order; gbrp is (G, B, R)
        // Policy (profiles/r03_exp21_planar_rgb_tube_vs_round1.txt): with the two-level chunk queue the tube kernel wins every 4-tap and
        // nearest launch (gbrp10le tetrahedral 8 / 16 / 128 frames 359 / 412 / 432 vs 282 / 353 / 427 Gpx/s, gbrp 519 / 583 / 637 vs
        // 379 / 447 / 593, gbrp nearest 885 = 0.66 of the peak); trilinear's eight 16-byte taps stay on round 1's kernel, which is
        // 16-20 % faster there at 128 frames.  LUTR_RGB2=all sends everything it can take to the tube kernels, =0 nothing.
        if (tube_first) {
            PlaneSet Q = P;
            const int from[3] = {2, 0, 1};
            for (int k = 0; k < 3; k++) {
                Q.s[k] = P.s[from[k]]; Q.d[k] = P.d[from[k]]; Q.ss[k] = P.ss[from[k]]; Q.ds[k] = P.ds[from[k]];
                Q.sfs[k] = P.sfs[from[k]]; Q.dfs[k] = P.dfs[from[k]];
            }
            const char *name = wide ? launch_rgb_tube_ly1(st, L, Q, G, depth, mode, 0, stats, queue)
                                    : launch_rgb_tube_ly0(st, L, Q, G, depth, mode, 0, stats, queue);
            if (name) return name;
        }
        if (tiles) return launch_rgb_tile(st, L, P, G, depth, mode, stats, queue);
        if (variant == VAR_VEC_LDS) variant = VAR_VEC_GLOBAL;          // a prelut the tube kernels could not take (layout): vector kernels
    }
    if (!tiles && variant == VAR_VEC_LDS) variant = VAR_VEC_GLOBAL;
    // A ragged width on aligned (padded) rows: the fast kernel takes the columns up to the last multiple of
    // its unit, the scalar kernel the few that remain (each pixel is independent, so any split is exact).
    const int wv = G.w / pxt * pxt;
    if (tiles && variant == VAR_AUTO && !vec_ok && wv > 0 && wv < G.w && !small_job(px) &&
        (mode == LUTR_INTERP_NEAREST || mode == LUTR_INTERP_TRILINEAR || mode == LUTR_INTERP_TETRAHEDRAL) &&
        (long long)wv * G.rows * G.nframes / pxt < 0x7fffffffll &&
        planes_aligned(P, 0, 16, G.nframes > 1) && planes_aligned(P, 1, 16, G.nframes > 1) &&
        planes_aligned(P, 2, 16, G.nframes > 1)) {
        FrameGeom Gv = G, Ge = G;
        Gv.w = wv;
        Ge.w = G.w - wv;
        PlaneSet Pe = P;
        for (int c = 0; c < 3; c++) { Pe.s[c] += (long long)wv * (wide ? 2 : 1); Pe.d[c] += (long long)wv * (wide ? 2 : 1); }
        const char *name = launch_rgb_tile(st, L, P, Gv, depth, mode, stats, queue);
        hipLaunchKernelGGL(k_rgb_generic, dim3(grid_for((long long)Ge.w * G.rows * G.nframes, 256 * 64)), dim3(256), 0, st,
                           L, Pe, Ge, wide, mode);
        return name;
    }
    if (!vec_ok) {
        if (variant == VAR_VEC_GLOBAL || variant == VAR_VEC_LDS) return nullptr;
        // grid-stride; enough blocks to fill 256 CUs x 8
        hipLaunchKernelGGL(k_rgb_generic, dim3(grid_for(px, 256 * 64)), dim3(256), 0, st, L, P, G, wide, mode);
        return "k_rgb_generic";
    }
    const dim3 grid(grid_for(px / (kVecBytes / (wide ? 2 : 1)))), block(256);
#define RGB_CASE(W, I) \
    if (wide == W && mode == I) { \
        hipLaunchKernelGGL((k_rgb_vec<W, I>), grid, block, 0, st, L, P, G); \
        return "k_rgb_vec<" #W "," #I ">"; \
    }
    RGB_CASE(0, 0) RGB_CASE(0, 1) RGB_CASE(0, 2)
    RGB_CASE(1, 0) RGB_CASE(1, 1) RGB_CASE(1, 2)
#undef RGB_CASE
    return nullptr;
}

// Round-2 tile kernels (lutr_tile2.hip): input and output depth are independent there, so source planes are checked
// against the input unit (16 bytes of luma, the matching chroma bytes) and destination planes against the output unit.
static bool plane_ok(const uint8_t *p, long long stride, long long fstride, long long a, bool batch)
{
    constexpr long long kMaxStride = (0xffffffffll - 64 * 32) / 32;
    return stride > 0 && stride <= kMaxStride && aligned_to(p, a) && stride % a == 0 && (!batch || (fstride >= 0 && fstride % a == 0));
}

static const char *try_tile2(hipStream_t st, const LutConsts &L, const YuvConsts &K, const PlaneSet &P, const FrameGeom &G,
                             int din, int dout, int lut_depth, int csx, int csy, int mode, bool fast, unsigned *stats, unsigned *queue)
{
    // a prelut: only one that is the same non-decreasing shaper on the three channels (LutConsts::pre_shared) and only up to 10 bit (the
    // coordinate table); anything else stays on the generic / vector kernels
    const char *nop = getenv("LUTR_NO_TILE2_PRELUT");
    if (getenv("LUTR_NO_TILE2") || (L.pre && (!L.pre_shared || lut_depth > 10 || (nop && nop[0] != '0')))) return nullptr;
    const int win = din > 8, wout = dout > 8, pxt = win ? 8 : 16, bh = 1 << csy;
    if (!(mode == LUTR_INTERP_NEAREST || mode == LUTR_INTERP_TRILINEAR || mode == LUTR_INTERP_TETRAHEDRAL)) return nullptr;
    if (G.w % pxt || G.row0 % bh || G.rows % bh || (csx == 0 && csy == 1)) return nullptr;
    if ((long long)(G.w / pxt) * (G.rows >> csy) * G.nframes >= 0x7fffffffll) return nullptr;
    const long long yi = 16, yo = (long long)pxt * (wout ? 2 : 1), ci = (long long)(pxt >> csx) * (win ? 2 : 1),
                    co = (long long)(pxt >> csx) * (wout ? 2 : 1);
    const bool batch = G.nframes > 1;
    if (!plane_ok(P.s[0], P.ss[0], P.sfs[0], yi, batch) || !plane_ok(P.d[0], P.ds[0], P.dfs[0], yo > 16 ? 16 : yo, batch)) return nullptr;
    for (int c = 1; c < 3; c++)
        if (!plane_ok(P.s[c], P.ss[c], P.sfs[c], ci, batch) || !plane_ok(P.d[c], P.ds[c], P.dfs[c], co, batch)) return nullptr;
#define T2_TRY(tag) return launch_yuv_tile2_##tag(st, L, K, P, G, din, dout, lut_depth, csx, csy, mode, fast, stats, queue)
    const int key = win * 1000 + wout * 100 + csx * 10 + csy;
    switch (key) {
    case 11:   T2_TRY(w00_c11);
    case 10:   T2_TRY(w00_c10);
    case 0:    T2_TRY(w00_c00);
    case 1111: T2_TRY(w11_c11);
    case 1110: T2_TRY(w11_c10);
    case 1100: T2_TRY(w11_c00);
    case 1011: T2_TRY(w10_c11);
    case 1010: T2_TRY(w10_c10);
    case 1000: T2_TRY(w10_c00);
    }
#undef T2_TRY
    return nullptr;
}

const char *launch_yuv(hipStream_t st, int variant, const LutConsts &L, const YuvConsts &K,
                       const PlaneSet &P, const FrameGeom &G, int din, int dout, int lut_depth, int csx, int csy, int mode,
                       bool fast, unsigned *stats, unsigned *queue)
{
    const int win = din > 8, wout = dout > 8;
    const int pxt = win ? 8 : 16;
    const int bh = 1 << csy;
    if (L.pre && !(L.pre_shared && lut_depth <= 10) && variant == VAR_VEC_LDS) variant = VAR_VEC_GLOBAL;   // a prelut the tile kernels cannot take
    if (variant == VAR_VEC_LDS || (variant == VAR_AUTO && !small_job((long long)G.w * G.rows * G.nframes, kSmallYuvMpx))) {
        if (const char *name = try_tile2(st, L, K, P, G, din, dout, lut_depth, csx, csy, mode, fast, stats, queue)) return name;
        if (variant == VAR_VEC_LDS) return nullptr;          // asked for the tile kernels, and they cannot take this call
    }
    const long long cbytes = (long long)(pxt >> csx) * (win ? 2 : 1);      // chroma bytes per thread
    bool vec_ok = (mode == LUTR_INTERP_NEAREST || mode == LUTR_INTERP_TRILINEAR || mode == LUTR_INTERP_TETRAHEDRAL) &&
                  win == wout && G.w % pxt == 0 && G.row0 % bh == 0 && G.rows % bh == 0 &&
                  !(csx == 0 && csy == 1) &&
                  (long long)(G.w / pxt) * (G.rows >> csy) * G.nframes < 0x7fffffffll;
    if (vec_ok) vec_ok = planes_aligned(P, 0, 16, G.nframes > 1) && planes_aligned(P, 1, cbytes, G.nframes > 1) &&
                         planes_aligned(P, 2, cbytes, G.nframes > 1);
    if (variant == VAR_GENERIC) vec_ok = false;
    // ragged width on aligned (padded) rows: tile kernel up to the last whole unit, scalar kernel for the rest
    // (the split falls on a chroma-block boundary: the unit is 8 or 16 luma samples wide)
    const int wv = G.w / pxt * pxt;
    if (variant == VAR_AUTO && wv > 0 && wv < G.w && !small_job((long long)G.w * G.rows * G.nframes, kSmallYuvMpx)) {
        FrameGeom Gv = G, Ge = G;
        Gv.w = wv;
        Ge.w = G.w - wv;
        if (const char *name = try_tile2(st, L, K, P, Gv, din, dout, lut_depth, csx, csy, mode, fast, stats, queue)) {
            PlaneSet Pe = P;
            const long long bsi = win ? 2 : 1, bso = wout ? 2 : 1;
            Pe.s[0] += wv * bsi; Pe.d[0] += wv * bso;
            for (int c = 1; c < 3; c++) { Pe.s[c] += (wv >> csx) * bsi; Pe.d[c] += (wv >> csx) * bso; }
            const long long eb = (long long)((Ge.w + (1 << csx) - 1) >> csx) * ((G.rows + bh - 1) >> csy) * G.nframes;
            hipLaunchKernelGGL(k_yuv_generic, dim3(grid_for(eb, 256 * 64)), dim3(256), 0, st, L, K, Pe, Ge, win, wout,
                               csx, csy, mode);
            return name;
        }
    }
    // 10-bit (or deeper) source written as 8 bit: the vector kernel with 16-byte luma units
    if (!vec_ok && variant != VAR_GENERIC && win == 1 && wout == 0 &&
        (mode == LUTR_INTERP_NEAREST || mode == LUTR_INTERP_TRILINEAR || mode == LUTR_INTERP_TETRAHEDRAL) &&
        G.w % 8 == 0 && G.row0 % bh == 0 && G.rows % bh == 0 && !(csx == 0 && csy == 1) &&
        (long long)(G.w / 8) * (G.rows >> csy) * G.nframes < 0x7fffffffll &&
        plane_ok(P.s[0], P.ss[0], P.sfs[0], 16, G.nframes > 1) && plane_ok(P.d[0], P.ds[0], P.dfs[0], 8, G.nframes > 1) &&
        plane_ok(P.s[1], P.ss[1], P.sfs[1], 16 >> csx, G.nframes > 1) && plane_ok(P.d[1], P.ds[1], P.dfs[1], 8 >> csx, G.nframes > 1) &&
        plane_ok(P.s[2], P.ss[2], P.sfs[2], 16 >> csx, G.nframes > 1) && plane_ok(P.d[2], P.ds[2], P.dfs[2], 8 >> csx, G.nframes > 1)) {
        const dim3 grid(grid_for((long long)(G.w / 8) * (G.rows >> csy) * G.nframes)), block(256);
#define YUV10_CASE(X, Y, I) \
        if (csx == X && csy == Y && mode == I) { \
            hipLaunchKernelGGL((k_yuv_vec<1, 0, X, Y, I>), grid, block, 0, st, L, K, P, G); \
            return "k_yuv_vec<10," #X "," #Y "," #I ">"; \
        }
#define YUV10_FMT(X, Y) YUV10_CASE(X, Y, 0) YUV10_CASE(X, Y, 1) YUV10_CASE(X, Y, 2)
        YUV10_FMT(1, 1) YUV10_FMT(1, 0) YUV10_FMT(0, 0)
#undef YUV10_FMT
#undef YUV10_CASE
    }
    if (!vec_ok) {
        if (variant == VAR_VEC_GLOBAL || variant == VAR_VEC_LDS) return nullptr;
        const long long blocks = (long long)((G.w + (1 << csx) - 1) >> csx) * ((G.rows + bh - 1) >> csy) * G.nframes;
        hipLaunchKernelGGL(k_yuv_generic, dim3(grid_for(blocks, 256 * 64)), dim3(256), 0, st, L, K, P, G, win, wout,
                           csx, csy, mode);
        return "k_yuv_generic";
    }
    const long long units = (long long)(G.w / (kVecBytes / (win ? 2 : 1))) * (G.rows >> csy) * G.nframes;
    const dim3 grid(grid_for(units)), block(256);
#define YUV_CASE(W, X, Y, I) \
    if (win == W && csx == X && csy == Y && mode == I) { \
        hipLaunchKernelGGL((k_yuv_vec<W, W, X, Y, I>), grid, block, 0, st, L, K, P, G); \
        return "k_yuv_vec<" #W "," #X "," #Y "," #I ">"; \
    }
#define YUV_FMT(W, X, Y) YUV_CASE(W, X, Y, 0) YUV_CASE(W, X, Y, 1) YUV_CASE(W, X, Y, 2)
    YUV_FMT(0, 1, 1) YUV_FMT(0, 1, 0) YUV_FMT(0, 0, 0)
    YUV_FMT(1, 1, 1) YUV_FMT(1, 1, 0) YUV_FMT(1, 0, 0)
#undef YUV_FMT
#undef YUV_CASE
    return nullptr;
}

}  // namespace lutr
