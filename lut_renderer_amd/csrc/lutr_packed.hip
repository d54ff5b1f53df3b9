// lutr_packed.hip -- lut3d on packed RGB (rgb24/bgr24, rgba/bgra/argb/abgr and the *0 variants,
// rgb48le/bgr48le, rgba64le/bgra64le): the interleaved formats FFmpeg's lut3d filter accepts
// next to the planar gbrp family (SURVEY.md A.3 "Supported pixel formats", 8f rank 2).
//
// What it replaces: the packed branch of the filter the reference reaches through
//   /root/reference/src/lut_renderer/ffmpeg.py:246   lut3d=file=...:interp=...
// when the frames it is given are packed RGB (e.g. a decoded image sequence).
//
// Per pixel exactly A.3: code * (1/M) -> clip(x * scale * (N-1)) -> interp -> (int)(v * M)
// clipped to [0, M], M = 255 or 65535; the fourth component (alpha or padding) is carried
// over from the source untouched, as FFmpeg does.
//
// Memory: 4 pixels per lane as whole dwords (12/16/24/32 bytes, contiguous across the
// wave), lattice taps gathered from L2.  HBM-bound in principle (6 B/px rgb24); the
// gather keeps it at the L2 rate, like k_rgb_vec.
#include "lutr_device.h"

namespace lutr {

// component `off` (runtime, wave-uniform) of a 4-component pixel held in one or two dwords
template <int WIDE>
__device__ __forceinline__ float comp4(const uint32_t *w, int i, int off)
{
    if constexpr (WIDE) {
        const unsigned long long pv = ((unsigned long long)w[2 * i + 1] << 32) | w[2 * i];
        return (float)(unsigned)((pv >> (off * 16)) & 0xffffull);
    } else {
        return (float)((w[i] >> (off * 8)) & 0xffu);
    }
}

// component k (compile time) of pixel i of a 3-component run
template <int WIDE>
__device__ __forceinline__ float comp3(const uint32_t *w, int i, int k)
{
    return word_sample<WIDE>(w, i * 3 + k);
}

template <int WIDE, int NC, int INTERP>
__global__ __launch_bounds__(256) void k_packed_vec(LutConsts L, PackedSet P, FrameGeom G)
{
    constexpr int NW = NC * (WIDE ? 2 : 1);               // dwords per 4 pixels
    const GFetch f(L);
    const unsigned uw = (unsigned)G.w / 4;
    const unsigned total = uw * (unsigned)G.rows * (unsigned)G.nframes;
    const unsigned u = blockIdx.x * 256u + threadIdx.x;
    if (u >= total) return;
    const unsigned xu = u % uw, t = u / uw;
    const int y = G.row0 + (int)(t % (unsigned)G.rows);
    const long long fr = t / (unsigned)G.rows;
    const uint32_t *sp = (const uint32_t *)(P.s + fr * P.sfs + (long long)y * P.ss) + (size_t)xu * NW;
    uint32_t *dp = (uint32_t *)(P.d + fr * P.dfs + (long long)y * P.ds) + (size_t)xu * NW;
    uint32_t in[NW], out[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) { in[k] = sp[k]; out[k] = 0; }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if constexpr (NC == 4) {
            const Rgb o = lut3d_px<INTERP>(L, f, comp4<WIDE>(in, i, P.ro), comp4<WIDE>(in, i, P.go),
                                           comp4<WIDE>(in, i, P.bo));
            if constexpr (WIDE) {
                const unsigned long long pv = ((unsigned long long)in[2 * i + 1] << 32) | in[2 * i];
                const unsigned long long q = (pv & (0xffffull << (P.ao * 16))) |
                                             ((unsigned long long)(unsigned)o.r << (P.ro * 16)) |
                                             ((unsigned long long)(unsigned)o.g << (P.go * 16)) |
                                             ((unsigned long long)(unsigned)o.b << (P.bo * 16));
                out[2 * i] = (uint32_t)q;
                out[2 * i + 1] = (uint32_t)(q >> 32);
            } else {
                out[i] = (in[i] & (0xffu << (P.ao * 8))) | ((unsigned)o.r << (P.ro * 8)) |
                         ((unsigned)o.g << (P.go * 8)) | ((unsigned)o.b << (P.bo * 8));
            }
        } else {
            const float c0 = comp3<WIDE>(in, i, 0), c1 = comp3<WIDE>(in, i, 1), c2 = comp3<WIDE>(in, i, 2);
            const bool swap = P.ro != 0;                  // bgr order
            const Rgb o = lut3d_px<INTERP>(L, f, swap ? c2 : c0, c1, swap ? c0 : c2);
            word_put<WIDE>(out, i * 3 + 0, swap ? o.b : o.r);
            word_put<WIDE>(out, i * 3 + 1, o.g);
            word_put<WIDE>(out, i * 3 + 2, swap ? o.r : o.b);
        }
    }
#pragma unroll
    for (int k = 0; k < NW; k++) dp[k] = out[k];
}

// any width / alignment (16-bit formats still need 2-byte aligned rows), all five modes
__global__ __launch_bounds__(256) void k_packed_generic(LutConsts L, PackedSet P, FrameGeom G, int wide, int nc, int mode)
{
    const GFetch f(L);
    const long long total = (long long)G.w * G.rows * G.nframes;
    for (long long u = blockIdx.x * 256ll + threadIdx.x; u < total; u += (long long)gridDim.x * 256ll) {
        const int x = (int)(u % G.w);
        const long long t = u / G.w;
        const int y = G.row0 + (int)(t % G.rows);
        const long long fr = t / G.rows;
        const uint8_t *srow = P.s + fr * P.sfs + (long long)y * P.ss;
        uint8_t *drow = P.d + fr * P.dfs + (long long)y * P.ds;
        const int e = x * nc;
        const Rgb o = lut3d_px_rt(mode, L, f, ld_sample(srow, e + P.ro, wide), ld_sample(srow, e + P.go, wide),
                                  ld_sample(srow, e + P.bo, wide));
        if (nc == 4) st_sample(drow, e + P.ao, wide, ld_sample(srow, e + P.ao, wide));
        st_sample(drow, e + P.ro, wide, o.r);
        st_sample(drow, e + P.go, wide, o.g);
        st_sample(drow, e + P.bo, wide, o.b);
    }
}

static inline bool mult4(long long v) { return (v & 3) == 0; }

// launches under this many pixels stay on the plain vector kernel (no start-up cost); LUTR_SMALL_JOB_MPX as in lutr_kernels.hip
static bool small_job_packed(long long px)
{
    long long mpx = 33;        // the tube kernels' crossover (two-level chunk queue): 8 UHD rgb24 frames run at 480 vs 343 Gpx/s
    if (const char *e = getenv("LUTR_SMALL_JOB_MPX")) { const long long v = atoll(e); if (v >= 0 && v <= 100000) mpx = v; }
    return px < mpx * 1000000ll;
}

const char *launch_packed(hipStream_t st, int variant, const LutConsts &L, const PackedSet &P, const FrameGeom &G,
                          int wide, int nc, int mode, unsigned *stats, unsigned *queue)
{
    const long long px = (long long)G.w * G.rows * G.nframes;
    // round 3: the tube kernels (lutr_rgb2.hip) for the component orders that keep R, G, B adjacent: R G B [x], B G R [x], x R G B, x B G R
    if ((variant == VAR_VEC_LDS || (variant == VAR_AUTO && !small_job_packed(px))) && !getenv("LUTR_NO_RGB2") &&
        !(getenv("LUTR_RGB2") && getenv("LUTR_RGB2")[0] == '0')) {
        const bool fwd = P.go == P.ro + 1 && P.bo == P.go + 1, bwd = P.go == P.bo + 1 && P.ro == P.go + 1;
        const int p0 = fwd ? P.ro : P.bo;
        if ((fwd || bwd) && (nc == 4 ? p0 <= 1 : p0 == 0)) {
            PlaneSet Q{};
            for (int k = 0; k < 3; k++) { Q.s[k] = P.s; Q.d[k] = P.d; Q.ss[k] = P.ss; Q.ds[k] = P.ds; Q.sfs[k] = P.sfs; Q.dfs[k] = P.dfs; }
            const int depth = wide ? 16 : 8, rev = bwd ? 1 : 0;
            const char *name = nullptr;
            if (nc == 3) name = wide ? launch_rgb_tube_ly3(st, L, Q, G, depth, mode, rev, stats, queue)
                                     : launch_rgb_tube_ly2(st, L, Q, G, depth, mode, rev, stats, queue);
            else if (!wide) name = p0 ? launch_rgb_tube_ly5(st, L, Q, G, depth, mode, rev, stats, queue)
                                      : launch_rgb_tube_ly4(st, L, Q, G, depth, mode, rev, stats, queue);
            else name = p0 ? launch_rgb_tube_ly7(st, L, Q, G, depth, mode, rev, stats, queue)
                           : launch_rgb_tube_ly6(st, L, Q, G, depth, mode, rev, stats, queue);
            if (name) return name;
        }
    }
    bool vec_ok = (mode == LUTR_INTERP_NEAREST || mode == LUTR_INTERP_TRILINEAR || mode == LUTR_INTERP_TETRAHEDRAL) &&
                  G.w % 4 == 0 && px / 4 < 0x7fffffffll && mult4((long long)(uintptr_t)P.s) &&
                  mult4((long long)(uintptr_t)P.d) && mult4(P.ss) && mult4(P.ds) &&
                  (G.nframes == 1 || (mult4(P.sfs) && mult4(P.dfs)));
    // the 3-component vector body knows RGB and BGR order only; any other permutation takes the scalar kernel
    if (nc == 3 && !(P.go == 1 && ((P.ro == 0 && P.bo == 2) || (P.ro == 2 && P.bo == 0)))) vec_ok = false;
    if (variant == VAR_GENERIC) vec_ok = false;
    if (!vec_ok) {
        if (variant == VAR_VEC_GLOBAL || variant == VAR_VEC_LDS) return nullptr;
        long long b = (px + 255) / 256;
        if (b < 1) b = 1;
        if (b > 256 * 64) b = 256 * 64;
        hipLaunchKernelGGL(k_packed_generic, dim3((unsigned)b), dim3(256), 0, st, L, P, G, wide, nc, mode);
        return "k_packed_generic";
    }
    const dim3 grid((unsigned)((px / 4 + 255) / 256)), block(256);
#define PK_CASE(W, C, I) \
    if (wide == W && nc == C && mode == I) { \
        hipLaunchKernelGGL((k_packed_vec<W, C, I>), grid, block, 0, st, L, P, G); \
        return "k_packed_vec<" #W "," #C "," #I ">"; \
    }
#define PK_FMT(W, C) PK_CASE(W, C, 0) PK_CASE(W, C, 1) PK_CASE(W, C, 2)
    PK_FMT(0, 3) PK_FMT(0, 4) PK_FMT(1, 3) PK_FMT(1, 4)
#undef PK_FMT
#undef PK_CASE
    return nullptr;
}

}  // namespace lutr
