// lutr_dither.hip -- error-diffusion dither of the final quantisation.
//
// What it replaces: `zscale=dither=error_diffusion`, which the reference appends to the filter chain at
//   /root/reference/src/lut_renderer/ffmpeg.py:305-307   (option `zscale_dither`, models.py:46, default "none")
// The arithmetic is the engine's own contract (zimg is neither vendored nor pinned):
// the RGB -> YUV stage keeps its unquantised value x (the fma chain of 3.2 without the +0.5), then per
// plane, rows top to bottom and left to right (Floyd-Steinberg, zimg's dither_ed order of operations):
//     err = 0; err += e[left]*7/16; err += e[up-right]*3/16; err += e[up]*5/16; err += e[up-left]*1/16
//     v = clip(x + err, 0, max_o);  q = rint(v) (half to even);  e = v - q
// Bit-exact against oracle/lut3d_oracle.c orc_apply_yuv_dither.
//
// Two kernels:
//   k_yuv_float   one thread per chroma block: YUV -> RGB -> lut3d -> unquantised YUV, written as float
//                 planes into scratch (taps gathered from L2 like the generic kernel).
//   k_dither_ed   the sequential part.  A pixel needs its left neighbour and three pixels of the row above,
//                 so row r can run a few columns behind row r-1: a wave takes a band of 64 rows, lane r on row
//                 r, skewed by 4 columns per lane (2 would do; 4 keeps all lanes on the same phase of a 4-column
//                 group, so inputs load as float4 and outputs store as one packed word); the error of the row
//                 above arrives from lane r-1 with one cross-lane move per step.  The NW waves of a workgroup pipeline consecutive bands of the
//                 same plane: the last lane of a band publishes its errors in an LDS row and a progress
//                 counter, the first lane of the next band (another wave of the same workgroup, co-resident by
//                 construction, so the wait cannot deadlock) trails it.  One workgroup per (frame, plane).
//                 This is latency-bound integer/float bookkeeping -- no roofline claim; it exists so that the
//                 option does something, at a rate far above the CPU's.
#include <mutex>
#include <set>

#include "lutr_device.h"

namespace lutr {

// ---------------------------------------------------------------- pass 1: unquantised YUV planes
__global__ __launch_bounds__(256) void k_yuv_float(LutConsts L, YuvConsts K, PlaneSet P, FrameGeom G, FloatPlanes F,
                                                   int win, int csx, int csy, int mode)
{
    const GFetch f(L);
    const int bw = 1 << csx, bh = 1 << csy;
    const int cw = (G.w + bw - 1) >> csx, ch = (G.h + bh - 1) >> csy;
    const long long total = (long long)cw * ch * G.nframes;
    for (long long u = blockIdx.x * 256ll + threadIdx.x; u < total; u += (long long)gridDim.x * 256ll) {
        const int cx = (int)(u % cw);
        const long long t = u / cw;
        const int cy = (int)(t % ch);
        const long long fr = t / ch;
        const float cbv = ld_sample(P.s[1] + fr * P.sfs[1] + (long long)cy * P.ss[1], cx, win);
        const float crv = ld_sample(P.s[2] + fr * P.sfs[2] + (long long)cy * P.ss[2], cx, win);
        const Chroma c = chroma_terms(K, cbv, crv);
        float rs = 0.f, gs = 0.f, bs = 0.f;
        for (int dy = 0; dy < bh; dy++) {
            const int yy = cy * bh + dy;
            const int y = yy < G.h ? yy : G.h - 1;
            for (int dx = 0; dx < bw; dx++) {
                const int xx = cx * bw + dx;
                const int x = xx < G.w ? xx : G.w - 1;
                const float yv = ld_sample(P.s[0] + fr * P.sfs[0] + (long long)y * P.ss[0], x, win);
                const Rgb q = yuv_to_rgb(K, yv, c);
                const Rgb o = lut3d_px_rt(mode, L, f, q.r, q.g, q.b);
                rs += o.r; gs += o.g; bs += o.b;
                if (yy < G.h && xx < G.w)
                    F.y[(fr * G.h + y) * G.w + x] = fma_(K.cyr, o.r, fma_(K.cyg, o.g, fma_(K.cyb, o.b, K.yob))) - 0.5f;
            }
        }
        F.cb[(fr * ch + cy) * cw + cx] = fma_(K.cbr, rs, fma_(K.cbg, gs, fma_(K.cbb, bs, K.cob))) - 0.5f;
        F.cr[(fr * ch + cy) * cw + cx] = fma_(K.crr, rs, fma_(K.crg, gs, fma_(K.crb, bs, K.cob))) - 0.5f;
    }
}

// ---------------------------------------------------------------- pass 2: error diffusion
extern __shared__ __attribute__((aligned(16))) char dither_smem[];

__device__ __forceinline__ int lds_read_progress(volatile int *p) { return *p; }

// SKEW columns between consecutive rows of a band (>= 2 for Floyd-Steinberg; 4 puts every lane on the same
// phase of a 4-column group, so the groups can be loaded as one float4 and stored as one packed word).
constexpr int SKEW = 4;
constexpr int PFG = 4;       // groups of 4 columns kept in flight per lane (software prefetch)

// VEC: plane widths are multiples of 4 and the destination rows are aligned for packed 4-sample stores.
template <bool VEC>
__global__ __launch_bounds__(512) void k_dither_ed(FloatPlanes F, PlaneSet P, FrameGeom G, int csx, int csy, float maxv, int wide)
{
    const int plane = blockIdx.x % 3;
    const long long fr = blockIdx.x / 3;
    const int w = plane ? (G.w + (1 << csx) - 1) >> csx : G.w;
    const int h = plane ? (G.h + (1 << csy) - 1) >> csy : G.h;
    const float *x = (plane == 0 ? F.y : plane == 1 ? F.cb : F.cr) + fr * (long long)w * h;
    uint8_t *dst = P.d[plane] + fr * P.dfs[plane];
    const long long dstride = P.ds[plane];

    const int nw = blockDim.x >> 6;                       // waves of this workgroup = bands in flight
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    // LDS: per wave one row of w + 2 errors (what its last lane produced, padded by one column each side),
    // then per wave one progress counter: band * (w + 1) + columns the last lane has finished in that band
    float *erow_all = (float *)dither_smem;
    const int rowlen = w + 2;
    volatile int *prog = (volatile int *)(erow_all + (size_t)nw * rowlen);
    float *my_erow = erow_all + (size_t)wave * rowlen;
    const int up_wave = (wave + nw - 1) % nw;
    const float *up_erow = erow_all + (size_t)up_wave * rowlen;
    for (int i = threadIdx.x; i < nw * rowlen; i += blockDim.x) erow_all[i] = 0.0f;
    if (threadIdx.x < nw) prog[threadIdx.x] = -1;
    __syncthreads();                                      // the only barrier; everything after is wave-to-wave

    // group of 4 columns starting at column c0 (a multiple of 4, possibly outside the row) of row xr
    auto load_group = [&](const float *xr, int c0) -> float4 {
        if constexpr (VEC) {
            return *(const float4 *)(xr + min(max(c0, 0), w - 4));
        } else {
            float4 g;
            g.x = xr[min(max(c0 + 0, 0), w - 1)];
            g.y = xr[min(max(c0 + 1, 0), w - 1)];
            g.z = xr[min(max(c0 + 2, 0), w - 1)];
            g.w = xr[min(max(c0 + 3, 0), w - 1)];
            return g;
        }
    };

    const int nbands = (h + 63) >> 6;
    for (int band = wave; band < nbands; band += nw) {
        const int row = band * 64 + lane;
        const int rowc = row < h ? row : h - 1;           // lanes past the last row shadow it (loads stay in range)
        const float *xr = x + (long long)rowc * w;
        uint8_t *drow = dst + (long long)rowc * dstride;
        const bool row_ok = row < h;
        const bool last_lane = (lane == 63) || (row == h - 1);
        // e_left: my previous error.  ul/u/ur: errors of the row above at columns j-1, j, j+1.  m1..m3: my
        // errors of the last three steps; the lane below is SKEW = 4 columns behind and needs the one from
        // three steps ago (column j + 1 of its row above).
        float e_left = 0.0f, ul = 0.0f, u = 0.0f, ur = 0.0f, m1 = 0.0f, m2 = 0.0f, m3 = 0.0f;
        float4 ring[PFG];
#pragma unroll
        for (int g = 0; g < PFG; g++) ring[g] = load_group(xr, 4 * g - SKEW * lane);
        const int steps = w + SKEW * 63;
        const int need_base = (band - 1) * (w + 1);
        // Lanes > 0 see columns -1 and 0 of the row above go by during their steps of skew; lane 0 starts
        // at column 0 right away, so its "up" error for column 0 is primed here.
        if (lane == 0 && band > 0) {
            while (lds_read_progress(&prog[up_wave]) < need_base + 1) __builtin_amdgcn_s_sleep(1);
            ur = ((volatile const float *)up_erow)[1];
        }
        for (int t0 = 0; t0 < steps; t0 += 4 * PFG) {
#pragma unroll
            for (int g = 0; g < PFG; g++) {
                const float4 xg = ring[g];
                const int j0 = t0 + 4 * g - SKEW * lane;             // first column of this group
                ring[g] = load_group(xr, j0 + 4 * PFG);
                unsigned qs[4];
                float es[4];
                // Lane 0 of a band below the first takes the errors of the row above from the LDS row of the wave
                // that owns band - 1: ONE wait per group (until that wave's last lane has passed column j0 + 4,
                // or the end of the row), then the four values.
                float upv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (lane == 0 && band > 0 && j0 + 1 < w) {
                    const int need = need_base + min(j0 + 5, w);
                    while (lds_read_progress(&prog[up_wave]) < need) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (j0 + k + 1 < w) upv[k] = ((volatile const float *)up_erow)[j0 + k + 2];
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = j0 + k;
                    // the error of the row above at column j + 1
                    float up_new = __shfl_up(m3, 1, 64);
                    if (lane == 0) up_new = upv[k];
                    if (j + 1 >= w) up_new = 0.0f;
                    ul = u; u = ur; ur = up_new;
                    const float xv = k == 0 ? xg.x : k == 1 ? xg.y : k == 2 ? xg.z : xg.w;
                    const bool act = row_ok && j >= 0 && j < w;
                    float e = 0.0f;
                    qs[k] = 0;
                    if (act) {
                        float err = 0.0f;
                        err += e_left * (7.0f / 16.0f);
                        err += ur * (3.0f / 16.0f);
                        err += u * (5.0f / 16.0f);
                        err += ul * (1.0f / 16.0f);
                        float v = xv + err;
                        v = fminf(fmaxf(v, 0.0f), maxv);
                        const float q = rintf(v);
                        e = v - q;
                        qs[k] = (unsigned)q;
                        if constexpr (!VEC) {
                            if (wide) ((uint16_t *)drow)[j] = (uint16_t)qs[k];
                            else drow[j] = (uint8_t)qs[k];
                        }
                    }
                    e_left = act ? e : 0.0f;
                    es[k] = e_left;
                    m3 = m2; m2 = m1; m1 = e_left;
                }
                // the band's last lane publishes its errors for the band below: the group's values, then (once
                // they are in LDS) the number of columns it has finished
                if (last_lane && row_ok && j0 + 3 >= 0 && j0 < w) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (j0 + k >= 0 && j0 + k < w) ((volatile float *)my_erow)[j0 + k + 1] = es[k];
                    __builtin_amdgcn_s_waitcnt(0xc07f);                // lgkmcnt(0)
                    prog[wave] = band * (w + 1) + min(j0 + 4, w);
                }
                if constexpr (VEC) {
                    // w % 4 == 0 and j0 % 4 == 0: a group is wholly inside the row or wholly outside
                    if (row_ok && j0 >= 0 && j0 < w) {
                        if (wide) *(uint2 *)(drow + 2 * j0) = make_uint2(qs[0] | (qs[1] << 16), qs[2] | (qs[3] << 16));
                        else *(uint32_t *)(drow + j0) = qs[0] | (qs[1] << 8) | (qs[2] << 16) | (qs[3] << 24);
                    }
                }
            }
        }
        // a band whose last lane wrote nothing past some column still has to release its consumer
        if (last_lane && row_ok) prog[wave] = band * (w + 1) + w + 1;
    }
}

// ---------------------------------------------------------------- launcher
const char *launch_yuv_dither(hipStream_t st, const LutConsts &L, const YuvConsts &K, const PlaneSet &P,
                              const FrameGeom &G, const FloatPlanes &F, int din, int dout, int csx, int csy, int mode)
{
    const int win = din > 8, wout = dout > 8;
    const long long blocks = (long long)((G.w + (1 << csx) - 1) >> csx) * ((G.h + (1 << csy) - 1) >> csy) * G.nframes;
    long long gb = (blocks + 255) / 256;
    if (gb < 1) gb = 1;
    if (gb > 256 * 64) gb = 256 * 64;
    hipLaunchKernelGGL(k_yuv_float, dim3((unsigned)gb), dim3(256), 0, st, L, K, P, G, F, win, csx, csy, mode);
    // waves per workgroup = bands of one plane in flight: as many as the LDS error rows allow (144 KB of the
    // CU's 160 KB: one workgroup per CU, and a batch of 64 frames is 192 workgroups for 256 CUs), at most 8
    int nw = (int)((144 * 1024) / ((size_t)(G.w + 2) * sizeof(float) + sizeof(int)));
    if (nw > 8) nw = 8;
    if (nw < 1) nw = 1;
    const int nbands = (G.h + 63) / 64;
    if (nw > nbands) nw = nbands;
    const size_t lds = (size_t)nw * ((size_t)(G.w + 2) * sizeof(float) + sizeof(int));
    if (lds > 160 * 1024) return nullptr;                 // rows wider than ~40,000 samples: not supported
    {   // dynamic LDS above 64 KB has to be allowed per kernel and device
        static std::set<int> done;
        static std::mutex mu;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(mu);
        if (!done.count(dev)) {
            (void)hipFuncSetAttribute((const void *)k_dither_ed<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_dither_ed<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            done.insert(dev);
        }
    }
    // packed path: every plane width a multiple of 4, destination rows aligned for 4-sample stores
    const int cwid = (G.w + (1 << csx) - 1) >> csx;
    bool vec = G.w % 4 == 0 && cwid % 4 == 0;
    const uintptr_t al = wout ? 8 : 4;
    for (int c = 0; c < 3 && vec; c++)
        vec = ((uintptr_t)P.d[c] % al) == 0 && (P.ds[c] % (long long)al) == 0 && (G.nframes == 1 || P.dfs[c] % (long long)al == 0);
    const dim3 grid((unsigned)(3 * G.nframes)), block(64 * nw);
    if (vec) hipLaunchKernelGGL((k_dither_ed<true>), grid, block, lds, st, F, P, G, csx, csy, K.max_o, wout);
    else hipLaunchKernelGGL((k_dither_ed<false>), grid, block, lds, st, F, P, G, csx, csy, K.max_o, wout);
    return "k_yuv_float+k_dither_ed";
}

}  // namespace lutr
