/*
 * oracle/lut3d_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the per-pixel path the reference delegates to FFmpeg:
 *   /root/reference/src/lut_renderer/ffmpeg.py:246   lut3d=file='...':interp=...
 *   /root/reference/src/lut_renderer/ffmpeg.py:225-235, :308-309  scale=/format= around it
 * The arithmetic itself lives in third-party FFmpeg (libavfilter/vf_lut3d.c,
 * libswscale), which is NOT vendored under /root/reference, is not pinned to
 * any version (absent from pyproject.toml / uv.lock) and is not installed in
 * this image.  This file restates FFmpeg's published algorithm (SURVEY.md
 * Appendix A) and the engine's own YUV contract (DESIGN.md "YUV contract").
 *
 * PARITY UNPINNED: the reference holds no pixel-level golden vectors and no
 * ffmpeg binary exists here, so this oracle is pinned only by the analytic
 * known-answer tests of SURVEY.md A.6 and by the argv fixtures captured from
 * the reference's own build_command (tests/golden/argv_cases.json).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (liblutr.so) never links or calls it.
 */
#ifndef LUT3D_ORACLE_H
#define LUT3D_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* interpolation modes, numbered like FFmpeg's enum interp_mode (vf_lut3d.c) */
enum { ORC_NEAREST = 0, ORC_TRILINEAR = 1, ORC_TETRAHEDRAL = 2, ORC_PYRAMID = 3, ORC_PRISM = 4 };

/* errors: negative errno-style, mirroring AVERROR(EINVAL)/AVERROR_INVALIDDATA */
#define ORC_EINVAL  (-22)
#define ORC_ENOENT  (-2)
#define ORC_ENOMEM  (-12)
#define ORC_EILSEQ  (-84)   /* "invalid data" */

typedef struct orc_lut {
    int    n;          /* lutsize, 2..256 */
    float  scale[3];   /* r,g,b : clip(1/(max-min),0,1) */
    float *rgb;        /* n*n*n*3 floats, index ((r*n+g)*n+b)*3 + c  (blue fastest) */
    /* lut3d's prelut (a 1D shaper ahead of the cube; only cineSpace .csp files carry one).  pre_size 0 = none, else 65536:
     * prelut[c * pre_size + i], sampled at pre_min[c] + i / pre_scale[c] (FFmpeg Lut3DPreLut). */
    int    pre_size;
    float  pre_min[3], pre_scale[3];
    float *prelut;
} orc_lut;

int  orc_cube_parse(const char *path, orc_lut *out);
/* any file lut3d reads (.cube .dat .3dl .m3d .csp), by extension */
int  orc_lut_file_parse(const char *path, orc_lut *out);
void orc_lut_free(orc_lut *lut);

/* Planar RGB, FFmpeg gbrp plane order: plane 0 = G, 1 = B, 2 = R.
 * depth 8 -> uint8 samples, 9..16 -> little-endian uint16 samples.
 * strides in bytes.  nthreads <= 1 -> scalar; else FFmpeg-style row slices. */
int orc_apply_planar_rgb(const orc_lut *lut, int depth, int interp, int w, int h,
                         const void *const src[3], const ptrdiff_t sstride[3],
                         void *const dst[3], const ptrdiff_t dstride[3], int nthreads);

/* packed one-pixel helper used by the analytic KATs: in/out are integer codes */
int orc_apply_pixel(const orc_lut *lut, int depth, int interp, const int in_rgb[3], int out_rgb[3]);

/* ---- YUV contract (the engine's own definition) ---- */
enum { ORC_MAT_BT709 = 0, ORC_MAT_BT601 = 1, ORC_MAT_BT2020 = 2 };
enum { ORC_RANGE_TV = 0, ORC_RANGE_PC = 1 };

typedef struct orc_yuv_consts {
    /* input side: integer YUV codes (depth din) -> integer RGB codes (depth dl) */
    float ky, yb;              /* yy = fmaf(ky, Y, yb)   (yb carries -ky*yoff + 0.5) */
    float coff;                /* chroma zero point of the input depth */
    float krv, kgu, kgv, kbu;
    float max_l;               /* 2^dl - 1 */
    /* output side: integer RGB codes (depth dl) -> integer YUV codes (depth dout) */
    float cyr, cyg, cyb, yob;  /* Y  = floor(fma(cyr,R,fma(cyg,G,fma(cyb,B,yob)))) */
    float cbr, cbg, cbb;       /* Cb = floor(fma(cbr,Rs,fma(cbg,Gs,fma(cbb,Bs,cob)))), Rs = sum over the chroma block */
    float crr, crg, crb, cob;
    float max_o;               /* 2^dout - 1 */
    /* optional full->limited prologue on the input codes (scale=in_range=pc:out_range=tv,
     * ffmpeg.py:225): Y' = floor(fma(py, Y, pyb)), C' = floor(fma(pc, C, pcb)); enabled iff pre != 0 */
    int   pre;
    float py, pyb, pc, pcb, pre_max;
} orc_yuv_consts;

/* din: depth of the input codes; dl: depth the LUT runs at (after an optional
 * prologue that also reduces to dl); dout: depth of the output codes.
 * chroma_n = number of luma samples sharing one chroma sample (1, 2 or 4). */
int orc_yuv_constants(int matrix_in, int range_in, int matrix_out, int range_out,
                      int din, int dl, int dout, int chroma_n, int prologue_pc_to_tv,
                      orc_yuv_consts *out);

/* planar YUV in -> planar YUV out through integer RGB + lut3d.
 * csx/csy: log2 chroma subsampling (4:2:0 -> 1,1; 4:2:2 -> 1,0; 4:4:4 -> 0,0), same on both sides.
 * din/dout 8 -> uint8 planes, else uint16 LE planes. */
int orc_apply_yuv(const orc_lut *lut, int interp, const orc_yuv_consts *k,
                  int din, int dl, int dout, int csx, int csy, int w, int h,
                  const void *const src[3], const ptrdiff_t sstride[3],
                  void *const dst[3], const ptrdiff_t dstride[3], int nthreads);

/* the same path with error-diffusion dither on the final quantisation (reference ffmpeg.py:305-307,
 * `zscale=dither=error_diffusion`); whole frames only.  orc_dither_plane is the Floyd-Steinberg step
 * alone: x = w*h unquantised values, dst = integer plane (uint8, or uint16 when wide). */
/* The FAST variant of the product (include/lutr.h lutr_ctx_set_precision): nodes as fp16 of value * (2^dl - 1), blend as
 * an fmaf chain with fp32 accumulation in the kernels' order; everything else as orc_apply_yuv.  dl must be 8 or 10,
 * interp nearest / trilinear / tetrahedral.  Bit-exact twin of csrc/lutr_tile2.hip V_FAST. */
int orc_apply_yuv_fast(const orc_lut *lut, int interp, const orc_yuv_consts *k,
                       int din, int dl, int dout, int csx, int csy, int w, int h,
                       const void *const src[3], const ptrdiff_t sstride[3],
                       void *const dst[3], const ptrdiff_t dstride[3], int nthreads);
/* fp32 <-> fp16 (round to nearest even) as used for the fast lattice */
uint16_t orc_f2h(float f);
float orc_h2f(uint16_t h);

int orc_apply_yuv_dither(const orc_lut *lut, int mode, const orc_yuv_consts *k,
                         int din, int dl, int dout, int csx, int csy, int w, int h,
                         const void *const src[3], const ptrdiff_t sstride[3],
                         void *const dst[3], const ptrdiff_t dstride[3], int nthreads);
void orc_dither_plane(const float *x, int w, int h, float maxv, int wide, void *dst, ptrdiff_t dstride);

#ifdef __cplusplus
}
#endif
#endif
