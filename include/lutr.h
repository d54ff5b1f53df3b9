/*
 * lutr.h -- C-ABI of the MI355X-native 3D-LUT apply engine (liblutr.so).
 *
 * This is the drop-in boundary for ONE path of ionlz/LUT-renderer: the per-pixel
 * work the reference delegates to the external `ffmpeg` process through the
 * filter string it builds at
 *     src/lut_renderer/ffmpeg.py:246   lut3d=file='<cube>':interp=<mode>
 * together with the conversions it (or ffmpeg's filter negotiation) puts around
 * that filter:
 *     src/lut_renderer/ffmpeg.py:212-236   scale=in_range=..:out_range=..:in_color_matrix=..
 *     src/lut_renderer/ffmpeg.py:224,233   format=<8-bit intermediate>
 *     src/lut_renderer/ffmpeg.py:304-310   format=<pix_fmt>
 * The reference has no FFI for this path (it spawns a process,
 * src/lut_renderer/task_manager.py:145-151); these entry points are what a
 * ctypes binding would load instead (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in signatures
 *     (a HIP stream crosses as void*).
 *   - every function returns 0 on success or a negative errno-style code;
 *     lutr_last_error() returns a thread-local message for the last failure.
 *   - pixel-plane pointers handed to lutr_apply_* are DEVICE pointers on the
 *     context's GPU.  The library never falls back to the CPU: with no usable
 *     GPU, lutr_ctx_create fails with LUTR_EIO.
 *   - the caller owns every buffer it passes; the library owns the context,
 *     its stream (unless one is injected) and the device copy of the lattice.
 *   - a context is not shared between threads; different contexts are independent.
 */
#ifndef LUTR_H
#define LUTR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LUTR_VERSION_STRING "0.1.0"

/* error codes (negative errno values; mirrors FFmpeg's AVERROR(EINVAL) / AVERROR_INVALIDDATA split) */
#define LUTR_OK        0
#define LUTR_ENOENT   (-2)    /* file not found */
#define LUTR_EIO      (-5)    /* HIP runtime failure / no GPU */
#define LUTR_ENOMEM   (-12)
#define LUTR_EINVAL   (-22)   /* bad argument, unsupported format, LUT_3D_SIZE outside [2,256] */
#define LUTR_EILSEQ   (-84)   /* malformed .cube ("invalid data"), unexpected EOF */

/* interpolation, numbered like FFmpeg's lut3d `interp` option values
 * (replaces the interp= half of ffmpeg.py:242-246) */
enum lutr_interp {
    LUTR_INTERP_NEAREST     = 0,
    LUTR_INTERP_TRILINEAR   = 1,
    LUTR_INTERP_TETRAHEDRAL = 2,
    LUTR_INTERP_PYRAMID     = 3,
    LUTR_INTERP_PRISM       = 4
};

/* YUV<->RGB matrices the reference can name (ffmpeg.py:119-125) */
enum lutr_matrix {
    LUTR_MATRIX_BT709  = 0,
    LUTR_MATRIX_BT601  = 1,   /* smpte170m and bt470bg */
    LUTR_MATRIX_BT2020 = 2    /* bt2020nc; bt2020c uses the same coefficients */
};

/* ranges (in_range=/out_range= at ffmpeg.py:225) */
enum lutr_range { LUTR_RANGE_TV = 0, LUTR_RANGE_PC = 1 };

/* planar YUV formats: depth | csx<<8 | csy<<9  (csx/csy = log2 chroma subsampling) */
#define LUTR_FMT(depth, csx, csy) ((depth) | ((csx) << 8) | ((csy) << 9))
#define LUTR_FMT_DEPTH(f) ((f) & 0xff)
#define LUTR_FMT_CSX(f)   (((f) >> 8) & 1)
#define LUTR_FMT_CSY(f)   (((f) >> 9) & 1)
#define LUTR_FMT_YUV420P     LUTR_FMT(8, 1, 1)
#define LUTR_FMT_YUV422P     LUTR_FMT(8, 1, 0)
#define LUTR_FMT_YUV444P     LUTR_FMT(8, 0, 0)
#define LUTR_FMT_YUV420P10   LUTR_FMT(10, 1, 1)
#define LUTR_FMT_YUV422P10   LUTR_FMT(10, 1, 0)
#define LUTR_FMT_YUV444P10   LUTR_FMT(10, 0, 0)
#define LUTR_FMT_YUV420P12   LUTR_FMT(12, 1, 1)
#define LUTR_FMT_YUV422P12   LUTR_FMT(12, 1, 0)
#define LUTR_FMT_YUV444P12   LUTR_FMT(12, 0, 0)

/* What the filter chain around lut3d does to a YUV frame (ffmpeg.py:212-236, :304-310).
 *   src codes (fmt_in, range_src)
 *     -> optional prologue to (lut_depth, range_in)       scale=in_range=pc:out_range=..,format=yuv4xxp
 *     -> integer RGB at lut_depth via matrix_in           (auto-inserted scaler ahead of lut3d)
 *     -> lut3d                                            ffmpeg.py:246
 *     -> YUV (fmt_out, range_out) via matrix_out          format=<pix_fmt>, ffmpeg.py:309
 * The prologue runs iff range_src != range_in or depth(fmt_in) != lut_depth.
 * fmt_in and fmt_out must have the same chroma subsampling. */
typedef struct lutr_yuv_params {
    int32_t fmt_in;
    int32_t fmt_out;
    int32_t lut_depth;
    int32_t matrix_in;
    int32_t matrix_out;
    int32_t range_src;
    int32_t range_in;
    int32_t range_out;
} lutr_yuv_params;

/* One frame (or a batch of equally laid out frames) of three planes.
 * stride = bytes between rows; frame_stride = bytes between consecutive frames of
 * a batch (ignored when nframes == 1).  Planar RGB uses FFmpeg's gbrp order
 * (plane 0 = G, 1 = B, 2 = R); YUV uses Y, Cb, Cr. */
typedef struct lutr_planes {
    void     *data[3];
    ptrdiff_t stride[3];
    int64_t   frame_stride[3];
} lutr_planes;

/* Packed (interleaved) RGB formats lut3d accepts: bits per component (8|16, 16 = little-endian
 * uint16), components per pixel (3|4) and the component index of R, G, B inside a pixel; the
 * remaining slot of a 4-component format (alpha or padding) is copied from source to destination. */
#define LUTR_PACKED(bits, ncomp, ro, go, bo) ((bits) | ((ncomp) << 8) | ((ro) << 12) | ((go) << 16) | ((bo) << 20))
#define LUTR_PACKED_BITS(f)  ((f) & 0xff)
#define LUTR_PACKED_NCOMP(f) (((f) >> 8) & 0xf)
#define LUTR_PACKED_RO(f)    (((f) >> 12) & 0xf)
#define LUTR_PACKED_GO(f)    (((f) >> 16) & 0xf)
#define LUTR_PACKED_BO(f)    (((f) >> 20) & 0xf)
#define LUTR_PK_RGB24    LUTR_PACKED(8, 3, 0, 1, 2)
#define LUTR_PK_BGR24    LUTR_PACKED(8, 3, 2, 1, 0)
#define LUTR_PK_RGBA     LUTR_PACKED(8, 4, 0, 1, 2)   /* also rgb0 */
#define LUTR_PK_BGRA     LUTR_PACKED(8, 4, 2, 1, 0)   /* also bgr0 */
#define LUTR_PK_ARGB     LUTR_PACKED(8, 4, 1, 2, 3)   /* also 0rgb */
#define LUTR_PK_ABGR     LUTR_PACKED(8, 4, 3, 2, 1)   /* also 0bgr */
#define LUTR_PK_RGB48LE  LUTR_PACKED(16, 3, 0, 1, 2)
#define LUTR_PK_BGR48LE  LUTR_PACKED(16, 3, 2, 1, 0)
#define LUTR_PK_RGBA64LE LUTR_PACKED(16, 4, 0, 1, 2)
#define LUTR_PK_BGRA64LE LUTR_PACKED(16, 4, 2, 1, 0)

/* one interleaved image, or a batch of equally laid out ones */
typedef struct lutr_packed {
    void     *data;
    ptrdiff_t stride;          /* bytes between rows */
    int64_t   frame_stride;    /* bytes between frames of a batch (ignored when nframes == 1) */
} lutr_packed;

typedef struct lutr_ctx lutr_ctx;

const char *lutr_version(void);
const char *lutr_last_error(void);

/* ---- .cube files (host side; what lut3d's file= option does, ffmpeg.py:246) ---- */
/* Parses `path` with FFmpeg's parse_cube semantics.  On success *rgb is a malloc'ed
 * n*n*n*3 float lattice indexed ((r*n+g)*n+b)*3+c (blue fastest), scale[c] =
 * clip(1/(DOMAIN_MAX[c]-DOMAIN_MIN[c]), 0, 1).  Free with lutr_cube_free. */
int  lutr_cube_parse(const char *path, float **rgb, int *n, float scale[3]);
void lutr_cube_free(float *rgb);
/* Any 3D LUT file lut3d's file= option accepts, chosen by extension like FFmpeg does: .cube (as above),
 * .dat, .3dl, .m3d, .csp (cineSpace).  Same outputs as lutr_cube_parse; free with lutr_cube_free.  Unknown extension: LUTR_EINVAL.
 * A .csp file whose three channels carry a pre-LUT (a 1D shaper ahead of the cube, lut3d's `prelut`) needs lutr_lut_parse_ex:
 * lutr_lut_parse returns LUTR_EINVAL for it rather than dropping the shaper. */
int  lutr_lut_parse(const char *path, float **rgb, int *n, float scale[3]);
/* As lutr_lut_parse, plus the file's prelut as lut3d builds it (/root/reference never sees it: FFmpeg's vf_lut3d.c parse_cinespace):
 * *prelut = 3 x *prelut_size floats (channel c at [c * size]; size is 65536) sampled at prelut_min[c] + i / prelut_scale[c],
 * or NULL / 0 when the file has none.  Free *prelut with lutr_cube_free.  Pass it to lutr_ctx_set_prelut after lutr_ctx_set_lut. */
int  lutr_lut_parse_ex(const char *path, float **rgb, int *n, float scale[3], float **prelut, int *prelut_size,
                       float prelut_min[3], float prelut_scale[3]);

/* ---- context ---- */
int  lutr_ctx_create(int device, lutr_ctx **out);
void lutr_ctx_destroy(lutr_ctx *ctx);
/* run on a caller-owned HIP stream (hipStream_t passed as void*); NULL selects HIP's default
 * (null) stream, which is what torch.cuda.current_stream() is unless the caller changed it.
 * Until this is called the context uses a private non-blocking stream.  Launches of one context never
 * overlap: if work issued on the previous stream may still be running, the new stream is made to wait
 * for it (hipStreamWaitEvent) -- the context's work queue and scratch are per context, not per stream. */
int  lutr_ctx_set_stream(lutr_ctx *ctx, void *hip_stream);
int  lutr_ctx_sync(lutr_ctx *ctx);

/* upload a host lattice (layout of lutr_cube_parse); drops a prelut set earlier */
int  lutr_ctx_set_lut(lutr_ctx *ctx, const float *rgb, int n, const float scale[3]);
/* lut3d's prelut for the lattice just set (layout of lutr_lut_parse_ex; size 0 or prelut NULL removes it).  Every sample then goes
 * through FFmpeg's apply_prelut (linear interpolation in the 1D table) before the cube.  With a prelut the calls run on the
 * generic / vector kernels (the LDS tile kernels and the fast precision do not take one). */
int  lutr_ctx_set_prelut(lutr_ctx *ctx, const float *prelut, int size, const float prelut_min[3], const float prelut_scale[3]);
/* multi-GPU: a non-root rank allocates the device lattice without filling it, the host
 * broadcasts into the pointer returned by lutr_ctx_lut_device (RCCL, root = the rank that
 * called lutr_ctx_set_lut), then every rank may apply. */
int  lutr_ctx_lut_alloc(lutr_ctx *ctx, int n, const float scale[3]);
int  lutr_ctx_lut_device(lutr_ctx *ctx, void **dptr, size_t *bytes);
/* call after writing the device lattice obtained from lutr_ctx_lut_device (e.g. once the broadcast
 * has landed): checks that every node is finite (LUTR_EINVAL otherwise, like lutr_ctx_set_lut) and
 * records the lattice's value range, which selects clip-free kernels for lattices inside [0, 1].
 * Optional: an unsealed lattice is applied with the general kernels. */
int  lutr_ctx_lut_seal(lutr_ctx *ctx);
/* One process, several GPUs (the reference is one GUI process with a thread pool,
 * src/lut_renderer/task_manager.py:229-235, so it cannot start one rank per GPU): copy the lattice of
 * ctxs[root] into every other context, GPU to GPU over xGMI (hipMemcpyPeerAsync on each receiver's stream;
 * a plain device copy when two contexts share a GPU).  Receivers are allocated as needed and inherit the
 * root's value-range seal.  Asynchronous: each context's later applies are ordered behind its copy.  This
 * is the single-process twin of the RCCL broadcast one-rank-per-GPU hosts do into lutr_ctx_lut_device. */
int  lutr_lut_broadcast(lutr_ctx **ctxs, int nctx, int root);
/* lutr_lut_broadcast with flags.  LUTR_BCAST_FORCE_PEER_COPY: take the cross-device call (hipMemcpyPeerAsync) also between
 * two contexts of ONE GPU (a self-peer copy is legal) -- the test hook that lets a one-GPU box execute what an 8-GPU node
 * runs.  The receiver's copy is recorded on its own event (a later lutr_ctx_set_stream waits for it) and the root waits
 * for outstanding copies before its lattice is overwritten or freed. */
enum { LUTR_BCAST_FORCE_PEER_COPY = 1 };
int  lutr_lut_broadcast_ex(lutr_ctx **ctxs, int nctx, int root, unsigned flags);
/* bytes of the device lattice layout for size n: (n+1)^3 nodes of 16 bytes */
size_t lutr_lattice_bytes(int n);

/* ---- apply (device pointers; asynchronous on the context's stream) ---- */
/* lut3d on planar RGB at `depth` bits (8 -> uint8 planes, 9..16 -> uint16 LE planes),
 * luma rows [row0, row0+rows) of each of nframes frames. */
int lutr_apply_planar_rgb(lutr_ctx *ctx, int depth, int interp, int w, int h, int nframes,
                          const lutr_planes *src, const lutr_planes *dst, int row0, int rows);

/* lut3d on packed RGB (`pfmt` = one of LUTR_PK_* / LUTR_PACKED(...)); M = 255 or 65535.  16-bit
 * formats need 2-byte aligned rows.  src == dst (in place) is allowed. */
int lutr_apply_packed_rgb(lutr_ctx *ctx, int pfmt, int interp, int w, int h, int nframes,
                          const lutr_packed *src, const lutr_packed *dst, int row0, int rows);

/* fused YUV -> RGB -> lut3d -> RGB -> YUV; row0 and rows must be multiples of the
 * chroma block height (2 for 4:2:0) unless row0+rows == h. */
int lutr_apply_yuv(lutr_ctx *ctx, const lutr_yuv_params *p, int interp, int w, int h, int nframes,
                   const lutr_planes *src, const lutr_planes *dst, int row0, int rows);

/* zscale_dither of the reference (models.py:46; the filter `zscale=dither=error_diffusion`, ffmpeg.py:305-307) */
enum lutr_dither { LUTR_DITHER_NONE = 0, LUTR_DITHER_ERROR_DIFFUSION = 1 };

/* lutr_apply_yuv with the final quantisation dithered: Floyd-Steinberg error diffusion per output plane
 * (rows top to bottom, left to right).  Rows are coupled, so this takes whole frames only;
 * shard a batch over GPUs by frames.  dither == LUTR_DITHER_NONE is lutr_apply_yuv on rows [0, h).
 * Uses context-owned device scratch of 4 bytes per output sample of the batch. */
int lutr_apply_yuv_dither(lutr_ctx *ctx, const lutr_yuv_params *p, int interp, int dither, int w, int h, int nframes,
                          const lutr_planes *src, const lutr_planes *dst);

/* ---- precision ---- */
/* STRICT (default): every kernel is a bit-exact restatement of FFmpeg's scalar C lut3d (vf_lut3d.c order of operations,
 * no fused multiply-add in the blend).  FAST: permission to use the tolerance-bounded tile kernels -- lattice staged as
 * fp16 of value * (2^depth - 1), blend as a fused multiply-add chain with fp32 accumulation, truncation as in FFmpeg.
 * Output differs from STRICT by at most ONE code at 8 and at 10 bit (north_star allows 1 / 2 against FFmpeg).  It is
 * used where it applies (fused YUV launches on the LDS-window kernels, LUT depth 8 or 10, lattice inside [0, 1]);
 * everything else runs the strict kernels.  lutr_ctx_last_kernel() names what ran (",fast"). */
enum lutr_precision { LUTR_PRECISION_STRICT = 0, LUTR_PRECISION_FAST = 1 };
int lutr_ctx_set_precision(lutr_ctx *ctx, int precision);

/* ---- tuning / introspection (bench and tests) ---- */
/* kernel variant: 0 = auto, 1 = generic (scalar, any layout), 2 = vector + global gather,
 * 3 = vector + LDS lattice window (persistent tile kernels).  Auto picks 3 for launches of 70 Mpx and more
 * (about 8 UHD frames), 2 below that (lower latency), 1 for layouts the vector kernels cannot take; a ragged
 * width on aligned rows is split between 3 and 1.  2 and 3 fail with LUTR_EINVAL on such layouts. */
int lutr_ctx_set_variant(lutr_ctx *ctx, int variant);
/* name of the kernel variant the last apply call launched ("" before the first) */
const char *lutr_ctx_last_kernel(lutr_ctx *ctx);
/* Statistics of the LDS-window tile kernels, accumulated since the previous call.  out[8] (may be
 * NULL) = { tiles, restage attempts (tiles neither the tube nor the wave's window could take), tiles done by the
 * global-gather body, windows staged, 0, 0, tiles served by the workgroup-shared grey tube (no window needed),
 * tiles that needed the exact (second-level) window test }.
 * Then collection is enabled (and zeroed) or disabled. */
int lutr_ctx_tile_stats(lutr_ctx *ctx, int enable, uint64_t out[8]);
/* the constant block the YUV kernels use, for cross-checking against the oracle: 32 floats */
int lutr_yuv_constants(const lutr_yuv_params *p, float out[32]);

#ifdef __cplusplus
}
#endif
#endif /* LUTR_H */
