// lutr_api.cpp -- C-ABI layer of liblutr.so: contexts, lattice upload, argument
// checking, and the YUV constant block.  Kernels live in lutr_kernels.hip.
//
// Boundary being replaced: the reference spawns `ffmpeg ... -vf ...lut3d=...` per task
// (/root/reference/src/lut_renderer/task_manager.py:145-151 with the argv from
// ffmpeg.py:179-414); include/lutr.h lists which filter-string fragment each entry
// point stands in for.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lutr_internal.h"

namespace lutr {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

static int hip_fail(hipError_t e, const char *what)
{
    set_error("%s: %s", what, hipGetErrorString(e));
    return LUTR_EIO;
}

#define HIP_TRY(call) \
    do { \
        hipError_t e_ = (call); \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

// ---------------------------------------------------------------- YUV constants
static bool matrix_k(int m, double *kr, double *kb)
{
    switch (m) {
    case LUTR_MATRIX_BT709:  *kr = 0.2126; *kb = 0.0722; return true;
    case LUTR_MATRIX_BT601:  *kr = 0.299;  *kb = 0.114;  return true;
    case LUTR_MATRIX_BT2020: *kr = 0.2627; *kb = 0.0593; return true;
    }
    return false;
}

// DESIGN.md "YUV contract".  All coefficients are formed in double and rounded to
// float once; the kernels then use exactly these floats.
int make_yuv_consts(const lutr_yuv_params &p, YuvConsts *o)
{
    const int din = LUTR_FMT_DEPTH(p.fmt_in), dout = LUTR_FMT_DEPTH(p.fmt_out), dl = p.lut_depth;
    if (din < 8 || din > 16 || dout < 8 || dout > 16 || dl < 8 || dl > 16) {
        set_error("unsupported bit depth (in %d, lut %d, out %d)", din, dl, dout);
        return LUTR_EINVAL;
    }
    if (LUTR_FMT_CSX(p.fmt_in) != LUTR_FMT_CSX(p.fmt_out) || LUTR_FMT_CSY(p.fmt_in) != LUTR_FMT_CSY(p.fmt_out)) {
        set_error("fmt_in and fmt_out must share chroma subsampling");
        return LUTR_EINVAL;
    }
    if (LUTR_FMT_CSX(p.fmt_in) == 0 && LUTR_FMT_CSY(p.fmt_in) == 1) {
        set_error("4:4:0 chroma layout is not supported");
        return LUTR_EINVAL;
    }
    auto range_ok = [](int r) { return r == LUTR_RANGE_TV || r == LUTR_RANGE_PC; };
    if (!range_ok(p.range_src) || !range_ok(p.range_in) || !range_ok(p.range_out)) {
        set_error("bad range value");
        return LUTR_EINVAL;
    }
    const bool prologue = (p.range_src != p.range_in) || (din != dl);
    if (prologue && p.range_src != LUTR_RANGE_PC) {
        // the reference only emits the prologue for full-range sources (ffmpeg.py:129-134, :212)
        set_error("a range/depth prologue is only defined for full-range (pc) sources");
        return LUTR_EINVAL;
    }
    double kr, kb;
    if (!matrix_k(p.matrix_in, &kr, &kb)) {
        set_error("bad matrix_in %d", p.matrix_in);
        return LUTR_EINVAL;
    }
    std::memset(o, 0, sizeof(*o));
    const int chroma_n = 1 << (LUTR_FMT_CSX(p.fmt_in) + LUTR_FMT_CSY(p.fmt_in));

    if (prologue) {
        const double mi = (double)((1 << din) - 1);
        const double sl = (double)(1 << (dl - 8));
        const double ml = (double)((1 << dl) - 1);
        const double half_in = (double)(1 << (din - 1));
        double py, pyo, pc, pco;
        if (p.range_in == LUTR_RANGE_TV) {
            py = 219.0 * sl / mi;  pyo = 16.0 * sl;
            pc = 224.0 * sl / mi;  pco = 128.0 * sl - half_in * pc;
        } else {
            py = ml / mi;          pyo = 0.0;
            pc = ml / mi;          pco = 128.0 * sl - half_in * pc;
        }
        o->pre = 1.0f;
        o->py = (float)py;  o->pyb = (float)(pyo + 0.5);
        o->pc = (float)pc;  o->pcb = (float)(pco + 0.5);
        o->pre_max = (float)ml;
    }
    {
        const double kg = 1.0 - kr - kb;
        const double s = (double)(1 << (dl - 8));
        const double m = (double)((1 << dl) - 1);
        double ky, yoff, kc;
        if (p.range_in == LUTR_RANGE_TV) { ky = m / (219.0 * s); yoff = 16.0 * s; kc = m / (224.0 * s); }
        else { ky = 1.0; yoff = 0.0; kc = 1.0; }
        o->ky = (float)ky;
        o->yb = (float)(-ky * yoff + 0.5);
        o->coff = (float)(128.0 * s);
        o->krv = (float)(2.0 * (1.0 - kr) * kc);
        o->kbu = (float)(2.0 * (1.0 - kb) * kc);
        o->kgu = (float)(-2.0 * kb * (1.0 - kb) / kg * kc);
        o->kgv = (float)(-2.0 * kr * (1.0 - kr) / kg * kc);
        o->max_l = (float)m;
    }
    if (!matrix_k(p.matrix_out, &kr, &kb)) {
        set_error("bad matrix_out %d", p.matrix_out);
        return LUTR_EINVAL;
    }
    {
        const double kg = 1.0 - kr - kb;
        const double so = (double)(1 << (dout - 8));
        const double mo = (double)((1 << dout) - 1);
        const double ml = (double)((1 << dl) - 1);
        const double n = (double)chroma_n;
        double ys, yoff, cs;
        if (p.range_out == LUTR_RANGE_TV) { ys = 219.0 * so; yoff = 16.0 * so; cs = 224.0 * so; }
        else { ys = mo; yoff = 0.0; cs = mo; }
        o->cyr = (float)(ys * kr / ml);
        o->cyg = (float)(ys * kg / ml);
        o->cyb = (float)(ys * kb / ml);
        o->yob = (float)(yoff + 0.5);
        o->cbr = (float)(cs * (-kr / (2.0 * (1.0 - kb))) / ml / n);
        o->cbg = (float)(cs * (-kg / (2.0 * (1.0 - kb))) / ml / n);
        o->cbb = (float)(cs * 0.5 / ml / n);
        o->crr = (float)(cs * 0.5 / ml / n);
        o->crg = (float)(cs * (-kg / (2.0 * (1.0 - kr))) / ml / n);
        o->crb = (float)(cs * (-kb / (2.0 * (1.0 - kr))) / ml / n);
        o->cob = (float)(128.0 * so + 0.5);
        o->max_o = (float)mo;
    }
    return LUTR_OK;
}

}  // namespace lutr

using namespace lutr;

struct lutr_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    float4 *lat = nullptr;
    size_t lat_bytes = 0;
    int n = 0;
    float scale[3] = {1.f, 1.f, 1.f};
    int variant = VAR_AUTO;
    std::string last_kernel;
    bool unit = false;               // every lattice node known to lie in [0, 1]
    unsigned *queue = nullptr;       // work-queue words of the tile kernels (device, 4 words: see lutr_ctx_create)
    float *fscratch = nullptr;       // float planes of the dither path
    size_t fscratch_floats = 0;
    unsigned *stats = nullptr;       // 8 device counters (4 reported + clock stamps), see lutr_ctx_tile_stats
    // The queue counter, the dither scratch and the stats block are per context, not per stream: launches of one
    // context must not overlap.  Every launch records `done` on the stream it ran on; binding another stream makes
    // that stream wait for it (lutr_ctx_set_stream), so applies issued from different streams serialise on the GPU.
    hipEvent_t done = nullptr;
    bool pending = false;            // `done` was recorded on `stream` and nothing has waited for it yet
    // fast variant (lutr_ctx_set_precision): fp16 copies of the lattice pre-multiplied by 2^depth - 1, built on first
    // use per depth (index 0: 8 bit, 1: 10 bit) and dropped whenever the lattice changes
    int precision = LUTR_PRECISION_STRICT;
    uint2 *lat16[2] = {nullptr, nullptr};
    // lut3d's prelut (lutr_ctx_set_prelut): the host copy, and per LUT depth a device table of the lattice coordinate of every
    // integer code -- the shaper, the scale and the clip folded into one lookup (pre_dev[depth - 8], 3 x pre_entries floats)
    std::vector<float> prelut;
    int pre_size = 0;
    float pre_min[3] = {0, 0, 0}, pre_scale[3] = {0, 0, 0};
    float *pre_dev[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int    pre_shared[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};     // per depth: the three tables agree and never fall (LutConsts::pre_shared)
    float  pre_kappa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // ... and their largest step between neighbouring codes
    std::vector<float> pre_host[9];                         // ... and the shared table itself, codes 0 .. 2^depth - 1 (LutConsts::pre_host)
    // lutr_lut_broadcast: copies other contexts are still reading out of THIS context's lattice (one event per receiver,
    // recorded on the receiver's stream behind its copy).  The lattice must not be overwritten or freed before they finish.
    std::vector<std::pair<int, hipEvent_t>> readers;      // (receiver's device, event)
};

// Wait for every peer copy that reads this context's lattice, then forget the events.
static void wait_readers(lutr_ctx *c)
{
    for (auto &r : c->readers) {
        (void)hipEventSynchronize(r.second);
        (void)hipSetDevice(r.first);
        (void)hipEventDestroy(r.second);
    }
    if (!c->readers.empty()) (void)hipSetDevice(c->device);
    c->readers.clear();
}

static void drop_lat16(lutr_ctx *c)
{
    for (auto &p : c->lat16)
        if (p) { (void)hipStreamSynchronize(c->stream); (void)hipFree(p); p = nullptr; }
}

static void drop_prelut_tables(lutr_ctx *c)
{
    for (auto &p : c->pre_dev)
        if (p) { (void)hipStreamSynchronize(c->stream); (void)hipFree(p); p = nullptr; }
}

extern "C" {

const char *lutr_version(void) { return LUTR_VERSION_STRING; }
const char *lutr_last_error(void) { return g_last_error.c_str(); }

size_t lutr_lattice_bytes(int n)
{
    if (n < 2 || n > 256) return 0;
    const size_t n1 = (size_t)n + 1;
    return n1 * n1 * n1 * sizeof(float4);
}

int lutr_yuv_constants(const lutr_yuv_params *p, float out[32])
{
    if (!p || !out) {
        set_error("lutr_yuv_constants: null argument");
        return LUTR_EINVAL;
    }
    YuvConsts k;
    const int rc = make_yuv_consts(*p, &k);
    if (rc) return rc;
    std::memcpy(out, &k, sizeof(k));
    return LUTR_OK;
}

int lutr_ctx_create(int device, lutr_ctx **out)
{
    if (!out) {
        set_error("lutr_ctx_create: null out pointer");
        return LUTR_EINVAL;
    }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no HIP device available (%s); liblutr has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return LUTR_EIO;
    }
    if (device < 0 || device >= count) {
        set_error("device %d out of range (have %d)", device, count);
        return LUTR_EINVAL;
    }
    HIP_TRY(hipSetDevice(device));
    lutr_ctx *c = new lutr_ctx();
    c->device = device;
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return hip_fail(e, "hipStreamCreateWithFlags");
    }
    c->stream = c->own_stream;
    // four words: {claims, waves done} of the round-3 tile kernels, which return them to zero themselves at the end of every launch
    // (no memset node per launch), word 2 for round 1's RGB kernel (set by its launcher), one spare
    e = hipMalloc((void **)&c->queue, 4 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(c->queue, 0, 4 * sizeof(unsigned));
    if (e != hipSuccess) {
        if (c->queue) (void)hipFree(c->queue);
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return hip_fail(e, "hipMalloc(queue)");
    }
    e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)hipFree(c->queue);
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return hip_fail(e, "hipEventCreateWithFlags");
    }
    *out = c;
    return LUTR_OK;
}

void lutr_ctx_destroy(lutr_ctx *c)
{
    if (!c) return;
    wait_readers(c);
    (void)hipSetDevice(c->device);
    drop_lat16(c);
    drop_prelut_tables(c);
    if (c->lat) (void)hipFree(c->lat);
    if (c->stats) (void)hipFree(c->stats);
    if (c->fscratch) (void)hipFree(c->fscratch);
    if (c->queue) (void)hipFree(c->queue);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int lutr_ctx_set_stream(lutr_ctx *c, void *hip_stream)
{
    if (!c) { set_error("null context"); return LUTR_EINVAL; }
    hipStream_t next = (hipStream_t)hip_stream;   // NULL is HIP's default (null) stream, e.g. torch's default
    if (next != c->stream && c->pending) {
        // work of this context may still be running on the old stream, and it owns the context's queue counter and
        // scratch: the new stream starts behind it
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamWaitEvent(next, c->done, 0));
    }
    c->stream = next;
    return LUTR_OK;
}

int lutr_ctx_sync(lutr_ctx *c)
{
    if (!c) { set_error("null context"); return LUTR_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LUTR_OK;
}

int lutr_ctx_set_precision(lutr_ctx *c, int precision)
{
    if (!c || (precision != LUTR_PRECISION_STRICT && precision != LUTR_PRECISION_FAST)) {
        set_error("bad precision %d", precision);
        return LUTR_EINVAL;
    }
    c->precision = precision;
    return LUTR_OK;
}

int lutr_ctx_set_variant(lutr_ctx *c, int variant)
{
    if (!c || variant < VAR_AUTO || variant > VAR_VEC_LDS) {
        set_error("bad variant %d", variant);
        return LUTR_EINVAL;
    }
    c->variant = variant;
    return LUTR_OK;
}

const char *lutr_ctx_last_kernel(lutr_ctx *c) { return c ? c->last_kernel.c_str() : ""; }

int lutr_ctx_tile_stats(lutr_ctx *c, int enable, uint64_t out[8])
{
    if (!c) { set_error("null context"); return LUTR_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (out) {
        for (int i = 0; i < 8; i++) out[i] = 0;
        if (c->stats) {
            unsigned h[32];
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipMemcpy(h, c->stats, sizeof(h), hipMemcpyDeviceToHost));
            for (int i = 0; i < 4; i++) out[i] = h[i];
            out[4] = h[30];                                         // mixed tiles: tube body + gather body for the few lanes outside the tube
            out[5] = 0;
            out[6] = h[12];                                         // tiles served by the workgroup's grey tube
            out[7] = h[6];                                          // tiles that needed the second-level (exact) window test
            if (getenv("LUTR_DEBUG"))
            {
                fprintf(stderr, "[lutr stats raw] %u %u %u %u | %u %u %u | %u %u %u | %u %u |", h[0], h[1], h[2], h[3], h[4], h[5], h[6],
                        h[7], h[8], h[9], h[10], h[11]);
                for (int i = 12; i < 32; i++) fprintf(stderr, " %u", h[i]);
                fprintf(stderr, "\n");
            }
        }
    }
    if (enable && !c->stats) {
        HIP_TRY(hipMalloc((void **)&c->stats, 32 * sizeof(unsigned)));
    } else if (!enable && c->stats) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipFree(c->stats);
        c->stats = nullptr;
    }
    if (c->stats) HIP_TRY(hipMemset(c->stats, 0, 32 * sizeof(unsigned)));
    return LUTR_OK;
}

static int alloc_lattice(lutr_ctx *c, int n, const float scale[3])
{
    if (!c || !scale) { set_error("null argument"); return LUTR_EINVAL; }
    if (n < 2 || n > 256) {
        set_error("too large or invalid 3D LUT size %d", n);
        return LUTR_EINVAL;
    }
    for (int i = 0; i < 3; i++)
        if (!(scale[i] >= 0.f && scale[i] <= 1.f)) {
            set_error("scale[%d] = %g outside [0,1]", i, (double)scale[i]);
            return LUTR_EINVAL;
        }
    HIP_TRY(hipSetDevice(c->device));
    wait_readers(c);                 // peers of an earlier lutr_lut_broadcast may still be copying out of the buffer
    drop_lat16(c);                   // they describe the previous lattice
    drop_prelut_tables(c);           // and so does a prelut: it belongs to the LUT file (set it again after the lattice)
    c->prelut.clear(); c->pre_size = 0;
    const size_t bytes = lutr_lattice_bytes(n);
    if (bytes != c->lat_bytes) {
        if (c->lat) { HIP_TRY(hipStreamSynchronize(c->stream)); (void)hipFree(c->lat); c->lat = nullptr; c->lat_bytes = 0; }
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return LUTR_ENOMEM; }
        c->lat = (float4 *)p;
        c->lat_bytes = bytes;
    }
    c->n = n;
    c->unit = false;                 // unknown until the nodes are seen (lutr_ctx_set_lut / lutr_ctx_lut_seal)
    std::memcpy(c->scale, scale, sizeof(c->scale));
    return LUTR_OK;
}

int lutr_ctx_lut_alloc(lutr_ctx *c, int n, const float scale[3]) { return alloc_lattice(c, n, scale); }

int lutr_ctx_set_lut(lutr_ctx *c, const float *rgb, int n, const float scale[3])
{
    if (!rgb) { set_error("null lattice"); return LUTR_EINVAL; }
    const size_t count = (size_t)(n > 0 ? n : 0) * n * n * 3;
    bool unit = true;
    for (size_t i = 0; i < count && n >= 2 && n <= 256; i++) {
        if (!std::isfinite(rgb[i])) {
            set_error("non-finite lattice value at float %zu", i);
            return LUTR_EINVAL;
        }
        unit = unit && rgb[i] >= 0.0f && rgb[i] <= 1.0f;
    }
    const int rc = alloc_lattice(c, n, scale);
    if (rc) return rc;
    // pack [r][g][b][3] -> (n+1)^3 float4 with the last node replicated on each axis
    const int n1 = n + 1;
    std::vector<float4> host((size_t)n1 * n1 * n1);
    for (int r = 0; r < n1; r++) {
        const int rr = r < n ? r : n - 1;
        for (int g = 0; g < n1; g++) {
            const int gg = g < n ? g : n - 1;
            for (int b = 0; b < n1; b++) {
                const int bb = b < n ? b : n - 1;
                const float *s = &rgb[(((size_t)rr * n + gg) * n + bb) * 3];
                host[((size_t)r * n1 + g) * n1 + b] = make_float4(s[0], s[1], s[2], 0.f);
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(c->lat, host.data(), c->lat_bytes, hipMemcpyHostToDevice));
    c->unit = unit;
    return LUTR_OK;
}

int lutr_ctx_set_prelut(lutr_ctx *c, const float *prelut, int size, const float vmin[3], const float vscale[3])
{
    if (!c) { set_error("null context"); return LUTR_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    drop_prelut_tables(c);
    c->prelut.clear(); c->pre_size = 0;
    if (!prelut || size == 0) return LUTR_OK;
    if (size < 2 || size > 65536 || !vmin || !vscale) { set_error("prelut size %d outside [2, 65536] or null ranges", size); return LUTR_EINVAL; }
    for (size_t i = 0; i < (size_t)3 * size; i++)
        if (!std::isfinite(prelut[i])) { set_error("non-finite prelut value at float %zu", i); return LUTR_EINVAL; }
    for (int i = 0; i < 3; i++)
        if (!std::isfinite(vmin[i]) || !std::isfinite(vscale[i])) { set_error("non-finite prelut range"); return LUTR_EINVAL; }
    c->prelut.assign(prelut, prelut + (size_t)3 * size);
    c->pre_size = size;
    std::memcpy(c->pre_min, vmin, sizeof(c->pre_min));
    std::memcpy(c->pre_scale, vscale, sizeof(c->pre_scale));
    return LUTR_OK;
}

// The lattice coordinate of every integer code at this LUT depth with the prelut in front: FFmpeg's
// prelut_interp_1d_linear on code * (1 / M), then * scale * (n - 1), clipped to [0, n - 1] -- per pixel in FFmpeg, per code here,
// float for float the same operations (this file is compiled without contraction).  256 entries for 8-bit containers, else 65536.
static int prelut_table(lutr_ctx *c, int depth, const float **dev, int *entries, int *shared, float *kappa, const float **host_tab)
{
    *dev = nullptr; *entries = 0; *shared = 0; *kappa = 0.0f; *host_tab = nullptr;
    if (!c->pre_size) return LUTR_OK;
    const int slot = depth - 8;
    if (slot < 0 || slot > 8) { set_error("prelut: LUT depth %d outside 8..16", depth); return LUTR_EINVAL; }
    const int ne = depth <= 8 ? 256 : 65536;
    *entries = ne;
    if (c->pre_dev[slot]) {
        *dev = c->pre_dev[slot]; *shared = c->pre_shared[slot]; *kappa = c->pre_kappa[slot];
        *host_tab = c->pre_shared[slot] ? c->pre_host[slot].data() : nullptr;
        return LUTR_OK;
    }
    const int maxi = (1 << depth) - 1, pmax = c->pre_size - 1;
    const float scale_f = 1.0f / (float)maxi, lut_max = (float)(c->n - 1);
    std::vector<float> host((size_t)3 * ne);
    for (int ch = 0; ch < 3; ch++) {
        const float sc = c->scale[ch] * lut_max;
        const float *tab = &c->prelut[(size_t)ch * c->pre_size];
        for (int code = 0; code < ne; code++) {
            const float s = (float)code * scale_f;
            const float scaled = (s - c->pre_min[ch]) * c->pre_scale[ch];
            const float x = scaled < 0.0f ? 0.0f : (scaled > (float)pmax ? (float)pmax : scaled);
            const int prev = (int)x, next = (prev + 1) < pmax ? prev + 1 : pmax;
            const float p = tab[prev], nn = tab[next], d = x - (float)prev;
            const float v = p + (nn - p) * d;
            const float t = v * sc;
            host[(size_t)ch * ne + code] = t < 0.0f ? 0.0f : (t > lut_max ? lut_max : t);
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, host.size() * sizeof(float));
    if (e != hipSuccess) { set_error("hipMalloc(prelut table): %s", hipGetErrorString(e)); return LUTR_ENOMEM; }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    c->pre_dev[slot] = (float *)p;
    *dev = c->pre_dev[slot];
    // one table for the three channels?  (what the fused YUV tile kernels can take: their coordinate table is indexed by code alone,
    // and their validity bounds want a monotone map with a known largest slope)
    bool same = true;
    float step = 0.0f;
    for (int code = 0; code <= maxi && same; code++) {
        const float v = host[code];
        same = host[(size_t)ne + code] == v && host[(size_t)2 * ne + code] == v;
        if (code) { const float d = v - host[code - 1]; if (d < 0.0f) same = false; else if (d > step) step = d; }
    }
    c->pre_shared[slot] = same ? 1 : 0;
    c->pre_kappa[slot] = same ? step : 0.0f;
    c->pre_host[slot].clear();
    if (same) c->pre_host[slot].assign(host.begin(), host.begin() + maxi + 1);
    *shared = c->pre_shared[slot]; *kappa = c->pre_kappa[slot];
    *host_tab = same ? c->pre_host[slot].data() : nullptr;
    return LUTR_OK;
}

int lutr_ctx_lut_seal(lutr_ctx *c)
{
    if (!c) { set_error("null argument"); return LUTR_EINVAL; }
    if (!c->lat) { set_error("no lattice set on this context"); return LUTR_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    std::vector<float4> host(c->lat_bytes / sizeof(float4));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(host.data(), c->lat, c->lat_bytes, hipMemcpyDeviceToHost));
    bool unit = true;
    for (size_t i = 0; i < host.size(); i++) {
        const float v[3] = {host[i].x, host[i].y, host[i].z};
        for (int k = 0; k < 3; k++) {
            if (!std::isfinite(v[k])) {
                c->unit = false;
                set_error("non-finite lattice value at node %zu", i);
                return LUTR_EINVAL;
            }
            unit = unit && v[k] >= 0.0f && v[k] <= 1.0f;
        }
    }
    drop_lat16(c);
    c->unit = unit;
    return LUTR_OK;
}

int lutr_lut_broadcast(lutr_ctx **ctxs, int nctx, int root) { return lutr_lut_broadcast_ex(ctxs, nctx, root, 0); }

int lutr_lut_broadcast_ex(lutr_ctx **ctxs, int nctx, int root, unsigned flags)
{
    if (!ctxs || nctx < 1 || root < 0 || root >= nctx || (flags & ~(unsigned)LUTR_BCAST_FORCE_PEER_COPY)) {
        set_error("lutr_lut_broadcast: bad arguments");
        return LUTR_EINVAL;
    }
    for (int i = 0; i < nctx; i++)
        if (!ctxs[i]) { set_error("lutr_lut_broadcast: null context %d", i); return LUTR_EINVAL; }
    lutr_ctx *r = ctxs[root];
    if (!r->lat) { set_error("the root context holds no lattice"); return LUTR_EINVAL; }
    // the root's upload (a blocking copy) has landed; order the peers' copies behind whatever the root's stream still runs
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipEventRecord(r->done, r->stream));
    r->pending = true;
    for (int i = 0; i < nctx; i++) {
        lutr_ctx *c = ctxs[i];
        if (c == r) continue;
        int rc = alloc_lattice(c, r->n, r->scale);        // selects c's device
        if (rc) return rc;
        HIP_TRY(hipStreamWaitEvent(c->stream, r->done, 0));
        // (LUTR_BCAST_FORCE_PEER_COPY: the cross-device call on a same-device pair -- a self-peer copy is legal -- so that a
        // one-GPU box executes the branch an 8-GPU node takes)
        if (c->device == r->device && !(flags & LUTR_BCAST_FORCE_PEER_COPY))
            HIP_TRY(hipMemcpyAsync(c->lat, r->lat, r->lat_bytes, hipMemcpyDeviceToDevice, c->stream));
        else
            HIP_TRY(hipMemcpyPeerAsync(c->lat, c->device, r->lat, r->device, r->lat_bytes, c->stream));   // xGMI, GPU to GPU
        c->unit = r->unit;           // same nodes: the root's scan of the value range holds for the copy
        // later applies of the receiver are ordered behind its copy even if it is rebound to another stream first
        HIP_TRY(hipEventRecord(c->done, c->stream));
        c->pending = true;
        // ... and the root must not overwrite or free the buffer while this copy reads it
        hipEvent_t ev = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev, c->stream));
        r->readers.emplace_back(c->device, ev);
    }
    HIP_TRY(hipSetDevice(r->device));
    return LUTR_OK;
}

int lutr_ctx_lut_device(lutr_ctx *c, void **dptr, size_t *bytes)
{
    if (!c || !dptr || !bytes) { set_error("null argument"); return LUTR_EINVAL; }
    if (!c->lat) { set_error("no lattice set on this context"); return LUTR_EINVAL; }
    *dptr = c->lat;
    *bytes = c->lat_bytes;
    return LUTR_OK;
}

static int check_common(lutr_ctx *c, int interp, int w, int h, int nframes, const void *src,
                        const void *dst, int row0, int rows)
{
    if (!c || !src || !dst) { set_error("null argument"); return LUTR_EINVAL; }
    if (!c->lat) { set_error("no lattice set on this context (call lutr_ctx_set_lut first)"); return LUTR_EINVAL; }
    if (interp < LUTR_INTERP_NEAREST || interp > LUTR_INTERP_PRISM) {
        set_error("unknown interpolation mode %d", interp);
        return LUTR_EINVAL;
    }
    if (w < 0 || h < 0 || nframes < 0 || row0 < 0 || rows < 0 || row0 + rows > h) {
        set_error("bad geometry w=%d h=%d nframes=%d row0=%d rows=%d", w, h, nframes, row0, rows);
        return LUTR_EINVAL;
    }
    return LUTR_OK;
}

static void fill_planes(PlaneSet *P, const lutr_planes *src, const lutr_planes *dst)
{
    for (int i = 0; i < 3; i++) {
        P->s[i] = (const uint8_t *)src->data[i];
        P->d[i] = (uint8_t *)dst->data[i];
        P->ss[i] = src->stride[i];
        P->ds[i] = dst->stride[i];
        P->sfs[i] = src->frame_stride[i];
        P->dfs[i] = dst->frame_stride[i];
    }
}

// The fast variant's lattice for `depth`, or nullptr when fast does not apply: strict precision selected, a depth
// other than 8 / 10 (the tolerance is only defined there), or a lattice outside [0, 1] (the fast kernels are clip-free).
static const uint2 *fast_lattice(lutr_ctx *c, int depth)
{
    if (c->precision != LUTR_PRECISION_FAST || !c->unit || (depth != 8 && depth != 10)) return nullptr;
    uint2 *&slot = c->lat16[depth == 8 ? 0 : 1];
    if (!slot) {
        const size_t nodes = c->lat_bytes / sizeof(float4);
        void *p = nullptr;
        if (hipMalloc(&p, nodes * sizeof(uint2)) != hipSuccess) return nullptr;
        slot = (uint2 *)p;
        launch_make_lat16(c->stream, c->lat, slot, nodes, (float)((1 << depth) - 1));   // same stream as the apply that follows
    }
    return slot;
}

static int fill_lut(LutConsts *L, lutr_ctx *c, int depth)
{
    const int maxi = (1 << depth) - 1;
    const int rc = prelut_table(c, depth, &L->pre, &L->pre_stride, &L->pre_shared, &L->pre_kappa, &L->pre_host);
    if (rc) return rc;
    L->lat = c->lat;
    L->lat16 = nullptr;
    L->n1 = c->n + 1;
    L->maxf = (float)maxi;
    L->unit = c->unit ? 1 : 0;
    L->scale_f = 1.0f / (float)maxi;
    L->lut_max = (float)(c->n - 1);
    for (int i = 0; i < 3; i++) L->sc[i] = c->scale[i] * L->lut_max;
    return LUTR_OK;
}

static int finish_launch(lutr_ctx *c, const char *name)
{
    if (!name) {
        set_error("the requested kernel variant cannot take this layout (alignment, width multiple, depth mix or mode)");
        return LUTR_EINVAL;
    }
    c->last_kernel = name;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, name);
    HIP_TRY(hipEventRecord(c->done, c->stream));
    c->pending = true;
    return LUTR_OK;
}

int lutr_apply_planar_rgb(lutr_ctx *c, int depth, int interp, int w, int h, int nframes,
                          const lutr_planes *src, const lutr_planes *dst, int row0, int rows)
{
    int rc = check_common(c, interp, w, h, nframes, src, dst, row0, rows);
    if (rc) return rc;
    if (depth < 8 || depth > 16) { set_error("unsupported depth %d", depth); return LUTR_EINVAL; }
    if (w == 0 || rows == 0 || nframes == 0) return LUTR_OK;
    for (int i = 0; i < 3; i++)
        if (!src->data[i] || !dst->data[i]) { set_error("null plane %d", i); return LUTR_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    LutConsts L; PlaneSet P; FrameGeom G{w, h, row0, rows, nframes};
    if (const int rc = fill_lut(&L, c, depth)) return rc;
    fill_planes(&P, src, dst);
    return finish_launch(c, launch_rgb(c->stream, c->variant, L, P, G, depth, interp, c->stats, c->queue));
}

int lutr_apply_packed_rgb(lutr_ctx *c, int pfmt, int interp, int w, int h, int nframes,
                          const lutr_packed *src, const lutr_packed *dst, int row0, int rows)
{
    int rc = check_common(c, interp, w, h, nframes, src, dst, row0, rows);
    if (rc) return rc;
    const int bits = LUTR_PACKED_BITS(pfmt), nc = LUTR_PACKED_NCOMP(pfmt);
    const int ro = LUTR_PACKED_RO(pfmt), go = LUTR_PACKED_GO(pfmt), bo = LUTR_PACKED_BO(pfmt);
    if ((bits != 8 && bits != 16) || (nc != 3 && nc != 4) || (pfmt >> 24) || ro >= nc || go >= nc || bo >= nc ||
        ro == go || go == bo || ro == bo) {
        set_error("unsupported packed format 0x%x", pfmt);
        return LUTR_EINVAL;
    }
    if (w == 0 || rows == 0 || nframes == 0) return LUTR_OK;
    if (!src->data || !dst->data) { set_error("null image"); return LUTR_EINVAL; }
    const int wide = bits == 16;
    if (wide && (((uintptr_t)src->data | (uintptr_t)dst->data | (uintptr_t)src->stride | (uintptr_t)dst->stride |
                  (nframes > 1 ? (uintptr_t)src->frame_stride | (uintptr_t)dst->frame_stride : 0)) & 1)) {
        set_error("16-bit packed formats need 2-byte aligned rows");
        return LUTR_EINVAL;
    }
    HIP_TRY(hipSetDevice(c->device));
    LutConsts L; FrameGeom G{w, h, row0, rows, nframes};
    if (const int rc = fill_lut(&L, c, bits)) return rc;
    PackedSet P;
    P.s = (const uint8_t *)src->data; P.d = (uint8_t *)dst->data;
    P.ss = src->stride; P.ds = dst->stride;
    P.sfs = src->frame_stride; P.dfs = dst->frame_stride;
    P.ro = ro; P.go = go; P.bo = bo; P.ao = nc == 4 ? 6 - ro - go - bo : 3;
    return finish_launch(c, launch_packed(c->stream, c->variant, L, P, G, wide, nc, interp, c->stats, c->queue));
}

int lutr_apply_yuv(lutr_ctx *c, const lutr_yuv_params *p, int interp, int w, int h, int nframes,
                   const lutr_planes *src, const lutr_planes *dst, int row0, int rows)
{
    int rc = check_common(c, interp, w, h, nframes, src, dst, row0, rows);
    if (rc) return rc;
    if (!p) { set_error("null yuv params"); return LUTR_EINVAL; }
    YuvConsts K;
    rc = make_yuv_consts(*p, &K);
    if (rc) return rc;
    const int csx = LUTR_FMT_CSX(p->fmt_in), csy = LUTR_FMT_CSY(p->fmt_in);
    const int bh = 1 << csy;
    if (row0 % bh || (rows % bh && row0 + rows != h)) {
        set_error("row0/rows must be multiples of the chroma block height %d", bh);
        return LUTR_EINVAL;
    }
    if (w == 0 || rows == 0 || nframes == 0) return LUTR_OK;
    for (int i = 0; i < 3; i++)
        if (!src->data[i] || !dst->data[i]) { set_error("null plane %d", i); return LUTR_EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    LutConsts L; PlaneSet P; FrameGeom G{w, h, row0, rows, nframes};
    if (const int rc = fill_lut(&L, c, p->lut_depth)) return rc;
    L.lat16 = fast_lattice(c, p->lut_depth);
    fill_planes(&P, src, dst);
    return finish_launch(c, launch_yuv(c->stream, c->variant, L, K, P, G, LUTR_FMT_DEPTH(p->fmt_in),
                                       LUTR_FMT_DEPTH(p->fmt_out), p->lut_depth, csx, csy, interp, L.lat16 != nullptr,
                                       c->stats, c->queue));
}

int lutr_apply_yuv_dither(lutr_ctx *c, const lutr_yuv_params *p, int interp, int dither, int w, int h, int nframes,
                          const lutr_planes *src, const lutr_planes *dst)
{
    if (dither == LUTR_DITHER_NONE) return lutr_apply_yuv(c, p, interp, w, h, nframes, src, dst, 0, h);
    if (dither != LUTR_DITHER_ERROR_DIFFUSION) { set_error("unknown dither mode %d", dither); return LUTR_EINVAL; }
    int rc = check_common(c, interp, w, h, nframes, src, dst, 0, h);
    if (rc) return rc;
    if (!p) { set_error("null yuv params"); return LUTR_EINVAL; }
    YuvConsts K;
    rc = make_yuv_consts(*p, &K);
    if (rc) return rc;
    if (w == 0 || h == 0 || nframes == 0) return LUTR_OK;
    for (int i = 0; i < 3; i++)
        if (!src->data[i] || !dst->data[i]) { set_error("null plane %d", i); return LUTR_EINVAL; }
    const int csx = LUTR_FMT_CSX(p->fmt_in), csy = LUTR_FMT_CSY(p->fmt_in);
    const size_t cw = (size_t)((w + (1 << csx) - 1) >> csx), ch = (size_t)((h + (1 << csy) - 1) >> csy);
    const size_t ny = (size_t)w * h * nframes, nc = cw * ch * nframes;
    HIP_TRY(hipSetDevice(c->device));
    if (ny + 2 * nc > c->fscratch_floats) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->fscratch) (void)hipFree(c->fscratch);
        c->fscratch = nullptr;
        c->fscratch_floats = 0;
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, (ny + 2 * nc) * sizeof(float));
        if (e != hipSuccess) { set_error("hipMalloc(%zu): %s", (ny + 2 * nc) * sizeof(float), hipGetErrorString(e)); return LUTR_ENOMEM; }
        c->fscratch = (float *)q;
        c->fscratch_floats = ny + 2 * nc;
    }
    LutConsts L; PlaneSet P; FrameGeom G{w, h, 0, h, nframes};
    if (const int rc = fill_lut(&L, c, p->lut_depth)) return rc;
    fill_planes(&P, src, dst);
    FloatPlanes F{c->fscratch, c->fscratch + ny, c->fscratch + ny + nc};
    return finish_launch(c, launch_yuv_dither(c->stream, L, K, P, G, F, LUTR_FMT_DEPTH(p->fmt_in),
                                              LUTR_FMT_DEPTH(p->fmt_out), csx, csy, interp));
}

}  // extern "C"
