// lut_formats.cpp -- the other 3D LUT files FFmpeg's lut3d reads besides .cube:
// .dat (DaVinci), .3dl (Autodesk/Lustre, fixed 17^3, 12-bit integers), .m3d (Pandora) and .csp
// (cineSpace, without a pre-LUT shaper).  lutr_lut_parse picks the reader from the file extension the
// way lut3d's file= option does (the reference passes the path through at ffmpeg.py:246; its GUI only
// offers *.cube, lut_manager.py:121, so these are breadth, not the hot path).
//
// Semantics follow FFmpeg's parse_dat / parse_3dl / parse_m3d / parse_cinespace [FFmpeg-recall,
// libavfilter/vf_lut3d.c]; the readers are written as a small record scanner, independently of the
// oracle's loop-for-loop restatement (oracle/lut3d_oracle.c) so the tests compare two implementations.
// All four store the lattice blue-fastest ((r*n+g)*n+b); scale is 1 except for .csp input ranges.
// Like the .cube reader, non-finite entries are rejected.
#include <algorithm>
#include <cctype>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lutr_internal.h"

namespace {

constexpr int kMaxLine = 512;
constexpr int kMaxLevel = 256;

struct Lines {
    FILE *f = nullptr;
    char buf[kMaxLine];
    ~Lines() { if (f) std::fclose(f); }
    bool raw() { return std::fgets(buf, sizeof(buf), f) != nullptr; }
    static bool skippable(const char *s)
    {
        while (*s && std::isspace((unsigned char)*s)) s++;
        return *s == 0 || *s == '#';
    }
    // next line that is neither blank nor a '#' comment
    bool record()
    {
        while (raw())
            if (!skippable(buf)) return true;
        return false;
    }
    bool starts(const char *prefix) const { return std::strncmp(buf, prefix, std::strlen(prefix)) == 0; }
};

std::string lower_ext(const char *path)
{
    const char *dot = std::strrchr(path, '.');
    std::string ext(dot ? dot + 1 : "");
    for (auto &ch : ext) ch = (char)std::tolower((unsigned char)ch);
    return ext;
}

int fail(int code, const char *path, const char *what)
{
    lutr::set_error("'%s': %s", path, what);
    return code;
}

bool finite3(const float *v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }

// .dat: optional "3DLUTSIZE n" (default 33), then n^3 float triplets, blue fastest
int read_dat(Lines &in, const char *path, std::vector<float> &tab, int &n)
{
    if (!in.record()) return fail(LUTR_EILSEQ, path, "unexpected EOF");
    n = 33;
    if (in.starts("3DLUTSIZE ")) {
        n = (int)std::strtol(in.buf + 10, nullptr, 0);
        if (n < 2 || n > kMaxLevel) return fail(LUTR_EINVAL, path, "too large or invalid 3D LUT size");
        if (!in.record()) return fail(LUTR_EILSEQ, path, "unexpected EOF");
    }
    const size_t count = (size_t)n * n * n;
    tab.resize(count * 3);
    for (size_t e = 0; e < count; e++) {
        if (e && !in.record()) return fail(LUTR_EILSEQ, path, "unexpected EOF");
        float *v = &tab[e * 3];
        if (std::sscanf(in.buf, "%f %f %f", v, v + 1, v + 2) != 3 || !finite3(v))
            return fail(LUTR_EILSEQ, path, "invalid data");
    }
    return LUTR_OK;
}

// .3dl: one header record (the input shaper row) is skipped, then 17^3 integer triplets / 4096
int read_3dl(Lines &in, const char *path, std::vector<float> &tab, int &n)
{
    n = 17;
    if (!in.record()) return fail(LUTR_EILSEQ, path, "unexpected EOF");
    const size_t count = (size_t)n * n * n;
    tab.resize(count * 3);
    for (size_t e = 0; e < count; e++) {
        int r, g, b;
        if (!in.record()) return fail(LUTR_EILSEQ, path, "unexpected EOF");
        if (std::sscanf(in.buf, "%d %d %d", &r, &g, &b) != 3) return fail(LUTR_EILSEQ, path, "invalid data");
        tab[e * 3 + 0] = (float)r / 4096.0f;
        tab[e * 3 + 1] = (float)g / 4096.0f;
        tab[e * 3 + 2] = (float)b / 4096.0f;
    }
    return LUTR_OK;
}

// .m3d: header lines "in N", "out M", "values <c> <c> <c>" (column order), then size^3 rows with
// size = ceil(cbrt(in)); values are divided by out - 1; rows are read verbatim (no comment skipping)
int read_m3d(Lines &in, const char *path, std::vector<float> &tab, int &n)
{
    long nin = -1, nout = -1;
    int col[3] = {0, 1, 2};
    while (in.raw()) {
        if (in.starts("in")) nin = std::strtol(in.buf + 2, nullptr, 0);
        else if (in.starts("out")) nout = std::strtol(in.buf + 3, nullptr, 0);
        else if (in.starts("values")) {
            const char *p = in.buf + 6;
            for (int id = 0; id < 3; id++) {
                while (std::isspace((unsigned char)*p)) p++;
                if (*p == 'r') col[id] = 0;
                else if (*p == 'g') col[id] = 1;
                else if (*p == 'b') col[id] = 2;
                while (*p && !std::isspace((unsigned char)*p)) p++;
            }
            break;
        }
    }
    if (nin == -1 || nout == -1) return fail(LUTR_EILSEQ, path, "in and out must be defined");
    const long cap = (long)kMaxLevel * kMaxLevel * kMaxLevel;
    if (nin < 2 || nout < 2 || nin > cap || nout > cap) return fail(LUTR_EILSEQ, path, "invalid in or out");
    n = 1;
    while ((long)n * n * n < nin) n++;
    if (n < 2 || n > kMaxLevel) return fail(LUTR_EINVAL, path, "too large or invalid 3D LUT size");
    const float k = (float)(1.0 / (double)(nout - 1));
    const size_t count = (size_t)n * n * n;
    tab.resize(count * 3);
    for (size_t e = 0; e < count; e++) {
        float v[3];
        if (!in.raw()) return fail(LUTR_EILSEQ, path, "unexpected EOF");
        if (std::sscanf(in.buf, "%f %f %f", v, v + 1, v + 2) != 3 || !finite3(v))
            return fail(LUTR_EILSEQ, path, "invalid data");
        tab[e * 3 + 0] = v[col[0]] * k;
        tab[e * 3 + 1] = v[col[1]] * k;
        tab[e * 3 + 2] = v[col[2]] * k;
    }
    return LUTR_OK;
}

// .csp: "CSPLUTV100", "3D", optional METADATA block, per channel either {2; in_min in_max; out_min out_max} or a pre-LUT
// {npoints; npoints inputs; npoints outputs} (monotonic), then "n n n" and n^3 triplets RED fastest, each multiplied by
// (out_max - out_min).  Without pre-LUTs the input ranges become the per-channel scale like .cube's DOMAIN.  With one on
// every channel lut3d resamples it to 65536 uniform entries (its prelut: a 1D shaper applied to the normalised sample ahead of
// the cube; the mix of the resampling lerp is the UNNORMALISED distance x - in[idx], as in FFmpeg) and the scale is 1.
constexpr int kPrelutSize = 65536;

struct Prelut { std::vector<float> tab; float vmin[3], vscale[3]; int size = 0; };

int nearest_sample_index(const std::vector<float> &data, float x, int low, int hi)
{
    if (x < data[low]) return low;
    if (x > data[hi]) return hi;
    while (hi - low > 1) {
        const int mid = (low + hi) / 2;
        if (x < data[mid]) hi = mid;
        else low = mid;
    }
    return low;
}

// bit tests: this file is compiled with -fno-honor-nans, where isnan / v != v may be folded away
bool finite_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return (u & 0x7f800000u) != 0x7f800000u; }

float sanitizef(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) {
        if (u & 0x007fffffu) return 0.0f;                          // NaN
        return (u & 0x80000000u) ? -FLT_MAX : FLT_MAX;             // +-inf
    }
    return f;
}

// next whitespace-delimited float of the stream (the pre-LUT points may span lines)
bool next_float(Lines &in, float &v) { return std::fscanf(in.f, "%f", &v) == 1; }

int read_csp(Lines &in, const char *path, std::vector<float> &tab, int &n, float scale[3], Prelut &pre)
{
    if (!in.record() || !in.starts("CSPLUTV100")) return fail(LUTR_EINVAL, path, "not cineSpace LUT format");
    if (!in.record() || !in.starts("3D")) return fail(LUTR_EINVAL, path, "not 3D LUT format");
    float imin[3] = {0, 0, 0}, imax[3] = {1, 1, 1}, omin[3] = {0, 0, 0}, omax[3] = {1, 1, 1};
    std::vector<float> pin[3], pout[3];
    bool meta = false;
    for (;;) {
        if (!in.record()) return fail(LUTR_EILSEQ, path, "unexpected EOF");
        if (in.starts("BEGIN METADATA")) { meta = true; continue; }
        if (in.starts("END METADATA")) { meta = false; continue; }
        if (!meta) break;
    }
    for (int c = 0; c < 3; c++) {
        const long npoints = std::strtol(in.buf, nullptr, 0);
        if (npoints > 2) {
            if (npoints > kPrelutSize) return fail(LUTR_EINVAL, path, "prelut size too large");
            pin[c].resize((size_t)npoints); pout[c].resize((size_t)npoints);
            imin[c] = omin[c] = FLT_MAX; imax[c] = omax[c] = -FLT_MAX;
            float v, last = 0.0f;
            for (long j = 0; j < npoints; j++) {
                if (!next_float(in, v) || !finite_bits(v)) return fail(LUTR_EILSEQ, path, "invalid data");
                imin[c] = std::min(imin[c], v); imax[c] = std::max(imax[c], v);
                pin[c][(size_t)j] = v;
                if (j > 0 && v < last) return fail(LUTR_EILSEQ, path, "invalid file has non-monotonic pre-lut");
                last = v;
            }
            for (long j = 0; j < npoints; j++) {
                if (!next_float(in, v) || !finite_bits(v)) return fail(LUTR_EILSEQ, path, "invalid data");
                omin[c] = std::min(omin[c], v); omax[c] = std::max(omax[c], v);
                pout[c][(size_t)j] = v;          // outputs: range only -- lut3d checks monotonicity on the INPUT points alone
            }
        } else if (npoints == 2) {
            if (!in.record() || std::sscanf(in.buf, "%f %f", &imin[c], &imax[c]) != 2) return fail(LUTR_EILSEQ, path, "invalid data");
            if (!in.record() || std::sscanf(in.buf, "%f %f", &omin[c], &omax[c]) != 2) return fail(LUTR_EILSEQ, path, "invalid data");
        } else {
            return fail(LUTR_EILSEQ, path, "unsupported number of pre-lut points");
        }
        if (!in.record()) return fail(LUTR_EILSEQ, path, "unexpected EOF");
    }
    int sr, sg, sb;
    if (std::sscanf(in.buf, "%d %d %d", &sr, &sg, &sb) != 3) return fail(LUTR_EILSEQ, path, "invalid data");
    if (sr != sg || sr != sb) return fail(LUTR_EILSEQ, path, "unsupported size combination");
    n = sr;
    if (n < 2 || n > kMaxLevel) return fail(LUTR_EINVAL, path, "too large or invalid 3D LUT size");
    const size_t count = (size_t)n * n * n;
    tab.resize(count * 3);
    for (size_t e = 0; e < count; e++) {
        float v[3];
        if (!in.record()) return fail(LUTR_EILSEQ, path, "unexpected EOF");
        if (std::sscanf(in.buf, "%f %f %f", v, v + 1, v + 2) != 3 || !finite3(v))
            return fail(LUTR_EILSEQ, path, "invalid data");
        const size_t r = e % n, g = (e / n) % n, b = e / ((size_t)n * n);
        float *dst = &tab[((r * n + g) * n + b) * 3];
        for (int c = 0; c < 3; c++) dst[c] = v[c] * (omax[c] - omin[c]);
    }
    if (!pin[0].empty() && !pin[1].empty() && !pin[2].empty()) {
        pre.size = kPrelutSize;
        pre.tab.resize((size_t)3 * kPrelutSize);
        for (int c = 0; c < 3; c++) {
            const int np = (int)pin[c].size();
            pre.vmin[c] = imin[c];
            pre.vscale[c] = (1.0f / (float)(imax[c] - imin[c])) * (float)(kPrelutSize - 1);
            for (int i = 0; i < kPrelutSize; i++) {
                float mix = (float)i / (float)(kPrelutSize - 1);
                const float x = imin[c] + (imax[c] - imin[c]) * mix;                 // lerpf(in_min, in_max, mix)
                int idx = nearest_sample_index(pin[c], x, 0, np - 1);
                if (idx + 1 >= np) idx = np - 2;
                const float a = pout[c][(size_t)idx], b = pout[c][(size_t)idx + 1];
                mix = x - pin[c][(size_t)idx];
                pre.tab[(size_t)c * kPrelutSize + i] = sanitizef(a + (b - a) * mix);
            }
            scale[c] = 1.0f;
        }
        return LUTR_OK;
    }
    for (int c = 0; c < 3; c++) {
        float s = (float)(1.0 / (double)(imax[c] - imin[c]));
        if (!(s == s)) s = 0.f;
        scale[c] = s < 0.f ? 0.f : (s > 1.f ? 1.f : s);
    }
    return LUTR_OK;
}

}  // namespace

extern "C" int lutr_lut_parse_ex(const char *path, float **rgb, int *n, float scale[3], float **prelut, int *prelut_size,
                                float prelut_min[3], float prelut_scale[3])
{
    if (!path || !rgb || !n || !scale) {
        lutr::set_error("lutr_lut_parse: null argument");
        return LUTR_EINVAL;
    }
    *rgb = nullptr;
    *n = 0;
    if (prelut) *prelut = nullptr;
    if (prelut_size) *prelut_size = 0;
    const std::string ext = lower_ext(path);
    if (ext == "cube") return lutr_cube_parse(path, rgb, n, scale);
    if (ext != "dat" && ext != "3dl" && ext != "m3d" && ext != "csp") {
        lutr::set_error("'%s': unrecognized '.%s' file type", path, ext.c_str());
        return LUTR_EINVAL;
    }
    Lines in;
    in.f = std::fopen(path, "r");
    if (!in.f) return fail(LUTR_ENOENT, path, "cannot open");
    std::vector<float> tab;
    Prelut pre;
    int size = 0;
    scale[0] = scale[1] = scale[2] = 1.0f;
    int rc;
    if (ext == "dat") rc = read_dat(in, path, tab, size);
    else if (ext == "3dl") rc = read_3dl(in, path, tab, size);
    else if (ext == "m3d") rc = read_m3d(in, path, tab, size);
    else rc = read_csp(in, path, tab, size, scale, pre);
    if (rc) return rc;
    if (pre.size && (!prelut || !prelut_size || !prelut_min || !prelut_scale))
        return fail(LUTR_EINVAL, path, "the file carries a pre-LUT: read it with lutr_lut_parse_ex and pass it to lutr_ctx_set_prelut");
    float *out = (float *)std::malloc(tab.size() * sizeof(float));
    if (!out) return fail(LUTR_ENOMEM, path, "out of memory");
    std::memcpy(out, tab.data(), tab.size() * sizeof(float));
    if (pre.size) {
        float *po = (float *)std::malloc(pre.tab.size() * sizeof(float));
        if (!po) { std::free(out); return fail(LUTR_ENOMEM, path, "out of memory"); }
        std::memcpy(po, pre.tab.data(), pre.tab.size() * sizeof(float));
        *prelut = po;
        *prelut_size = pre.size;
        for (int c = 0; c < 3; c++) { prelut_min[c] = pre.vmin[c]; prelut_scale[c] = pre.vscale[c]; }
    }
    *rgb = out;
    *n = size;
    return LUTR_OK;
}

extern "C" int lutr_lut_parse(const char *path, float **rgb, int *n, float scale[3])
{
    return lutr_lut_parse_ex(path, rgb, n, scale, nullptr, nullptr, nullptr, nullptr);
}
