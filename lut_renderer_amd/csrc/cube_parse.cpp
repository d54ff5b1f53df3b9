// cube_parse.cpp -- the product's .cube reader (host side of lutr_cube_parse).
//
// Replaces what FFmpeg's lut3d does with the file= option the reference passes at
// /root/reference/src/lut_renderer/ffmpeg.py:246 (the GUI only admits *.cube files:
// lut_manager.py:121).  Semantics follow FFmpeg's parse_cube as distilled in
// SURVEY.md Appendix A.2; the implementation is a small line-driven state machine,
// independent of the oracle's restatement (oracle/lut3d_oracle.c) so that the parity
// tests compare two separately written readers.
//
// Deliberate deviation: a lattice entry that is not finite (nan/inf are accepted by
// sscanf("%f")) is rejected as invalid data, because the kernels' tie handling in the
// tetrahedral blend relies on 0 * c == 0 (DESIGN.md "Kernels").
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lutr_internal.h"

namespace {

constexpr int kMaxLine = 512;    // fgets buffer FFmpeg uses: longer lines arrive in pieces
constexpr int kMaxLevel = 256;

bool ends_with_cube(const char *path)
{
    const char *dot = std::strrchr(path, '.');
    if (!dot) return false;
    std::string ext(dot + 1);
    for (auto &ch : ext) ch = (char)std::tolower((unsigned char)ch);
    return ext == "cube";
}

bool blank_or_comment(const char *s)
{
    while (*s && std::isspace((unsigned char)*s)) s++;
    return *s == 0 || *s == '#';
}

struct Reader {
    FILE *f = nullptr;
    char buf[kMaxLine];
    ~Reader() { if (f) std::fclose(f); }
    bool next() { return std::fgets(buf, sizeof(buf), f) != nullptr; }
    bool starts(const char *prefix) const { return std::strncmp(buf, prefix, std::strlen(prefix)) == 0; }
};

}  // namespace

extern "C" int lutr_cube_parse(const char *path, float **rgb, int *n, float scale[3])
{
    if (!path || !rgb || !n || !scale) {
        lutr::set_error("lutr_cube_parse: null argument");
        return LUTR_EINVAL;
    }
    *rgb = nullptr;
    *n = 0;
    if (!ends_with_cube(path)) {
        lutr::set_error("'%s': unrecognised LUT extension (only .cube is supported)", path);
        return LUTR_EINVAL;
    }
    Reader rd;
    rd.f = std::fopen(path, "r");
    if (!rd.f) {
        lutr::set_error("'%s': cannot open", path);
        return LUTR_ENOENT;
    }

    // phase 1: everything up to the LUT_3D_SIZE line is ignored (DOMAIN_/TITLE included)
    int size = 0;
    while (rd.next()) {
        if (rd.starts("LUT_3D_SIZE")) {
            size = (int)std::strtol(rd.buf + 12, nullptr, 0);
            if (size < 2 || size > kMaxLevel) {
                lutr::set_error("'%s': too large or invalid 3D LUT size %d", path, size);
                return LUTR_EINVAL;
            }
            break;
        }
    }
    if (size == 0) {
        lutr::set_error("'%s': 3D LUT is empty (no LUT_3D_SIZE)", path);
        return LUTR_EILSEQ;
    }

    // phase 2: size^3 triplets in file order (red fastest), stored blue fastest
    float dmin[3] = {0.f, 0.f, 0.f}, dmax[3] = {1.f, 1.f, 1.f};
    const size_t count = (size_t)size * size * size;
    std::vector<float> table(count * 3);
    size_t filled = 0;
    while (filled < count) {
        if (!rd.next()) {
            lutr::set_error("'%s': unexpected EOF after %zu of %zu entries", path, filled, count);
            return LUTR_EILSEQ;
        }
        if (rd.starts("DOMAIN_")) {
            float *dst = rd.starts("DOMAIN_MIN ") ? dmin : rd.starts("DOMAIN_MAX ") ? dmax : nullptr;
            if (!dst) {
                lutr::set_error("'%s': malformed DOMAIN_ line", path);
                return LUTR_EILSEQ;
            }
            std::sscanf(rd.buf + 11, "%f %f %f", dst, dst + 1, dst + 2);
            continue;
        }
        if (rd.starts("TITLE") || blank_or_comment(rd.buf))
            continue;
        float v[3];
        if (std::sscanf(rd.buf, "%f %f %f", &v[0], &v[1], &v[2]) != 3) {
            lutr::set_error("'%s': invalid data at entry %zu", path, filled);
            return LUTR_EILSEQ;
        }
        if (!std::isfinite(v[0]) || !std::isfinite(v[1]) || !std::isfinite(v[2])) {
            lutr::set_error("'%s': non-finite lattice value at entry %zu", path, filled);
            return LUTR_EILSEQ;
        }
        // entry index = r + size*(g + size*b)
        const size_t r = filled % size, g = (filled / size) % size, b = filled / ((size_t)size * size);
        float *dst = &table[((r * size + g) * size + b) * 3];
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
        filled++;
    }

    for (int c = 0; c < 3; c++) {
        // vf_lut3d.c: av_clipf(1. / (max[c] - min[c]), 0.f, 1.f): float subtraction, double division, float clip
        const float span = dmax[c] - dmin[c];
        float s = (float)(1.0 / (double)span);
        if (!(s == s)) s = 0.f;
        s = s < 0.f ? 0.f : s;
        s = s > 1.f ? 1.f : s;
        scale[c] = s;
    }
    float *out = (float *)std::malloc(table.size() * sizeof(float));
    if (!out) {
        lutr::set_error("out of memory");
        return LUTR_ENOMEM;
    }
    std::memcpy(out, table.data(), table.size() * sizeof(float));
    *rgb = out;
    *n = size;
    return LUTR_OK;
}

extern "C" void lutr_cube_free(float *rgb) { std::free(rgb); }
