/*
 * oracle/lut3d_oracle.c -- TEST INFRASTRUCTURE ONLY (see lut3d_oracle.h header note).
 *
 * Plain scalar C restatement, compiled with -ffp-contract=off so every
 * product and sum below rounds exactly where the C source says it does.
 *
 * What it follows:
 *   - .cube parsing and the lut3d per-pixel pipeline: FFmpeg libavfilter/vf_lut3d.c
 *     as distilled in SURVEY.md Appendix A.1-A.5 (the filter the reference emits
 *     at /root/reference/src/lut_renderer/ffmpeg.py:246).  FFmpeg is a third-party,
 *     un-vendored, un-pinned dependency of the reference (readme.md:29), so the
 *     published algorithm is restated here.  PARITY UNPINNED vs a live ffmpeg.
 *   - YUV<->RGB / range handling: the filters the reference puts around lut3d
 *     (ffmpeg.py:212-236 scale=..., :224/:233 format=<8-bit intermediate>,
 *     :304-310 format=<pix_fmt>).  swscale's fixed-point arithmetic is unpinned
 *     and version dependent, so the engine defines its own
 *     real-arithmetic contract (DESIGN.md "YUV contract"); this file is that
 *     contract's executable definition.
 */
#include "lut3d_oracle.h"

#include <ctype.h>
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_LINE_SIZE 512   /* vf_lut3d.c MAX_LINE_SIZE */
#define MAX_LEVEL     256   /* vf_lut3d.c MAX_LEVEL */

typedef struct { float r, g, b; } rgbvec;

/* ------------------------------------------------------------------ */
/* .cube parser: SURVEY.md A.2 (vf_lut3d.c parse_cube)                 */
/* ------------------------------------------------------------------ */

static int skip_line(const char *p)
{
    while (*p && isspace((unsigned char)*p))
        p++;
    return !*p || *p == '#';
}

static int has_ext_cube(const char *path)
{
    const char *dot = strrchr(path, '.');
    if (!dot)
        return 0;
    dot++;
    return (tolower((unsigned char)dot[0]) == 'c' && tolower((unsigned char)dot[1]) == 'u' &&
            tolower((unsigned char)dot[2]) == 'b' && tolower((unsigned char)dot[3]) == 'e' && dot[4] == 0);
}

int orc_cube_parse(const char *path, orc_lut *out)
{
    char line[MAX_LINE_SIZE];
    float min[3] = {0.0f, 0.0f, 0.0f};
    float max[3] = {1.0f, 1.0f, 1.0f};
    FILE *f;
    int found = 0;

    if (!path || !out)
        return ORC_EINVAL;
    memset(out, 0, sizeof(*out));
    /* lut3d picks the parser by extension; no/unknown extension is an error */
    if (!has_ext_cube(path))
        return ORC_EINVAL;
    f = fopen(path, "r");
    if (!f)
        return ORC_ENOENT;

    /* every line before LUT_3D_SIZE is ignored, including DOMAIN_xxx and TITLE (quirk, A.2) */
    while (fgets(line, sizeof(line), f)) {
        if (!strncmp(line, "LUT_3D_SIZE", 11)) {
            const int size = (int)strtol(line + 12, NULL, 0);
            int i, j, k;
            if (size < 2 || size > MAX_LEVEL) {
                fclose(f);
                return ORC_EINVAL;
            }
            out->n = size;
            out->rgb = (float *)malloc((size_t)size * size * size * 3 * sizeof(float));
            if (!out->rgb) {
                fclose(f);
                return ORC_ENOMEM;
            }
            /* file order: R fastest, then G, then B; stored blue-fastest lut[i][j][k] */
            for (k = 0; k < size; k++) {
                for (j = 0; j < size; j++) {
                    for (i = 0; i < size; i++) {
                        float *vec = &out->rgb[(((size_t)i * size + j) * size + k) * 3];
                        for (;;) {
                            if (!fgets(line, sizeof(line), f)) {   /* "Unexpected EOF" */
                                fclose(f);
                                orc_lut_free(out);
                                return ORC_EILSEQ;
                            }
                            if (!strncmp(line, "DOMAIN_", 7)) {
                                float *vals = NULL;
                                if (!strncmp(line + 7, "MIN ", 4))
                                    vals = min;
                                else if (!strncmp(line + 7, "MAX ", 4))
                                    vals = max;
                                if (!vals) {
                                    fclose(f);
                                    orc_lut_free(out);
                                    return ORC_EILSEQ;
                                }
                                sscanf(line + 11, "%f %f %f", vals, vals + 1, vals + 2);
                                continue;
                            }
                            if (!strncmp(line, "TITLE", 5))
                                continue;
                            if (skip_line(line))
                                continue;
                            break;
                        }
                        if (sscanf(line, "%f %f %f", &vec[0], &vec[1], &vec[2]) != 3) {
                            fclose(f);
                            orc_lut_free(out);
                            return ORC_EILSEQ;
                        }
                    }
                }
            }
            found = 1;
            break;
        }
    }
    fclose(f);
    if (!found) {          /* "3D LUT is empty" */
        orc_lut_free(out);
        return ORC_EILSEQ;
    }
    /* scale = clip(1/(max-min), 0, 1); min is never subtracted from the input (A.2) */
    for (int c = 0; c < 3; c++) {
        /* vf_lut3d.c: av_clipf(1. / (max[c] - min[c]), 0.f, 1.f) -- the subtraction is float - float (C evaluates it
         * in float), the division is double (the literal 1. is a double), the clip's parameter makes it float again */
        const float span = max[c] - min[c];
        float s = (float)(1.0 / (double)span);
        if (s < 0.f) s = 0.f;     /* av_clipf(x, 0, 1) = min(max(x,0),1) */
        if (s > 1.f) s = 1.f;
        if (s != s) s = 0.f;      /* NaN from 0/0 domain: treat as degenerate */
        out->scale[c] = s;
    }
    return 0;
}

void orc_lut_free(orc_lut *lut)
{
    if (lut) {
        free(lut->rgb);
        lut->rgb = NULL;
        lut->n = 0;
        free(lut->prelut);
        lut->prelut = NULL;
        lut->pre_size = 0;
    }
}

/* ------------------------------------------------------------------ */
/* the other lut3d file formats [FFmpeg-recall:   */
/* vf_lut3d.c parse_dat / parse_3dl / parse_m3d / parse_cinespace]      */
/* ------------------------------------------------------------------ */

/* NEXT_LINE(loop_cond): read lines while loop_cond holds; EOF -> "Unexpected EOF", invalid data */
#define NEXT_LINE(loop_cond) do {                         \
        if (!fgets(line, sizeof(line), f)) { rc = ORC_EILSEQ; goto done; } \
    } while (loop_cond)

static int alloc_lut(orc_lut *out, int size)
{
    if (size < 2 || size > MAX_LEVEL)        /* allocate_3dlut: "Too large or invalid 3D LUT size" */
        return ORC_EINVAL;
    out->n = size;
    out->rgb = (float *)malloc((size_t)size * size * size * 3 * sizeof(float));
    return out->rgb ? 0 : ORC_ENOMEM;
}

static int parse_dat(FILE *f, orc_lut *out)
{
    char line[MAX_LINE_SIZE];
    int rc = 0, i, j, k, size = 33;
    NEXT_LINE(skip_line(line));
    if (!strncmp(line, "3DLUTSIZE ", 10)) {
        size = (int)strtol(line + 10, NULL, 0);
        NEXT_LINE(skip_line(line));
    }
    if ((rc = alloc_lut(out, size)))
        goto done;
    for (k = 0; k < size; k++)
        for (j = 0; j < size; j++)
            for (i = 0; i < size; i++) {
                float *vec = &out->rgb[(((size_t)k * size + j) * size + i) * 3];
                if (k != 0 || j != 0 || i != 0)
                    NEXT_LINE(skip_line(line));
                if (sscanf(line, "%f %f %f", &vec[0], &vec[1], &vec[2]) != 3) { rc = ORC_EILSEQ; goto done; }
            }
done:
    return rc;
}

static int parse_3dl(FILE *f, orc_lut *out)
{
    char line[MAX_LINE_SIZE];
    int rc, i, j, k;
    const int size = 17;
    const float scale = 16 * 16 * 16;
    if ((rc = alloc_lut(out, size)))
        goto done;
    NEXT_LINE(skip_line(line));
    for (k = 0; k < size; k++)
        for (j = 0; j < size; j++)
            for (i = 0; i < size; i++) {
                int r, g, b;
                float *vec = &out->rgb[(((size_t)k * size + j) * size + i) * 3];
                NEXT_LINE(skip_line(line));
                if (sscanf(line, "%d %d %d", &r, &g, &b) != 3) { rc = ORC_EILSEQ; goto done; }
                vec[0] = r / scale;
                vec[1] = g / scale;
                vec[2] = b / scale;
            }
done:
    return rc;
}

static int parse_m3d(FILE *f, orc_lut *out)
{
    float scale;
    int rc = 0, i, j, k, size, in = -1, outv = -1;
    char line[MAX_LINE_SIZE];
    unsigned char rgb_map[3] = {0, 1, 2};

    while (fgets(line, sizeof(line), f)) {
        if (!strncmp(line, "in", 2)) in = (int)strtol(line + 2, NULL, 0);
        else if (!strncmp(line, "out", 3)) outv = (int)strtol(line + 3, NULL, 0);
        else if (!strncmp(line, "values", 6)) {
            const char *p = line + 6;
            for (int id = 0; id < 3; id++) {
                while (isspace((unsigned char)*p)) p++;
                switch (*p) {
                case 'r': rgb_map[id] = 0; break;
                case 'g': rgb_map[id] = 1; break;
                case 'b': rgb_map[id] = 2; break;
                }
                while (*p && !isspace((unsigned char)*p)) p++;
            }
            break;
        }
    }
    if (in == -1 || outv == -1)
        return ORC_EILSEQ;
    if (in < 2 || outv < 2 || in > MAX_LEVEL * MAX_LEVEL * MAX_LEVEL || outv > MAX_LEVEL * MAX_LEVEL * MAX_LEVEL)
        return ORC_EILSEQ;
    for (size = 1; size * size * size < in; size++)
        ;
    if ((rc = alloc_lut(out, size)))
        goto done;
    scale = 1. / (outv - 1);
    for (k = 0; k < size; k++)
        for (j = 0; j < size; j++)
            for (i = 0; i < size; i++) {
                float *vec = &out->rgb[(((size_t)k * size + j) * size + i) * 3];
                float val[3];
                NEXT_LINE(0);
                if (sscanf(line, "%f %f %f", val, val + 1, val + 2) != 3) { rc = ORC_EILSEQ; goto done; }
                vec[0] = val[rgb_map[0]] * scale;
                vec[1] = val[rgb_map[1]] * scale;
                vec[2] = val[rgb_map[2]] * scale;
            }
done:
    return rc;
}

/* cineSpace .csp [FFmpeg-recall: vf_lut3d.c parse_cinespace].  Per channel either 2 points (input / output ranges) or a
 * pre-LUT of up to 65536 monotonic points; when all three channels have one, lut3d resamples it to 65536 uniform entries
 * (nearest_sample_index + lerpf with the UNNORMALISED distance `x - in[idx]` as the mix, as FFmpeg does) and applies it to the
 * normalised sample before the cube (apply_prelut below). */
#define PRELUT_SIZE 65536

static inline float lerpf(float v0, float v1, float f) { return v0 + (v1 - v0) * f; }

static int nearest_sample_index(const float *data, float x, int low, int hi)
{
    int mid;
    if (x < data[low]) return low;
    if (x > data[hi]) return hi;
    for (;;) {
        if (hi - low <= 1) return low;
        mid = (low + hi) / 2;
        if (x < data[mid]) hi = mid;
        else low = mid;
    }
}

static float sanitizef(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) {
        if (u & 0x007fffffu) return 0.0f;                 /* NaN */
        return (u & 0x80000000u) ? -FLT_MAX : FLT_MAX;    /* +-inf */
    }
    return f;
}

static int next_float(FILE *f, float *v)
{
    return fscanf(f, "%f", v) == 1;
}

static int parse_cinespace(FILE *f, orc_lut *out)
{
    char line[MAX_LINE_SIZE];
    float in_min[3] = {0.0f, 0.0f, 0.0f}, in_max[3] = {1.0f, 1.0f, 1.0f};
    float out_min[3] = {0.0f, 0.0f, 0.0f}, out_max[3] = {1.0f, 1.0f, 1.0f};
    int inside_metadata = 0, size, rc = 0, prelut = 0;
    int prelut_sizes[3] = {0, 0, 0};
    float *in_prelut[3] = {NULL, NULL, NULL}, *out_prelut[3] = {NULL, NULL, NULL};

#define CSP_FAIL(code) do { rc = (code); goto done; } while (0)
#undef NEXT_LINE_CSP
#define NEXT_LINE_CSP() do { \
        do { if (!fgets(line, sizeof(line), f)) CSP_FAIL(ORC_EILSEQ); } while (skip_line(line)); \
    } while (0)
    NEXT_LINE_CSP();
    if (strncmp(line, "CSPLUTV100", 10)) CSP_FAIL(ORC_EINVAL);
    NEXT_LINE_CSP();
    if (strncmp(line, "3D", 2)) CSP_FAIL(ORC_EINVAL);
    while (1) {
        NEXT_LINE_CSP();
        if (!strncmp(line, "BEGIN METADATA", 14)) { inside_metadata = 1; continue; }
        if (!strncmp(line, "END METADATA", 12)) { inside_metadata = 0; continue; }
        if (inside_metadata == 0) {
            int size_r, size_g, size_b;
            for (int i = 0; i < 3; i++) {
                int npoints = (int)strtol(line, NULL, 0);
                if (npoints > 2) {
                    float v, last = 0.0f;
                    if (npoints > PRELUT_SIZE) CSP_FAIL(ORC_EINVAL);
                    if (in_prelut[i] || out_prelut[i]) CSP_FAIL(ORC_EINVAL);
                    in_prelut[i] = (float *)malloc((size_t)npoints * sizeof(float));
                    out_prelut[i] = (float *)malloc((size_t)npoints * sizeof(float));
                    if (!in_prelut[i] || !out_prelut[i]) CSP_FAIL(ORC_ENOMEM);
                    prelut_sizes[i] = npoints;
                    in_min[i] = FLT_MAX; in_max[i] = -FLT_MAX;
                    out_min[i] = FLT_MAX; out_max[i] = -FLT_MAX;
                    for (int j = 0; j < npoints; j++) {
                        if (!next_float(f, &v)) CSP_FAIL(ORC_EILSEQ);
                        in_min[i] = v < in_min[i] ? v : in_min[i];
                        in_max[i] = v > in_max[i] ? v : in_max[i];
                        in_prelut[i][j] = v;
                        if (j > 0 && v < last) CSP_FAIL(ORC_EILSEQ);     /* "Invalid file has non-monotonic pre-lut" */
                        last = v;
                    }
                    for (int j = 0; j < npoints; j++) {
                        if (!next_float(f, &v)) CSP_FAIL(ORC_EILSEQ);
                        out_min[i] = v < out_min[i] ? v : out_min[i];
                        out_max[i] = v > out_max[i] ? v : out_max[i];
                        out_prelut[i][j] = v;      /* no monotonicity check on outputs: parse_cinespace has it on the inputs only */
                    }
                } else if (npoints == 2) {
                    NEXT_LINE_CSP();
                    if (sscanf(line, "%f %f", &in_min[i], &in_max[i]) != 2) CSP_FAIL(ORC_EILSEQ);
                    NEXT_LINE_CSP();
                    if (sscanf(line, "%f %f", &out_min[i], &out_max[i]) != 2) CSP_FAIL(ORC_EILSEQ);
                } else {
                    CSP_FAIL(ORC_EILSEQ);
                }
                NEXT_LINE_CSP();
            }
            if (sscanf(line, "%d %d %d", &size_r, &size_g, &size_b) != 3) CSP_FAIL(ORC_EILSEQ);
            if (size_r != size_g || size_r != size_b) CSP_FAIL(ORC_EILSEQ);
            size = size_r;
            if (prelut_sizes[0] && prelut_sizes[1] && prelut_sizes[2]) prelut = 1;
            if ((rc = alloc_lut(out, size)))
                goto done;
            for (int k = 0; k < size; k++)
                for (int j = 0; j < size; j++)
                    for (int i = 0; i < size; i++) {
                        float *vec = &out->rgb[(((size_t)i * size + j) * size + k) * 3];
                        NEXT_LINE_CSP();
                        if (sscanf(line, "%f %f %f", &vec[0], &vec[1], &vec[2]) != 3) CSP_FAIL(ORC_EILSEQ);
                        vec[0] *= out_max[0] - out_min[0];
                        vec[1] *= out_max[1] - out_min[1];
                        vec[2] *= out_max[2] - out_min[2];
                    }
            break;
        }
    }
    if (prelut) {
        out->prelut = (float *)malloc((size_t)3 * PRELUT_SIZE * sizeof(float));
        if (!out->prelut) CSP_FAIL(ORC_ENOMEM);
        out->pre_size = PRELUT_SIZE;
        for (int c = 0; c < 3; c++) {
            out->pre_min[c] = in_min[c];
            out->pre_scale[c] = (1.0f / (float)(in_max[c] - in_min[c])) * (float)(PRELUT_SIZE - 1);
            for (int i = 0; i < PRELUT_SIZE; ++i) {
                float mix = (float)i / (float)(PRELUT_SIZE - 1);
                const float x = lerpf(in_min[c], in_max[c], mix);
                int idx = nearest_sample_index(in_prelut[c], x, 0, prelut_sizes[c] - 1);
                float a, b;
                if (idx + 1 >= prelut_sizes[c]) idx = prelut_sizes[c] - 2;      /* FFmpeg asserts idx + 1 < npoints */
                a = out_prelut[c][idx + 0];
                b = out_prelut[c][idx + 1];
                mix = x - in_prelut[c][idx];
                out->prelut[(size_t)c * PRELUT_SIZE + i] = sanitizef(lerpf(a, b, mix));
            }
            out->scale[c] = 1.00f;
        }
    } else {
        for (int c = 0; c < 3; c++) {
            float s = (float)(1. / (in_max[c] - in_min[c]));
            if (s < 0.f) s = 0.f;
            if (s > 1.f) s = 1.f;
            if (s != s) s = 0.f;
            out->scale[c] = s;
        }
    }
done:
    for (int c = 0; c < 3; c++) { free(in_prelut[c]); free(out_prelut[c]); }
    return rc;
#undef CSP_FAIL
}

/* lut3d's config: the parser is chosen by the (case-insensitive) extension; none / unknown -> EINVAL */
int orc_lut_file_parse(const char *path, orc_lut *out)
{
    const char *ext;
    char e[8] = {0};
    FILE *f;
    int rc;
    if (!path || !out)
        return ORC_EINVAL;
    ext = strrchr(path, '.');
    if (!ext)
        return ORC_EINVAL;
    ext++;
    for (int i = 0; i < 7 && ext[i]; i++)
        e[i] = (char)tolower((unsigned char)ext[i]);
    if (!strcmp(e, "cube"))
        return orc_cube_parse(path, out);
    if (strcmp(e, "dat") && strcmp(e, "3dl") && strcmp(e, "m3d") && strcmp(e, "csp"))
        return ORC_EINVAL;
    memset(out, 0, sizeof(*out));
    out->scale[0] = out->scale[1] = out->scale[2] = 1.0f;
    f = fopen(path, "r");
    if (!f)
        return ORC_ENOENT;
    if (!strcmp(e, "dat")) rc = parse_dat(f, out);
    else if (!strcmp(e, "3dl")) rc = parse_3dl(f, out);
    else if (!strcmp(e, "m3d")) rc = parse_m3d(f, out);
    else rc = parse_cinespace(f, out);
    fclose(f);
    if (rc)
        orc_lut_free(out);
    return rc;
}

/* ------------------------------------------------------------------ */
/* interpolation: SURVEY.md A.4 / A.5 (vf_lut3d.c interp_*)            */
/* ------------------------------------------------------------------ */

static inline rgbvec node(const orc_lut *l, int r, int g, int b)
{
    const float *p = &l->rgb[(((size_t)r * l->n + g) * l->n + b) * 3];
    rgbvec v = {p[0], p[1], p[2]};
    return v;
}

#define PREV(x) ((int)(x))
#define NEXT(x, n) (((int)(x) + 1) < (n) - 1 ? ((int)(x) + 1) : (n) - 1)
/* vf_lut3d.c: #define NEAR(x) ((int)((x) + .5)) -- .5 is a double literal, so the float coordinate is promoted and the
 * sum is exact; with .5f the sum would round up to the next integer for x = k + 0.49999997 (the float below k + 1/2). */
#define NEAR(x) ((int)((double)(x) + .5))


static inline rgbvec lerp(rgbvec a, rgbvec b, float f)
{
    rgbvec v = {lerpf(a.r, b.r, f), lerpf(a.g, b.g, f), lerpf(a.b, b.b, f)};
    return v;
}

static rgbvec interp_nearest(const orc_lut *l, rgbvec s)
{
    return node(l, NEAR(s.r), NEAR(s.g), NEAR(s.b));
}

static rgbvec interp_trilinear(const orc_lut *l, rgbvec s)
{
    const int n = l->n;
    const int p[3] = {PREV(s.r), PREV(s.g), PREV(s.b)};
    const int x[3] = {NEXT(s.r, n), NEXT(s.g, n), NEXT(s.b, n)};
    const rgbvec d = {s.r - p[0], s.g - p[1], s.b - p[2]};
    const rgbvec c000 = node(l, p[0], p[1], p[2]);
    const rgbvec c001 = node(l, p[0], p[1], x[2]);
    const rgbvec c010 = node(l, p[0], x[1], p[2]);
    const rgbvec c011 = node(l, p[0], x[1], x[2]);
    const rgbvec c100 = node(l, x[0], p[1], p[2]);
    const rgbvec c101 = node(l, x[0], p[1], x[2]);
    const rgbvec c110 = node(l, x[0], x[1], p[2]);
    const rgbvec c111 = node(l, x[0], x[1], x[2]);
    const rgbvec c00 = lerp(c000, c100, d.r);
    const rgbvec c10 = lerp(c010, c110, d.r);
    const rgbvec c01 = lerp(c001, c101, d.r);
    const rgbvec c11 = lerp(c011, c111, d.r);
    const rgbvec c0 = lerp(c00, c10, d.g);
    const rgbvec c1 = lerp(c01, c11, d.g);
    return lerp(c0, c1, d.b);
}

#define TETRA(w0, v0, w1, v1, w2, v2, w3, v3) \
    do { \
        c.r = (w0) * v0.r + (w1) * v1.r + (w2) * v2.r + (w3) * v3.r; \
        c.g = (w0) * v0.g + (w1) * v1.g + (w2) * v2.g + (w3) * v3.g; \
        c.b = (w0) * v0.b + (w1) * v1.b + (w2) * v2.b + (w3) * v3.b; \
    } while (0)

static rgbvec interp_tetrahedral(const orc_lut *l, rgbvec s)
{
    const int n = l->n;
    const int p[3] = {PREV(s.r), PREV(s.g), PREV(s.b)};
    const int x[3] = {NEXT(s.r, n), NEXT(s.g, n), NEXT(s.b, n)};
    const rgbvec d = {s.r - p[0], s.g - p[1], s.b - p[2]};
    const rgbvec c000 = node(l, p[0], p[1], p[2]);
    const rgbvec c111 = node(l, x[0], x[1], x[2]);
    rgbvec c;
    if (d.r > d.g) {
        if (d.g > d.b) {
            const rgbvec c100 = node(l, x[0], p[1], p[2]);
            const rgbvec c110 = node(l, x[0], x[1], p[2]);
            TETRA(1 - d.r, c000, d.r - d.g, c100, d.g - d.b, c110, d.b, c111);
        } else if (d.r > d.b) {
            const rgbvec c100 = node(l, x[0], p[1], p[2]);
            const rgbvec c101 = node(l, x[0], p[1], x[2]);
            TETRA(1 - d.r, c000, d.r - d.b, c100, d.b - d.g, c101, d.g, c111);
        } else {
            const rgbvec c001 = node(l, p[0], p[1], x[2]);
            const rgbvec c101 = node(l, x[0], p[1], x[2]);
            TETRA(1 - d.b, c000, d.b - d.r, c001, d.r - d.g, c101, d.g, c111);
        }
    } else {
        if (d.b > d.g) {
            const rgbvec c001 = node(l, p[0], p[1], x[2]);
            const rgbvec c011 = node(l, p[0], x[1], x[2]);
            TETRA(1 - d.b, c000, d.b - d.g, c001, d.g - d.r, c011, d.r, c111);
        } else if (d.b > d.r) {
            const rgbvec c010 = node(l, p[0], x[1], p[2]);
            const rgbvec c011 = node(l, p[0], x[1], x[2]);
            TETRA(1 - d.g, c000, d.g - d.b, c010, d.b - d.r, c011, d.r, c111);
        } else {
            const rgbvec c010 = node(l, p[0], x[1], p[2]);
            const rgbvec c110 = node(l, x[0], x[1], p[2]);
            TETRA(1 - d.g, c000, d.g - d.r, c010, d.r - d.b, c110, d.b, c111);
        }
    }
    return c;
}

/* pyramid / prism: newer FFmpeg modes the reference whitelists (ffmpeg.py:243) but
 * the GUI never offers (main_window.py:649-651).  Restated from recall; SURVEY 8f rank 2. */
static rgbvec interp_pyramid(const orc_lut *l, rgbvec s)
{
    const int n = l->n;
    const int p[3] = {PREV(s.r), PREV(s.g), PREV(s.b)};
    const int x[3] = {NEXT(s.r, n), NEXT(s.g, n), NEXT(s.b, n)};
    const rgbvec d = {s.r - p[0], s.g - p[1], s.b - p[2]};
    const rgbvec c000 = node(l, p[0], p[1], p[2]);
    const rgbvec c111 = node(l, x[0], x[1], x[2]);
    rgbvec c;
#define PYR(ch) \
    if (d.g > d.r && d.b > d.r) { \
        c.ch = c000.ch + (c111.ch - c011.ch) * d.r + (c010.ch - c000.ch) * d.g + (c001.ch - c000.ch) * d.b + \
               (c011.ch - c001.ch - c010.ch + c000.ch) * d.g * d.b; \
    } else if (d.r > d.g && d.b > d.g) { \
        c.ch = c000.ch + (c100.ch - c000.ch) * d.r + (c111.ch - c101.ch) * d.g + (c001.ch - c000.ch) * d.b + \
               (c101.ch - c001.ch - c100.ch + c000.ch) * d.r * d.b; \
    } else { \
        c.ch = c000.ch + (c100.ch - c000.ch) * d.r + (c010.ch - c000.ch) * d.g + (c111.ch - c110.ch) * d.b + \
               (c110.ch - c100.ch - c010.ch + c000.ch) * d.r * d.g; \
    }
    const rgbvec c001 = node(l, p[0], p[1], x[2]);
    const rgbvec c010 = node(l, p[0], x[1], p[2]);
    const rgbvec c011 = node(l, p[0], x[1], x[2]);
    const rgbvec c100 = node(l, x[0], p[1], p[2]);
    const rgbvec c101 = node(l, x[0], p[1], x[2]);
    const rgbvec c110 = node(l, x[0], x[1], p[2]);
    PYR(r) PYR(g) PYR(b)
#undef PYR
    return c;
}

static rgbvec interp_prism(const orc_lut *l, rgbvec s)
{
    const int n = l->n;
    const int p[3] = {PREV(s.r), PREV(s.g), PREV(s.b)};
    const int x[3] = {NEXT(s.r, n), NEXT(s.g, n), NEXT(s.b, n)};
    const rgbvec d = {s.r - p[0], s.g - p[1], s.b - p[2]};
    const rgbvec c000 = node(l, p[0], p[1], p[2]);
    const rgbvec c001 = node(l, p[0], p[1], x[2]);
    const rgbvec c010 = node(l, p[0], x[1], p[2]);
    const rgbvec c011 = node(l, p[0], x[1], x[2]);
    const rgbvec c100 = node(l, x[0], p[1], p[2]);
    const rgbvec c101 = node(l, x[0], p[1], x[2]);
    const rgbvec c110 = node(l, x[0], x[1], p[2]);
    const rgbvec c111 = node(l, x[0], x[1], x[2]);
    rgbvec c;
#define PRI(ch) \
    if (d.b > d.r) { \
        c.ch = c000.ch + (c001.ch - c000.ch) * d.b + (c101.ch - c001.ch) * d.r + (c010.ch - c000.ch) * d.g + \
               (c000.ch - c010.ch - c001.ch + c011.ch) * d.b * d.g + \
               (c001.ch - c011.ch - c101.ch + c111.ch) * d.r * d.g; \
    } else { \
        c.ch = c000.ch + (c101.ch - c100.ch) * d.b + (c100.ch - c000.ch) * d.r + (c010.ch - c000.ch) * d.g + \
               (c100.ch - c110.ch - c101.ch + c111.ch) * d.b * d.g + \
               (c000.ch - c010.ch - c100.ch + c110.ch) * d.r * d.g; \
    }
    PRI(r) PRI(g) PRI(b)
#undef PRI
    return c;
}

static inline rgbvec interp(const orc_lut *l, int mode, rgbvec s)
{
    switch (mode) {
    case ORC_NEAREST:     return interp_nearest(l, s);
    case ORC_TRILINEAR:   return interp_trilinear(l, s);
    case ORC_PYRAMID:     return interp_pyramid(l, s);
    case ORC_PRISM:       return interp_prism(l, s);
    default:              return interp_tetrahedral(l, s);
    }
}

static inline float clipf(float a, float lo, float hi)
{
    /* av_clipf: FFMIN(FFMAX(a, amin), amax) */
    const float t = a > lo ? a : lo;
    return t > hi ? hi : t;
}

/* C float->int conversion truncates toward zero; out-of-int-range is UB in C, the
 * oracle (like the GPU's v_cvt_i32_f32) saturates.  Then av_clip_uintp2. */
static inline int quant(float v, float maxf, int maxi)
{
    float t = v * maxf;
    int i;
    if (!(t > -2147483648.0f)) i = INT32_MIN;      /* also NaN -> lowest */
    else if (t >= 2147483648.0f) i = INT32_MAX;
    else i = (int)t;
    if (i < 0) return 0;
    if (i > maxi) return maxi;
    return i;
}

/* [FFmpeg-recall: vf_lut3d.c prelut_interp_1d_linear / apply_prelut] the shaper ahead of the cube, on the normalised sample */
static inline float prelut_interp_1d_linear(const orc_lut *l, int idx, float s)
{
    const int lut_max = l->pre_size - 1;
    const float scaled = (s - l->pre_min[idx]) * l->pre_scale[idx];
    const float x = clipf(scaled, 0.0f, (float)lut_max);
    const int prev = PREV(x);
    const int next = ((int)(x) + 1) < lut_max ? ((int)(x) + 1) : lut_max;
    const float p = l->prelut[(size_t)idx * l->pre_size + prev];
    const float n = l->prelut[(size_t)idx * l->pre_size + next];
    const float d = x - (float)prev;
    return lerpf(p, n, d);
}

static inline rgbvec apply_prelut(const orc_lut *l, rgbvec s)
{
    rgbvec c;
    if (l->pre_size <= 0 || !l->prelut)
        return s;
    c.r = prelut_interp_1d_linear(l, 0, s.r);
    c.g = prelut_interp_1d_linear(l, 1, s.g);
    c.b = prelut_interp_1d_linear(l, 2, s.b);
    return c;
}

/* one pixel of the A.3 pipeline: integer codes in, integer codes out */
static inline void lut_pixel(const orc_lut *l, int mode, float scale_f, const float scale_c[3],
                             float lut_max, float maxf, int maxi,
                             int r, int g, int b, int *ro, int *go, int *bo)
{
    const rgbvec rgb0 = {(float)r * scale_f, (float)g * scale_f, (float)b * scale_f};
    const rgbvec rgb = apply_prelut(l, rgb0);
    const rgbvec s = {clipf(rgb.r * scale_c[0], 0, lut_max),
                      clipf(rgb.g * scale_c[1], 0, lut_max),
                      clipf(rgb.b * scale_c[2], 0, lut_max)};
    const rgbvec v = interp(l, mode, s);
    *ro = quant(v.r, maxf, maxi);
    *go = quant(v.g, maxf, maxi);
    *bo = quant(v.b, maxf, maxi);
}

typedef struct lut_consts {
    float scale_f, scale_c[3], lut_max, maxf;
    int maxi;
} lut_consts;

static void make_lut_consts(const orc_lut *l, int depth, lut_consts *k)
{
    k->maxi = (1 << depth) - 1;
    k->maxf = (float)k->maxi;
    k->scale_f = 1.0f / (float)k->maxi;
    k->lut_max = (float)(l->n - 1);
    for (int c = 0; c < 3; c++)
        k->scale_c[c] = l->scale[c] * k->lut_max;
}

int orc_apply_pixel(const orc_lut *lut, int depth, int mode, const int in_rgb[3], int out_rgb[3])
{
    lut_consts k;
    if (!lut || !lut->rgb || depth < 8 || depth > 16)
        return ORC_EINVAL;
    make_lut_consts(lut, depth, &k);
    lut_pixel(lut, mode, k.scale_f, k.scale_c, k.lut_max, k.maxf, k.maxi,
              in_rgb[0], in_rgb[1], in_rgb[2], &out_rgb[0], &out_rgb[1], &out_rgb[2]);
    return 0;
}

/* ------------------------------------------------------------------ */
/* row-slice threading, FFmpeg style: rows [h*j/n, h*(j+1)/n)  (3.4)   */
/* ------------------------------------------------------------------ */

typedef void (*slice_fn)(void *arg, int y0, int y1);
typedef struct { slice_fn fn; void *arg; int y0, y1; } slice_job;

static void *slice_tramp(void *p)
{
    slice_job *j = (slice_job *)p;
    j->fn(j->arg, j->y0, j->y1);
    return NULL;
}

static void run_slices(slice_fn fn, void *arg, int h, int align, int nthreads)
{
    /* units of `align` rows so chroma blocks never straddle two slices */
    const int units = (h + align - 1) / align;
    int n = nthreads < 1 ? 1 : nthreads;
    if (n > units) n = units;
    if (n <= 1) {
        fn(arg, 0, h);
        return;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n);
    slice_job *jobs = (slice_job *)malloc(sizeof(slice_job) * n);
    for (int j = 0; j < n; j++) {
        int a = (int)((long long)units * j / n) * align;
        int b = (int)((long long)units * (j + 1) / n) * align;
        if (b > h) b = h;
        jobs[j].fn = fn; jobs[j].arg = arg; jobs[j].y0 = a; jobs[j].y1 = b;
        pthread_create(&th[j], NULL, slice_tramp, &jobs[j]);
    }
    for (int j = 0; j < n; j++)
        pthread_join(th[j], NULL);
    free(th);
    free(jobs);
}

/* ------------------------------------------------------------------ */
/* planar RGB (gbrp order G,B,R): SURVEY.md A.3                        */
/* ------------------------------------------------------------------ */

typedef struct {
    const orc_lut *lut; int depth, mode, w;
    const uint8_t *src[3]; ptrdiff_t ss[3];
    uint8_t *dst[3]; ptrdiff_t ds[3];
    lut_consts k;
} rgb_job;

static inline int ld(const uint8_t *row, int x, int wide)
{
    return wide ? ((const uint16_t *)row)[x] : row[x];
}

static inline void st(uint8_t *row, int x, int wide, int v)
{
    if (wide) ((uint16_t *)row)[x] = (uint16_t)v;
    else row[x] = (uint8_t)v;
}

static void rgb_slice(void *arg, int y0, int y1)
{
    rgb_job *j = (rgb_job *)arg;
    const int wide = j->depth > 8;
    for (int y = y0; y < y1; y++) {
        const uint8_t *sg = j->src[0] + y * j->ss[0];
        const uint8_t *sb = j->src[1] + y * j->ss[1];
        const uint8_t *sr = j->src[2] + y * j->ss[2];
        uint8_t *dg = j->dst[0] + y * j->ds[0];
        uint8_t *db = j->dst[1] + y * j->ds[1];
        uint8_t *dr = j->dst[2] + y * j->ds[2];
        for (int x = 0; x < j->w; x++) {
            int ro, go, bo;
            lut_pixel(j->lut, j->mode, j->k.scale_f, j->k.scale_c, j->k.lut_max, j->k.maxf, j->k.maxi,
                      ld(sr, x, wide), ld(sg, x, wide), ld(sb, x, wide), &ro, &go, &bo);
            st(dr, x, wide, ro);
            st(dg, x, wide, go);
            st(db, x, wide, bo);
        }
    }
}

int orc_apply_planar_rgb(const orc_lut *lut, int depth, int mode, int w, int h,
                         const void *const src[3], const ptrdiff_t sstride[3],
                         void *const dst[3], const ptrdiff_t dstride[3], int nthreads)
{
    rgb_job j;
    if (!lut || !lut->rgb || depth < 8 || depth > 16 || w < 0 || h < 0 ||
        mode < ORC_NEAREST || mode > ORC_PRISM)
        return ORC_EINVAL;
    if (w == 0 || h == 0)
        return 0;
    j.lut = lut; j.depth = depth; j.mode = mode; j.w = w;
    for (int c = 0; c < 3; c++) {
        j.src[c] = (const uint8_t *)src[c]; j.ss[c] = sstride[c];
        j.dst[c] = (uint8_t *)dst[c]; j.ds[c] = dstride[c];
    }
    make_lut_consts(lut, depth, &j.k);
    run_slices(rgb_slice, &j, h, 1, nthreads);
    return 0;
}

/* ------------------------------------------------------------------ */
/* YUV contract: constants                                 */
/* ------------------------------------------------------------------ */

static int matrix_coeffs(int m, double *kr, double *kb)
{
    switch (m) {
    case ORC_MAT_BT709:  *kr = 0.2126; *kb = 0.0722; return 0;
    case ORC_MAT_BT601:  *kr = 0.299;  *kb = 0.114;  return 0;   /* smpte170m == bt470bg */
    case ORC_MAT_BT2020: *kr = 0.2627; *kb = 0.0593; return 0;   /* bt2020c uses nc coefficients */
    }
    return ORC_EINVAL;
}

int orc_yuv_constants(int matrix_in, int range_in, int matrix_out, int range_out,
                      int din, int dl, int dout, int chroma_n, int prologue_pc_to_tv,
                      orc_yuv_consts *o)
{
    double kr, kb, kg;
    if (!o || din < 8 || din > 16 || dl < 8 || dl > 16 || dout < 8 || dout > 16 ||
        (chroma_n != 1 && chroma_n != 2 && chroma_n != 4))
        return ORC_EINVAL;
    if (!prologue_pc_to_tv && din != dl)
        return ORC_EINVAL;      /* only the prologue changes depth ahead of the LUT */
    memset(o, 0, sizeof(*o));

    /* optional prologue: scale=in_range=pc:out_range=<r>,format=<dl-bit> (ffmpeg.py:224-233).
     * Input is full range at depth din; result is range_in's meaning at depth dl:
     * after it the LUT-side conversion sees (range_in, dl). */
    if (prologue_pc_to_tv) {
        const double mi = (double)((1 << din) - 1);
        const double sl = (double)(1 << (dl - 8));
        const double ml = (double)((1 << dl) - 1);
        const double half_in = (double)(1 << (din - 1));
        double py, pyo, pc, pco;
        if (range_in == ORC_RANGE_TV) {       /* pc -> tv */
            py = 219.0 * sl / mi;  pyo = 16.0 * sl;
            pc = 224.0 * sl / mi;  pco = 128.0 * sl - half_in * pc;
        } else {                              /* pc -> pc, depth change only */
            py = ml / mi;          pyo = 0.0;
            pc = ml / mi;          pco = 128.0 * sl - half_in * pc;
        }
        o->pre = 1;
        o->py = (float)py;  o->pyb = (float)(pyo + 0.5);
        o->pc = (float)pc;  o->pcb = (float)(pco + 0.5);
        o->pre_max = (float)ml;
    }

    /* input side, at depth dl */
    if (matrix_coeffs(matrix_in, &kr, &kb))
        return ORC_EINVAL;
    kg = 1.0 - kr - kb;
    {
        const double s = (double)(1 << (dl - 8));
        const double m = (double)((1 << dl) - 1);
        double ky, yoff, kc;
        if (range_in == ORC_RANGE_TV) { ky = m / (219.0 * s); yoff = 16.0 * s; kc = m / (224.0 * s); }
        else if (range_in == ORC_RANGE_PC) { ky = 1.0; yoff = 0.0; kc = 1.0; }
        else return ORC_EINVAL;
        o->ky = (float)ky;
        o->yb = (float)(-ky * yoff + 0.5);
        o->coff = (float)(128.0 * s);
        o->krv = (float)(2.0 * (1.0 - kr) * kc);
        o->kbu = (float)(2.0 * (1.0 - kb) * kc);
        o->kgu = (float)(-2.0 * kb * (1.0 - kb) / kg * kc);
        o->kgv = (float)(-2.0 * kr * (1.0 - kr) / kg * kc);
        o->max_l = (float)m;
    }
    /* output side: RGB codes at depth dl (full range) -> YUV codes at depth dout */
    if (matrix_coeffs(matrix_out, &kr, &kb))
        return ORC_EINVAL;
    kg = 1.0 - kr - kb;
    {
        const double so = (double)(1 << (dout - 8));
        const double mo = (double)((1 << dout) - 1);
        const double ml = (double)((1 << dl) - 1);
        const double n = (double)chroma_n;
        double ys, yoff, cs;
        if (range_out == ORC_RANGE_TV) { ys = 219.0 * so; yoff = 16.0 * so; cs = 224.0 * so; }
        else if (range_out == ORC_RANGE_PC) { ys = mo; yoff = 0.0; cs = mo; }
        else return ORC_EINVAL;
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
    return 0;
}

/* ------------------------------------------------------------------ */
/* YUV contract: pixels                                                */
/* ------------------------------------------------------------------ */

typedef struct {
    const orc_lut *lut; int mode; const orc_yuv_consts *k;
    int din, dl, dout, csx, csy, w, h;
    const uint8_t *src[3]; ptrdiff_t ss[3];
    uint8_t *dst[3]; ptrdiff_t ds[3];
    lut_consts lk;
    float *fdst[3]; int fw[3];      /* dither path: unquantised output planes (values before the +0.5 / floor) */
    const uint16_t *lat16;          /* fast variant: nodes as fp16 of value * M, [r][g][b][3]; NULL = strict */
} yuv_job;

/* ------------------------------------------------------------------ */
/* FAST variant (lutr_ctx_set_precision(FAST), csrc/lutr_tile2.hip V_FAST): the same pixel pipeline with two      */
/* changes -- lattice nodes are fp16 of value * (2^depth - 1) (round to nearest even), and the blend is a fused    */
/* multiply-add chain with fp32 accumulation in the kernels' order.  Coordinates, truncation and the YUV contract  */
/* are the strict ones.  This restatement lets the GPU be checked bit for bit; tests bound |fast - strict| <= 1.   */
/* ------------------------------------------------------------------ */
static uint16_t f2h_rne(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));   /* inf / nan */
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                       /* rounds to inf */
    if (x < 0x33000001u) return (uint16_t)sign;                                                     /* rounds to zero */
    if (x < 0x38800000u) {                                   /* subnormal half: value * 2^24 rounded to an integer */
        const int e = (int)(x >> 23);                        /* biased float exponent, 102..112 */
        const uint32_t m = (x & 0x7fffffu) | 0x800000u;      /* 24-bit significand */
        const int sh = 126 - e;                              /* bits to drop: 14..24 */
        uint32_t h = m >> sh;
        const uint32_t rem = m & ((1u << sh) - 1), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (h & 1))) h++;
        return (uint16_t)(sign | h);
    }
    {
        uint32_t h = ((x - 0x38000000u) >> 13);              /* rebias exponent, keep 10 fraction bits */
        const uint32_t rem = x & 0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
        return (uint16_t)(sign | h);
    }
}

static float h2f(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
    float f;
    if (e == 0) {
        if (m == 0) x = sign;
        else { f = (float)m * (1.0f / 16777216.0f); memcpy(&x, &f, 4); x |= sign; }     /* m * 2^-24, exact */
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    memcpy(&f, &x, 4);
    return f;
}

uint16_t orc_f2h(float f) { return f2h_rne(f); }
float orc_h2f(uint16_t h) { return h2f(h); }

static inline rgbvec node16(const uint16_t *l16, int n, int r, int g, int b)
{
    const uint16_t *p = &l16[(((size_t)r * n + g) * n + b) * 3];
    rgbvec v = {h2f(p[0]), h2f(p[1]), h2f(p[2])};
    return v;
}

static inline float med3f(float a, float b, float c)
{
    const float lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}

/* one pixel of the fast variant; returns codes before the YUV side */
static inline void lut_pixel_fast(const uint16_t *l16, int n, int mode, const lut_consts *k,
                                  int r, int g, int b, int *ro, int *go, int *bo)
{
    const rgbvec rgb = {(float)r * k->scale_f, (float)g * k->scale_f, (float)b * k->scale_f};
    const rgbvec s = {clipf(rgb.r * k->scale_c[0], 0, k->lut_max), clipf(rgb.g * k->scale_c[1], 0, k->lut_max),
                      clipf(rgb.b * k->scale_c[2], 0, k->lut_max)};
    rgbvec v;
    if (mode == ORC_NEAREST) {
        v = node16(l16, n, NEAR(s.r), NEAR(s.g), NEAR(s.b));
    } else {
        const int p[3] = {PREV(s.r), PREV(s.g), PREV(s.b)};
        const int x[3] = {NEXT(s.r, n), NEXT(s.g, n), NEXT(s.b, n)};
        const float dr = s.r - p[0], dg = s.g - p[1], db = s.b - p[2];
        if (mode == ORC_TRILINEAR) {
            const rgbvec c000 = node16(l16, n, p[0], p[1], p[2]), c001 = node16(l16, n, p[0], p[1], x[2]);
            const rgbvec c010 = node16(l16, n, p[0], x[1], p[2]), c011 = node16(l16, n, p[0], x[1], x[2]);
            const rgbvec c100 = node16(l16, n, x[0], p[1], p[2]), c101 = node16(l16, n, x[0], p[1], x[2]);
            const rgbvec c110 = node16(l16, n, x[0], x[1], p[2]), c111 = node16(l16, n, x[0], x[1], x[2]);
            /* The kernels stage each node with the fp16 difference to its r + 1 neighbour (csrc/lutr_tile2.hip Node::rec): the four  */
            /* lerps along r use that rounded difference (the subtraction itself is exact in fp32), the later ones are plain fp32.  */
#define FL(a, b, f) fmaf((b) - (a), (f), (a))
#define FLR(a, b, f) fmaf(h2f(f2h_rne((b) - (a))), (f), (a))
#define TRI16(ch) \
            { \
                const float c00 = FLR(c000.ch, c100.ch, dr), c10 = FLR(c010.ch, c110.ch, dr); \
                const float c01 = FLR(c001.ch, c101.ch, dr), c11 = FLR(c011.ch, c111.ch, dr); \
                const float c0 = FL(c00, c10, dg), c1 = FL(c01, c11, dg); \
                v.ch = FL(c0, c1, db); \
            }
            TRI16(r) TRI16(g) TRI16(b)
#undef TRI16
#undef FLR
#undef FL
        } else {
            /* the kernels' sorted form: (1-x) c000 + (x-y) cA + (y-z) cB + z c111, cA one step along the axis of the
             * largest fraction, cB all steps but the one along the smallest */
            const float mx = fmaxf(fmaxf(dr, dg), db), md = med3f(dr, dg, db), mn = fminf(fminf(dr, dg), db);
            const int rg = dr > dg, gb = dg > db, rb = dr > db;
            const int amax = (rg && rb) ? 0 : (gb ? 1 : 2);
            const int amin = (gb && rb) ? 2 : (rg ? 1 : 0);
            int a[3] = {p[0], p[1], p[2]}, bq[3] = {x[0], x[1], x[2]};
            a[amax] = x[amax];
            bq[amin] = p[amin];
            const rgbvec c0 = node16(l16, n, p[0], p[1], p[2]), c1 = node16(l16, n, a[0], a[1], a[2]);
            const rgbvec c2 = node16(l16, n, bq[0], bq[1], bq[2]), c3 = node16(l16, n, x[0], x[1], x[2]);
            const float w0 = 1.0f - mx, w1 = mx - md, w2 = md - mn, w3 = mn;
            v.r = fmaf(w3, c3.r, fmaf(w2, c2.r, fmaf(w1, c1.r, w0 * c0.r)));
            v.g = fmaf(w3, c3.g, fmaf(w2, c2.g, fmaf(w1, c1.g, w0 * c0.g)));
            v.b = fmaf(w3, c3.b, fmaf(w2, c2.b, fmaf(w1, c1.b, w0 * c0.b)));
        }
    }
    /* nodes are already scaled by M: truncate, clip */
    {
        const float t[3] = {truncf(v.r), truncf(v.g), truncf(v.b)};
        int *o[3] = {ro, go, bo};
        for (int c = 0; c < 3; c++) {
            int i = t[c] >= 2147483648.0f ? INT32_MAX : (t[c] > -2147483648.0f ? (int)t[c] : INT32_MIN);
            *o[c] = i < 0 ? 0 : (i > k->maxi ? k->maxi : i);
        }
    }
}


static inline float clip_floor(float v, float hi)
{
    return clipf(floorf(v), 0.0f, hi);
}

static void yuv_slice(void *arg, int y0, int y1)
{
    yuv_job *j = (yuv_job *)arg;
    const orc_yuv_consts *k = j->k;
    const int wi = j->din > 8, wo = j->dout > 8;
    const int bw = 1 << j->csx, bh = 1 << j->csy;
    /* one chroma block at a time: bw x bh luma samples share one (Cb,Cr) */
    for (int by = y0; by < y1; by += bh) {
        for (int bx = 0; bx < j->w; bx += bw) {
            const int cx = bx >> j->csx, cy = by >> j->csy;
            float cbv = (float)ld(j->src[1] + cy * j->ss[1], cx, wi);
            float crv = (float)ld(j->src[2] + cy * j->ss[2], cx, wi);
            float rs = 0.f, gs = 0.f, bs = 0.f;
            if (k->pre) {
                cbv = clip_floor(fmaf(k->pc, cbv, k->pcb), k->pre_max);
                crv = clip_floor(fmaf(k->pc, crv, k->pcb), k->pre_max);
            }
            const float cb = cbv - k->coff;
            const float cr = crv - k->coff;
            const float rv = k->krv * cr;
            const float gv = fmaf(k->kgu, cb, k->kgv * cr);
            const float bu = k->kbu * cb;
            for (int dy = 0; dy < bh; dy++) {
                /* odd sizes: the edge sample is replicated into the block */
                const int y = by + dy < j->h ? by + dy : j->h - 1;
                for (int dx = 0; dx < bw; dx++) {
                    const int x = bx + dx < j->w ? bx + dx : j->w - 1;
                    float yv = (float)ld(j->src[0] + y * j->ss[0], x, wi);
                    int ro, go, bo;
                    if (k->pre)
                        yv = clip_floor(fmaf(k->py, yv, k->pyb), k->pre_max);
                    const float yy = fmaf(k->ky, yv, k->yb);
                    const float rf = clip_floor(yy + rv, k->max_l);
                    const float gf = clip_floor(yy + gv, k->max_l);
                    const float bf = clip_floor(yy + bu, k->max_l);
                    if (j->lat16)
                        lut_pixel_fast(j->lat16, j->lut->n, j->mode, &j->lk, (int)rf, (int)gf, (int)bf, &ro, &go, &bo);
                    else
                        lut_pixel(j->lut, j->mode, j->lk.scale_f, j->lk.scale_c, j->lk.lut_max,
                                  j->lk.maxf, j->lk.maxi, (int)rf, (int)gf, (int)bf, &ro, &go, &bo);
                    rs += (float)ro; gs += (float)go; bs += (float)bo;
                    if (by + dy < j->h && bx + dx < j->w) {
                        const float yf = fmaf(k->cyr, (float)ro, fmaf(k->cyg, (float)go, fmaf(k->cyb, (float)bo, k->yob)));
                        if (j->fdst[0])
                            j->fdst[0][(size_t)y * j->fw[0] + x] = yf - 0.5f;
                        else
                            st(j->dst[0] + y * j->ds[0], x, wo, (int)clip_floor(yf, k->max_o));
                    }
                }
            }
            {
                const float cbf = fmaf(k->cbr, rs, fmaf(k->cbg, gs, fmaf(k->cbb, bs, k->cob)));
                const float crf = fmaf(k->crr, rs, fmaf(k->crg, gs, fmaf(k->crb, bs, k->cob)));
                if (j->fdst[1]) {
                    j->fdst[1][(size_t)cy * j->fw[1] + cx] = cbf - 0.5f;
                    j->fdst[2][(size_t)cy * j->fw[2] + cx] = crf - 0.5f;
                } else {
                    st(j->dst[1] + cy * j->ds[1], cx, wo, (int)clip_floor(cbf, k->max_o));
                    st(j->dst[2] + cy * j->ds[2], cx, wo, (int)clip_floor(crf, k->max_o));
                }
            }
        }
    }
}

int orc_apply_yuv(const orc_lut *lut, int mode, const orc_yuv_consts *k,
                  int din, int dl, int dout, int csx, int csy, int w, int h,
                  const void *const src[3], const ptrdiff_t sstride[3],
                  void *const dst[3], const ptrdiff_t dstride[3], int nthreads)
{
    yuv_job j;
    if (!lut || !lut->rgb || !k || w < 0 || h < 0 || csx < 0 || csx > 1 || csy < 0 || csy > 1 ||
        din < 8 || din > 16 || dl < 8 || dl > 16 || dout < 8 || dout > 16 ||
        mode < ORC_NEAREST || mode > ORC_PRISM)
        return ORC_EINVAL;
    if (w == 0 || h == 0)
        return 0;
    j.lut = lut; j.mode = mode; j.k = k;
    j.din = din; j.dl = dl; j.dout = dout; j.csx = csx; j.csy = csy; j.w = w; j.h = h;
    for (int c = 0; c < 3; c++) {
        j.src[c] = (const uint8_t *)src[c]; j.ss[c] = sstride[c];
        j.dst[c] = (uint8_t *)dst[c]; j.ds[c] = dstride[c];
        j.fdst[c] = NULL; j.fw[c] = 0;
    }
    j.lat16 = NULL;
    make_lut_consts(lut, dl, &j.lk);
    run_slices(yuv_slice, &j, h, 1 << csy, nthreads);
    return 0;
}

int orc_apply_yuv_fast(const orc_lut *lut, int mode, const orc_yuv_consts *k,
                       int din, int dl, int dout, int csx, int csy, int w, int h,
                       const void *const src[3], const ptrdiff_t sstride[3],
                       void *const dst[3], const ptrdiff_t dstride[3], int nthreads)
{
    yuv_job j;
    if (!lut || !lut->rgb || !k || w < 0 || h < 0 || csx < 0 || csx > 1 || csy < 0 || csy > 1 ||
        din < 8 || din > 16 || (dl != 8 && dl != 10) || dout < 8 || dout > 16 ||
        mode < ORC_NEAREST || mode > ORC_TETRAHEDRAL)
        return ORC_EINVAL;
    if (w == 0 || h == 0)
        return 0;
    const size_t count = (size_t)lut->n * lut->n * lut->n * 3;
    uint16_t *l16 = (uint16_t *)malloc(count * sizeof(uint16_t));
    if (!l16) return ORC_ENOMEM;
    const float m = (float)((1 << dl) - 1);
    for (size_t i = 0; i < count; i++) l16[i] = f2h_rne(lut->rgb[i] * m);
    j.lut = lut; j.mode = mode; j.k = k;
    j.din = din; j.dl = dl; j.dout = dout; j.csx = csx; j.csy = csy; j.w = w; j.h = h;
    for (int c = 0; c < 3; c++) {
        j.src[c] = (const uint8_t *)src[c]; j.ss[c] = sstride[c];
        j.dst[c] = (uint8_t *)dst[c]; j.ds[c] = dstride[c];
        j.fdst[c] = NULL; j.fw[c] = 0;
    }
    j.lat16 = l16;
    make_lut_consts(lut, dl, &j.lk);
    run_slices(yuv_slice, &j, h, 1 << csy, nthreads);
    free(l16);
    return 0;
}

/* ------------------------------------------------------------------ */
/* error-diffusion dither of the final quantisation (SURVEY.md 8a a9,   */
/* reference ffmpeg.py:305-307 `zscale=dither=error_diffusion`).        */
/* The engine's own contract, modelled on zimg's dither_ed [recall]:    */
/* Floyd-Steinberg, every row left to right, errors of the row above    */
/* read with zero padding; float ops in exactly this order.             */
/* ------------------------------------------------------------------ */
void orc_dither_plane(const float *x, int w, int h, float maxv, int wide, void *dst, ptrdiff_t dstride)
{
    float *top = (float *)calloc((size_t)w + 2, sizeof(float));
    float *cur = (float *)calloc((size_t)w + 2, sizeof(float));
    if (!top || !cur) { free(top); free(cur); return; }
    for (int y = 0; y < h; y++) {
        float err_left = 0.0f;
        uint8_t *drow = (uint8_t *)dst + y * dstride;
        for (int j = 0; j < w; j++) {
            /* error rows are padded by one on each side: index j+1 is column j */
            float v = x[(size_t)y * w + j];
            float err = 0.0f;
            float q;
            err += err_left * (7.0f / 16.0f);
            err += top[j + 2] * (3.0f / 16.0f);
            err += top[j + 1] * (5.0f / 16.0f);
            err += top[j] * (1.0f / 16.0f);
            v += err;
            v = fminf(fmaxf(v, 0.0f), maxv);
            q = rintf(v);                       /* round half to even, like lrintf in the default mode */
            err_left = v - q;
            cur[j + 1] = err_left;
            st(drow, j, wide, (int)q);
        }
        { float *t = top; top = cur; cur = t; }
    }
    free(top);
    free(cur);
}

int orc_apply_yuv_dither(const orc_lut *lut, int mode, const orc_yuv_consts *k,
                         int din, int dl, int dout, int csx, int csy, int w, int h,
                         const void *const src[3], const ptrdiff_t sstride[3],
                         void *const dst[3], const ptrdiff_t dstride[3], int nthreads)
{
    yuv_job j;
    if (!lut || !lut->rgb || !k || w < 0 || h < 0 || csx < 0 || csx > 1 || csy < 0 || csy > 1 ||
        din < 8 || din > 16 || dl < 8 || dl > 16 || dout < 8 || dout > 16 ||
        mode < ORC_NEAREST || mode > ORC_PRISM)
        return ORC_EINVAL;
    if (w == 0 || h == 0)
        return 0;
    const int cw = (w + (1 << csx) - 1) >> csx, chh = (h + (1 << csy) - 1) >> csy;
    j.lut = lut; j.mode = mode; j.k = k;
    j.din = din; j.dl = dl; j.dout = dout; j.csx = csx; j.csy = csy; j.w = w; j.h = h;
    for (int c = 0; c < 3; c++) {
        j.src[c] = (const uint8_t *)src[c]; j.ss[c] = sstride[c];
        j.dst[c] = (uint8_t *)dst[c]; j.ds[c] = dstride[c];
        j.fw[c] = c ? cw : w;
        j.fdst[c] = (float *)malloc((size_t)j.fw[c] * (c ? chh : h) * sizeof(float));
        if (!j.fdst[c]) {
            for (int d = 0; d < c; d++) free(j.fdst[d]);
            return ORC_ENOMEM;
        }
    }
    j.lat16 = NULL;
    make_lut_consts(lut, dl, &j.lk);
    run_slices(yuv_slice, &j, h, 1 << csy, nthreads);
    for (int c = 0; c < 3; c++) {
        orc_dither_plane(j.fdst[c], j.fw[c], c ? chh : h, k->max_o, dout > 8, dst[c], dstride[c]);
        free(j.fdst[c]);
    }
    return 0;
}
