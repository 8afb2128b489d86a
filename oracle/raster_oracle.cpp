/*
 * raster_oracle.cpp -- CPU restatement of the reference's draw path.
 *
 * TEST INFRASTRUCTURE ONLY (see raster_oracle.h).  PARITY UNPINNED: the
 * reference delegates this arithmetic to the Vulkan ICD and holds no fixtures.
 *
 * The pipeline restated here is the one the reference records and submits:
 *   vertex fetch (obj_loader.cpp:98-109, 173-205: pos3 [normal3] [uv2], u32 indices)
 *   -> vertex shader (test_ktx.vert:26-34, push_constants.vert:26-34, dummy.vert.glsl:20-26)
 *   -> clip / divide / viewport (Vulkan 1.0 "Primitive Clipping", "Coordinate Transformations")
 *   -> cull + top-left-rule rasterisation (state: test_ktx_app.cpp:435-464)
 *   -> depth test/write on D24 (test_ktx_app.cpp:476-485, 90-100)
 *   -> fragment shader (test_ktx.frag:13-17, push_constants.frag:11-15, dummy.frag.glsl:11-14)
 *   -> colour store with sRGB encode (vk_app.cpp:197-221 prefers *_SRGB), 4x resolve
 *      (dummy_obj_viewer_app.cpp:468-520).
 *
 * Every floating-point step is written as an explicit sequence of IEEE-754
 * binary32 operations (+ - * / sqrt fma, int<->float conversions).  Build with
 * -ffp-contract=off and without -ffast-math: then each step has exactly one
 * correctly rounded result, and an independent implementation that follows the
 * same sequence reproduces it bit for bit.
 */
#include "raster_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

namespace {

/* ---------------------------------------------------------------- scalar rules */

inline float fma32(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

inline bool finite32(float v) { return fabsf(v) <= 3.4028234663852886e38f; }

const int SUBPIXEL_BITS = 8; /* 24.8 window coordinates */
const int SUBPIXEL_ONE = 1 << SUBPIXEL_BITS;
const int SUBPIXEL_HALF = SUBPIXEL_ONE / 2;
const float GUARD_BAND = 2.0f; /* x,y are clipped to [-2w, 2w]; the scissor does the rest */

/* window coordinate -> fixed point: floor(v * 256 + 0.5), one fma then floor */
inline int32_t snap(float v)
{
    float t = floorf(fma32(v, 256.0f, 0.5f));
    t = fminf(fmaxf(t, -2147483520.0f), 2147483520.0f);
    return (int32_t)t;
}

/* z in [0,1] -> D24: the float product z * (2^24 - 1) rounded to the nearest integer (ties to even).
 * (Adding 0.5 and truncating is not that: from 2^23 on a float has no fraction bit left, the sum is itself
 * rounded to an integer, and every value whose fraction is below one half comes out one too high.) */
inline uint32_t depth24(float z)
{
    z = fminf(fmaxf(z, 0.0f), 1.0f);
    uint32_t d = (uint32_t)rintf(z * 16777215.0f);
    return d > 0xFFFFFFu ? 0xFFFFFFu : d;
}

/* sRGB tables.  decode[k] is the float nearest to the exact linear value of
 * code k; encode thresholds thr[k] (k=1..255) are the floats nearest to the
 * linear value of code k-0.5, and a linear value c encodes to the number of
 * thresholds <= c.  That is round-to-nearest of 255*srgb(c) up to the float
 * rounding of the thresholds. */
struct srgb_tables {
    float decode[256];
    float thr[257];
    srgb_tables()
    {
        for (int k = 0; k < 256; ++k) {
            decode[k] = (float)to_linear(k / 255.0);
        }
        thr[0] = -INFINITY;
        for (int k = 1; k < 256; ++k) {
            thr[k] = (float)to_linear((k - 0.5) / 255.0);
        }
        thr[256] = INFINITY;
    }
    static double to_linear(double s)
    {
        return s <= 0.04045 ? s / 12.92 : pow((s + 0.055) / 1.055, 2.4);
    }
};
const srgb_tables g_srgb;

inline uint32_t srgb_encode(float c)
{
    if (!(c > 0.0f)) return 0; /* also NaN */
    /* number of k in 1..255 with thr[k] <= c */
    int lo = 0, hi = 255; /* invariant: thr[lo] <= c, answer in [lo, hi] */
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (g_srgb.thr[mid] <= c) lo = mid; else hi = mid - 1;
    }
    return (uint32_t)lo;
}

inline uint32_t unorm8(float c)
{
    if (!(c > 0.0f)) return 0;
    c = fminf(c, 1.0f);
    return (uint32_t)fma32(c, 255.0f, 0.5f);
}

/* log2 for LOD selection: exponent + degree-5 polynomial on the mantissa */
inline float lod_log2(float x)
{
    if (!(x >= 1.17549435e-38f)) return -126.0f; /* zero, denormal, negative, NaN */
    if (!finite32(x)) return 128.0f;
    uint32_t bits;
    memcpy(&bits, &x, 4);
    int e = (int)((bits >> 23) & 0xFF) - 127;
    uint32_t mb = (bits & 0x7FFFFFu) | 0x3F800000u;
    float m;
    memcpy(&m, &mb, 4);
    float t = m - 1.0f;
    float p = -0.03459521010518074f;
    p = fma32(p, t, 0.14643360674381256f);
    p = fma32(p, t, -0.30338966846466064f);
    p = fma32(p, t, 0.469301700592041f);
    p = fma32(p, t, -0.7204423546791077f);
    p = fma32(p, t, 1.442683219909668f);
    return fma32(p, t, (float)e);
}

/* ---------------------------------------------------------------- matrices (column-major m[c*4+r]) */

/* C = A*B with one product then three fmas per element */
void mat4_mul_pinned(const float* a, const float* b, float* out)
{
    float r[16];
    for (int c = 0; c < 4; ++c) {
        for (int row = 0; row < 4; ++row) {
            float acc = a[0 * 4 + row] * b[c * 4 + 0];
            acc = fma32(a[1 * 4 + row], b[c * 4 + 1], acc);
            acc = fma32(a[2 * 4 + row], b[c * 4 + 2], acc);
            acc = fma32(a[3 * 4 + row], b[c * 4 + 3], acc);
            r[c * 4 + row] = acc;
        }
    }
    memcpy(out, r, sizeof(r));
}

/* M * vec4(p, 1) */
inline void mat4_point(const float* m, const float* p, float* out)
{
    for (int row = 0; row < 4; ++row) {
        float acc = fma32(m[2 * 4 + row], p[2], m[3 * 4 + row]);
        acc = fma32(m[1 * 4 + row], p[1], acc);
        acc = fma32(m[0 * 4 + row], p[0], acc);
        out[row] = acc;
    }
}

/* inverse(transpose(mat3(M))) = cofactor(A)/det(A), A = upper-left 3x3 of M.  out is column-major 3x3. */
void normal_matrix(const float* m4, float* out)
{
    /* a[c][r] */
    float a00 = m4[0], a01 = m4[1], a02 = m4[2];
    float a10 = m4[4], a11 = m4[5], a12 = m4[6];
    float a20 = m4[8], a21 = m4[9], a22 = m4[10];
    /* cofactors, c[c][r] of A (column c, row r) */
    float c00 = fma32(a11, a22, -(a12 * a21));
    float c01 = fma32(a12, a20, -(a10 * a22));
    float c02 = fma32(a10, a21, -(a11 * a20));
    float c10 = fma32(a21, a02, -(a22 * a01));
    float c11 = fma32(a22, a00, -(a20 * a02));
    float c12 = fma32(a20, a01, -(a21 * a00));
    float c20 = fma32(a01, a12, -(a02 * a11));
    float c21 = fma32(a02, a10, -(a00 * a12));
    float c22 = fma32(a00, a11, -(a01 * a10));
    float det = fma32(a02, c02, fma32(a01, c01, a00 * c00));
    float inv = 1.0f / det;
    out[0] = c00 * inv; out[1] = c01 * inv; out[2] = c02 * inv;
    out[3] = c10 * inv; out[4] = c11 * inv; out[5] = c12 * inv;
    out[6] = c20 * inv; out[7] = c21 * inv; out[8] = c22 * inv;
}

inline void mat3_vec(const float* m, const float* v, float* out)
{
    for (int row = 0; row < 3; ++row) {
        float acc = m[0 * 3 + row] * v[0];
        acc = fma32(m[1 * 3 + row], v[1], acc);
        acc = fma32(m[2 * 3 + row], v[2], acc);
        out[row] = acc;
    }
}

/* ---------------------------------------------------------------- texture unit */

inline uint32_t level_dim(uint32_t d, uint32_t l) { uint32_t v = d >> l; return v ? v : 1u; }

inline void fetch_texel(const orc_texture* t, uint32_t level, int x, int y, float* rgba)
{
    uint32_t w = level_dim(t->width, level);
    const uint8_t* p = t->level_data[level] + ((size_t)y * w + (size_t)x) * 4;
    if (t->srgb) {
        rgba[0] = g_srgb.decode[p[0]];
        rgba[1] = g_srgb.decode[p[1]];
        rgba[2] = g_srgb.decode[p[2]];
    } else {
        rgba[0] = (float)p[0] / 255.0f;
        rgba[1] = (float)p[1] / 255.0f;
        rgba[2] = (float)p[2] / 255.0f;
    }
    rgba[3] = (float)p[3] / 255.0f;
}

inline int wrap_coord(int i, int n, int tiled)
{
    if (tiled) {
        int m = i % n;
        return m < 0 ? m + n : m;
    }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

void sample_level(const orc_texture* t, uint32_t level, float u, float v, float* rgba)
{
    int w = (int)level_dim(t->width, level), h = (int)level_dim(t->height, level);
    if (!finite32(u)) u = 0.0f;
    if (!finite32(v)) v = 0.0f;
    if (t->filter_linear) {
        float fu = fma32(u, (float)w, -0.5f);
        float fv = fma32(v, (float)h, -0.5f);
        fu = fminf(fmaxf(fu, -1.0e9f), 1.0e9f);
        fv = fminf(fmaxf(fv, -1.0e9f), 1.0e9f);
        float iu = floorf(fu), iv = floorf(fv);
        float ax = fu - iu, ay = fv - iv;
        int x0 = wrap_coord((int)iu, w, t->tiled), x1 = wrap_coord((int)iu + 1, w, t->tiled);
        int y0 = wrap_coord((int)iv, h, t->tiled), y1 = wrap_coord((int)iv + 1, h, t->tiled);
        float c00[4], c10[4], c01[4], c11[4];
        fetch_texel(t, level, x0, y0, c00);
        fetch_texel(t, level, x1, y0, c10);
        fetch_texel(t, level, x0, y1, c01);
        fetch_texel(t, level, x1, y1, c11);
        for (int k = 0; k < 4; ++k) {
            float top = fma32(ax, c10[k] - c00[k], c00[k]);
            float bot = fma32(ax, c11[k] - c01[k], c01[k]);
            rgba[k] = fma32(ay, bot - top, top);
        }
    } else {
        float fu = fminf(fmaxf(u * (float)w, -1.0e9f), 1.0e9f);
        float fv = fminf(fmaxf(v * (float)h, -1.0e9f), 1.0e9f);
        int x = wrap_coord((int)floorf(fu), w, t->tiled);
        int y = wrap_coord((int)floorf(fv), h, t->tiled);
        fetch_texel(t, level, x, y, rgba);
    }
}

/* texture() with implicit derivatives; sampler of tools.cpp:60-82: lod in [0, levels-1], bias 0 */
void sample_texture(const orc_texture* t, float u, float v, float dudx, float dvdx, float dudy, float dvdy, float* rgba)
{
    float w0 = (float)t->width, h0 = (float)t->height;
    float mxu = dudx * w0, mxv = dvdx * h0, myu = dudy * w0, myv = dvdy * h0;
    float rx = fma32(mxv, mxv, mxu * mxu);
    float ry = fma32(myv, myv, myu * myu);
    float rho2 = fmaxf(rx, ry);
    float lambda = 0.5f * lod_log2(rho2);
    float max_lod = (float)(t->levels - 1);
    lambda = fminf(fmaxf(lambda, 0.0f), max_lod);
    if (!(lambda > 0.0f) || t->levels == 1) {
        sample_level(t, 0, u, v, rgba); /* magnification, or a single level */
        return;
    }
    if (t->filter_linear) {
        float fl = floorf(lambda);
        uint32_t lo = (uint32_t)fl;
        uint32_t hi = lo + 1 < t->levels ? lo + 1 : t->levels - 1;
        float frac = lambda - fl;
        float a[4], b[4];
        sample_level(t, lo, u, v, a);
        sample_level(t, hi, u, v, b);
        for (int k = 0; k < 4; ++k) rgba[k] = fma32(frac, b[k] - a[k], a[k]);
    } else {
        /* MIPMAP_MODE_NEAREST: level = ceil(lambda + 0.5) - 1 */
        float lv = ceilf(lambda + 0.5f) - 1.0f;
        lv = fminf(fmaxf(lv, 0.0f), max_lod);
        sample_level(t, (uint32_t)lv, u, v, rgba);
    }
}

/* ---------------------------------------------------------------- geometry */

struct clip_vertex {
    float x, y, z, w;
    float b[3]; /* weights over the three corners of the parent primitive */
};

struct snapped_vertex {
    int32_t x, y; /* 24.8 */
    float z;      /* viewport depth */
    float iw;     /* 1/w */
};

struct view_state {
    float vx, vy, vw, vh, zmin, zmax;
    float half_w, half_h, ox, oy, pz;
    int cx0, cy0, cx1, cy1; /* inclusive pixel clamp rectangle */
};

void make_view_state(const orc_pass* p, int band_y0, int band_y1, view_state* vs)
{
    vs->vx = p->viewport[0]; vs->vy = p->viewport[1];
    vs->vw = p->viewport[2]; vs->vh = p->viewport[3];
    vs->zmin = p->viewport[4]; vs->zmax = p->viewport[5];
    vs->half_w = vs->vw * 0.5f;
    vs->half_h = vs->vh * 0.5f;
    vs->ox = vs->vx + vs->half_w;
    vs->oy = vs->vy + vs->half_h;
    vs->pz = vs->zmax - vs->zmin;
    /* pixels whose centre lies in the viewport rectangle */
    int vx0 = (int)ceilf(vs->vx - 0.5f), vy0 = (int)ceilf(vs->vy - 0.5f);
    int vx1 = (int)ceilf(vs->vx + vs->vw - 0.5f) - 1, vy1 = (int)ceilf(vs->vy + vs->vh - 0.5f) - 1;
    int x0 = std::max(std::max(vx0, p->scissor[0]), std::max(p->area[0], 0));
    int y0 = std::max(std::max(vy0, p->scissor[1]), std::max(p->area[1], 0));
    int x1 = std::min(std::min(vx1, p->scissor[0] + p->scissor[2] - 1), std::min(p->area[0] + p->area[2] - 1, (int)p->width - 1));
    int y1 = std::min(std::min(vy1, p->scissor[1] + p->scissor[3] - 1), std::min(p->area[1] + p->area[3] - 1, (int)p->height - 1));
    vs->cx0 = x0; vs->cx1 = x1;
    vs->cy0 = std::max(y0, band_y0); vs->cy1 = std::min(y1, band_y1);
}

/* outcode bits */
enum {
    OC_XNEG = 1, OC_XPOS = 2, OC_YNEG = 4, OC_YPOS = 8, OC_ZNEG = 16, OC_ZPOS = 32,
    OC_GUARD = 64,  /* outside the guard band */
    OC_INVALID = 128 /* non-finite coordinate or w <= 0 where a divide would be needed */
};

inline uint32_t outcode(const float* c)
{
    float x = c[0], y = c[1], z = c[2], w = c[3];
    if (!(finite32(x) && finite32(y) && finite32(z) && finite32(w))) return OC_INVALID;
    uint32_t oc = 0;
    if (x < -w) oc |= OC_XNEG;
    if (x > w) oc |= OC_XPOS;
    if (y < -w) oc |= OC_YNEG;
    if (y > w) oc |= OC_YPOS;
    if (z < 0.0f) oc |= OC_ZNEG;
    if (z > w) oc |= OC_ZPOS;
    float gw = GUARD_BAND * w;
    if (x < -gw || x > gw || y < -gw || y > gw) oc |= OC_GUARD;
    return oc;
}

inline void project(const view_state* vs, float x, float y, float z, float w, snapped_vertex* out)
{
    float iw = 1.0f / w;
    float xd = x * iw, yd = y * iw, zd = z * iw;
    float xf = fma32(xd, vs->half_w, vs->ox);
    float yf = fma32(yd, vs->half_h, vs->oy);
    out->x = snap(xf);
    out->y = snap(yf);
    out->z = fma32(zd, vs->pz, vs->zmin);
    out->iw = iw;
}

/* signed distance of a clip-space point to plane k (inside: d >= 0) */
inline float plane_distance(int k, const clip_vertex& v)
{
    switch (k) {
    case 0: return v.z;
    case 1: return v.w - v.z;
    case 2: return fma32(GUARD_BAND, v.w, v.x);
    case 3: return fma32(GUARD_BAND, v.w, -v.x);
    case 4: return fma32(GUARD_BAND, v.w, v.y);
    default: return fma32(GUARD_BAND, v.w, -v.y);
    }
}

/* Sutherland-Hodgman against near, far and the four guard-band planes.  The
 * crossing point is always computed from the inside end to the outside end so
 * two primitives sharing an edge produce the same new vertex. */
int clip_polygon(clip_vertex* poly /* in/out, capacity 12 */, int n)
{
    clip_vertex tmp[12];
    for (int k = 0; k < 6; ++k) {
        float d[12];
        bool any_out = false;
        for (int i = 0; i < n; ++i) {
            d[i] = plane_distance(k, poly[i]);
            if (!(d[i] >= 0.0f)) any_out = true;
        }
        if (!any_out) continue;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            int j = (i + 1 == n) ? 0 : i + 1;
            bool in_i = d[i] >= 0.0f, in_j = d[j] >= 0.0f;
            if (in_i) tmp[m++] = poly[i];
            if (in_i != in_j) {
                const clip_vertex& a = in_i ? poly[i] : poly[j]; /* inside end */
                const clip_vertex& b = in_i ? poly[j] : poly[i]; /* outside end */
                float da = in_i ? d[i] : d[j], db = in_i ? d[j] : d[i];
                float t = da / (da - db);
                clip_vertex r;
                r.x = fma32(t, b.x - a.x, a.x);
                r.y = fma32(t, b.y - a.y, a.y);
                r.z = fma32(t, b.z - a.z, a.z);
                r.w = fma32(t, b.w - a.w, a.w);
                r.b[0] = fma32(t, b.b[0] - a.b[0], a.b[0]);
                r.b[1] = fma32(t, b.b[1] - a.b[1], a.b[1]);
                r.b[2] = fma32(t, b.b[2] - a.b[2], a.b[2]);
                tmp[m++] = r;
            }
        }
        n = m;
        for (int i = 0; i < n; ++i) poly[i] = tmp[i];
        if (n < 3) return 0;
    }
    return n;
}

/* A triangle ready for rasterisation: oriented so the edge functions are
 * positive inside. w[k] weights corner k. */
struct raster_tri {
    snapped_vertex v[3];
    float bary[3][3]; /* bary[k] = weights of corner k over the parent's corners */
    bool clipped;
    int64_t area2;    /* > 0 */
    float inv_area;
};

/* sample offsets in 1/256 pixel */
const int SAMPLE_X1[1] = {128};
const int SAMPLE_Y1[1] = {128};
const int SAMPLE_X4[4] = {96, 224, 32, 160};  /* (0.375,0.125) (0.875,0.375) (0.125,0.625) (0.625,0.875) */
const int SAMPLE_Y4[4] = {32, 96, 160, 224};

inline int64_t edge_fn(const snapped_vertex& a, const snapped_vertex& b, int64_t px, int64_t py)
{
    return (int64_t)(b.x - a.x) * (py - a.y) - (int64_t)(b.y - a.y) * (px - a.x);
}

/* top-left rule: with positive-inside orientation an edge a->b owns its own
 * points when it goes up (dy < 0: a left edge) or is horizontal going right
 * (dy == 0, dx > 0: a top edge) in y-down window space. */
inline bool owns_edge(const snapped_vertex& a, const snapped_vertex& b)
{
    int dy = b.y - a.y, dx = b.x - a.x;
    return dy < 0 || (dy == 0 && dx > 0);
}

/* orient + cull.  returns false if the triangle produces no fragments */
bool setup_triangle(raster_tri* t, int cull_mode, int front_face)
{
    int64_t s = edge_fn(t->v[0], t->v[1], t->v[2].x, t->v[2].y);
    if (s == 0) return false;
    /* Vulkan area a = -s/2 in y-down window space; CCW front face <=> a > 0 <=> s < 0 */
    bool front = (front_face == 0) ? (s < 0) : (s > 0);
    if ((cull_mode & 1) && front) return false;
    if ((cull_mode & 2) && !front) return false;
    if (s < 0) {
        std::swap(t->v[1], t->v[2]);
        for (int j = 0; j < 3; ++j) std::swap(t->bary[1][j], t->bary[2][j]);
        s = -s;
    }
    t->area2 = s;
    t->inv_area = 1.0f / (float)s;
    return true;
}

/* ---------------------------------------------------------------- per-draw state */

struct draw_state {
    const orc_draw* d;
    float clip_m[16];
    float nrm_m[9];
    uint32_t tri_count;
    uint32_t prim_base; /* running primitive number of triangle 0 */
};

void make_draw_state(const orc_draw* d, uint32_t prim_base, draw_state* ds)
{
    ds->d = d;
    ds->tri_count = d->index_count / 3;
    ds->prim_base = prim_base;
    const float* mvp = d->ubo + 64;
    const float* model = d->ubo + 48;
    if (d->shader == ORC_SHADER_DUMMY) {
        memcpy(ds->clip_m, mvp, 64);
        memset(ds->nrm_m, 0, sizeof(ds->nrm_m));
    } else {
        /* gl_Position = u_mvp * transform * vec4(pos,1): (mvp*transform) first (test_ktx.vert:28) */
        mat4_mul_pinned(mvp, d->push + 4, ds->clip_m);
        float mt[16];
        mat4_mul_pinned(model, d->push + 4, mt);
        normal_matrix(mt, ds->nrm_m); /* inverse(transpose(mat3(u_model*transform))) (test_ktx.vert:30) */
    }
}

inline const uint8_t* vertex_ptr(const orc_draw* d, uint32_t index)
{
    int64_t vi = (int64_t)index + d->vertex_offset;
    return d->vb + (size_t)vi * d->stride;
}

inline bool vertex_in_range(const orc_draw* d, uint32_t index)
{
    int64_t vi = (int64_t)index + d->vertex_offset;
    return vi >= 0 && (uint64_t)(vi + 1) * d->stride <= d->vb_bytes;
}

inline void load_attr(const orc_draw* d, uint32_t index, int off, int n, float* out)
{
    for (int k = 0; k < n; ++k) out[k] = 0.0f;
    if (off < 0) return;
    memcpy(out, vertex_ptr(d, index) + off, (size_t)n * 4);
}

/* vertex shader position output, including the reference's clip-space "y = 1 - y" */
inline void vs_position(const draw_state* ds, uint32_t index, float* clip)
{
    float pos[3];
    load_attr(ds->d, index, ds->d->off_pos, 3, pos);
    mat4_point(ds->clip_m, pos, clip);
    clip[1] = 1.0f - clip[1];
}

/* varyings used by the fragment stage: the light term, uv (2), colour (3).
 * The fragment shaders use v_normal only in dot(v_normal, normalize(vec3(0, 1, -1))) (test_ktx.frag:14-15,
 * push_constants.frag:12-13).  The dot product is linear in the normal, so it is taken here, at the vertex, and
 * interpolated: the same value as interpolating the normal and taking the dot per fragment, up to float rounding. */
struct varyings { float light; float uv[2]; float col[3]; };

inline void vs_varyings(const draw_state* ds, uint32_t index, varyings* out)
{
    const orc_draw* d = ds->d;
    float nrm[3], uv[2];
    load_attr(d, index, d->off_nrm, 3, nrm);
    load_attr(d, index, d->off_uv, 2, uv);
    if (d->shader == ORC_SHADER_DUMMY) {
        out->light = 0.0f; /* the dummy shader does not light */
        out->uv[0] = uv[0];
        out->uv[1] = 1.0f - uv[1]; /* dummy.vert.glsl:25 */
        out->col[0] = out->col[1] = out->col[2] = 0.0f;
    } else {
        float len2 = fma32(nrm[2], nrm[2], fma32(nrm[1], nrm[1], nrm[0] * nrm[0]));
        float len = sqrtf(len2);
        float nn[3] = {nrm[0] / len, nrm[1] / len, nrm[2] / len};
        float tn[3];
        mat3_vec(ds->nrm_m, nn, tn); /* v_normal (test_ktx.vert:30) */
        /* normalize(vec3(0,1,-1)) as float constants */
        const float lx = 0.0f, ly = 0.70710678118654752f, lz = -0.70710678118654752f;
        out->light = fma32(tn[2], lz, fma32(tn[1], ly, tn[0] * lx));
        out->uv[0] = uv[0]; out->uv[1] = uv[1];
        out->col[0] = d->push[0]; out->col[1] = d->push[1]; out->col[2] = d->push[2];
    }
}

/* ---------------------------------------------------------------- primitive expansion */

struct prim_tris {
    raster_tri tri[8];
    uint8_t sub[8];
    int count;
};

/* Fetch, transform, reject, clip, project, cull: every triangle this primitive rasterises. */
void expand_primitive(const draw_state* ds, const view_state* vs, uint32_t tri, prim_tris* out)
{
    out->count = 0;
    const orc_draw* d = ds->d;
    uint32_t idx[3];
    for (int k = 0; k < 3; ++k) {
        idx[k] = d->ib[(size_t)d->first_index + (size_t)tri * 3 + k];
        if (!vertex_in_range(d, idx[k])) return; /* out-of-bounds fetch: primitive dropped */
    }
    float clip[3][4];
    uint32_t oc[3];
    for (int k = 0; k < 3; ++k) {
        vs_position(ds, idx[k], clip[k]);
        oc[k] = outcode(clip[k]);
    }
    uint32_t any = oc[0] | oc[1] | oc[2];
    if (any & OC_INVALID) return;
    if (oc[0] & oc[1] & oc[2] & 63u) return; /* all outside one frustum plane */
    if (!(any & (OC_ZNEG | OC_ZPOS | OC_GUARD))) {
        /* z in [0,w] for all three, so w >= 0; w == 0 cannot be projected */
        if (!(clip[0][3] > 0.0f && clip[1][3] > 0.0f && clip[2][3] > 0.0f)) return;
        raster_tri t;
        for (int k = 0; k < 3; ++k) {
            project(vs, clip[k][0], clip[k][1], clip[k][2], clip[k][3], &t.v[k]);
            for (int j = 0; j < 3; ++j) t.bary[k][j] = (j == k) ? 1.0f : 0.0f;
        }
        t.clipped = false;
        if (setup_triangle(&t, d->cull_mode, d->front_face)) {
            out->tri[0] = t;
            out->sub[0] = 0;
            out->count = 1;
        }
        return;
    }
    clip_vertex poly[12];
    for (int k = 0; k < 3; ++k) {
        poly[k].x = clip[k][0]; poly[k].y = clip[k][1]; poly[k].z = clip[k][2]; poly[k].w = clip[k][3];
        for (int j = 0; j < 3; ++j) poly[k].b[j] = (j == k) ? 1.0f : 0.0f;
    }
    int n = clip_polygon(poly, 3);
    if (n < 3) return;
    snapped_vertex sv[12];
    for (int i = 0; i < n; ++i) {
        if (!(poly[i].w > 0.0f) || !finite32(poly[i].x) || !finite32(poly[i].y) || !finite32(poly[i].z) || !finite32(poly[i].w)) return;
        project(vs, poly[i].x, poly[i].y, poly[i].z, poly[i].w, &sv[i]);
    }
    for (int k = 1; k + 1 < n; ++k) { /* fan (0, k, k+1), sub index k-1 */
        raster_tri t;
        const int c[3] = {0, k, k + 1};
        for (int q = 0; q < 3; ++q) {
            t.v[q] = sv[c[q]];
            for (int j = 0; j < 3; ++j) t.bary[q][j] = poly[c[q]].b[j];
        }
        t.clipped = true;
        if (setup_triangle(&t, d->cull_mode, d->front_face)) {
            out->tri[out->count] = t;
            out->sub[out->count] = (uint8_t)(k - 1);
            out->count++;
        }
    }
}

/* ---------------------------------------------------------------- fragment stage */

/* weights over the parent's corners at window position (px,py) in 24.8 */
inline void parent_weights(const raster_tri* t, int64_t px, int64_t py, float* lam)
{
    float b0 = (float)edge_fn(t->v[1], t->v[2], px, py) * t->inv_area;
    float b1 = (float)edge_fn(t->v[2], t->v[0], px, py) * t->inv_area;
    float b2 = (float)edge_fn(t->v[0], t->v[1], px, py) * t->inv_area;
    float p0 = b0 * t->v[0].iw, p1 = b1 * t->v[1].iw, p2 = b2 * t->v[2].iw;
    float den = (p0 + p1) + p2;
    float rden = 1.0f / den;
    float q0 = p0 * rden, q1 = p1 * rden, q2 = p2 * rden;
    if (!t->clipped) {
        /* corners may have been swapped by orientation; bary rows carry that */
        for (int j = 0; j < 3; ++j) {
            lam[j] = t->bary[0][j] != 0.0f ? q0 : (t->bary[1][j] != 0.0f ? q1 : q2);
        }
    } else {
        for (int j = 0; j < 3; ++j) {
            lam[j] = fma32(q2, t->bary[2][j], fma32(q1, t->bary[1][j], q0 * t->bary[0][j]));
        }
    }
}

inline float interp(const float* lam, float a0, float a1, float a2)
{
    return fma32(lam[2], a2, fma32(lam[1], a1, lam[0] * a0));
}

void shade_fragment(const draw_state* ds, const orc_texture* textures, int n_textures,
                    uint32_t tri, const raster_tri* t, int x, int y, float* rgba)
{
    const orc_draw* d = ds->d;
    uint32_t idx[3];
    for (int k = 0; k < 3; ++k) idx[k] = d->ib[(size_t)d->first_index + (size_t)tri * 3 + k];
    varyings v[3];
    for (int k = 0; k < 3; ++k) vs_varyings(ds, idx[k], &v[k]);
    int64_t px = (int64_t)x * SUBPIXEL_ONE + SUBPIXEL_HALF, py = (int64_t)y * SUBPIXEL_ONE + SUBPIXEL_HALF;
    float lam[3];
    parent_weights(t, px, py, lam);
    float tex[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    bool textured = (d->shader == ORC_SHADER_DUMMY || d->shader == ORC_SHADER_TEST_KTX);
    if (textured) {
        if (d->texture >= 0 && d->texture < n_textures) {
            float uv[2] = {interp(lam, v[0].uv[0], v[1].uv[0], v[2].uv[0]), interp(lam, v[0].uv[1], v[1].uv[1], v[2].uv[1])};
            /* derivatives: differences inside the 2x2 pixel quad, same primitive extrapolated */
            int qx0 = x & ~1, qy0 = y & ~1;
            float lx0[3], lx1[3], ly0[3], ly1[3];
            parent_weights(t, (int64_t)qx0 * SUBPIXEL_ONE + SUBPIXEL_HALF, py, lx0);
            parent_weights(t, (int64_t)(qx0 + 1) * SUBPIXEL_ONE + SUBPIXEL_HALF, py, lx1);
            parent_weights(t, px, (int64_t)qy0 * SUBPIXEL_ONE + SUBPIXEL_HALF, ly0);
            parent_weights(t, px, (int64_t)(qy0 + 1) * SUBPIXEL_ONE + SUBPIXEL_HALF, ly1);
            float dudx = interp(lx1, v[0].uv[0], v[1].uv[0], v[2].uv[0]) - interp(lx0, v[0].uv[0], v[1].uv[0], v[2].uv[0]);
            float dvdx = interp(lx1, v[0].uv[1], v[1].uv[1], v[2].uv[1]) - interp(lx0, v[0].uv[1], v[1].uv[1], v[2].uv[1]);
            float dudy = interp(ly1, v[0].uv[0], v[1].uv[0], v[2].uv[0]) - interp(ly0, v[0].uv[0], v[1].uv[0], v[2].uv[0]);
            float dvdy = interp(ly1, v[0].uv[1], v[1].uv[1], v[2].uv[1]) - interp(ly0, v[0].uv[1], v[1].uv[1], v[2].uv[1]);
            sample_texture(&textures[d->texture], uv[0], uv[1], dudx, dvdx, dudy, dvdy, tex);
        } else {
            tex[0] = tex[1] = tex[2] = 0.0f; /* unbound sampler reads black */
        }
    }
    if (d->shader == ORC_SHADER_DUMMY) {
        rgba[0] = tex[0]; rgba[1] = tex[1]; rgba[2] = tex[2]; rgba[3] = 1.0f; /* dummy.frag.glsl:13 */
        return;
    }
    float diff = fmaxf(interp(lam, v[0].light, v[1].light, v[2].light), 0.0f);
    if (d->shader == ORC_SHADER_TEST_KTX) {
        rgba[0] = tex[0] * diff; rgba[1] = tex[1] * diff; rgba[2] = tex[2] * diff; /* test_ktx.frag:16 */
    } else {
        float c[3] = {interp(lam, v[0].col[0], v[1].col[0], v[2].col[0]), interp(lam, v[0].col[1], v[1].col[1], v[2].col[1]),
                      interp(lam, v[0].col[2], v[1].col[2], v[2].col[2])};
        rgba[0] = c[0] * diff; rgba[1] = c[1] * diff; rgba[2] = c[2] * diff; /* push_constants.frag:14 */
    }
    rgba[3] = 1.0f;
}

/* ---------------------------------------------------------------- band renderer */

struct winner { uint32_t draw, tri; uint8_t sub; uint8_t valid; };

struct band_job {
    const orc_pass* pass;
    const std::vector<draw_state>* draws;
    const orc_texture* textures;
    int n_textures;
    int y0, y1;
    uint8_t* out_color;
    uint32_t* out_depth;
    uint32_t* out_prim;
    int literal;
    const uint32_t* row_range; /* per running primitive number: (ymin << 16) | ymax, or 0xFFFFFFFF when nothing is drawn */
};

inline void store_color(const orc_pass* p, const float* rgba, uint8_t* dst)
{
    uint32_t r, g, b;
    if (p->color_srgb) { r = srgb_encode(rgba[0]); g = srgb_encode(rgba[1]); b = srgb_encode(rgba[2]); }
    else { r = unorm8(rgba[0]); g = unorm8(rgba[1]); b = unorm8(rgba[2]); }
    uint32_t a = unorm8(rgba[3]);
    if (p->color_bgra) { dst[0] = (uint8_t)b; dst[1] = (uint8_t)g; dst[2] = (uint8_t)r; }
    else { dst[0] = (uint8_t)r; dst[1] = (uint8_t)g; dst[2] = (uint8_t)b; }
    dst[3] = (uint8_t)a;
}

void render_band(const band_job* job)
{
    const orc_pass* p = job->pass;
    const int W = (int)p->width, S = (int)p->samples;
    const int* sx = S == 4 ? SAMPLE_X4 : SAMPLE_X1;
    const int* sy = S == 4 ? SAMPLE_Y4 : SAMPLE_Y1;
    view_state vs;
    make_view_state(p, job->y0, job->y1, &vs);
    const int rows = job->y1 - job->y0 + 1;
    if (rows <= 0) return;

    /* per-sample linear colour (literal mode) or winner (deferred mode) */
    std::vector<float> sample_rgba;
    std::vector<winner> win;
    if (job->literal) sample_rgba.resize((size_t)rows * W * S * 4);
    else win.resize((size_t)rows * W * S);
    const uint32_t dclear = depth24(p->clear_depth);
    for (int y = job->y0; y <= job->y1; ++y) {
        for (int x = 0; x < W; ++x) {
            for (int s = 0; s < S; ++s) {
                size_t gi = ((size_t)y * W + x) * S + s;
                size_t li = ((size_t)(y - job->y0) * W + x) * S + s;
                /* loadOp CLEAR applies to the render area; pixels outside it are not touched
                 * (they read back as the zeros a fresh target holds) */
                const bool in_area = x >= p->area[0] && x < p->area[0] + p->area[2] && y >= p->area[1] && y < p->area[1] + p->area[3];
                job->out_depth[gi] = in_area ? dclear : 0u;
                if (job->out_prim) job->out_prim[gi] = 0xFFFFFFFFu;
                if (job->literal) memcpy(&sample_rgba[li * 4], p->clear_color, 16);
                else win[li].valid = 0;
            }
        }
    }

    if (vs.cx0 <= vs.cx1 && vs.cy0 <= vs.cy1) {
        for (size_t di = 0; di < job->draws->size(); ++di) {
            const draw_state* ds = &(*job->draws)[di];
            const orc_draw* d = ds->d;
            for (uint32_t tri = 0; tri < ds->tri_count; ++tri) {
                const uint32_t rr = job->row_range[ds->prim_base + tri];
                if (rr == 0xFFFFFFFFu || (int)(rr >> 16) > vs.cy1 || (int)(rr & 0xFFFFu) < vs.cy0) continue;
                prim_tris pt;
                expand_primitive(ds, &vs, tri, &pt);
                for (int ti = 0; ti < pt.count; ++ti) {
                    const raster_tri* t = &pt.tri[ti];
                    int minx = std::min(t->v[0].x, std::min(t->v[1].x, t->v[2].x));
                    int maxx = std::max(t->v[0].x, std::max(t->v[1].x, t->v[2].x));
                    int miny = std::min(t->v[0].y, std::min(t->v[1].y, t->v[2].y));
                    int maxy = std::max(t->v[0].y, std::max(t->v[1].y, t->v[2].y));
                    /* pixels that own at least one sample position inside the box */
                    int x0 = std::max(minx >> SUBPIXEL_BITS, vs.cx0), x1 = std::min(maxx >> SUBPIXEL_BITS, vs.cx1);
                    int y0 = std::max(miny >> SUBPIXEL_BITS, vs.cy0), y1 = std::min(maxy >> SUBPIXEL_BITS, vs.cy1);
                    if (x0 > x1 || y0 > y1) continue;
                    const bool own0 = owns_edge(t->v[1], t->v[2]);
                    const bool own1 = owns_edge(t->v[2], t->v[0]);
                    const bool own2 = owns_edge(t->v[0], t->v[1]);
                    for (int y = y0; y <= y1; ++y) {
                        for (int x = x0; x <= x1; ++x) {
                            bool shaded = false;
                            float frag[4];
                            for (int s = 0; s < S; ++s) {
                                int64_t px = (int64_t)x * SUBPIXEL_ONE + sx[s], py = (int64_t)y * SUBPIXEL_ONE + sy[s];
                                int64_t e0 = edge_fn(t->v[1], t->v[2], px, py);
                                int64_t e1 = edge_fn(t->v[2], t->v[0], px, py);
                                int64_t e2 = edge_fn(t->v[0], t->v[1], px, py);
                                bool in0 = e0 > 0 || (e0 == 0 && own0);
                                bool in1 = e1 > 0 || (e1 == 0 && own1);
                                bool in2 = e2 > 0 || (e2 == 0 && own2);
                                if (!(in0 && in1 && in2)) continue;
                                /* depth is affine over the window: z0 plus the edge functions of corners 1 and 2 times
                                 * the depth's change per unit of each (a fragment costs two multiply-adds) */
                                float gz1 = (t->v[1].z - t->v[0].z) * t->inv_area, gz2 = (t->v[2].z - t->v[0].z) * t->inv_area;
                                float z = fma32((float)e2, gz2, fma32((float)e1, gz1, t->v[0].z));
                                uint32_t dz = depth24(z);
                                size_t gi = ((size_t)y * W + x) * S + s;
                                size_t li = ((size_t)(y - job->y0) * W + x) * S + s;
                                if (d->depth_test) {
                                    uint32_t stored = job->out_depth[gi];
                                    bool pass = d->depth_op == 1 ? dz < stored : dz <= stored;
                                    if (!pass) continue;
                                    if (d->depth_write) job->out_depth[gi] = dz;
                                }
                                if (job->out_prim) job->out_prim[gi] = ds->prim_base + tri;
                                if (job->literal) {
                                    if (!shaded) {
                                        shade_fragment(ds, job->textures, job->n_textures, tri, t, x, y, frag);
                                        shaded = true;
                                    }
                                    memcpy(&sample_rgba[li * 4], frag, 16);
                                } else {
                                    win[li].draw = (uint32_t)di; win[li].tri = tri; win[li].sub = pt.sub[ti]; win[li].valid = 1;
                                }
                            }
                        }
                    }
                }
            }
        }
    }

    /* resolve + store */
    for (int y = job->y0; y <= job->y1; ++y) {
        for (int x = 0; x < W; ++x) {
            if (!(x >= p->area[0] && x < p->area[0] + p->area[2] && y >= p->area[1] && y < p->area[1] + p->area[3])) {
                memset(job->out_color + ((size_t)y * W + x) * 4, 0, 4);
                continue;
            }
            float acc[4];
            float smp[4][4];
            for (int s = 0; s < S; ++s) {
                size_t li = ((size_t)(y - job->y0) * W + x) * S + s;
                if (job->literal) {
                    memcpy(smp[s], &sample_rgba[li * 4], 16);
                    continue;
                }
                const winner& wn = win[li];
                if (!wn.valid) { memcpy(smp[s], p->clear_color, 16); continue; }
                /* one shading per (pixel, primitive): reuse an earlier sample's result */
                int reuse = -1;
                for (int s2 = 0; s2 < s; ++s2) {
                    const winner& w2 = win[li - s + s2];
                    if (w2.valid && w2.draw == wn.draw && w2.tri == wn.tri && w2.sub == wn.sub) { reuse = s2; break; }
                }
                if (reuse >= 0) { memcpy(smp[s], smp[reuse], 16); continue; }
                const draw_state* ds = &(*job->draws)[wn.draw];
                prim_tris pt;
                expand_primitive(ds, &vs, wn.tri, &pt);
                const raster_tri* t = nullptr;
                for (int ti = 0; ti < pt.count; ++ti) if (pt.sub[ti] == wn.sub) t = &pt.tri[ti];
                if (t) shade_fragment(ds, job->textures, job->n_textures, wn.tri, t, x, y, smp[s]);
                else memcpy(smp[s], p->clear_color, 16);
            }
            if (S == 1) memcpy(acc, smp[0], 16);
            else for (int k = 0; k < 4; ++k) acc[k] = ((smp[0][k] + smp[1][k]) + (smp[2][k] + smp[3][k])) * 0.25f;
            store_color(p, acc, job->out_color + ((size_t)y * W + x) * 4);
        }
    }
}

/* Pre-pass shared by all bands: which pixel rows each primitive can touch. */
struct range_job {
    const orc_pass* pass;
    const std::vector<draw_state>* draws;
    uint32_t* row_range;
    uint32_t first, last; /* running primitive numbers [first, last) */
};

void compute_row_ranges(const range_job* job)
{
    view_state vs;
    make_view_state(job->pass, 0, (int)job->pass->height - 1, &vs);
    const std::vector<draw_state>& draws = *job->draws;
    size_t di = 0;
    for (uint32_t p = job->first; p < job->last; ++p) {
        while (di + 1 < draws.size() && p >= draws[di].prim_base + draws[di].tri_count) ++di;
        prim_tris pt;
        expand_primitive(&draws[di], &vs, p - draws[di].prim_base, &pt);
        int lo = 1 << 30, hi = -1;
        for (int ti = 0; ti < pt.count; ++ti) {
            const raster_tri* t = &pt.tri[ti];
            int minx = std::min(t->v[0].x, std::min(t->v[1].x, t->v[2].x)) >> SUBPIXEL_BITS;
            int maxx = std::max(t->v[0].x, std::max(t->v[1].x, t->v[2].x)) >> SUBPIXEL_BITS;
            int miny = std::min(t->v[0].y, std::min(t->v[1].y, t->v[2].y)) >> SUBPIXEL_BITS;
            int maxy = std::max(t->v[0].y, std::max(t->v[1].y, t->v[2].y)) >> SUBPIXEL_BITS;
            if (std::max(minx, vs.cx0) > std::min(maxx, vs.cx1)) continue;
            miny = std::max(miny, vs.cy0); maxy = std::min(maxy, vs.cy1);
            if (miny > maxy) continue;
            lo = std::min(lo, miny); hi = std::max(hi, maxy);
        }
        job->row_range[p] = hi < 0 ? 0xFFFFFFFFu : (((uint32_t)lo << 16) | (uint32_t)hi);
    }
}

} // namespace

/* ---------------------------------------------------------------- C interface */

extern "C" {

int orc_render(const orc_pass* pass, const orc_draw* draws, int n_draws,
               const orc_texture* textures, int n_textures,
               uint8_t* out_color, uint32_t* out_depth, uint32_t* out_prim,
               int literal, int threads)
{
    if (!pass || !out_color || !out_depth) return -1;
    if (pass->samples != 1 && pass->samples != 4) return -2;
    std::vector<draw_state> ds((size_t)n_draws);
    uint32_t prim_base = 0;
    for (int i = 0; i < n_draws; ++i) {
        if (draws[i].depth_test && draws[i].depth_op != 1 && draws[i].depth_op != 3) return -3;
        make_draw_state(&draws[i], prim_base, &ds[(size_t)i]);
        prim_base += ds[(size_t)i].tri_count;
    }
    int H = (int)pass->height;
    if (H > 65535) return -4;
    if (threads < 1) threads = 1;
    if (threads > H) threads = H > 0 ? H : 1;
    std::vector<uint32_t> row_range((size_t)prim_base + 1);
    {
        std::vector<range_job> rj((size_t)threads);
        std::vector<std::thread> rpool;
        for (int t = 0; t < threads; ++t) {
            rj[(size_t)t].pass = pass; rj[(size_t)t].draws = &ds; rj[(size_t)t].row_range = row_range.data();
            rj[(size_t)t].first = (uint32_t)((uint64_t)prim_base * t / threads);
            rj[(size_t)t].last = (uint32_t)((uint64_t)prim_base * (t + 1) / threads);
        }
        if (threads == 1) compute_row_ranges(&rj[0]);
        else {
            for (int t = 0; t < threads; ++t) if (rj[(size_t)t].first < rj[(size_t)t].last) rpool.emplace_back(compute_row_ranges, &rj[(size_t)t]);
            for (auto& th : rpool) th.join();
        }
    }
    std::vector<band_job> jobs((size_t)threads);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) {
        band_job& j = jobs[(size_t)t];
        j.pass = pass; j.draws = &ds; j.textures = textures; j.n_textures = n_textures;
        j.y0 = (int)((int64_t)H * t / threads);
        j.y1 = (int)((int64_t)H * (t + 1) / threads) - 1;
        j.out_color = out_color; j.out_depth = out_depth; j.out_prim = out_prim; j.literal = literal; j.row_range = row_range.data();
    }
    if (threads == 1) {
        render_band(&jobs[0]);
    } else {
        for (int t = 0; t < threads; ++t) pool.emplace_back(render_band, &jobs[(size_t)t]);
        for (auto& th : pool) th.join();
    }
    return 0;
}

uint32_t orc_mip_level_count(uint32_t w, uint32_t h, int gen_mips)
{
    if (!gen_mips) return 1;
    /* tools.cpp:514: uint32_t mip_levels = log2(max(w,h)) -- truncation, one short of a full chain */
    uint32_t m = w > h ? w : h;
    uint32_t l = (uint32_t)log2((double)m);
    return l ? l : 1;
}

/* vkCmdBlitImage LINEAR of the whole source level into the whole destination
 * level (tools.cpp:670-707): destination texel centre i+0.5 maps to source
 * coordinate (i+0.5)*sw/dw, filtered bilinearly with clamp-to-edge. */
int orc_gen_mip(const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh, int srgb)
{
    float scale_x = (float)sw / (float)dw, scale_y = (float)sh / (float)dh;
    for (uint32_t y = 0; y < dh; ++y) {
        for (uint32_t x = 0; x < dw; ++x) {
            float fu = fma32((float)x + 0.5f, scale_x, -0.5f);
            float fv = fma32((float)y + 0.5f, scale_y, -0.5f);
            float iu = floorf(fu), iv = floorf(fv);
            float ax = fu - iu, ay = fv - iv;
            int x0 = wrap_coord((int)iu, (int)sw, 0), x1 = wrap_coord((int)iu + 1, (int)sw, 0);
            int y0 = wrap_coord((int)iv, (int)sh, 0), y1 = wrap_coord((int)iv + 1, (int)sh, 0);
            const uint8_t* p00 = src + ((size_t)y0 * sw + x0) * 4;
            const uint8_t* p10 = src + ((size_t)y0 * sw + x1) * 4;
            const uint8_t* p01 = src + ((size_t)y1 * sw + x0) * 4;
            const uint8_t* p11 = src + ((size_t)y1 * sw + x1) * 4;
            uint8_t* o = dst + ((size_t)y * dw + x) * 4;
            for (int k = 0; k < 4; ++k) {
                bool lin = !(srgb && k < 3);
                float c00 = lin ? (float)p00[k] / 255.0f : g_srgb.decode[p00[k]];
                float c10 = lin ? (float)p10[k] / 255.0f : g_srgb.decode[p10[k]];
                float c01 = lin ? (float)p01[k] / 255.0f : g_srgb.decode[p01[k]];
                float c11 = lin ? (float)p11[k] / 255.0f : g_srgb.decode[p11[k]];
                float top = fma32(ax, c10 - c00, c00);
                float bot = fma32(ax, c11 - c01, c01);
                float c = fma32(ay, bot - top, top);
                o[k] = (uint8_t)(lin ? unorm8(c) : srgb_encode(c));
            }
        }
    }
    return 0;
}

uint32_t orc_srgb_encode(float linear) { return srgb_encode(linear); }
float orc_srgb_decode(uint32_t code) { return g_srgb.decode[code & 255u]; }
float orc_log2(float x) { return lod_log2(x); }
int32_t orc_snap(float v) { return snap(v); }
uint32_t orc_depth24(float z) { return depth24(z); }
void orc_sample_texture(const orc_texture* t, float u, float v, float dudx, float dvdx, float dudy, float dvdy, float out_rgba[4])
{
    sample_texture(t, u, v, dudx, dvdx, dudy, dvdy, out_rgba);
}

/* ---- glm-equivalent host matrices: plain products and sums in glm's source order ---- */

void orc_mat4_mul(const float a[16], const float b[16], float out[16])
{
    float r[16];
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row)
            r[c * 4 + row] = a[0 + row] * b[c * 4 + 0] + a[4 + row] * b[c * 4 + 1] + a[8 + row] * b[c * 4 + 2] + a[12 + row] * b[c * 4 + 3];
    memcpy(out, r, sizeof(r));
}

static void perspective_common(float fov, float w, float h, float out[16], float* hh)
{
    memset(out, 0, 64);
    float hv = cosf(0.5f * fov) / sinf(0.5f * fov);
    float wv = hv * h / w;
    out[0] = wv;
    out[5] = hv;
    out[11] = -1.0f;
    *hh = hv;
}

void orc_perspective_fov_rh_no(float fov, float w, float h, float n, float f, float out[16])
{
    float hh;
    perspective_common(fov, w, h, out, &hh);
    out[10] = -(f + n) / (f - n);
    out[14] = -(2.0f * f * n) / (f - n);
}

void orc_perspective_fov_rh_zo(float fov, float w, float h, float n, float f, float out[16])
{
    float hh;
    perspective_common(fov, w, h, out, &hh);
    out[10] = f / (n - f);
    out[14] = -(f * n) / (f - n);
}

static inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(const float* a, const float* b, float* o)
{
    float r[3] = {a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]};
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}
static inline void normalize3(const float* a, float* o)
{
    float inv = 1.0f / sqrtf(dot3(a, a));
    o[0] = a[0] * inv; o[1] = a[1] * inv; o[2] = a[2] * inv;
}

void orc_look_at_rh(const float eye[3], const float center[3], const float up[3], float out[16])
{
    float d[3] = {center[0] - eye[0], center[1] - eye[1], center[2] - eye[2]};
    float f[3], s[3], u[3], t[3];
    normalize3(d, f);
    cross3(f, up, t);
    normalize3(t, s);
    cross3(s, f, u);
    memset(out, 0, 64);
    out[0] = s[0]; out[4] = s[1]; out[8] = s[2];
    out[1] = u[0]; out[5] = u[1]; out[9] = u[2];
    out[2] = -f[0]; out[6] = -f[1]; out[10] = -f[2];
    out[12] = -dot3(s, eye); out[13] = -dot3(u, eye); out[14] = dot3(f, eye);
    out[15] = 1.0f;
}

void orc_translate(const float m[16], const float v[3], float out[16])
{
    float r[16];
    memcpy(r, m, 64);
    for (int row = 0; row < 4; ++row) r[12 + row] = m[0 + row] * v[0] + m[4 + row] * v[1] + m[8 + row] * v[2] + m[12 + row];
    memcpy(out, r, 64);
}

void orc_scale(const float m[16], const float v[3], float out[16])
{
    float r[16];
    for (int row = 0; row < 4; ++row) {
        r[0 + row] = m[0 + row] * v[0];
        r[4 + row] = m[4 + row] * v[1];
        r[8 + row] = m[8 + row] * v[2];
        r[12 + row] = m[12 + row];
    }
    memcpy(out, r, 64);
}

void orc_rotate(const float m[16], float angle, const float axis_in[3], float out[16])
{
    float c = cosf(angle), s = sinf(angle);
    float axis[3];
    normalize3(axis_in, axis);
    float temp[3] = {(1.0f - c) * axis[0], (1.0f - c) * axis[1], (1.0f - c) * axis[2]};
    float R[3][3];
    R[0][0] = c + temp[0] * axis[0];
    R[0][1] = temp[0] * axis[1] + s * axis[2];
    R[0][2] = temp[0] * axis[2] - s * axis[1];
    R[1][0] = temp[1] * axis[0] - s * axis[2];
    R[1][1] = c + temp[1] * axis[1];
    R[1][2] = temp[1] * axis[2] + s * axis[0];
    R[2][0] = temp[2] * axis[0] + s * axis[1];
    R[2][1] = temp[2] * axis[1] - s * axis[0];
    R[2][2] = c + temp[2] * axis[2];
    float r[16];
    for (int i = 0; i < 3; ++i)
        for (int row = 0; row < 4; ++row)
            r[i * 4 + row] = m[0 + row] * R[i][0] + m[4 + row] * R[i][1] + m[8 + row] * R[i][2];
    for (int row = 0; row < 4; ++row) r[12 + row] = m[12 + row];
    memcpy(out, r, 64);
}

void orc_yaw_pitch_roll(float yaw, float pitch, float roll, float out[16])
{
    float ch = cosf(yaw), sh = sinf(yaw), cp = cosf(pitch), sp = sinf(pitch), cb = cosf(roll), sb = sinf(roll);
    memset(out, 0, 64);
    out[0] = ch * cb + sh * sp * sb;
    out[1] = sb * cp;
    out[2] = -sh * cb + ch * sp * sb;
    out[4] = -ch * sb + sh * sp * cb;
    out[5] = cb * cp;
    out[6] = sb * sh + ch * sp * cb;
    out[8] = sh * cp;
    out[9] = -sp;
    out[10] = ch * cp;
    out[15] = 1.0f;
}

} // extern "C"
