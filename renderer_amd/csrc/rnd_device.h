/*
 * rnd_device.h -- device-side arithmetic of the draw path (gfx950).
 *
 * Every floating-point step is an explicit IEEE-754 binary32 operation
 * (+ - * / sqrt fma and int<->float conversions); the file is compiled with
 * -ffp-contract=off so nothing is fused or reassociated behind the source.
 * Two implementations that follow the same sequence agree bit for bit, which
 * is what makes coverage and depth comparable word for word with a CPU run.
 *
 * Pipeline rules implemented here (Vulkan 1.0 fixed function driven by the
 * state the reference sets in test_ktx_app.cpp:412-625):
 *   clip volume 0<=z<=w plus a guard band of 2w in x and y; viewport transform;
 *   24.8 fixed-point window coordinates; top-left fill rule; D24 depth;
 *   perspective-correct varyings; sRGB store.
 */
#ifndef RND_DEVICE_H
#define RND_DEVICE_H

#include "rnd_internal.h"

namespace rnd {

__device__ __forceinline__ float fma32(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

/* Pointers read from a record in memory (a draw's vertex, index and uniform buffers) are device
 * memory; saying so lets the load be a global one instead of a flat one. */
typedef const __attribute__((address_space(1))) float* GlobalF32;
typedef const __attribute__((address_space(1))) uint32_t* GlobalU32;
__device__ __forceinline__ float load_f32(const float* p) { return *(GlobalF32)p; }
__device__ __forceinline__ uint32_t load_u32(const uint32_t* p) { return *(GlobalU32)p; }
__device__ __forceinline__ bool finite32(float v) { return fabsf(v) <= 3.4028234663852886e38f; }

/* 1.0f / x, rounded as the division rounds it, in half the instructions.  The hardware's estimate
 * refined by one Newton step in fused multiply-adds equals the IEEE quotient for every x whose
 * magnitude is a normal number below 2^126: tools/micro/rcp_exact.hip compares the two over all 2^32
 * bit patterns.  Zero, subnormals, the top two binades, infinities and NaN take the division. */
__device__ __forceinline__ float recip32(float x)
{
    const float ax = fabsf(x);
    if (ax >= 1.17549435082228750797e-38f && ax < 8.50705917302346158658e37f) {
        const float r = __builtin_amdgcn_rcpf(x);
        return fma32(fma32(-x, r, 1.0f), r, r);
    }
    return 1.0f / x;
}

/* the two halves of recip32, for callers that take several reciprocals behind one test */
__device__ __forceinline__ bool recip32_range(float x)
{
    const float ax = fabsf(x);
    return ax >= 1.17549435082228750797e-38f && ax < 8.50705917302346158658e37f;
}
__device__ __forceinline__ float recip32_in_range(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    return fma32(fma32(-x, r, 1.0f), r, r);
}

/* the same for an x known to be a non-zero integer below 2^63 in magnitude (a triangle's doubled area in
 * 24.8 units): always in the range where the refined estimate is the quotient, so no test */
__device__ __forceinline__ float recip32_of_area(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    return fma32(fma32(-x, r, 1.0f), r, r);
}

/* window coordinate -> 24.8: floor(fma(v, 256, 0.5)) */
__device__ __forceinline__ int32_t snap(float v)
{
    float t = floorf(fma32(v, 256.0f, 0.5f));
    t = fminf(fmaxf(t, -2147483520.0f), 2147483520.0f);
    return (int32_t)t;
}

/* z -> D24: the float product clamp(z) * (2^24 - 1) rounded to the nearest integer, ties to even (v_rndne_f32) */
__device__ __forceinline__ uint32_t depth24(float z)
{
    z = fminf(fmaxf(z, 0.0f), 1.0f);
    return (uint32_t)__builtin_rintf(z * 16777215.0f); /* z <= 1, so the product is at most 2^24 - 1, which is a float */
}

/* linear -> sRGB code: number of thresholds <= c; the bucket table gives the number of thresholds <= the
 * bucket's lower edge, and a bucket holds at most one threshold (checked where the table is built).
 * Without a branch: c is first brought into the table's range [2^-13, 1) -- below thr[1] (which lies above 2^-13)
 * every value has code 0 like 2^-13 itself, from thr[255] (which lies below the largest float under 1) on every value
 * has code 255 like that float; a NaN comes out of the median as the lower bound, code 0.  The three channels of a
 * pixel then run their two table reads side by side instead of one branch after the other. */
template <typename Tables>
__device__ __forceinline__ uint32_t srgb_encode(const Tables& tb, float c)
{
    const float cc = __builtin_amdgcn_fmed3f(c, 1.220703125e-4f, 0.99999994f);
    const uint32_t key = (__float_as_uint(cc) >> SRGB_LUT_SHIFT) - SRGB_LUT_FIRST;
    uint32_t k = tb.lut[key];
    k += (cc >= tb.thr[k + 1]) ? 1u : 0u;
    return k;
}

__device__ __forceinline__ uint32_t unorm8(float c)
{
    if (!(c > 0.0f)) return 0;
    c = fminf(c, 1.0f);
    return (uint32_t)fma32(c, 255.0f, 0.5f);
}

/* log2 for LOD selection: exponent + degree-5 polynomial on the mantissa */
__device__ __forceinline__ float lod_log2(float x)
{
    if (!(x >= 1.17549435e-38f)) return -126.0f;
    if (!finite32(x)) return 128.0f;
    uint32_t bits = __float_as_uint(x);
    int e = (int)((bits >> 23) & 0xFF) - 127;
    float m = __uint_as_float((bits & 0x7FFFFFu) | 0x3F800000u);
    float t = m - 1.0f;
    float p = -0.03459521010518074f;
    p = fma32(p, t, 0.14643360674381256f);
    p = fma32(p, t, -0.30338966846466064f);
    p = fma32(p, t, 0.469301700592041f);
    p = fma32(p, t, -0.7204423546791077f);
    p = fma32(p, t, 1.442683219909668f);
    return fma32(p, t, (float)e);
}

/* ---------------------------------------------------------------- matrices (column-major) */

__device__ __forceinline__ void mat4_mul_pinned(const float* a, const float* b, float* out)
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int row = 0; row < 4; ++row) {
            float acc = a[0 * 4 + row] * b[c * 4 + 0];
            acc = fma32(a[1 * 4 + row], b[c * 4 + 1], acc);
            acc = fma32(a[2 * 4 + row], b[c * 4 + 2], acc);
            acc = fma32(a[3 * 4 + row], b[c * 4 + 3], acc);
            out[c * 4 + row] = acc;
        }
    }
}

__device__ __forceinline__ void mat4_point(const float* m, float x, float y, float z, float* out)
{
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        float acc = fma32(m[2 * 4 + row], z, m[3 * 4 + row]);
        acc = fma32(m[1 * 4 + row], y, acc);
        acc = fma32(m[0 * 4 + row], x, acc);
        out[row] = acc;
    }
}

/* inverse(transpose(mat3(M))) as cofactors over the determinant; out column-major 3x3 */
__device__ __forceinline__ void normal_matrix(const float* m4, float* out)
{
    float a00 = m4[0], a01 = m4[1], a02 = m4[2];
    float a10 = m4[4], a11 = m4[5], a12 = m4[6];
    float a20 = m4[8], a21 = m4[9], a22 = m4[10];
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

__device__ __forceinline__ void mat3_vec(const float* m, const float* v, float* out)
{
#pragma unroll
    for (int row = 0; row < 3; ++row) {
        float acc = m[0 * 3 + row] * v[0];
        acc = fma32(m[1 * 3 + row], v[1], acc);
        acc = fma32(m[2 * 3 + row], v[2], acc);
        out[row] = acc;
    }
}

/* ---------------------------------------------------------------- vertex stage */

__device__ __forceinline__ uint32_t outcode(float x, float y, float z, float w)
{
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

struct Projected {
    int32_t x, y;
    float z, iw;
};

__device__ __forceinline__ Projected project(const FrameParams& fp, float x, float y, float z, float w)
{
    Projected p;
    float iw = recip32(w);
    float xd = x * iw, yd = y * iw, zd = z * iw;
    p.x = snap(fma32(xd, fp.half_w, fp.ox));
    p.y = snap(fma32(yd, fp.half_h, fp.oy));
    p.z = fma32(zd, fp.pz, fp.zmin);
    p.iw = iw;
    return p;
}

/* vertex shader position: M * vec4(pos,1), then the reference's clip-space y = 1 - y
 * (test_ktx.vert:28-29, push_constants.vert:28-29, dummy.vert.glsl:22-23) */
__device__ __forceinline__ void vs_position(const DrawStatic& d, const float* clip_m, uint32_t v, float* clip)
{
    const float* p = reinterpret_cast<const float*>(d.vb + (size_t)v * d.stride + d.off_pos);
    mat4_point(clip_m, load_f32(p), load_f32(p + 1), load_f32(p + 2), clip);
    clip[1] = 1.0f - clip[1];
}

/* ---------------------------------------------------------------- primitive */

/* A triangle oriented so that its edge functions are positive inside */
struct RTri {
    int32_t x[3], y[3]; /* 24.8 */
    float z[3], iw[3];
    int64_t area2;
    float inv_area;
    uint8_t corner[3];  /* parent corner held by each slot (unclipped primitives) */
};

__device__ __forceinline__ int64_t edge64(int32_t ax, int32_t ay, int32_t bx, int32_t by, int64_t px, int64_t py)
{
    return (int64_t)(bx - ax) * (py - ay) - (int64_t)(by - ay) * (px - ax);
}

/* the edge a->b owns points lying exactly on it when it is a left or a top edge */
__device__ __forceinline__ bool owns_edge(int32_t ax, int32_t ay, int32_t bx, int32_t by)
{
    int dy = by - ay, dx = bx - ax;
    return dy < 0 || (dy == 0 && dx > 0);
}

/* orient + cull; false: no fragments */
__device__ __forceinline__ bool setup_tri(RTri& t, uint32_t cull_mode, uint32_t front_face)
{
    int64_t s = edge64(t.x[0], t.y[0], t.x[1], t.y[1], t.x[2], t.y[2]);
    if (s == 0) return false;
    bool front = (front_face == 0) ? (s < 0) : (s > 0);
    if ((cull_mode & 1u) && front) return false;
    if ((cull_mode & 2u) && !front) return false;
    if (s < 0) {
        int32_t ti; float tf; uint8_t tc;
        ti = t.x[1]; t.x[1] = t.x[2]; t.x[2] = ti;
        ti = t.y[1]; t.y[1] = t.y[2]; t.y[2] = ti;
        tf = t.z[1]; t.z[1] = t.z[2]; t.z[2] = tf;
        tf = t.iw[1]; t.iw[1] = t.iw[2]; t.iw[2] = tf;
        tc = t.corner[1]; t.corner[1] = t.corner[2]; t.corner[2] = tc;
        s = -s;
    }
    t.area2 = s;
    t.inv_area = recip32((float)s);
    return true;
}

/* clip-space vertex with its weights over the parent's corners */
struct ClipV {
    float x, y, z, w;
    float b0, b1, b2;
};

__device__ __forceinline__ float plane_distance(int k, const ClipV& v)
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

/* Polygon storage of the clipper: two buffers of up to 12 vertices x 7 floats per clipping lane,
 * kept in LDS ([buffer][vertex][component][lane], so the lanes' accesses are conflict free)
 * instead of per-thread arrays, which would become scratch memory.  A wave clips CLIP_LANES
 * primitives at a time. */
constexpr int CLIP_LANES = 4;
constexpr int CLIP_MAX_VERTS = 12;
constexpr int CLIP_LDS_FLOATS = 2 * CLIP_MAX_VERTS * 7 * CLIP_LANES; /* per wave */

struct ClipStore {
    float* base; /* the wave's LDS block + this lane's index */
    __device__ __forceinline__ float* at(int buf, int i, int comp) const
    {
        return base + ((buf * CLIP_MAX_VERTS + i) * 7 + comp) * CLIP_LANES;
    }
    __device__ __forceinline__ ClipV get(int buf, int i) const
    {
        ClipV v;
        v.x = *at(buf, i, 0); v.y = *at(buf, i, 1); v.z = *at(buf, i, 2); v.w = *at(buf, i, 3);
        v.b0 = *at(buf, i, 4); v.b1 = *at(buf, i, 5); v.b2 = *at(buf, i, 6);
        return v;
    }
    __device__ __forceinline__ void put(int buf, int i, const ClipV& v) const
    {
        *at(buf, i, 0) = v.x; *at(buf, i, 1) = v.y; *at(buf, i, 2) = v.z; *at(buf, i, 3) = v.w;
        *at(buf, i, 4) = v.b0; *at(buf, i, 5) = v.b1; *at(buf, i, 6) = v.b2;
    }
};

/* Sutherland-Hodgman against near, far and the guard band; crossings are computed from the
 * inside end so a shared edge yields the same vertex from both sides.  The polygon starts in
 * buffer 0; returns the vertex count and the buffer holding the result (0 vertices: nothing left). */
__device__ __forceinline__ int clip_polygon(const ClipStore& st, int n, int& buf)
{
    buf = 0;
    for (int k = 0; k < 6; ++k) {
        bool any_out = false;
        for (int i = 0; i < n; ++i) {
            if (!(plane_distance(k, st.get(buf, i)) >= 0.0f)) any_out = true;
        }
        if (!any_out) continue;
        const int dst = buf ^ 1;
        int m = 0;
        ClipV cur = st.get(buf, 0);
        float dc = plane_distance(k, cur);
        for (int i = 0; i < n; ++i) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            const ClipV nxt = st.get(buf, j);
            const float dn = plane_distance(k, nxt);
            const bool in_i = dc >= 0.0f, in_j = dn >= 0.0f;
            if (in_i) st.put(dst, m++, cur);
            if (in_i != in_j) {
                const ClipV a = in_i ? cur : nxt; /* inside end */
                const ClipV b = in_i ? nxt : cur; /* outside end */
                const float da = in_i ? dc : dn, db = in_i ? dn : dc;
                const float t = da / (da - db);
                ClipV r;
                r.x = fma32(t, b.x - a.x, a.x);
                r.y = fma32(t, b.y - a.y, a.y);
                r.z = fma32(t, b.z - a.z, a.z);
                r.w = fma32(t, b.w - a.w, a.w);
                r.b0 = fma32(t, b.b0 - a.b0, a.b0);
                r.b1 = fma32(t, b.b1 - a.b1, a.b1);
                r.b2 = fma32(t, b.b2 - a.b2, a.b2);
                st.put(dst, m++, r);
            }
            cur = nxt; dc = dn;
        }
        n = m;
        buf = dst;
        if (n < 3) return 0;
    }
    return n;
}

/* Clips one primitive.  Returns the number of corners of the clipped polygon (0: nothing survives,
 * or a corner cannot be projected) and the buffer that holds it. */
__device__ __forceinline__ int build_clip_fan(const DrawStatic& d, const float* clip_m, uint32_t i0, uint32_t i1, uint32_t i2,
                                              const ClipStore& st, int& buf)
{
    const uint32_t idx[3] = {i0, i1, i2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float c[4];
        vs_position(d, clip_m, idx[k], c);
        ClipV v;
        v.x = c[0]; v.y = c[1]; v.z = c[2]; v.w = c[3];
        v.b0 = k == 0 ? 1.0f : 0.0f; v.b1 = k == 1 ? 1.0f : 0.0f; v.b2 = k == 2 ? 1.0f : 0.0f;
        st.put(0, k, v);
    }
    const int n = clip_polygon(st, 3, buf);
    if (n < 3) return 0;
    for (int i = 0; i < n; ++i) {
        const ClipV v = st.get(buf, i);
        if (!(v.w > 0.0f) || !finite32(v.x) || !finite32(v.y) || !finite32(v.z) || !finite32(v.w)) return 0;
    }
    return n;
}

/* fan triangle k (1 <= k <= n-2): corners (0, k, k+1), projected, oriented, culled.
 * bary[slot][j] = weight of the parent's corner j at the triangle's corner in that slot */
__device__ __forceinline__ bool fan_triangle(const FrameParams& fp, const ClipStore& st, int buf, int k,
                                             uint32_t cull_mode, uint32_t front_face, RTri& t, float bary[3][3])
{
    const int c[3] = {0, k, k + 1};
    ClipV cv[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        cv[q] = st.get(buf, c[q]);
        const Projected p = project(fp, cv[q].x, cv[q].y, cv[q].z, cv[q].w);
        t.x[q] = p.x; t.y[q] = p.y; t.z[q] = p.z; t.iw[q] = p.iw;
        t.corner[q] = (uint8_t)q;
    }
    if (!setup_tri(t, cull_mode, front_face)) return false;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const ClipV& v = t.corner[q] == 0 ? cv[0] : (t.corner[q] == 1 ? cv[1] : cv[2]);
        bary[q][0] = v.b0; bary[q][1] = v.b1; bary[q][2] = v.b2;
    }
    return true;
}

} // namespace rnd
#endif
