/*
 * rnd_kernels.hip -- gfx950 kernels of the draw path.
 *
 * One submit of a recorded pass runs:
 *   k_constants  per-draw matrices from the uniform buffer (read at execution time,
 *                base_obj_viewer_app.cpp:134-138) and the push constants
 *   k_vertex     vertex fetch + position shader + clip codes + viewport + 24.8 snap
 *   k_setup      per primitive: reject / cull / tile box, per-tile counts
 *   k_scan       exclusive scan of the tile counts
 *   k_fill       primitive slots grouped by tile
 *   k_tile       one workgroup per 32x32 tile: edge-function rasteriser into an LDS
 *                depth/owner word per sample, then one shading per covered pixel and
 *                coalesced colour/depth stores
 *
 * Depth resolution is order independent: each fragment offers the 64-bit word
 * (D24 << 40 | inverted submission number << 8 | fan index) and the tile keeps the
 * minimum.  Under LESS_OR_EQUAL with depth writes that is exactly what in-order
 * processing leaves behind: the smallest depth, and among equal depths the last
 * primitive submitted.
 */
#include "rnd_device.h"

namespace rnd {

/* ------------------------------------------------------------------ helpers */

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t wave_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ uint32_t pack_bbox(int tx0, int ty0, int tx1, int ty1)
{
    return (uint32_t)tx0 | ((uint32_t)ty0 << 8) | ((uint32_t)tx1 << 16) | ((uint32_t)ty1 << 24);
}

/* primitive slot -> draw, triangle */
__device__ __forceinline__ void slot_to_prim(const FrameParams& fp, uint32_t slot, uint32_t& draw, uint32_t& tri)
{
    Chunk c = fp.tchunks[slot >> 6];
    draw = c.draw;
    tri = c.first + (slot & 63u);
}

struct PrimFetch {
    uint32_t idx[3];
    SVtx v[3];
    uint32_t oc_any, oc_all;
};

/* false: an index points outside the vertex buffer, the primitive is dropped */
__device__ __forceinline__ bool fetch_prim(const FrameParams& fp, const DrawStatic& d, uint32_t tri, PrimFetch& pf)
{
    const uint32_t* ip = d.ib + (size_t)tri * 3;
    pf.idx[0] = ip[0]; pf.idx[1] = ip[1]; pf.idx[2] = ip[2];
    if (pf.idx[0] >= d.vert_count || pf.idx[1] >= d.vert_count || pf.idx[2] >= d.vert_count) return false;
    const SVtx* sv = fp.svtx + d.svtx_base;
    pf.v[0] = sv[pf.idx[0]]; pf.v[1] = sv[pf.idx[1]]; pf.v[2] = sv[pf.idx[2]];
    uint32_t o0 = (uint32_t)pf.v[0].xf & 255u, o1 = (uint32_t)pf.v[1].xf & 255u, o2 = (uint32_t)pf.v[2].xf & 255u;
    pf.oc_any = o0 | o1 | o2;
    pf.oc_all = o0 & o1 & o2;
    return true;
}

/* 0: rejected, 1: rasterise v[] directly, 2: needs the clipper */
__device__ __forceinline__ int classify_prim(const PrimFetch& pf)
{
    if (pf.oc_any & OC_INVALID) return 0;
    if (pf.oc_all & 63u) return 0;
    if (pf.oc_any & (OC_ZNEG | OC_ZPOS | OC_GUARD)) return 2;
    return 1;
}

__device__ __forceinline__ void tri_from_fetch(const PrimFetch& pf, RTri& t)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        t.x[k] = pf.v[k].xf >> 8; /* arithmetic shift drops the outcode byte */
        t.y[k] = pf.v[k].y;
        t.z[k] = pf.v[k].z;
        t.iw[k] = pf.v[k].iw;
        t.corner[k] = (uint8_t)k;
    }
}

/* inclusive pixel box of the samples a triangle can cover, clipped to the clamp rectangle */
__device__ __forceinline__ bool pixel_box(const FrameParams& fp, const RTri& t, int& x0, int& y0, int& x1, int& y1)
{
    int minx = min(t.x[0], min(t.x[1], t.x[2])), maxx = max(t.x[0], max(t.x[1], t.x[2]));
    int miny = min(t.y[0], min(t.y[1], t.y[2])), maxy = max(t.y[0], max(t.y[1], t.y[2]));
    if (fp.samples == 1) {
        /* pixel centres inside [min, max] */
        x0 = (minx + 127) >> 8; x1 = (maxx - 128) >> 8;
        y0 = (miny + 127) >> 8; y1 = (maxy - 128) >> 8;
    } else {
        x0 = minx >> 8; x1 = maxx >> 8;
        y0 = miny >> 8; y1 = maxy >> 8;
    }
    x0 = max(x0, fp.clamp_x0); y0 = max(y0, fp.clamp_y0);
    x1 = min(x1, fp.clamp_x1); y1 = min(y1, fp.clamp_y1);
    return x0 <= x1 && y0 <= y1;
}

/* ------------------------------------------------------------------ record-time: largest index of a range */

__global__ __launch_bounds__(256) void k_index_range(const IndexRangeJob* jobs, uint32_t* out_max)
{
    const IndexRangeJob job = jobs[blockIdx.y];
    uint32_t m = 0;
    bool any = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < job.count; i += gridDim.x * blockDim.x) {
        m = max(m, job.ib[i]);
        any = true;
    }
    if (any) atomicMax(&out_max[blockIdx.y], m);
}

/* ------------------------------------------------------------------ per-draw matrices */

__global__ __launch_bounds__(64) void k_constants(FrameParams fp)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= fp.n_draws) return;
    const DrawStatic& d = fp.draws[i];
    DrawDerived out;
    float mvp[16], model[16], tr[16];
    for (int k = 0; k < 16; ++k) { mvp[k] = d.ubo[64 + k]; model[k] = d.ubo[48 + k]; tr[k] = d.push[4 + k]; }
    if (d.shader == 0) {
        for (int k = 0; k < 16; ++k) out.clip_m[k] = mvp[k];
        for (int k = 0; k < 12; ++k) out.nrm_m[k] = 0.0f;
    } else {
        float mt[16];
        mat4_mul_pinned(mvp, tr, out.clip_m);
        mat4_mul_pinned(model, tr, mt);
        normal_matrix(mt, out.nrm_m);
        out.nrm_m[9] = out.nrm_m[10] = out.nrm_m[11] = 0.0f;
    }
    fp.derived[i] = out;
}

/* ------------------------------------------------------------------ vertex kernel */

__global__ __launch_bounds__(256) void k_vertex(FrameParams fp)
{
    uint32_t chunk = wave_uniform(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (chunk >= fp.n_vchunks) return;
    const Chunk c = fp.vchunks[chunk];
    const DrawStatic& d = fp.draws[c.draw];
    uint32_t v = c.first + lane_id();
    if (v >= d.vert_count) return;
    const float* m = fp.derived[c.draw].clip_m;
    float clip[4];
    vs_position(d, m, v, clip);
    uint32_t oc = outcode(clip[0], clip[1], clip[2], clip[3]);
    SVtx out;
    if (oc & (OC_ZNEG | OC_ZPOS | OC_GUARD | OC_INVALID)) {
        out.xf = (int32_t)oc; out.y = 0; out.z = 0.0f; out.iw = 0.0f;
    } else if (!(clip[3] > 0.0f)) {
        out.xf = (int32_t)OC_INVALID; out.y = 0; out.z = 0.0f; out.iw = 0.0f;
    } else {
        Projected p = project(fp, clip[0], clip[1], clip[2], clip[3]);
        out.xf = (int32_t)(((uint32_t)p.x << 8) | oc);
        out.y = p.y; out.z = p.z; out.iw = p.iw;
    }
    fp.svtx[d.svtx_base + v] = out;
}

/* ------------------------------------------------------------------ setup: tile box per primitive + tile counts */

/* tile box of everything a primitive rasterises; BBOX_CULLED when nothing */
__device__ uint32_t prim_tile_box(const FrameParams& fp, const DrawStatic& d, const float* clip_m, uint32_t tri)
{
    PrimFetch pf;
    if (!fetch_prim(fp, d, tri, pf)) return BBOX_CULLED;
    int cls = classify_prim(pf);
    if (cls == 0) return BBOX_CULLED;
    int bx0 = 1 << 30, by0 = 1 << 30, bx1 = -1, by1 = -1;
    if (cls == 1) {
        RTri t;
        tri_from_fetch(pf, t);
        if (!setup_tri(t, d.cull_mode, d.front_face)) return BBOX_CULLED;
        if (!pixel_box(fp, t, bx0, by0, bx1, by1)) return BBOX_CULLED;
    } else {
        ClipFan fan;
        build_clip_fan(fp, d, clip_m, pf.idx[0], pf.idx[1], pf.idx[2], fan);
        for (int k = 1; k + 1 < fan.n; ++k) {
            RTri t; float bary[3][3];
            if (!fan_triangle(fan, k, d.cull_mode, d.front_face, t, bary)) continue;
            int x0, y0, x1, y1;
            if (!pixel_box(fp, t, x0, y0, x1, y1)) continue;
            bx0 = min(bx0, x0); by0 = min(by0, y0); bx1 = max(bx1, x1); by1 = max(by1, y1);
        }
        if (bx1 < 0) return BBOX_CULLED;
    }
    return pack_bbox(bx0 >> TILE_SHIFT, by0 >> TILE_SHIFT, bx1 >> TILE_SHIFT, by1 >> TILE_SHIFT);
}

/* Walks the tiles of every lane's box with wave-uniform control flow.  For tile
 * number k of its box a lane gets a slot from the tile's counter; lanes of the wave
 * that hit the same tile share one atomic. */
template <bool WRITE>
__device__ __forceinline__ void bin_wave(const FrameParams& fp, uint32_t box, uint32_t slot)
{
    const uint32_t lane = lane_id();
    int tx0 = box & 255u, ty0 = (box >> 8) & 255u, tx1 = (box >> 16) & 255u, ty1 = box >> 24;
    int w = tx1 - tx0 + 1, h = ty1 - ty0 + 1;
    int n = (box == BBOX_CULLED) ? 0 : w * h;
    bool large = n > MAX_TILES_PER_PRIM;
    if (large) n = 0;
    uint32_t* counter = WRITE ? fp.tile_cursor : fp.tile_count;
    for (int k = 0;; ++k) {
        bool act = k < n;
        if (__ballot(act) == 0ull) break;
        int tile = -1;
        if (act) {
            int ty = ty0 + k / w, tx = tx0 + k % w;
            int t = ty * (int)fp.tiles_x + tx;
            if ((uint32_t)t % fp.shard_count == fp.shard_index) tile = t;
        }
        uint64_t pending = __ballot(tile >= 0);
        while (pending) {
            int leader = __ffsll((unsigned long long)pending) - 1;
            int lt = __shfl(tile, leader);
            uint64_t same = __ballot(tile == lt);
            uint32_t base = 0;
            if ((int)lane == leader) base = atomicAdd(&counter[lt], (uint32_t)__popcll(same));
            if (WRITE) {
                base = (uint32_t)__shfl((int)base, leader);
                if (tile == lt) {
                    uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                    fp.refs[fp.tile_offset[lt] + base + rank] = slot;
                }
            }
            pending &= ~same;
        }
    }
    if (WRITE) {
        uint64_t lm = __ballot(large);
        if (lm) {
            int leader = __ffsll((unsigned long long)lm) - 1;
            uint32_t base = 0;
            if ((int)lane == leader) base = atomicAdd(fp.large_count, (uint32_t)__popcll(lm));
            base = (uint32_t)__shfl((int)base, leader);
            if (large) fp.large_list[base + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull))] = slot;
        }
    }
}

__global__ __launch_bounds__(256) void k_setup(FrameParams fp)
{
    uint32_t chunk = wave_uniform(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (chunk >= fp.n_tchunks) return;
    const Chunk c = fp.tchunks[chunk];
    const DrawStatic& d = fp.draws[c.draw];
    uint32_t tri = c.first + lane_id();
    uint32_t slot = chunk * 64u + lane_id();
    uint32_t box = BBOX_CULLED;
    if (tri < d.tri_count) box = prim_tile_box(fp, d, fp.derived[c.draw].clip_m, tri);
    fp.prim_bbox[slot] = box;
    bin_wave<false>(fp, box, slot);
}

__global__ __launch_bounds__(256) void k_fill(FrameParams fp)
{
    uint32_t chunk = wave_uniform(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (chunk >= fp.n_tchunks) return;
    uint32_t slot = chunk * 64u + lane_id();
    bin_wave<true>(fp, fp.prim_bbox[slot], slot);
}

/* exclusive scan of tile_count into tile_offset by one workgroup; clears the cursors */
__global__ __launch_bounds__(1024) void k_scan(FrameParams fp)
{
    __shared__ uint32_t part[1024];
    const uint32_t n = fp.tiles_x * fp.tiles_y;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t b = threadIdx.x * per, e = min(b + per, n);
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; ++i) sum += fp.tile_count[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t i = b; i < e; ++i) {
        fp.tile_offset[i] = run;
        run += fp.tile_count[i];
        fp.tile_cursor[i] = 0;
    }
}

/* ------------------------------------------------------------------ tile kernel */

__constant__ int c_sample_x4[4] = {96, 224, 32, 160}; /* (0.375,0.125) (0.875,0.375) (0.125,0.625) (0.625,0.875) */
__constant__ int c_sample_y4[4] = {32, 96, 160, 224};

struct TileCtx {
    int px0, py0;         /* first pixel of the tile */
    int rx0, ry0, rx1, ry1; /* inclusive pixel rectangle: tile ∩ clamp */
    unsigned long long* keys;
    uint32_t samples;
};

/* Rasterise one oriented triangle into the tile.  I is int32_t when every edge
 * product fits 32 bits (extent below 64 pixels), else int64_t; both give the same
 * integers. */
template <typename I>
__device__ __forceinline__ void raster_tri(const FrameParams& fp, const TileCtx& tc, const RTri& t, uint64_t payload)
{
    int x0, y0, x1, y1;
    if (!pixel_box(fp, t, x0, y0, x1, y1)) return;
    x0 = max(x0, tc.rx0); y0 = max(y0, tc.ry0); x1 = min(x1, tc.rx1); y1 = min(y1, tc.ry1);
    if (x0 > x1 || y0 > y1) return;
    /* edge k is opposite corner k: e0 = v1->v2, e1 = v2->v0, e2 = v0->v1 */
    const I ax0 = t.x[1], ay0 = t.y[1], bx0 = t.x[2], by0 = t.y[2];
    const I ax1 = t.x[2], ay1 = t.y[2], bx1 = t.x[0], by1 = t.y[0];
    const I ax2 = t.x[0], ay2 = t.y[0], bx2 = t.x[1], by2 = t.y[1];
    const I dx0 = bx0 - ax0, dy0 = by0 - ay0, dx1 = bx1 - ax1, dy1 = by1 - ay1, dx2 = bx2 - ax2, dy2 = by2 - ay2;
    /* a point on an edge is inside only if the edge owns it: fold that into a bias of -1 */
    const I bias0 = owns_edge(t.x[1], t.y[1], t.x[2], t.y[2]) ? 0 : 1;
    const I bias1 = owns_edge(t.x[2], t.y[2], t.x[0], t.y[0]) ? 0 : 1;
    const I bias2 = owns_edge(t.x[0], t.y[0], t.x[1], t.y[1]) ? 0 : 1;
    const float dz1 = t.z[1] - t.z[0], dz2 = t.z[2] - t.z[0];
    const uint32_t S = tc.samples;
    for (int y = y0; y <= y1; ++y) {
        for (int x = x0; x <= x1; ++x) {
            const uint32_t pix = (uint32_t)((y - tc.py0) * TILE_W + (x - tc.px0));
            for (uint32_t s = 0; s < S; ++s) {
                const I px = (I)(x * 256 + (S == 1 ? 128 : c_sample_x4[s]));
                const I py = (I)(y * 256 + (S == 1 ? 128 : c_sample_y4[s]));
                const I e0 = dx0 * (py - ay0) - dy0 * (px - ax0);
                const I e1 = dx1 * (py - ay1) - dy1 * (px - ax1);
                const I e2 = dx2 * (py - ay2) - dy2 * (px - ax2);
                if (((e0 - bias0) | (e1 - bias1) | (e2 - bias2)) < 0) continue;
                const float b1 = (float)e1 * t.inv_area, b2 = (float)e2 * t.inv_area;
                const float z = fma32(b2, dz2, fma32(b1, dz1, t.z[0]));
                const uint32_t d24 = fp.depth_test ? depth24(z) : 0u;
                const unsigned long long key = ((unsigned long long)d24 << 40) | payload;
                unsigned long long* slot = &tc.keys[pix * S + s];
                if (key < *slot) atomicMin(slot, key); /* early-out read, then the LDS atomic */
            }
        }
    }
}

__device__ __forceinline__ void raster_any(const FrameParams& fp, const TileCtx& tc, const RTri& t, uint64_t payload)
{
    int ext = max(max(t.x[0], max(t.x[1], t.x[2])) - min(t.x[0], min(t.x[1], t.x[2])),
                  max(t.y[0], max(t.y[1], t.y[2])) - min(t.y[0], min(t.y[1], t.y[2])));
    if (ext < (1 << 14)) raster_tri<int32_t>(fp, tc, t, payload);
    else raster_tri<int64_t>(fp, tc, t, payload);
}

/* low 40 bits of a fragment's word: submission order (inverted unless LESS) and fan index */
__device__ __forceinline__ uint64_t make_payload(const FrameParams& fp, uint32_t slot, uint32_t sub)
{
    uint32_t ord = fp.depth_less ? slot : ~slot;
    return ((uint64_t)ord << 8) | (uint64_t)sub;
}

__device__ void raster_slot(const FrameParams& fp, const TileCtx& tc, uint32_t slot)
{
    uint32_t draw, tri;
    slot_to_prim(fp, slot, draw, tri);
    const DrawStatic& d = fp.draws[draw];
    PrimFetch pf;
    if (!fetch_prim(fp, d, tri, pf)) return;
    int cls = classify_prim(pf);
    if (cls == 1) {
        RTri t;
        tri_from_fetch(pf, t);
        if (!setup_tri(t, d.cull_mode, d.front_face)) return;
        raster_any(fp, tc, t, make_payload(fp, slot, 0));
    } else if (cls == 2) {
        ClipFan fan;
        build_clip_fan(fp, d, fp.derived[draw].clip_m, pf.idx[0], pf.idx[1], pf.idx[2], fan);
        for (int k = 1; k + 1 < fan.n; ++k) {
            RTri t; float bary[3][3];
            if (!fan_triangle(fan, k, d.cull_mode, d.front_face, t, bary)) continue;
            raster_any(fp, tc, t, make_payload(fp, slot, (uint32_t)(k - 1)));
        }
    }
}

/* ---- fragment stage ---- */

struct Varyings { float n[3]; float uv[2]; float col[3]; };

__device__ __forceinline__ void vs_varyings(const DrawStatic& d, const float* nrm_m, uint32_t v, Varyings& out)
{
    const uint8_t* base = d.vb + (size_t)v * d.stride;
    float nrm[3] = {0.0f, 0.0f, 0.0f}, uv[2] = {0.0f, 0.0f};
    if (d.off_nrm >= 0) { const float* p = reinterpret_cast<const float*>(base + d.off_nrm); nrm[0] = p[0]; nrm[1] = p[1]; nrm[2] = p[2]; }
    if (d.off_uv >= 0) { const float* p = reinterpret_cast<const float*>(base + d.off_uv); uv[0] = p[0]; uv[1] = p[1]; }
    if (d.shader == 0) {
        out.n[0] = nrm[0]; out.n[1] = nrm[1]; out.n[2] = nrm[2];
        out.uv[0] = uv[0]; out.uv[1] = 1.0f - uv[1];
        out.col[0] = out.col[1] = out.col[2] = 0.0f;
    } else {
        float len2 = fma32(nrm[2], nrm[2], fma32(nrm[1], nrm[1], nrm[0] * nrm[0]));
        float len = sqrtf(len2);
        float nn[3] = {nrm[0] / len, nrm[1] / len, nrm[2] / len};
        mat3_vec(nrm_m, nn, out.n);
        out.uv[0] = uv[0]; out.uv[1] = uv[1];
        out.col[0] = d.push[0]; out.col[1] = d.push[1]; out.col[2] = d.push[2];
    }
}

/* perspective-correct weights over the parent's corners at a window position */
__device__ __forceinline__ void parent_weights(const RTri& t, bool clipped, const float bary[3][3], int64_t px, int64_t py, float* lam)
{
    float b0 = (float)edge64(t.x[1], t.y[1], t.x[2], t.y[2], px, py) * t.inv_area;
    float b1 = (float)edge64(t.x[2], t.y[2], t.x[0], t.y[0], px, py) * t.inv_area;
    float b2 = (float)edge64(t.x[0], t.y[0], t.x[1], t.y[1], px, py) * t.inv_area;
    float p0 = b0 * t.iw[0], p1 = b1 * t.iw[1], p2 = b2 * t.iw[2];
    float den = (p0 + p1) + p2;
    float rden = 1.0f / den;
    float q[3] = {p0 * rden, p1 * rden, p2 * rden};
    if (!clipped) {
        lam[t.corner[0]] = q[0]; lam[t.corner[1]] = q[1]; lam[t.corner[2]] = q[2];
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) lam[j] = fma32(q[2], bary[2][j], fma32(q[1], bary[1][j], q[0] * bary[0][j]));
    }
}

__device__ __forceinline__ float interp(const float* lam, float a0, float a1, float a2)
{
    return fma32(lam[2], a2, fma32(lam[1], a1, lam[0] * a0));
}

__device__ __forceinline__ uint32_t level_dim(uint32_t d, uint32_t l) { uint32_t v = d >> l; return v ? v : 1u; }

__device__ __forceinline__ int wrap_coord(int i, int n, uint32_t tiled)
{
    if (tiled) { int m = i % n; return m < 0 ? m + n : m; }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

__device__ __forceinline__ void fetch_texel(const ShadeTables& tb, const TexDesc& t, const uint8_t* lv, int w, int x, int y, float* rgb)
{
    uint32_t p = *reinterpret_cast<const uint32_t*>(lv + ((size_t)y * (size_t)w + (size_t)x) * 4);
    uint32_t r = p & 255u, g = (p >> 8) & 255u, b = (p >> 16) & 255u;
    if (t.srgb) { rgb[0] = tb.decode[r]; rgb[1] = tb.decode[g]; rgb[2] = tb.decode[b]; }
    else { rgb[0] = (float)r / 255.0f; rgb[1] = (float)g / 255.0f; rgb[2] = (float)b / 255.0f; }
}

__device__ void sample_level(const ShadeTables& tb, const TexDesc& t, uint32_t level, float u, float v, float* rgb)
{
    int w = (int)level_dim(t.width, level), h = (int)level_dim(t.height, level);
    const uint8_t* lv = t.level[level];
    if (!finite32(u)) u = 0.0f;
    if (!finite32(v)) v = 0.0f;
    if (t.filter_linear) {
        float fu = fma32(u, (float)w, -0.5f), fv = fma32(v, (float)h, -0.5f);
        fu = fminf(fmaxf(fu, -1.0e9f), 1.0e9f);
        fv = fminf(fmaxf(fv, -1.0e9f), 1.0e9f);
        float iu = floorf(fu), iv = floorf(fv);
        float ax = fu - iu, ay = fv - iv;
        int x0 = wrap_coord((int)iu, w, t.tiled), x1 = wrap_coord((int)iu + 1, w, t.tiled);
        int y0 = wrap_coord((int)iv, h, t.tiled), y1 = wrap_coord((int)iv + 1, h, t.tiled);
        float c00[3], c10[3], c01[3], c11[3];
        fetch_texel(tb, t, lv, w, x0, y0, c00);
        fetch_texel(tb, t, lv, w, x1, y0, c10);
        fetch_texel(tb, t, lv, w, x0, y1, c01);
        fetch_texel(tb, t, lv, w, x1, y1, c11);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float top = fma32(ax, c10[k] - c00[k], c00[k]);
            float bot = fma32(ax, c11[k] - c01[k], c01[k]);
            rgb[k] = fma32(ay, bot - top, top);
        }
    } else {
        float fu = fminf(fmaxf(u * (float)w, -1.0e9f), 1.0e9f);
        float fv = fminf(fmaxf(v * (float)h, -1.0e9f), 1.0e9f);
        int x = wrap_coord((int)floorf(fu), w, t.tiled), y = wrap_coord((int)floorf(fv), h, t.tiled);
        fetch_texel(tb, t, lv, w, x, y, rgb);
    }
}

/* texture() with implicit derivatives: lod = log2(longest footprint axis), clamped to
 * [0, levels-1], trilinear (sampler of tools.cpp:60-82) */
__device__ void sample_texture(const ShadeTables& tb, const TexDesc& t, float u, float v,
                               float dudx, float dvdx, float dudy, float dvdy, float* rgb)
{
    float w0 = (float)t.width, h0 = (float)t.height;
    float mxu = dudx * w0, mxv = dvdx * h0, myu = dudy * w0, myv = dvdy * h0;
    float rx = fma32(mxv, mxv, mxu * mxu), ry = fma32(myv, myv, myu * myu);
    float lambda = 0.5f * lod_log2(fmaxf(rx, ry));
    float max_lod = (float)(t.levels - 1);
    lambda = fminf(fmaxf(lambda, 0.0f), max_lod);
    if (!(lambda > 0.0f) || t.levels == 1) { sample_level(tb, t, 0, u, v, rgb); return; }
    if (t.filter_linear) {
        float fl = floorf(lambda);
        uint32_t lo = (uint32_t)fl;
        uint32_t hi = lo + 1 < t.levels ? lo + 1 : t.levels - 1;
        float frac = lambda - fl;
        float a[3], b[3];
        sample_level(tb, t, lo, u, v, a);
        sample_level(tb, t, hi, u, v, b);
#pragma unroll
        for (int k = 0; k < 3; ++k) rgb[k] = fma32(frac, b[k] - a[k], a[k]);
    } else {
        float lv = ceilf(lambda + 0.5f) - 1.0f;
        lv = fminf(fmaxf(lv, 0.0f), max_lod);
        sample_level(tb, t, (uint32_t)lv, u, v, rgb);
    }
}

/* colour of the fragment of (slot, sub) at pixel (x, y) */
__device__ void shade_pixel(const FrameParams& fp, const ShadeTables& tb, uint32_t slot, uint32_t sub, int x, int y, float* rgba)
{
    uint32_t draw, tri;
    slot_to_prim(fp, slot, draw, tri);
    const DrawStatic& d = fp.draws[draw];
    const DrawDerived& dd = fp.derived[draw];
    PrimFetch pf;
    fetch_prim(fp, d, tri, pf);
    RTri t;
    float bary[3][3];
    bool clipped = classify_prim(pf) == 2;
    if (!clipped) {
        tri_from_fetch(pf, t);
        setup_tri(t, d.cull_mode, d.front_face);
    } else {
        ClipFan fan;
        build_clip_fan(fp, d, dd.clip_m, pf.idx[0], pf.idx[1], pf.idx[2], fan);
        fan_triangle(fan, (int)sub + 1, d.cull_mode, d.front_face, t, bary);
    }
    Varyings v0, v1, v2;
    vs_varyings(d, dd.nrm_m, pf.idx[0], v0);
    vs_varyings(d, dd.nrm_m, pf.idx[1], v1);
    vs_varyings(d, dd.nrm_m, pf.idx[2], v2);
    const int64_t px = (int64_t)x * 256 + 128, py = (int64_t)y * 256 + 128;
    float lam[3];
    parent_weights(t, clipped, bary, px, py, lam);
    float tex[3] = {1.0f, 1.0f, 1.0f};
    if (d.shader == 0 || d.shader == 1) {
        if (d.texture >= 0) {
            float u = interp(lam, v0.uv[0], v1.uv[0], v2.uv[0]), v = interp(lam, v0.uv[1], v1.uv[1], v2.uv[1]);
            /* derivatives inside the 2x2 pixel quad: the neighbour in x and the neighbour in y, same primitive */
            float lnx[3], lny[3];
            parent_weights(t, clipped, bary, (int64_t)(x ^ 1) * 256 + 128, py, lnx);
            parent_weights(t, clipped, bary, px, (int64_t)(y ^ 1) * 256 + 128, lny);
            float unx = interp(lnx, v0.uv[0], v1.uv[0], v2.uv[0]), vnx = interp(lnx, v0.uv[1], v1.uv[1], v2.uv[1]);
            float uny = interp(lny, v0.uv[0], v1.uv[0], v2.uv[0]), vny = interp(lny, v0.uv[1], v1.uv[1], v2.uv[1]);
            float dudx = (x & 1) ? u - unx : unx - u, dvdx = (x & 1) ? v - vnx : vnx - v;
            float dudy = (y & 1) ? u - uny : uny - u, dvdy = (y & 1) ? v - vny : vny - v;
            sample_texture(tb, fp.textures[d.texture], u, v, dudx, dvdx, dudy, dvdy, tex);
        } else {
            tex[0] = tex[1] = tex[2] = 0.0f;
        }
    }
    if (d.shader == 0) { rgba[0] = tex[0]; rgba[1] = tex[1]; rgba[2] = tex[2]; rgba[3] = 1.0f; return; }
    float n[3] = {interp(lam, v0.n[0], v1.n[0], v2.n[0]), interp(lam, v0.n[1], v1.n[1], v2.n[1]), interp(lam, v0.n[2], v1.n[2], v2.n[2])};
    const float lx = 0.0f, ly = 0.70710678118654752f, lz = -0.70710678118654752f;
    float diff = fmaxf(fma32(n[2], lz, fma32(n[1], ly, n[0] * lx)), 0.0f);
    if (d.shader == 1) {
        rgba[0] = tex[0] * diff; rgba[1] = tex[1] * diff; rgba[2] = tex[2] * diff;
    } else {
        float c0 = interp(lam, v0.col[0], v1.col[0], v2.col[0]), c1 = interp(lam, v0.col[1], v1.col[1], v2.col[1]),
              c2 = interp(lam, v0.col[2], v1.col[2], v2.col[2]);
        rgba[0] = c0 * diff; rgba[1] = c1 * diff; rgba[2] = c2 * diff;
    }
    rgba[3] = 1.0f;
}

__device__ __forceinline__ uint32_t pack_color(const FrameParams& fp, const ShadeTables& tb, const float* rgba)
{
    uint32_t r, g, b;
    if (fp.color_srgb) { r = srgb_encode(tb, rgba[0]); g = srgb_encode(tb, rgba[1]); b = srgb_encode(tb, rgba[2]); }
    else { r = unorm8(rgba[0]); g = unorm8(rgba[1]); b = unorm8(rgba[2]); }
    uint32_t a = unorm8(rgba[3]);
    return fp.color_bgra ? (b | (g << 8) | (r << 16) | (a << 24)) : (r | (g << 8) | (b << 16) | (a << 24));
}

template <int S>
__global__ __launch_bounds__(TILE_THREADS) void k_tile(FrameParams fp)
{
    __shared__ unsigned long long keys[TILE_PIXELS * S];
    __shared__ ShadeTables tb;

    const uint32_t tile = blockIdx.x * fp.shard_count + fp.shard_index;
    const int tx = (int)(tile % fp.tiles_x), ty = (int)(tile / fp.tiles_x);
    TileCtx tc;
    tc.px0 = tx * TILE_W; tc.py0 = ty * TILE_H;
    tc.rx0 = max(tc.px0, fp.clamp_x0); tc.ry0 = max(tc.py0, fp.clamp_y0);
    tc.rx1 = min(tc.px0 + TILE_W - 1, fp.clamp_x1); tc.ry1 = min(tc.py0 + TILE_H - 1, fp.clamp_y1);
    tc.keys = keys;
    tc.samples = S;

    const unsigned long long clear_key = ((unsigned long long)(fp.depth_test ? fp.clear_d24 : 0xFFFFFFu) << 40) |
                                         (fp.depth_less ? 0ull : 0xFFFFFFFFFFull);
    for (uint32_t i = threadIdx.x; i < TILE_PIXELS * S; i += TILE_THREADS) keys[i] = clear_key;
    for (uint32_t i = threadIdx.x; i < 256; i += TILE_THREADS) tb.decode[i] = fp.srgb_decode[i];
    for (uint32_t i = threadIdx.x; i < 257; i += TILE_THREADS) tb.thr[i] = fp.srgb_thr[i];
    for (uint32_t i = threadIdx.x; i < SRGB_LUT_SIZE; i += TILE_THREADS) tb.lut[i] = fp.srgb_lut[i];
    __syncthreads();

    if (tc.rx0 <= tc.rx1 && tc.ry0 <= tc.ry1) {
        const uint32_t first = fp.tile_offset[tile], count = fp.tile_count[tile];
        for (uint32_t i = threadIdx.x; i < count; i += TILE_THREADS) raster_slot(fp, tc, fp.refs[first + i]);
        const uint32_t n_large = *fp.large_count;
        for (uint32_t i = threadIdx.x; i < n_large; i += TILE_THREADS) {
            const uint32_t slot = fp.large_list[i];
            const uint32_t box = fp.prim_bbox[slot];
            const int bx0 = box & 255u, by0 = (box >> 8) & 255u, bx1 = (box >> 16) & 255u, by1 = box >> 24;
            if (tx < bx0 || tx > bx1 || ty < by0 || ty > by1) continue;
            raster_slot(fp, tc, slot);
        }
    }
    __syncthreads();

    /* shade and store: consecutive lanes take consecutive pixels of a row, so a wave
     * stores two full 128-byte row segments per step */
    for (uint32_t p = threadIdx.x; p < TILE_PIXELS; p += TILE_THREADS) {
        const int x = tc.px0 + (int)(p & (TILE_W - 1)), y = tc.py0 + (int)(p >> TILE_SHIFT);
        if (x < fp.area_x0 || x > fp.area_x1 || y < fp.area_y0 || y > fp.area_y1) continue;
        float acc[4];
        float smp[S][4];
        unsigned long long k[S];
#pragma unroll
        for (int s = 0; s < S; ++s) k[s] = keys[p * S + s];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            /* a sample still holding the clear word was never won by a fragment */
            const bool covered = k[s] != clear_key;
            /* one shading per (pixel, primitive): samples owned by the same fan triangle share it */
            int reuse = -1;
#pragma unroll
            for (int s2 = 0; s2 < s; ++s2) {
                if (reuse < 0 && ((k[s2] ^ k[s]) & 0xFFFFFFFFFFull) == 0ull && k[s2] != clear_key) reuse = s2;
            }
            if (!covered) {
                smp[s][0] = fp.clear_color[0]; smp[s][1] = fp.clear_color[1]; smp[s][2] = fp.clear_color[2]; smp[s][3] = fp.clear_color[3];
            } else if (reuse >= 0) {
                smp[s][0] = smp[reuse][0]; smp[s][1] = smp[reuse][1]; smp[s][2] = smp[reuse][2]; smp[s][3] = smp[reuse][3];
            } else {
                const uint32_t low = (uint32_t)(k[s] >> 8);
                const uint32_t slot = fp.depth_less ? low : ~low;
                shade_pixel(fp, tb, slot, (uint32_t)(k[s] & 255ull), x, y, smp[s]);
            }
            const uint32_t d24 = fp.depth_test ? (uint32_t)(k[s] >> 40) : fp.clear_d24;
            fp.depth[((size_t)y * fp.width + (size_t)x) * S + s] = d24;
        }
        if constexpr (S == 1) {
            acc[0] = smp[0][0]; acc[1] = smp[0][1]; acc[2] = smp[0][2]; acc[3] = smp[0][3];
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = ((smp[0][c] + smp[1][c]) + (smp[2][c] + smp[3][c])) * 0.25f;
        }
        reinterpret_cast<uint32_t*>(fp.color)[(size_t)y * fp.width + (size_t)x] = pack_color(fp, tb, acc);
    }
}

/* ------------------------------------------------------------------ mip generation (blit LINEAR, tools.cpp:670-707) */

__global__ __launch_bounds__(256) void k_mip(const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh,
                                             int srgb, const float* g_decode, const float* g_thr, const uint8_t* g_lut)
{
    __shared__ ShadeTables tb;
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) tb.decode[i] = g_decode[i];
    for (uint32_t i = threadIdx.x; i < 257; i += blockDim.x) tb.thr[i] = g_thr[i];
    for (uint32_t i = threadIdx.x; i < SRGB_LUT_SIZE; i += blockDim.x) tb.lut[i] = g_lut[i];
    __syncthreads();
    const float scale_x = (float)sw / (float)dw, scale_y = (float)sh / (float)dh;
    const size_t total = (size_t)dw * dh;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)(i % dw), y = (uint32_t)(i / dw);
        float fu = fma32((float)x + 0.5f, scale_x, -0.5f), fv = fma32((float)y + 0.5f, scale_y, -0.5f);
        float iu = floorf(fu), iv = floorf(fv);
        float ax = fu - iu, ay = fv - iv;
        int x0 = wrap_coord((int)iu, (int)sw, 0), x1 = wrap_coord((int)iu + 1, (int)sw, 0);
        int y0 = wrap_coord((int)iv, (int)sh, 0), y1 = wrap_coord((int)iv + 1, (int)sh, 0);
        uint32_t p00 = reinterpret_cast<const uint32_t*>(src)[(size_t)y0 * sw + x0];
        uint32_t p10 = reinterpret_cast<const uint32_t*>(src)[(size_t)y0 * sw + x1];
        uint32_t p01 = reinterpret_cast<const uint32_t*>(src)[(size_t)y1 * sw + x0];
        uint32_t p11 = reinterpret_cast<const uint32_t*>(src)[(size_t)y1 * sw + x1];
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bool lin = !(srgb && k < 3);
            uint32_t b00 = (p00 >> (8 * k)) & 255u, b10 = (p10 >> (8 * k)) & 255u, b01 = (p01 >> (8 * k)) & 255u, b11 = (p11 >> (8 * k)) & 255u;
            float c00 = lin ? (float)b00 / 255.0f : tb.decode[b00];
            float c10 = lin ? (float)b10 / 255.0f : tb.decode[b10];
            float c01 = lin ? (float)b01 / 255.0f : tb.decode[b01];
            float c11 = lin ? (float)b11 / 255.0f : tb.decode[b11];
            float top = fma32(ax, c10 - c00, c00);
            float bot = fma32(ax, c11 - c01, c01);
            float c = fma32(ay, bot - top, top);
            out |= (lin ? unorm8(c) : srgb_encode(tb, c)) << (8 * k);
        }
        reinterpret_cast<uint32_t*>(dst)[i] = out;
    }
}

/* ------------------------------------------------------------------ launch wrappers */

void launch_index_range(const IndexRangeJob* jobs, uint32_t* out_max, uint32_t n_jobs, hipStream_t s)
{
    if (!n_jobs) return;
    hipLaunchKernelGGL(k_index_range, dim3(64, n_jobs), dim3(256), 0, s, jobs, out_max);
}

void launch_constants(const FrameParams& fp, hipStream_t s)
{
    if (!fp.n_draws) return;
    hipLaunchKernelGGL(k_constants, dim3((fp.n_draws + 63) / 64), dim3(64), 0, s, fp);
}

void launch_vertex(const FrameParams& fp, hipStream_t s)
{
    if (!fp.n_vchunks) return;
    hipLaunchKernelGGL(k_vertex, dim3((fp.n_vchunks + 3) / 4), dim3(256), 0, s, fp);
}

void launch_setup(const FrameParams& fp, hipStream_t s)
{
    if (!fp.n_tchunks) return;
    hipLaunchKernelGGL(k_setup, dim3((fp.n_tchunks + 3) / 4), dim3(256), 0, s, fp);
}

void launch_scan(const FrameParams& fp, hipStream_t s)
{
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, fp);
}

void launch_fill(const FrameParams& fp, hipStream_t s)
{
    if (!fp.n_tchunks) return;
    hipLaunchKernelGGL(k_fill, dim3((fp.n_tchunks + 3) / 4), dim3(256), 0, s, fp);
}

void launch_tile(const FrameParams& fp, hipStream_t s)
{
    const uint32_t n_tiles = fp.tiles_x * fp.tiles_y;
    const uint32_t mine = n_tiles > fp.shard_index ? (n_tiles - fp.shard_index + fp.shard_count - 1) / fp.shard_count : 0;
    if (!mine) return;
    if (fp.samples == 4) hipLaunchKernelGGL(k_tile<4>, dim3(mine), dim3(TILE_THREADS), 0, s, fp);
    else hipLaunchKernelGGL(k_tile<1>, dim3(mine), dim3(TILE_THREADS), 0, s, fp);
}

void launch_mip(const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh, int srgb,
                const float* srgb_decode, const float* srgb_thr, const uint8_t* srgb_lut, hipStream_t s)
{
    size_t total = (size_t)dw * dh;
    uint32_t blocks = (uint32_t)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_mip, dim3(blocks), dim3(256), 0, s, src, sw, sh, dst, dw, dh, srgb, srgb_decode, srgb_thr, srgb_lut);
}

} // namespace rnd
