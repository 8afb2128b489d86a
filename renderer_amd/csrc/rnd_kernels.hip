/*
 * rnd_kernels.hip -- gfx950 kernels of the draw path.
 *
 * One submit of a recorded pass runs three kernels, after a fourth in a pass of many draws or many vertices:
 *   k_draws      one lane per draw: the draw's matrices and whether its box can show at all; in a pass of
 *                many draws also the dense lists of the visible draws' chunks, which the next two walk
 *   k_vertex     vertex fetch, the draw's matrices built per wave from the uniform buffer as it
 *                is at execution time (base_obj_viewer_app.cpp:134-138) and the push constants,
 *                position shader, clip codes, viewport, 24.8 snap
 *   k_setup      per primitive: reject / cull / orient, then append an entry to the list of
 *                every tile its box touches (fixed-capacity lists, one atomic per wave and tile
 *                through ballot + popcount prefix); the few primitives crossing the near / far
 *                planes or the guard band are clipped here, fanned, and binned
 *   k_tile       one workgroup per 32x32 tile: edge-function rasteriser into an LDS depth/owner
 *                word per sample, then per pixel one shading per owning primitive, resolve,
 *                sRGB encode, coalesced depth and colour stores
 *
 * Depth resolution is order independent: each fragment offers the 64-bit word
 * (D24 << 40 | inverted submission number << 8 | fan index) and the tile keeps the
 * minimum.  Under LESS_OR_EQUAL with depth writes that is exactly what in-order
 * processing leaves behind: the smallest depth, and among equal depths the last
 * primitive submitted.
 */
#include "rnd_device.h"

#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

/* Scratch of rnd_draw_indexed_noalloc: every buffer a frame's kernels hand to one another (screen-space vertices, tile
 * lists, counters, ...), laid out by rnd::NoAllocLayout.  In host code this name is only a handle for
 * hipGetSymbolAddress; kernels get the address that call returns. */
static __device__ __align__(256) unsigned char scratch[rnd::NoAllocLayout::total];

namespace rnd {

hipError_t noalloc_scratch_address(void** out) { return hipGetSymbolAddress(out, HIP_SYMBOL(scratch)); }

/* ------------------------------------------------------------------ helpers */

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t wave_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

/* Workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share an L2).  A grid that
 * covers `cols` x rows cells is walked in groups of 8 rows: inside a group, id r goes to row r % 8
 * (= its XCD) and column r / 8.  Every XCD thus gets whole rows -- horizontal neighbours, which
 * gather the same vertices and texels, meet in one L2 -- while the rows stay spread over all XCDs
 * for balance.  The grid is padded to a multiple of 8 rows; cells past the last row do nothing.
 * The one division, id / (8 * cols), is a multiply-high by magic = ceil(2^32 / (8 * cols)), exact
 * while id * 8 * cols < 2^32 (ids stay below 2^18, 8 * cols below 2^12).  Placement only: results
 * do not depend on it. */
__device__ __forceinline__ void rows_of_eight(uint32_t id, uint32_t cols, uint32_t magic, uint32_t& col, uint32_t& row)
{
    const uint32_t group = 8u * cols;
    const uint32_t g = __umulhi(id, magic);
    const uint32_t r = id - g * group;
    col = r >> 3;
    row = g * 8u + (r & 7u);
}

static uint32_t rows_of_eight_magic(uint32_t cols) { return (uint32_t)((0x100000000ull + 8ull * cols - 1ull) / (8ull * cols)); }
static uint32_t rows_of_eight_grid(uint32_t cols, uint32_t rows) { return ((rows + 7u) / 8u) * 8u * cols; }

__device__ __forceinline__ uint32_t pack_bbox(int tx0, int ty0, int tx1, int ty1)
{
    return (uint32_t)tx0 | ((uint32_t)ty0 << 8) | ((uint32_t)tx1 << 16) | ((uint32_t)ty1 << 24);
}

/* row of tiles of pixel row y, which pixel_box has clamped to the target (never negative, below 2^16): the division by
 * the pass's tile height as the high word of one multiplication (FrameParams::tile_h_magic) */
__device__ __forceinline__ int tile_row(const FrameParams& fp, int y) { return (int)__umulhi((uint32_t)y, fp.tile_h_magic); }

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
    const uint32_t* ip = d.ib + (size_t)tri * 3; /* three adjacent words: the compiler merges them into one 12-byte load */
    pf.idx[0] = load_u32(ip); pf.idx[1] = load_u32(ip + 1); pf.idx[2] = load_u32(ip + 2);
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
        /* a pixel's samples lie between 1/8 and 7/8 of it on both axes (32 .. 224 of 256): pixels with a sample inside [min, max] */
        x0 = (minx + 31) >> 8; x1 = (maxx - 32) >> 8;
        y0 = (miny + 31) >> 8; y1 = (maxy - 32) >> 8;
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

/* ------------------------------------------------------------------ record-time: box of a draw's vertices */

/* floats to unsigned integers of the same order (NaN sorts above +inf or below -inf and is skipped by the caller) */
__device__ __forceinline__ uint32_t float_to_ordered(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

/* A vertex with a NaN coordinate is left out: every primitive that uses it is dropped by setup. */
__global__ __launch_bounds__(256) void k_bounds(const BoundsJob* jobs, uint32_t* out_min_max)
{
    const BoundsJob job = jobs[blockIdx.y];
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < job.count; i += gridDim.x * blockDim.x) {
        const float* p = reinterpret_cast<const float*>(job.vb + (size_t)i * job.stride + job.off_pos);
        const float x = load_f32(p), y = load_f32(p + 1), z = load_f32(p + 2);
        if (x != x || y != y || z != z) continue;
        const uint32_t o[3] = {float_to_ordered(x), float_to_ordered(y), float_to_ordered(z)};
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[k] = min(lo[k], o[k]); hi[k] = max(hi[k], o[k]); }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo[k] = min(lo[k], (uint32_t)__shfl_xor((int)lo[k], d));
            hi[k] = max(hi[k], (uint32_t)__shfl_xor((int)hi[k], d));
        }
    }
    if (lane_id() == 0) {
        uint32_t* out = out_min_max + (size_t)blockIdx.y * 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (lo[k] != 0xFFFFFFFFu) atomicMin(&out[k], lo[k]);
            if (hi[k] != 0u) atomicMax(&out[3 + k], hi[k]);
        }
    }
}

/* ------------------------------------------------------------------ vertex kernel */

/* a float of a uniform buffer as the frame must see it: from the pending host write when that
 * covers the address, else from memory */
struct VertexArgs {
    FrameParams fp;
    UniformPatch up;
};

__device__ __forceinline__ float uniform_read(const UniformPatch& up, const float* p)
{
    const size_t off = (size_t)(reinterpret_cast<const uint8_t*>(p) - up.dst);
    if (up.dst && off < (size_t)up.bytes) {
        /* the words are indexed per lane: read them where the kernel arguments already are in
         * memory (VertexArgs is the kernel's only argument), so the compiler need not copy
         * the block into private memory to index it */
        typedef const __attribute__((address_space(4))) uint32_t* KernargWords;
        KernargWords w = (KernargWords)((const __attribute__((address_space(4))) uint8_t*)__builtin_amdgcn_kernarg_segment_ptr() +
                                        offsetof(VertexArgs, up) + offsetof(UniformPatch, words));
        return __uint_as_float(w[off >> 2]);
    }
    return load_f32(p);
}

/* the same without a branch, for a lane that reads many floats in a row: both places are always readable (the kernel's
 * arguments; the uniform buffer, whose range was checked when the draw was recorded), so the loads all go out together
 * instead of each waiting behind the previous one's branch */
__device__ __forceinline__ float uniform_read_both(const UniformPatch& up, const float* p)
{
    const size_t off = (size_t)(reinterpret_cast<const uint8_t*>(p) - up.dst);
    const bool patched = up.dst && off < (size_t)up.bytes;
    typedef const __attribute__((address_space(4))) uint32_t* KernargWords;
    KernargWords w = (KernargWords)((const __attribute__((address_space(4))) uint8_t*)__builtin_amdgcn_kernarg_segment_ptr() +
                                    offsetof(VertexArgs, up) + offsetof(UniformPatch, words));
    const uint32_t from_patch = w[patched ? off >> 2 : 0];
    const float from_memory = load_f32(p);
    return patched ? __uint_as_float(from_patch) : from_memory;
}

__device__ __forceinline__ float lane_value(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

/* Element k (column-major) of a draw's matrix A * push.transform, A being the 4x4 at `a` in the uniform buffer as it
 * is when the frame executes (base_obj_viewer_app.cpp:126-138 writes it every frame): clip_m = mvp * push.transform,
 * and the normal matrix is inverse(transpose(mat3(model * push.transform))) (test_ktx.vert:28-30).  The dummy shader
 * uses mvp alone (dummy.vert.glsl:22).  One four-term multiply-add chain per element, whoever runs it. */
__device__ __forceinline__ float matrix_element(const UniformPatch& up, const DrawStatic& d, int shader, const float* a, uint32_t k)
{
    const uint32_t row = k & 3u, col = k >> 2;
    if (shader == 0) return uniform_read(up, a + k);
    const float* b = d.push + 4 + col * 4;
    float acc = uniform_read(up, a + 0 * 4 + row) * b[0];
    acc = fma32(uniform_read(up, a + 1 * 4 + row), b[1], acc);
    acc = fma32(uniform_read(up, a + 2 * 4 + row), b[2], acc);
    acc = fma32(uniform_read(up, a + 3 * 4 + row), b[3], acc);
    return acc;
}

/* The same built by a wave for its own use: lane k < 16 computes element k of mvp * transform, lane 16 + k element k
 * of model * transform; the results are then broadcast. */
__device__ __forceinline__ float wave_matrix_element(const UniformPatch& up, const DrawStatic& d, int shader, const float* ubo)
{
    const uint32_t lane = lane_id();
    return matrix_element(up, d, shader, ubo + ((lane & 16u) && shader != 0 ? 48 : 64), lane & 15u); /* model : mvp */
}

/* object space -> clip space, from lanes 0..15 */
__device__ __forceinline__ void wave_clip_matrix(float element, float* clip_m)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) clip_m[i] = lane_value(element, i);
}

/* the normal matrix, from lanes 16..26 */
__device__ __forceinline__ void wave_normal_matrix(int shader, float element, float* nrm_m)
{
    if (shader == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) nrm_m[i] = 0.0f;
    } else {
        float mt[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) mt[i] = 0.0f;
        mt[0] = lane_value(element, 16); mt[1] = lane_value(element, 17); mt[2] = lane_value(element, 18);
        mt[4] = lane_value(element, 20); mt[5] = lane_value(element, 21); mt[6] = lane_value(element, 22);
        mt[8] = lane_value(element, 24); mt[9] = lane_value(element, 25); mt[10] = lane_value(element, 26);
        normal_matrix(mt, nrm_m);
    }
}

/* True when no fragment of the draw can land inside the clamp rectangle: the eight corners of the box
 * of its vertices, taken through the draw's matrix and the shaders' y = 1 - y, all lie beyond one and the
 * same plane out of the rectangle's four (grown by a pixel), near and far.  A plane function is affine
 * in the object-space position, so its largest value over the box is at a corner; a corner counts as
 * beyond a plane only by more than TOL times the sum of the magnitudes that went into the function,
 * which covers the rounding of this evaluation and of the vertices' own transform.  corner_beyond gives,
 * for one corner, a bit per plane. */
__device__ __forceinline__ uint32_t corner_beyond(const FrameParams& fp, const DrawStatic& d, const float* m, uint32_t corner)
{
    constexpr float TOL = 1.0e-5f;
    const float p[3] = {(corner & 1u) ? d.bounds_max[0] : d.bounds_min[0], (corner & 2u) ? d.bounds_max[1] : d.bounds_min[1],
                        (corner & 4u) ? d.bounds_max[2] : d.bounds_min[2]};
    float clip[4], mag[4];
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        clip[row] = fma32(m[0 * 4 + row], p[0], fma32(m[1 * 4 + row], p[1], fma32(m[2 * 4 + row], p[2], m[3 * 4 + row])));
        mag[row] = fabsf(m[0 * 4 + row] * p[0]) + fabsf(m[1 * 4 + row] * p[1]) + fabsf(m[2 * 4 + row] * p[2]) + fabsf(m[3 * 4 + row]);
    }
    clip[1] = 1.0f - clip[1];
    mag[1] = mag[1] + 1.0f;
    const float x = clip[0], y = clip[1], z = clip[2], w = clip[3];
    uint32_t out = 0u;
    out |= fma32(-fp.cull_lo_x, w, x) < -TOL * fma32(fabsf(fp.cull_lo_x), mag[3], mag[0]) ? 1u : 0u;
    out |= fma32(fp.cull_hi_x, w, -x) < -TOL * fma32(fabsf(fp.cull_hi_x), mag[3], mag[0]) ? 2u : 0u;
    out |= fma32(-fp.cull_lo_y, w, y) < -TOL * fma32(fabsf(fp.cull_lo_y), mag[3], mag[1]) ? 4u : 0u;
    out |= fma32(fp.cull_hi_y, w, -y) < -TOL * fma32(fabsf(fp.cull_hi_y), mag[3], mag[1]) ? 8u : 0u;
    out |= z < -TOL * mag[2] ? 16u : 0u;
    out |= w - z < -TOL * (mag[3] + mag[2]) ? 32u : 0u;
    return out;
}

/* by a wave: lane k < 8 takes corner k; the answer is the same in every lane */
__device__ __forceinline__ bool draw_is_outside(const FrameParams& fp, const DrawStatic& d, const float* m)
{
    const uint32_t mine = corner_beyond(fp, d, m, lane_id() & 7u);
    bool outside = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) outside = outside || (__ballot((mine >> k) & 1u) & 0xFFull) == 0xFFull;
    return outside;
}

/* ... and a pass of few draws with many vertices: every vertex wave would build its draw's matrices again (two thirds of
 * such a wave's vector instructions), which for some thousand waves is more than the extra launch costs */
constexpr uint32_t PER_DRAW_KERNEL_MIN_WAVES = 2048;

/* Scans over the wave by data-parallel-primitive moves (row shifts inside each 16-lane row, then the rows' last
 * lanes broadcast to the rows after them): six vector instructions, where shuffles through the LDS crossbar take a
 * round trip per step. */
template <typename Op>
__device__ __forceinline__ int wave_inclusive_scan(int v, int identity, Op op)
{
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x111, 0xf, 0xf, false)); /* row_shr:1 */
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x112, 0xf, 0xf, false)); /* row_shr:2 */
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x114, 0xf, 0xf, false)); /* row_shr:4 */
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x118, 0xf, 0xf, false)); /* row_shr:8 */
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x142, 0xa, 0xf, false)); /* row_bcast:15 into rows 1 and 3 */
    v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x143, 0xc, 0xf, false)); /* row_bcast:31 into rows 2 and 3 */
    return v;
}

/* `count_v` chunks of 64 vertices and `count_t` chunks of 64 primitives of this lane's draw join the two dense lists.
 * The workgroup takes one range of each list for all its draws with ONE atomic per list (some thousand atomics a frame on
 * one address cost ten nanoseconds each, which for a pass of 40 000 draws was most of the kernel when every wave issued its
 * own): the waves scan their lanes' counts, leave their totals in LDS, thread 0 adds the workgroup's total to the list's
 * length and leaves every wave its first place.  Each lane then writes its draw's chunks; a draw of very many chunks has
 * them written by its whole wave. */
constexpr uint32_t DRAWS_THREADS = 512;
__device__ __forceinline__ void append_chunks(const FrameParams& fp, uint32_t draw, uint32_t count_v, uint32_t count_t)
{
    __shared__ uint32_t wave_first[2][DRAWS_THREADS / 64];
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t count[2] = {count_v, count_t}, incl[2];
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        incl[which] = (uint32_t)wave_inclusive_scan((int)count[which], 0, [](int x, int y) { return x + y; });
        if (lane == 63u) wave_first[which][wave] = incl[which];
    }
    __syncthreads();
    if (threadIdx.x < 2u) {
        const uint32_t which = threadIdx.x;
        uint32_t total = 0u;
        for (uint32_t w = 0; w < blockDim.x / 64u; ++w) total += wave_first[which][w];
        uint32_t first = total ? atomicAdd(&fp.counters[visible_length(fp.list_parity, which)], total) : 0u;
        for (uint32_t w = 0; w < blockDim.x / 64u; ++w) { const uint32_t t = wave_first[which][w]; wave_first[which][w] = first; first += t; }
    }
    __syncthreads();
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        Chunk* list = which ? fp.visible_tchunks : fp.visible_vchunks;
        const uint32_t n = count[which];
        const uint32_t at = wave_first[which][wave] + incl[which] - n;
        constexpr uint32_t FEW = 8u;
        if (n <= FEW)
            for (uint32_t j = 0; j < n; ++j) list[at + j] = Chunk{draw, j * 64u};
        uint64_t many = __ballot(n > FEW);
        while (many) {
            const int l = __ffsll((unsigned long long)many) - 1;
            many &= many - 1ull;
            const uint32_t m = (uint32_t)__shfl((int)n, l), first = (uint32_t)__shfl((int)at, l), dr = (uint32_t)__shfl((int)draw, l);
            for (uint32_t j = lane; j < m; j += 64u) list[first + j] = Chunk{dr, j * 64u};
        }
    }
}

/* A pass of many small draws: EIGHT LANES per draw build what the draw's vertex waves would each build again (a draw of
 * 162 vertices has three of them, and the matrices are most of such a wave's instructions) -- the two matrices and the
 * verdict on the draw's box.  Lane r of a draw's eight computes elements r and r + 8 of the clip matrix and one (lane 0: two)
 * of the nine elements of model * transform that the normal matrix is made of, every load of a lane requested at once; the
 * eight exchange their elements, lane r takes corner r of the box, and the verdicts are combined over the eight.  The
 * chunks of the visible draws then go to two dense lists, which the vertex and setup kernels walk: a culled draw costs
 * them nothing (a device that renders one band of a frame culls most of the pass). */
constexpr uint32_t DRAW_LANES = 8;
constexpr uint32_t DRAWS_PER_BLOCK = DRAWS_THREADS / DRAW_LANES;

/* element k of A * transform with every operand requested before the first is used (the chain of matrix_element) */
__device__ __forceinline__ float matrix_element_at_once(const UniformPatch& up, const DrawStatic& d, int shader, const float* a, uint32_t k)
{
    const uint32_t row = k & 3u, col = k >> 2;
    if (shader == 0) return uniform_read_both(up, a + k);
    const float a0 = uniform_read_both(up, a + 0 * 4 + row), a1 = uniform_read_both(up, a + 1 * 4 + row),
                a2 = uniform_read_both(up, a + 2 * 4 + row), a3 = uniform_read_both(up, a + 3 * 4 + row);
    const float* b = d.push + 4 + col * 4;
    const float b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
    float acc = a0 * b0;
    acc = fma32(a1, b1, acc);
    acc = fma32(a2, b2, acc);
    acc = fma32(a3, b3, acc);
    return acc;
}

__global__ __launch_bounds__(DRAWS_THREADS) void k_draws(VertexArgs args)
{
    const FrameParams& fp = args.fp;
    const uint32_t r = threadIdx.x & (DRAW_LANES - 1u);
    const uint32_t draw = blockIdx.x * DRAWS_PER_BLOCK + threadIdx.x / DRAW_LANES;
    /* the pair of lengths the previous frame counted into is the next frame's: that frame's kernels have all finished */
    if (fp.compact && blockIdx.x == 0u && threadIdx.x < 2u) fp.counters[visible_length(fp.list_parity ^ 1u, threadIdx.x)] = 0u;
    const bool live = draw < fp.n_draws;
    const DrawStatic& d = fp.draws[live ? draw : 0u]; /* the lanes of a draw past the end go along (the exchanges want every lane) */
    const int shader = d.shader;
    const float* ubo = d.ubo;
    /* the nine elements of model * transform that matter, in the order the lanes take them: lane r has element
     * r + r / 3 (0 1 2 4 5 6 8 9), lane 0 element 10 as well */
    const uint32_t mine_of_model = r + r / 3u;
    const float c_lo = matrix_element_at_once(args.up, d, shader, ubo + 64, r);
    const float c_hi = matrix_element_at_once(args.up, d, shader, ubo + 64, r + 8u);
    float t_a = 0.0f, t_b = 0.0f;
    if (shader != 0) {
        t_a = matrix_element_at_once(args.up, d, shader, ubo + 48, mine_of_model);
        if (r == 0u) t_b = matrix_element_at_once(args.up, d, shader, ubo + 48, 10u);
    }
    float m[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) { m[k] = __shfl(c_lo, k, DRAW_LANES); m[k + 8] = __shfl(c_hi, k, DRAW_LANES); }
    uint32_t beyond = corner_beyond(fp, d, m, r);
    beyond &= (uint32_t)__shfl_xor((int)beyond, 1, DRAW_LANES);
    beyond &= (uint32_t)__shfl_xor((int)beyond, 2, DRAW_LANES);
    beyond &= (uint32_t)__shfl_xor((int)beyond, 4, DRAW_LANES);
    bool outside = !live || d.vert_count == 0u;
    if (d.has_bounds != 0u && fp.cull_draws != 0u) outside = outside || beyond != 0u;
    float mt[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) mt[k] = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) mt[j + j / 3] = __shfl(t_a, j, DRAW_LANES);
    mt[10] = __shfl(t_b, 0, DRAW_LANES);
    float nm[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) nm[i] = 0.0f;
    if (shader != 0) normal_matrix(mt, nm);
    uint32_t n_v = 0u, n_t = 0u;
    if (live) {
        DrawDerived& out = fp.derived[draw];
        if (r == 0u) out.visible = outside ? 0u : 1u;
        if (!outside) {
            out.clip_m[r] = c_lo; out.clip_m[r + 8u] = c_hi;
            float mine = nm[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) if (r == (uint32_t)i) mine = nm[i];
            out.nrm_m[r] = mine;
            if (r == 0u) {
                out.nrm_m[8] = nm[8];
                n_v = (d.vert_count + 63u) / 64u;
                n_t = (d.tri_count + 63u) / 64u;
            }
        }
    }
    if (!fp.compact) return;
    append_chunks(fp, draw, n_v, n_t);
}

/* The same for a pass of few draws with many vertices each (no lists): one WAVE per draw, its lanes sharing out the
 * matrices' elements and the box's corners -- a tenth of the latency of one lane doing all of it, which is what a pass of
 * one draw waits for before its vertex kernel starts. */
__global__ __launch_bounds__(256) void k_draws_by_wave(VertexArgs args)
{
    const FrameParams& fp = args.fp;
    const uint32_t draw = wave_uniform(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (draw >= fp.n_draws) return;
    const DrawStatic& d = fp.draws[draw];
    float m[16], nm[9];
    const float mine = wave_matrix_element(args.up, d, d.shader, d.ubo);
    wave_clip_matrix(mine, m);
    bool outside = d.vert_count == 0u;
    if (d.has_bounds != 0u && fp.cull_draws != 0u) outside = outside || draw_is_outside(fp, d, m);
    DrawDerived& out = fp.derived[draw];
    if (lane_id() == 0) out.visible = outside ? 0u : 1u;
    if (outside) return;
    wave_normal_matrix(d.shader, mine, nm);
    if (lane_id() < 16u) out.clip_m[lane_id()] = mine;
#pragma unroll
    for (int i = 0; i < 9; ++i) if (lane_id() == (uint32_t)i) out.nrm_m[i] = nm[i];
}

/* Tables that no kernel of the frame writes while this one runs, read through the constant address space: a wave-uniform
 * address there is always a scalar load, also after the kernel's own stores (a kernel that walks a list of chunks has
 * stored the previous chunk's vertices when it reads the next chunk's draw) */
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* constant_table(const T* p)
{
    return (const __attribute__((address_space(4))) T*)reinterpret_cast<uintptr_t>(p);
}

/* One wave, one chunk of 64 vertices of one draw.  PREPARED: k_draws has run. */
template <bool PREPARED, bool ZERO_COUNTERS>
__device__ __forceinline__ void vertex_chunk(const VertexArgs& args, Chunk c)
{
    const FrameParams& fp = args.fp;
    const UniformPatch& up = args.up;
    /* The chunk is the same for the whole wave, and said so; and the draw's record is copied here, before the wave's
     * first store.  Both are needed for the record to be read with scalar loads, all at once: a field read after a
     * store that might have changed it is read with a vector load of its own, and the kernel began with six of those in
     * a row, each waiting for the one before. */
    c.draw = wave_uniform(c.draw); c.first = wave_uniform(c.first);
    const DrawStatic& dm = fp.draws[c.draw]; /* the record in memory: what is indexed per lane (push constants, box) */
    const auto* dk = constant_table(fp.draws) + c.draw;
    struct {
        const uint8_t* vb; const float* ubo; uint32_t stride; int32_t off_pos, off_nrm, off_uv; int32_t shader;
        uint32_t vert_count, svtx_base, has_bounds;
    } const d = {dk->vb, dk->ubo, dk->stride, dk->off_pos, dk->off_nrm, dk->off_uv, dk->shader, dk->vert_count, dk->svtx_base, dk->has_bounds};
    uint32_t visible = 1u;
    if constexpr (PREPARED) visible = constant_table(fp.derived)[c.draw].visible;
    /* the list counters of the previous frame were last read by its tile kernel; binning starts after this kernel
     * (the first three threads of the grid belong to chunk 0, which every launch has) */
    if constexpr (ZERO_COUNTERS) if (blockIdx.x == 0 && threadIdx.x < CTR_FRAME_COUNT) fp.counters[threadIdx.x] = 0;
    float m[16], nm[9], mine = 0.0f;
    if (!visible) return;
    const bool cullable = !PREPARED && d.has_bounds != 0u && fp.cull_draws != 0u;
    /* A draw with a box is tested before anything of it is fetched: a culled one costs its waves the matrices
     * and nothing else, and setup skips its primitives.  Not so the one draw of a one-draw pass, which is what
     * the pass is there to render: its waves request their vertices first and test under that latency. */
    const bool test_first = cullable && !fp.direct_index;
    if (test_first) {
        mine = wave_matrix_element(up, dm, d.shader, d.ubo);
        wave_clip_matrix(mine, m);
        const bool outside = draw_is_outside(fp, dm, m);
        if (c.first == 0 && lane_id() == 0) fp.derived[c.draw].visible = outside ? 0u : 1u;
        if (outside) return;
    }
    /* the vertex's attributes are requested first: they do not depend on the matrices, whose
     * construction (loads, multiply-adds, broadcasts) then runs under their latency */
    const uint32_t v = c.first + lane_id();
    const bool live = v < d.vert_count;
    float pos[3] = {0.0f, 0.0f, 0.0f}, nrm[3] = {0.0f, 0.0f, 0.0f}, uv[2] = {0.0f, 0.0f};
    /* A record in the reference's full layout -- position, normal, uv interleaved, 32 bytes (obj_loader.cpp:98-109,
     * 181-196) -- is fetched as two 16-byte loads, each of which uses half of every cache line it touches; any other
     * layout word by word.  (The wave's 64 records fetched as 16-byte pieces in lane order and dealt to the lanes
     * through LDS use whole lines per instruction, and were slower: the hop through LDS costs more than the second
     * touch of each line, which the vector cache serves.) */
    const bool full_layout = d.stride == 32u && d.off_pos == 0 && d.off_nrm == 12 && d.off_uv == 24;
    if (live && full_layout) {
        typedef float Half __attribute__((ext_vector_type(4), aligned(4)));
        typedef const __attribute__((address_space(1))) Half* GlobalHalf;
        const uint8_t* rec = d.vb + (size_t)v * 32u;
        const Half a = *(GlobalHalf)rec, b = *(GlobalHalf)(rec + 16);
        pos[0] = a.x; pos[1] = a.y; pos[2] = a.z;
        nrm[0] = a.w; nrm[1] = b.x; nrm[2] = b.y;
        uv[0] = b.z; uv[1] = b.w;
    } else if (live) {
        const uint8_t* rec = d.vb + (size_t)v * d.stride;
        const float* p = reinterpret_cast<const float*>(rec + d.off_pos);
        pos[0] = load_f32(p); pos[1] = load_f32(p + 1); pos[2] = load_f32(p + 2);
        if (d.off_uv >= 0) {
            const float* q = reinterpret_cast<const float*>(rec + d.off_uv);
            uv[0] = load_f32(q); uv[1] = load_f32(q + 1);
        }
        if (d.shader != 0 && d.off_nrm >= 0) {
            const float* q = reinterpret_cast<const float*>(rec + d.off_nrm);
            nrm[0] = load_f32(q); nrm[1] = load_f32(q + 1); nrm[2] = load_f32(q + 2);
        }
    }
    if constexpr (PREPARED) {
        /* wave-uniform addresses: scalar loads */
        const auto* dd = constant_table(fp.derived) + c.draw;
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = dd->clip_m[i];
#pragma unroll
        for (int i = 0; i < 9; ++i) nm[i] = dd->nrm_m[i];
    } else {
        if (!test_first) {
            mine = wave_matrix_element(up, dm, d.shader, d.ubo);
            wave_clip_matrix(mine, m);
            const bool outside = cullable && draw_is_outside(fp, dm, m);
            if (c.first == 0 && lane_id() == 0) fp.derived[c.draw].visible = outside ? 0u : 1u;
            if (outside) return;
        }
        wave_normal_matrix(d.shader, mine, nm);
        if (c.first == 0 && lane_id() < 16u) fp.derived[c.draw].clip_m[lane_id()] = mine;
    }
    if (!live) return;
    /* vertex shader position: M * vec4(pos,1), then the reference's clip-space y = 1 - y
     * (test_ktx.vert:28-29, push_constants.vert:28-29, dummy.vert.glsl:22-23) */
    float clip[4];
    mat4_point(m, pos[0], pos[1], pos[2], clip);
    clip[1] = 1.0f - clip[1];
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
    /* the rest of the vertex shader: v_normal (test_ktx.vert:30, push_constants.vert:30) and v_uv */
    SVary vo;
    vo.light = 0.0f; vo.u = uv[0];
    vo.v = d.shader == 0 ? 1.0f - uv[1] : uv[1]; /* dummy.vert.glsl:25 flips v */
    if (d.shader != 0) {
        const float len2 = fma32(nrm[2], nrm[2], fma32(nrm[1], nrm[1], nrm[0] * nrm[0]));
        const float len = sqrtf(len2);
        const float nn[3] = {nrm[0] / len, nrm[1] / len, nrm[2] / len};
        float tn[3];
        mat3_vec(nm, nn, tn);
        /* the light term of the fragment shaders, at the vertex (see SVary) */
        const float lx = 0.0f, ly = 0.70710678118654752f, lz = -0.70710678118654752f;
        vo.light = fma32(tn[2], lz, fma32(tn[1], ly, tn[0] * lx));
    }
    fp.svary[d.svtx_base + v] = vo;
}

/* how many workgroups walk a dense list of chunks: enough to fill every CU at full occupancy, no more */
constexpr uint32_t LIST_WALK_BLOCKS = 256u * 8u;

/* One wave per chunk of 64 vertices; COMPACT: the waves of a fixed grid share out the list of visible chunks. */
template <bool PREPARED, bool COMPACT>
__global__ __launch_bounds__(256) void k_vertex(VertexArgs args)
{
    const FrameParams& fp = args.fp;
    RND_KERNEL_BEGIN(fp, 0);
    const uint32_t wave = wave_uniform(blockIdx.x * 4u + (threadIdx.x >> 6));
    if constexpr (COMPACT) {
        /* the list counters of the previous frame were last read by its tile kernel; binning starts after this kernel */
        if (blockIdx.x == 0 && threadIdx.x < CTR_FRAME_COUNT) fp.counters[threadIdx.x] = 0;
        const uint32_t listed = min(constant_table(fp.counters)[visible_length(fp.list_parity, 0u)], fp.n_vchunks);
        const auto* list = constant_table(fp.visible_vchunks);
        for (uint32_t i = wave; i < listed; i += gridDim.x * 4u) {
            const Chunk c = {list[i].draw, list[i].first};
            vertex_chunk<true, false>(args, c);
        }
    } else {
        if (wave >= fp.n_vchunks) return;
        Chunk c = {0u, wave * 64u}; /* what the chunk table of a one-draw pass holds */
        if (!fp.direct_index) c = fp.vchunks[wave];
        vertex_chunk<PREPARED, true>(args, c);
    }
    RND_KERNEL_END(fp, 0u);
}

/* ------------------------------------------------------------------ setup: cull, orient, bin (one pass) */

__device__ __forceinline__ void flag_overflow(const FrameParams& fp, uint32_t bit) { atomicOr(&fp.counters[CTR_OVERFLOW], bit); }

/* append to the pass-wide list; one lane at a time */
__device__ __forceinline__ void push_large(const FrameParams& fp, const TileEntry& e, uint32_t box)
{
    uint32_t pos = atomicAdd(&fp.counters[CTR_LARGE], 1u);
    if (pos >= fp.large_capacity) { flag_overflow(fp, OVF_LARGE); return; }
    LargeEntry le;
    le.e = e; le.box = box; le.pad[0] = le.pad[1] = le.pad[2] = 0;
    fp.large_list[pos] = le;
}

/* store an entry at a position of a tile's list, or spill it to the pass-wide list */
__device__ __forceinline__ void put_entry(const FrameParams& fp, int tile, uint32_t pos, const TileEntry& e)
{
    if (pos < fp.tile_capacity) {
        fp.tile_lists[(size_t)tile * fp.tile_stride + pos] = e;
    } else {
        const int tx = tile % (int)fp.tiles_x, ty = tile / (int)fp.tiles_x;
        push_large(fp, e, pack_bbox(tx, ty, tx, ty));
    }
}

/* Appends every lane's entry to the lists of the tiles in its box.
 *
 * Tile 0 of every box (for small triangles the only one) is handled for the whole wave at
 * once: lanes that hit the same tile are grouped with ballots (no memory traffic), then the
 * leader of every group adds the group's size to its tile's counter -- all leaders in ONE
 * vector atomic, so the wave pays one atomic round trip -- and each lane takes base + its
 * rank inside the group (popcount of the lower lanes of the group).  Tiles 1..3 of a box
 * issue their own atomics back to back before any result is used; boxes of more than 4
 * tiles finish in a plain loop, more than MAX_TILES_PER_PRIM go to the pass-wide list. */
__device__ __forceinline__ void bin_wave(const FrameParams& fp, bool valid, const TileEntry& e, uint32_t box)
{
    const uint32_t lane = lane_id();
    const int tx0 = box & 255u, ty0 = (box >> 8) & 255u, tx1 = (box >> 16) & 255u, ty1 = box >> 24;
    const int w = tx1 - tx0 + 1, h = ty1 - ty0 + 1;
    int n = valid ? w * h : 0;
    const bool large = n > MAX_TILES_PER_PRIM;
    if (large) n = 0;
    /* tile (row, col) of the box, if this device renders it (one device renders every tile: no remainder to take) */
    auto tile_at = [&](int row, int col) -> int {
        const int t = (ty0 + row) * (int)fp.tiles_x + tx0 + col;
        if (fp.shard_count != 1u && (uint32_t)t % fp.shard_count != fp.shard_index) return -1;
        return t;
    };
    auto tile_of = [&](int k) -> int {
        if (k >= n) return -1;
        return tile_at(k / w, k % w);
    };
    /* the box's first four tiles in row order without the division: their row is how many multiples of the
     * width they have passed */
    auto tile_of_first = [&](int k) -> int {
        if (k >= n) return -1;
        const int row = (k >= w ? 1 : 0) + (k >= 2 * w ? 1 : 0) + (k >= 3 * w ? 1 : 0);
        return tile_at(row, k - row * w);
    };
    /* group the wave's first tiles */
    const int t0 = tile_of_first(0);
    uint64_t pending = __ballot(t0 >= 0);
    int leader = -1;
    uint32_t rank = 0, group = 0;
    while (pending) {
        const int l = __ffsll((unsigned long long)pending) - 1;
        const int lt = __shfl(t0, l);
        const uint64_t same = __ballot(t0 == lt);
        if (t0 == lt) {
            leader = l;
            rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            group = (uint32_t)__popcll(same);
        }
        pending &= ~same;
    }
    /* issue every atomic before using any result */
    uint32_t base0 = 0, p1 = 0, p2 = 0, p3 = 0;
    /* a wave of a fine mesh often has no box of more than one tile: the second to fourth tiles are not even worked out then */
    int t1 = -1, t2 = -1, t3 = -1;
    const bool several = __ballot(n > 1) != 0ull;
    if (several) { t1 = tile_of_first(1); t2 = tile_of_first(2); t3 = tile_of_first(3); }
    if ((int)lane == leader) base0 = atomicAdd(&fp.tile_count[(size_t)t0 * TILE_COUNT_STRIDE], group);
    if (several) {
        if (t1 >= 0) p1 = atomicAdd(&fp.tile_count[(size_t)t1 * TILE_COUNT_STRIDE], 1u);
        if (t2 >= 0) p2 = atomicAdd(&fp.tile_count[(size_t)t2 * TILE_COUNT_STRIDE], 1u);
        if (t3 >= 0) p3 = atomicAdd(&fp.tile_count[(size_t)t3 * TILE_COUNT_STRIDE], 1u);
    }
    base0 = (uint32_t)__shfl((int)base0, leader < 0 ? (int)lane : leader);
    if (t0 >= 0) put_entry(fp, t0, base0 + rank, e);
    if (several) {
        if (t1 >= 0) put_entry(fp, t1, p1, e);
        if (t2 >= 0) put_entry(fp, t2, p2, e);
        if (t3 >= 0) put_entry(fp, t3, p3, e);
    }
    for (int k = 4; k < n; ++k) {
        const int t = tile_of(k);
        if (t >= 0) put_entry(fp, t, atomicAdd(&fp.tile_count[(size_t)t * TILE_COUNT_STRIDE], 1u), e);
    }
    if (large) push_large(fp, e, box);
}

/* same for a single lane (the clip kernel) */
__device__ __forceinline__ void bin_one(const FrameParams& fp, const TileEntry& e, uint32_t box)
{
    const int tx0 = box & 255u, ty0 = (box >> 8) & 255u, tx1 = (box >> 16) & 255u, ty1 = box >> 24;
    if ((tx1 - tx0 + 1) * (ty1 - ty0 + 1) > MAX_TILES_PER_PRIM) { push_large(fp, e, box); return; }
    for (int ty = ty0; ty <= ty1; ++ty) {
        for (int tx = tx0; tx <= tx1; ++tx) {
            const int t = ty * (int)fp.tiles_x + tx;
            if ((uint32_t)t % fp.shard_count != fp.shard_index) continue;
            const uint32_t pos = atomicAdd(&fp.tile_count[(size_t)t * TILE_COUNT_STRIDE], 1u);
            if (pos < fp.tile_capacity) fp.tile_lists[(size_t)t * fp.tile_stride + pos] = e;
            else push_large(fp, e, pack_bbox(tx, ty, tx, ty));
        }
    }
}

/* Clips one primitive against near, far and the guard band, writes its fan of triangles and
 * bins them.  One lane per primitive. */
__device__ __forceinline__ void clip_primitive(const FrameParams& fp, const DrawStatic& d, uint32_t draw, uint32_t slot,
                                               uint32_t i0, uint32_t i1, uint32_t i2, const ClipStore& st)
{
    int buf = 0;
    const int corners = build_clip_fan(d, fp.derived[draw].clip_m, i0, i1, i2, st, buf);
    if (corners < 3) return;
    const uint32_t count = (uint32_t)corners - 2u;
    const uint32_t first = atomicAdd(&fp.counters[CTR_CLIP_TRIS], count);
    if (first + count > fp.clip_tri_capacity) { flag_overflow(fp, OVF_CLIP_TRIS); return; }
    PrimRec pr = {first, 0u, 0u, draw | FLAG_CLIPPED};
    fp.prim_rec[slot] = pr;
#pragma unroll 1
    for (int k = 1; k + 1 < corners; ++k) {
        RTri t; float bary[3][3];
        ClipTri ct;
        const bool ok = fan_triangle(fp, st, buf, k, d.cull_mode, d.front_face, t, bary);
        int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
        const bool vis = ok && pixel_box(fp, t, x0, y0, x1, y1);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            ct.x[q] = t.x[q]; ct.y[q] = t.y[q]; ct.z[q] = t.z[q]; ct.iw[q] = t.iw[q];
#pragma unroll
            for (int j = 0; j < 3; ++j) ct.bary[q * 3 + j] = ok ? bary[q][j] : 0.0f;
        }
        ct.v[0] = d.svtx_base + i0; ct.v[1] = d.svtx_base + i1; ct.v[2] = d.svtx_base + i2;
        ct.draw = draw;
        ct.valid = vis ? 1u : 0u;
#pragma unroll
        for (int j = 0; j < 5; ++j) ct.pad[j] = 0;
        const uint32_t idx = first + (uint32_t)(k - 1);
        fp.clip_tris[idx] = ct;
        if (vis) {
            TileEntry e = {idx, 0u, 0u, slot | ((uint32_t)(k - 1) << ENTRY_SUB_SHIFT) | FLAG_CLIPPED};
            bin_one(fp, e, pack_bbox(x0 >> TILE_SHIFT, tile_row(fp, y0), x1 >> TILE_SHIFT, tile_row(fp, y1)));
        }
    }
}

/* One wave, one chunk of 64 primitives of one draw (chunk number `chunk` of the pass: its primitives own slots
 * 64 * chunk ..): fetch, classify, cull and orient, bin; then clip the few that need it. */
__device__ __forceinline__ void setup_chunk(const FrameParams& fp, uint32_t chunk, Chunk c, float (*clip_lds)[CLIP_LDS_FLOATS])
{
    const DrawStatic& d = fp.draws[c.draw];
    const uint32_t lane = lane_id();
    const uint32_t tri = c.first + lane;
    const uint32_t slot = chunk * 64u + lane;
    const uint32_t svtx_base = d.svtx_base; /* read before the record below is stored: after it the compiler reads the draw's record again, per lane */
    bool valid = false, needs_clip = false;
    TileEntry e = {0, 0, 0, slot};
    uint32_t box = BBOX_CULLED;
    PrimFetch pf;
    pf.idx[0] = pf.idx[1] = pf.idx[2] = 0;
    if (tri < d.tri_count && fetch_prim(fp, d, tri, pf)) {
        const int cls = RND_PROBE(fp, 4u) ? 0 : classify_prim(pf);
        if (RND_PROBE(fp, 4u)) { if (pf.oc_any == 0xFFFFu) fp.prim_rec[slot].v0 = 1; }
        if (cls == 2) {
            needs_clip = true;
        } else if (cls == 1) {
            RTri t;
            tri_from_fetch(pf, t);
            int x0, y0, x1, y1;
            if (setup_tri(t, d.cull_mode, d.front_face) && pixel_box(fp, t, x0, y0, x1, y1)) {
                valid = true;
                box = pack_bbox(x0 >> TILE_SHIFT, tile_row(fp, y0), x1 >> TILE_SHIFT, tile_row(fp, y1));
                const uint32_t v0 = svtx_base + pf.idx[0], v1 = svtx_base + pf.idx[1], v2 = svtx_base + pf.idx[2];
                /* what shading needs, in the primitive's own corner order */
                PrimRec pr = {v0, v1, v2, c.draw};
                if (!fp.direct_index) fp.prim_rec[slot] = pr;
                /* what rasterisation needs: corners ordered for positive area */
                e.a = svtx_base + pf.idx[t.corner[0]];
                e.b = svtx_base + pf.idx[t.corner[1]];
                e.c = svtx_base + pf.idx[t.corner[2]];
            }
        }
    }
    if (!RND_PROBE(fp, 8u)) bin_wave(fp, valid, e, box);

    /* Primitives that cross the near plane, the far plane or the guard band (rare): the wave clips
     * them CLIP_LANES at a time, lane j taking the j-th of them, with the polygons in LDS. */
    uint64_t cm = __ballot(needs_clip);
    if (!cm) return;
    ClipStore st;
    st.base = clip_lds[threadIdx.x >> 6] + (lane & (CLIP_LANES - 1));
    while (cm) {
        int src = -1;
#pragma unroll
        for (int j = 0; j < CLIP_LANES; ++j) {
            const int l = cm ? __ffsll((unsigned long long)cm) - 1 : -1;
            if ((int)lane == j) src = l;
            cm &= cm - 1ull; /* no-op once empty */
        }
        const int from = src < 0 ? (int)lane : src;
        const uint32_t i0 = (uint32_t)__shfl((int)pf.idx[0], from), i1 = (uint32_t)__shfl((int)pf.idx[1], from),
                       i2 = (uint32_t)__shfl((int)pf.idx[2], from);
        if (src >= 0 && lane < (uint32_t)CLIP_LANES) clip_primitive(fp, d, c.draw, chunk * 64u + (uint32_t)src, i0, i1, i2, st);
    }
}

/* One wave per chunk of 64 primitives; COMPACT: the waves of a fixed grid share out the list of visible chunks. */
template <bool COMPACT>
__global__ __launch_bounds__(256, 8) void k_setup(FrameParams fp)
{
    __shared__ float clip_lds[4][CLIP_LDS_FLOATS];
    RND_KERNEL_BEGIN(fp, 1);
    if constexpr (COMPACT) {
        const uint32_t listed = min(constant_table(fp.counters)[visible_length(fp.list_parity, 1u)], fp.n_tchunks);
        const auto* list = constant_table(fp.visible_tchunks);
        for (uint32_t i = wave_uniform(blockIdx.x * 4u + (threadIdx.x >> 6)); i < listed; i += gridDim.x * 4u) {
            const Chunk c = {wave_uniform(list[i].draw), wave_uniform(list[i].first)};
            setup_chunk(fp, constant_table(fp.draws)[c.draw].first_tchunk + c.first / 64u, c, clip_lds);
        }
    } else {
        /* Workgroups are dealt round-robin over the 8 XCDs.  In a pass of one draw -- a mesh, whose neighbouring chunks of
         * primitives share vertices and append to the same tile lists -- each XCD takes a contiguous eighth of the chunks, so
         * that neighbours meet in one L2 (the grid is then a multiple of 8 workgroups).  A pass of many small draws keeps
         * the plain order: measured slower there, its draws being scattered over the screen anyway. */
        uint32_t block = blockIdx.x;
        if (fp.direct_index) block = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
        const uint32_t chunk = wave_uniform(block * 4u + (threadIdx.x >> 6));
        if (chunk >= fp.n_tchunks) return;
        /* a pass without a chunk table is one draw whose chunks are its triangles in order */
        const Chunk c = fp.tchunks ? fp.tchunks[chunk] : Chunk{0u, chunk * 64u};
        if (!fp.derived[c.draw].visible) return; /* the draw's box lies outside what this device renders */
        setup_chunk(fp, chunk, c, clip_lds);
    }
    RND_KERNEL_END(fp, (fp.n_vchunks + 3u) / 4u);
}

/* ------------------------------------------------------------------ tile kernel */

__constant__ int c_sample_x4[4] = {96, 224, 32, 160}; /* (0.375,0.125) (0.875,0.375) (0.125,0.625) (0.625,0.875) */
__constant__ int c_sample_y4[4] = {32, 96, 160, 224};

struct TileCtx {
    int px0, py0;           /* first pixel of the tile */
    int rx0, ry0, rx1, ry1; /* inclusive pixel rectangle: tile ∩ clamp */
    unsigned long long* keys;
    signed char* wave_row;  /* 64 bytes of LDS of this wave's own */
};

/* A triangle as the rasteriser uses it: corners ordered for positive area */
struct RasterTri {
    int32_t x[3], y[3];
    float z[3];
    uint32_t payload_lo; /* bits 0..2 fan index, bit 3 clipped; the order word is added per depth mode */
    uint32_t slot;
};

/* Per-sample test + depth + LDS min for one pixel.  I is int32_t when every edge product
 * fits 32 bits (extent below 64 pixels), else int64_t; both give the same integers. */
__device__ __forceinline__ int32_t edge_mul(int32_t a, int32_t b) { return __mul24(a, b); } /* operands below 2^15: full-rate multiply */
__device__ __forceinline__ int64_t edge_mul(int64_t a, int64_t b) { return a * b; }

template <typename I, int S>
struct EdgeSetup {
    I ax0, ay0, ax1, ay1, ax2, ay2;
    I dx0, dy0, dx1, dy1, dx2, dy2;
    I bias0, bias1, bias2;
    float z0, gz1, gz2;     /* depth at corner 0 and its change per unit of edge functions 1 and 2 */
    uint32_t pay_lo, pay_hi; /* the low 40 bits of a fragment's word: order and fan index */

    __device__ __forceinline__ void init(const FrameParams& fp, const RasterTri& t)
    {
        /* edge k is opposite corner k: e0 = v1->v2, e1 = v2->v0, e2 = v0->v1 */
        ax0 = t.x[1]; ay0 = t.y[1]; ax1 = t.x[2]; ay1 = t.y[2]; ax2 = t.x[0]; ay2 = t.y[0];
        dx0 = (I)t.x[2] - (I)t.x[1]; dy0 = (I)t.y[2] - (I)t.y[1];
        dx1 = (I)t.x[0] - (I)t.x[2]; dy1 = (I)t.y[0] - (I)t.y[2];
        dx2 = (I)t.x[1] - (I)t.x[0]; dy2 = (I)t.y[1] - (I)t.y[0];
        /* a point on an edge is inside only if the edge owns it: a bias of -1 otherwise */
        bias0 = owns_edge(t.x[1], t.y[1], t.x[2], t.y[2]) ? 0 : 1;
        bias1 = owns_edge(t.x[2], t.y[2], t.x[0], t.y[0]) ? 0 : 1;
        bias2 = owns_edge(t.x[0], t.y[0], t.x[1], t.y[1]) ? 0 : 1;
        const I area2 = edge_mul(dx2, (I)t.y[2] - (I)t.y[0]) - edge_mul(dy2, (I)t.x[2] - (I)t.x[0]);
        const float inv_area = recip32_of_area((float)area2); /* setup lists no triangle of zero area */
        z0 = t.z[0]; gz1 = (t.z[1] - t.z[0]) * inv_area; gz2 = (t.z[2] - t.z[0]) * inv_area;
        if (!fp.depth_test) z0 = gz1 = gz2 = 0.0f; /* no depth test: every fragment's depth is 0 and the order decides */
        /* Equal depths: under LESS the first fragment submitted stays, otherwise (LESS_OR_EQUAL, no depth test) the last one
         * wins, and the minimum over the words must say the same -- between primitives (their slots) and between the fan
         * triangles of one clipped primitive, which overlap where snapping has folded a sliver: the fan index counts the
         * same way as the slot (complemented when the last is to win) */
        const uint32_t ord = fp.depth_less ? t.slot : ~t.slot;
        pay_lo = (ord << 8) | (fp.depth_less ? t.payload_lo : (t.payload_lo ^ 7u));
        pay_hi = ord >> 24;
    }

    /* depth of a covered sample from its edge values, then the LDS minimum.  No read first:
     * the atomic needs no result, so the lane does not wait for the LDS */
    __device__ __forceinline__ void emit(const TileCtx& tc, uint32_t word_index, I e1, I e2) const
    {
        const float z = fma32((float)e2, gz2, fma32((float)e1, gz1, z0));
        const uint32_t hi = (depth24(z) << 8) | pay_hi;
        atomicMin(&tc.keys[word_index], ((unsigned long long)hi << 32) | pay_lo);
    }

    /* the same with the word's byte offset in the tile's array, which the walks step instead of an index */
    __device__ __forceinline__ void emit_at(const TileCtx& tc, uint32_t byte_offset, I e1, I e2) const
    {
        const float z = fma32((float)e2, gz2, fma32((float)e1, gz1, z0));
        const uint32_t hi = (depth24(z) << 8) | pay_hi;
        atomicMin(reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(tc.keys) + byte_offset), ((unsigned long long)hi << 32) | pay_lo);
    }

    __device__ __forceinline__ void pixel(const TileCtx& tc, int x, int y) const
    {
        const uint32_t pix = (uint32_t)((y - tc.py0) * TILE_W + (x - tc.px0));
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const I px = (I)(x * 256 + (S == 1 ? 128 : c_sample_x4[s]));
            const I py = (I)(y * 256 + (S == 1 ? 128 : c_sample_y4[s]));
            const I e0 = edge_mul(dx0, py - ay0) - edge_mul(dy0, px - ax0);
            const I e1 = edge_mul(dx1, py - ay1) - edge_mul(dy1, px - ax1);
            const I e2 = edge_mul(dx2, py - ay2) - edge_mul(dy2, px - ax2);
            if (((e0 - bias0) | (e1 - bias1) | (e2 - bias2)) < 0) continue;
            emit(tc, pix * S + s, e1, e2);
        }
    }
};

/* What a single-sample walk over part of a small triangle's box needs, whichever part it is: the three edge
 * values at the box's first pixel (ownership bias folded in: covered <=> none negative), their steps per pixel
 * right and down, the depth plane and the fragment word's low bits.  Taken once per triangle by the lane that
 * holds it; a lane that is dealt a chunk of the box fetches these sixteen words from that lane instead of the
 * triangle's corners, and starts its walk with six multiply-adds.
 *
 * The walk goes over the box in strips of two rows, left to right, and takes both pixels of a column in one step:
 * what a step costs beyond its two tests -- moving on, the end of a strip -- is paid once for two pixels. */
constexpr int WALK_ROWS = 2;

__device__ __forceinline__ int walk_steps(int w, int h) { return __mul24(w, (h + WALK_ROWS - 1) / WALK_ROWS); }

struct WalkSetup {
    int32_t e0, e1, e2;     /* at the box's first pixel */
    int32_t sx0, sx1, sx2;  /* one pixel right */
    int32_t sy0, sy1, sy2;  /* one pixel down */
    float z0, gz1, gz2;
    uint32_t pay_lo, pay_hi;
    uint32_t box;           /* bits 0..4 first column in the tile, 5..10 first row, 11..15 width - 1, 16..21 height - 1 (a tile has up to 40 rows), 26/27 bias of edges 1/2 */

    __device__ __forceinline__ void init(const FrameParams& fp, const TileCtx& tc, const RasterTri& t, int x0, int y0, int w, int h)
    {
        EdgeSetup<int32_t, 1> es;
        es.init(fp, t);
        const int32_t px = x0 * 256 + 128, py = y0 * 256 + 128;
        e0 = edge_mul(es.dx0, py - es.ay0) - edge_mul(es.dy0, px - es.ax0) - es.bias0;
        e1 = edge_mul(es.dx1, py - es.ay1) - edge_mul(es.dy1, px - es.ax1) - es.bias1;
        e2 = edge_mul(es.dx2, py - es.ay2) - edge_mul(es.dy2, px - es.ax2) - es.bias2;
        sx0 = -(es.dy0 << 8); sx1 = -(es.dy1 << 8); sx2 = -(es.dy2 << 8);
        sy0 = es.dx0 << 8; sy1 = es.dx1 << 8; sy2 = es.dx2 << 8;
        z0 = es.z0; gz1 = es.gz1; gz2 = es.gz2; pay_lo = es.pay_lo; pay_hi = es.pay_hi;
        box = (uint32_t)(x0 - tc.px0) | ((uint32_t)(y0 - tc.py0) << 5) | ((uint32_t)(w - 1) << 11) | ((uint32_t)(h - 1) << 16) |
              ((uint32_t)es.bias1 << 26) | ((uint32_t)es.bias2 << 27);
    }

    __device__ __forceinline__ WalkSetup from_lane(int o) const
    {
        WalkSetup u;
        u.e0 = __shfl(e0, o); u.e1 = __shfl(e1, o); u.e2 = __shfl(e2, o);
        u.sx0 = __shfl(sx0, o); u.sx1 = __shfl(sx1, o); u.sx2 = __shfl(sx2, o);
        u.sy0 = __shfl(sy0, o); u.sy1 = __shfl(sy1, o); u.sy2 = __shfl(sy2, o);
        u.z0 = __shfl(z0, o); u.gz1 = __shfl(gz1, o); u.gz2 = __shfl(gz2, o);
        u.pay_lo = (uint32_t)__shfl((int)pay_lo, o); u.pay_hi = (uint32_t)__shfl((int)pay_hi, o);
        u.box = (uint32_t)__shfl((int)box, o);
        return u;
    }

    __device__ __forceinline__ int steps() const { return walk_steps((int)((box >> 11) & 31u) + 1, (int)((box >> 16) & 63u) + 1); }

    /* steps k0 .. k1-1 of the box: step k is column k % width of strip k / width */
    __device__ __forceinline__ void walk(const TileCtx& tc, int k0, int k1) const
    {
        const int ow = (int)((box >> 11) & 31u) + 1, oh = (int)((box >> 16) & 63u) + 1;
        const int32_t b1 = (int32_t)((box >> 26) & 1u), b2 = (int32_t)((box >> 27) & 1u);
        /* strip and column of step k0 (k0 < 640, ow <= 32): the float quotient by the hardware's reciprocal
         * estimate, whose error is far below one, put right by the remainder */
        int strip = (int)((float)k0 * __builtin_amdgcn_rcpf((float)ow));
        int col = k0 - strip * ow;
        if (col < 0) { --strip; col += ow; } else if (col >= ow) { ++strip; col -= ow; }
        const int row = strip * WALK_ROWS;
        int32_t a0 = e0 + __mul24(col, sx0) + __mul24(row, sy0);
        int32_t a1 = e1 + __mul24(col, sx1) + __mul24(row, sy1);
        int32_t a2 = e2 + __mul24(col, sx2) + __mul24(row, sy2);
        /* from the last column of a strip to the first of the next: two down, ow - 1 back */
        const int32_t nl0 = WALK_ROWS * sy0 - __mul24(ow - 1, sx0), nl1 = WALK_ROWS * sy1 - __mul24(ow - 1, sx1), nl2 = WALK_ROWS * sy2 - __mul24(ow - 1, sx2);
        uint32_t pix8 = ((((box >> 5) & 63u) + (uint32_t)row) * TILE_W + (box & 31u) + (uint32_t)col) * 8u; /* byte offset of the pixel's word */
        /* a box of odd height has no second row in its last strip, whose steps are those from `single` on */
        const int single = (oh & 1) ? __mul24(ow, oh >> 1) : 0x7fffffff;
        EdgeSetup<int32_t, 1> es; /* only its depth plane is used */
        es.z0 = z0; es.gz1 = gz1; es.gz2 = gz2; es.pay_lo = pay_lo; es.pay_hi = pay_hi;
        const uint32_t next_strip8 = (uint32_t)(WALK_ROWS * TILE_W - ow + 1) * 8u;
        for (int i = k0; i < k1; ++i) {
            if ((a0 | a1 | a2) >= 0) es.emit_at(tc, pix8, a1 + b1, a2 + b2);
            const int32_t c0 = a0 + sy0, c1 = a1 + sy1, c2 = a2 + sy2;
            if ((c0 | c1 | c2) >= 0 && i < single) es.emit_at(tc, pix8 + TILE_W * 8u, c1 + b1, c2 + b2);
            const bool last = ++col == ow;
            a0 += last ? nl0 : sx0; a1 += last ? nl1 : sx1; a2 += last ? nl2 : sx2;
            pix8 += last ? next_strip8 : 8u;
            if (last) col = 0;
        }
    }
};

/* Multisample walk of pixels k0 .. k1-1 (row order) of a box whose first pixel is (x0, y0) and whose
 * width is w, starting at (row, col).  The edge values are stepped per pixel at the pixel's corner;
 * a sample's value is the corner's plus a constant per edge and sample (ownership bias folded in),
 * so a pixel costs adds only. */
template <int S>
__device__ __forceinline__ void walk_samples(const EdgeSetup<int32_t, S>& es, const TileCtx& tc, int x0, int y0, int w,
                                             int k0, int k1, int row, int col)
{
    const int32_t px = (x0 + col) * 256, py = (y0 + row) * 256;
    int32_t e0 = edge_mul(es.dx0, py - es.ay0) - edge_mul(es.dy0, px - es.ax0);
    int32_t e1 = edge_mul(es.dx1, py - es.ay1) - edge_mul(es.dy1, px - es.ax1);
    int32_t e2 = edge_mul(es.dx2, py - es.ay2) - edge_mul(es.dy2, px - es.ax2);
    int32_t o0[S], o1[S], o2[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        o0[s] = edge_mul(es.dx0, (int32_t)c_sample_y4[s]) - edge_mul(es.dy0, (int32_t)c_sample_x4[s]) - es.bias0;
        o1[s] = edge_mul(es.dx1, (int32_t)c_sample_y4[s]) - edge_mul(es.dy1, (int32_t)c_sample_x4[s]) - es.bias1;
        o2[s] = edge_mul(es.dx2, (int32_t)c_sample_y4[s]) - edge_mul(es.dy2, (int32_t)c_sample_x4[s]) - es.bias2;
    }
    const int32_t sx0 = -(es.dy0 << 8), sx1 = -(es.dy1 << 8), sx2 = -(es.dy2 << 8);
    const int32_t nl0 = (es.dx0 << 8) - (w - 1) * sx0, nl1 = (es.dx1 << 8) - (w - 1) * sx1, nl2 = (es.dx2 << 8) - (w - 1) * sx2;
    uint32_t pix = (uint32_t)((y0 + row - tc.py0) * TILE_W + (x0 + col - tc.px0));
    for (int i = k0; i < k1; ++i) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int32_t a0 = e0 + o0[s], a1 = e1 + o1[s], a2 = e2 + o2[s];
            if ((a0 | a1 | a2) >= 0) es.emit(tc, pix * S + s, a1 + es.bias1, a2 + es.bias2);
        }
        const bool last = ++col == w;
        e0 += last ? nl0 : sx0; e1 += last ? nl1 : sx1; e2 += last ? nl2 : sx2;
        pix += last ? (uint32_t)(TILE_W - w + 1) : 1u;
        if (last) col = 0;
    }
}

/* What a multisample walk over part of a small triangle's box needs, whichever part it is (the four-sample counterpart of
 * WalkSetup): the three edge values at the corner of the box's first pixel, their steps per pixel right and down, each
 * sample's constant per edge (ownership bias folded in), the depth plane and the fragment word's low bits.  Taken once per
 * triangle by the lane that holds it; a lane that is dealt a chunk of the box fetches these words from that lane and starts
 * its walk with six multiply-adds, instead of setting the triangle up again for every chunk. */
template <int S>
struct SampleWalkSetup {
    int32_t e0, e1, e2;     /* at the corner of the box's first pixel */
    int32_t sx0, sx1, sx2;  /* one pixel right */
    int32_t sy0, sy1, sy2;  /* one pixel down */
    int32_t o0[S], o1[S], o2[S];
    float z0, gz1, gz2;
    uint32_t pay_lo, pay_hi;
    uint32_t box;           /* bits 0..4 first column in the tile, 5..10 first row, 11..15 width - 1, 16..26 pixels - 1, 30/31 bias of edges 1/2 */

    __device__ __forceinline__ void init(const FrameParams& fp, const TileCtx& tc, const RasterTri& t, int x0, int y0, int w, int area)
    {
        EdgeSetup<int32_t, S> es;
        es.init(fp, t);
        const int32_t px = x0 * 256, py = y0 * 256;
        e0 = edge_mul(es.dx0, py - es.ay0) - edge_mul(es.dy0, px - es.ax0);
        e1 = edge_mul(es.dx1, py - es.ay1) - edge_mul(es.dy1, px - es.ax1);
        e2 = edge_mul(es.dx2, py - es.ay2) - edge_mul(es.dy2, px - es.ax2);
        sx0 = -(es.dy0 << 8); sx1 = -(es.dy1 << 8); sx2 = -(es.dy2 << 8);
        sy0 = es.dx0 << 8; sy1 = es.dx1 << 8; sy2 = es.dx2 << 8;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            o0[s] = edge_mul(es.dx0, (int32_t)c_sample_y4[s]) - edge_mul(es.dy0, (int32_t)c_sample_x4[s]) - es.bias0;
            o1[s] = edge_mul(es.dx1, (int32_t)c_sample_y4[s]) - edge_mul(es.dy1, (int32_t)c_sample_x4[s]) - es.bias1;
            o2[s] = edge_mul(es.dx2, (int32_t)c_sample_y4[s]) - edge_mul(es.dy2, (int32_t)c_sample_x4[s]) - es.bias2;
        }
        z0 = es.z0; gz1 = es.gz1; gz2 = es.gz2; pay_lo = es.pay_lo; pay_hi = es.pay_hi;
        box = (uint32_t)(x0 - tc.px0) | ((uint32_t)(y0 - tc.py0) << 5) | ((uint32_t)(w - 1) << 11) | ((uint32_t)(area - 1) << 16) |
              ((uint32_t)es.bias1 << 30) | ((uint32_t)es.bias2 << 31);
    }

    __device__ __forceinline__ SampleWalkSetup from_lane(int o) const
    {
        SampleWalkSetup u;
        u.e0 = __shfl(e0, o); u.e1 = __shfl(e1, o); u.e2 = __shfl(e2, o);
        u.sx0 = __shfl(sx0, o); u.sx1 = __shfl(sx1, o); u.sx2 = __shfl(sx2, o);
        u.sy0 = __shfl(sy0, o); u.sy1 = __shfl(sy1, o); u.sy2 = __shfl(sy2, o);
#pragma unroll
        for (int s = 0; s < S; ++s) { u.o0[s] = __shfl(o0[s], o); u.o1[s] = __shfl(o1[s], o); u.o2[s] = __shfl(o2[s], o); }
        u.z0 = __shfl(z0, o); u.gz1 = __shfl(gz1, o); u.gz2 = __shfl(gz2, o);
        u.pay_lo = (uint32_t)__shfl((int)pay_lo, o); u.pay_hi = (uint32_t)__shfl((int)pay_hi, o);
        u.box = (uint32_t)__shfl((int)box, o);
        return u;
    }

    __device__ __forceinline__ int pixels() const { return (int)((box >> 16) & 2047u) + 1; }

    /* pixels k0 .. k1-1 of the box in row order */
    __device__ __forceinline__ void walk(const TileCtx& tc, int k0, int k1) const
    {
        const int w = (int)((box >> 11) & 31u) + 1;
        const int32_t b1 = (int32_t)((box >> 30) & 1u), b2 = (int32_t)(box >> 31);
        int row = (int)((float)k0 * __builtin_amdgcn_rcpf((float)w)); /* within one of k0 / w */
        int col = k0 - row * w;
        if (col < 0) { --row; col += w; } else if (col >= w) { ++row; col -= w; }
        int32_t a0 = e0 + __mul24(col, sx0) + __mul24(row, sy0);
        int32_t a1 = e1 + __mul24(col, sx1) + __mul24(row, sy1);
        int32_t a2 = e2 + __mul24(col, sx2) + __mul24(row, sy2);
        const int32_t nl0 = sy0 - __mul24(w - 1, sx0), nl1 = sy1 - __mul24(w - 1, sx1), nl2 = sy2 - __mul24(w - 1, sx2);
        uint32_t pix = (((box >> 5) & 63u) + (uint32_t)row) * TILE_W + (box & 31u) + (uint32_t)col;
        EdgeSetup<int32_t, S> es; /* only its depth plane is used */
        es.z0 = z0; es.gz1 = gz1; es.gz2 = gz2; es.pay_lo = pay_lo; es.pay_hi = pay_hi;
        for (int i = k0; i < k1; ++i) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int32_t c0 = a0 + o0[s], c1 = a1 + o1[s], c2 = a2 + o2[s];
                if ((c0 | c1 | c2) >= 0) es.emit(tc, pix * S + s, c1 + b1, c2 + b2);
            }
            const bool last = ++col == w;
            a0 += last ? nl0 : sx0; a1 += last ? nl1 : sx1; a2 += last ? nl2 : sx2;
            pix += last ? (uint32_t)(TILE_W - w + 1) : 1u;
            if (last) col = 0;
        }
    }
};

__device__ __forceinline__ bool tri_box_in_tile(const FrameParams& fp, const TileCtx& tc, const RasterTri& t,
                                                int& x0, int& y0, int& x1, int& y1)
{
    const int minx = min(t.x[0], min(t.x[1], t.x[2])), maxx = max(t.x[0], max(t.x[1], t.x[2]));
    const int miny = min(t.y[0], min(t.y[1], t.y[2])), maxy = max(t.y[0], max(t.y[1], t.y[2]));
    if (fp.samples == 1) { x0 = (minx + 127) >> 8; x1 = (maxx - 128) >> 8; y0 = (miny + 127) >> 8; y1 = (maxy - 128) >> 8; }
    else { x0 = (minx + 31) >> 8; x1 = (maxx - 32) >> 8; y0 = (miny + 31) >> 8; y1 = (maxy - 32) >> 8; } /* samples lie at 32 .. 224 of a pixel's 256 */
    x0 = max(x0, tc.rx0); y0 = max(y0, tc.ry0); x1 = min(x1, tc.rx1); y1 = min(y1, tc.ry1);
    return x0 <= x1 && y0 <= y1;
}

__device__ __forceinline__ bool tri_is_small(const RasterTri& t)
{
    const int ext = max(max(t.x[0], max(t.x[1], t.x[2])) - min(t.x[0], min(t.x[1], t.x[2])),
                        max(t.y[0], max(t.y[1], t.y[2])) - min(t.y[0], min(t.y[1], t.y[2])));
    return ext < (1 << 14);
}

/* one lane walks its own (small) triangle's box, as ONE loop over the box's pixels in row order:
 * the wave then runs for as many steps as its largest box has pixels.  (As a loop over rows around
 * a loop over columns it would run for the tallest box's height times the widest box's width, which
 * for triangles of mixed orientation is several times more.)  With one sample per pixel the three
 * edge values are stepped incrementally: one add per edge per pixel, multiplies only at the start. */
template <int S>
__device__ __forceinline__ void raster_lane(const FrameParams& fp, const TileCtx& tc, const RasterTri& t, int x0, int y0, int x1, int y1)
{
    EdgeSetup<int32_t, S> es;
    es.init(fp, t);
    const int w = x1 - x0 + 1;
    const int count = w * (y1 - y0 + 1);
    if constexpr (S == 1) {
        const int32_t px = x0 * 256 + 128, py = y0 * 256 + 128;
        /* the ownership bias is folded into the start values and taken out again for the depth */
        int32_t e0 = edge_mul(es.dx0, py - es.ay0) - edge_mul(es.dy0, px - es.ax0) - es.bias0;
        int32_t e1 = edge_mul(es.dx1, py - es.ay1) - edge_mul(es.dy1, px - es.ax1) - es.bias1;
        int32_t e2 = edge_mul(es.dx2, py - es.ay2) - edge_mul(es.dy2, px - es.ax2) - es.bias2;
        const int32_t sx0 = -(es.dy0 << 8), sx1 = -(es.dy1 << 8), sx2 = -(es.dy2 << 8); /* one pixel right */
        const int32_t sy0 = es.dx0 << 8, sy1 = es.dx1 << 8, sy2 = es.dx2 << 8;          /* one pixel down */
        /* strips of two rows, both pixels of a column in one step (see WalkSetup); from the last column of a strip to
         * the first of the next: two down, w - 1 back */
        const int32_t nl0 = WALK_ROWS * sy0 - (w - 1) * sx0, nl1 = WALK_ROWS * sy1 - (w - 1) * sx1, nl2 = WALK_ROWS * sy2 - (w - 1) * sx2;
        const int h = y1 - y0 + 1;
        const int steps = walk_steps(w, h);
        const int single = (h & 1) ? steps - w : 0x7fffffff; /* the steps of an odd box's last strip have one row */
        uint32_t pix8 = (uint32_t)((y0 - tc.py0) * TILE_W + (x0 - tc.px0)) * 8u; /* byte offset of the pixel's word */
        const uint32_t next_strip8 = (uint32_t)(WALK_ROWS * TILE_W - w + 1) * 8u;
        int col = 0;
        for (int i = 0; i < steps; ++i) {
            if ((e0 | e1 | e2) >= 0) es.emit_at(tc, pix8, e1 + es.bias1, e2 + es.bias2);
            const int32_t c0 = e0 + sy0, c1 = e1 + sy1, c2 = e2 + sy2;
            if ((c0 | c1 | c2) >= 0 && i < single) es.emit_at(tc, pix8 + TILE_W * 8u, c1 + es.bias1, c2 + es.bias2);
            const bool last = ++col == w;
            e0 += last ? nl0 : sx0; e1 += last ? nl1 : sx1; e2 += last ? nl2 : sx2;
            pix8 += last ? next_strip8 : 8u;
            if (last) col = 0;
        }
    } else {
        walk_samples<S>(es, tc, x0, y0, w, 0, count, 0, 0);
    }
}

/* the whole wave walks one triangle's box in 8x8 pixel blocks (t and the box are wave-uniform):
 * the path of triangles 64 pixels or more across, whose edge products need 64 bits */
template <int S>
__device__ __forceinline__ void raster_wave(const FrameParams& fp, const TileCtx& tc, const RasterTri& t, int x0, int y0, int x1, int y1)
{
    const int lx = (int)(lane_id() & 7u), ly = (int)(lane_id() >> 3);
    EdgeSetup<int64_t, S> es;
    es.init(fp, t);
    for (int by = y0; by <= y1; by += 8) for (int bx = x0; bx <= x1; bx += 8) {
        const int x = bx + lx, y = by + ly;
        if (x <= x1 && y <= y1) es.pixel(tc, x, y);
    }
}

/* entry -> triangle; false when the entry must not be drawn */
__device__ __forceinline__ bool load_raster_tri(const FrameParams& fp, const TileEntry& e, RasterTri& t)
{
    t.slot = e.slot_flags & SLOT_MASK;
    if (e.slot_flags & FLAG_CLIPPED) {
        const ClipTri* ct = &fp.clip_tris[e.a];
#pragma unroll
        for (int k = 0; k < 3; ++k) { t.x[k] = ct->x[k]; t.y[k] = ct->y[k]; t.z[k] = ct->z[k]; }
        t.payload_lo = ((e.slot_flags >> ENTRY_SUB_SHIFT) & 7u) | 8u;
        return true;
    }
    const SVtx a = fp.svtx[e.a], b = fp.svtx[e.b], c = fp.svtx[e.c];
    t.x[0] = a.xf >> 8; t.y[0] = a.y; t.z[0] = a.z;
    t.x[1] = b.xf >> 8; t.y[1] = b.y; t.z[1] = b.z;
    t.x[2] = c.xf >> 8; t.y[2] = c.y; t.z[2] = c.z;
    t.payload_lo = 0;
    return true;
}

/* Who walks a triangle's box.
 *
 * A lane walking its own box costs the wave as many steps as the largest box among its lanes, so it
 * is used as it stands only while every box of the wave has at most DIRECT_PIXELS pixels (a mesh of
 * pixel-sized triangles).  Otherwise the boxes are cut into chunks of CHUNK_PIXELS consecutive
 * pixels (in the row order of the one-loop walk) and the chunks, not the triangles, are dealt to the
 * lanes: an exclusive scan of the chunk counts numbers them, lane l of round r takes chunk 64 r + l,
 * finds the triangle that owns it by a binary search over the scan (ds_bpermute), fetches that
 * triangle from its lane and walks the chunk.  Every lane then has at most CHUNK_PIXELS steps per
 * round whatever the mix of box sizes, and nothing is left to the whole-wave path except triangles
 * too large for 32-bit edge products. */
constexpr int DIRECT_PIXELS = 32; /* measured: 16-pixel chunks balance best, but a wave of boxes up to 32 pixels is quicker walked as it is; the same holds at four samples per pixel */
constexpr int CHUNK_PIXELS = 8; /* several samples per pixel; measured on the 4K instances frame at four samples: 16 pixels +1.1 %, 24 +6 %, 32 +10 % */
constexpr int CHUNK_STEPS = 8; /* one sample per pixel: steps of the two-row walk, 16 pixels; measured: 6 steps +0.4 %, 12 steps +2.3 %, 16 steps +3.9 % on the 4K instances frame */

template <int S>
__device__ __forceinline__ void raster_chunks(const FrameParams& fp, const TileCtx& tc, bool active, const RasterTri& t,
                                              int x0, int y0, int x1, int y1)
{
    const int lane = (int)lane_id();
    const int w = x1 - x0 + 1, h = y1 - y0 + 1;
    const int area = active ? w * h : 0;
    /* one sample per pixel: chunks of CHUNK_STEPS steps of the two-row walk; else of CHUNK_PIXELS pixels */
    const int n = S == 1 ? (active ? (walk_steps(w, h) + CHUNK_STEPS - 1) / CHUNK_STEPS : 0) : (area + CHUNK_PIXELS - 1) / CHUNK_PIXELS;
    const int incl = wave_inclusive_scan(n, 0, [](int a, int b) { return a + b; });
    const int excl = incl - n;
    const int total = __builtin_amdgcn_readlane(incl, 63);
    typename std::conditional<S == 1, WalkSetup, SampleWalkSetup<S>>::type mine;
    if constexpr (S == 1) mine.init(fp, tc, t, x0, y0, active ? w : 1, active ? h : 1);
    else mine.init(fp, tc, t, x0, y0, active ? w : 1, active ? area : 1);
    for (int base = 0; base < total; base += 64) {
        const int c = base + lane;
        /* owner of chunk c: the last lane that has chunks and whose first chunk number is not above c.  Every such lane
         * whose first chunk falls in this round writes its number into the wave's row of LDS at that chunk's place; a
         * running maximum over the row, started from the last lane whose first chunk lies before the round, gives every
         * place its owner. */
        volatile signed char* row = tc.wave_row; /* lanes talk to each other through it: every access is to happen as written */
        row[lane] = -1;
        if (n > 0 && excl >= base && excl < base + 64) row[excl - base] = (signed char)lane;
        const unsigned long long before = __ballot(n > 0 && excl <= base);
        const int carried = before ? 63 - __builtin_clzll(before) : 0;
        const int o = max(wave_inclusive_scan((int)row[lane], -1, [](int a, int b) { return max(a, b); }), carried);
        const auto u = mine.from_lane(o);
        const int ofirst = __shfl(excl, o);
        if (c >= total) continue;
        if constexpr (S == 1) {
            const int k0 = (c - ofirst) * CHUNK_STEPS, k1 = min(k0 + CHUNK_STEPS, u.steps());
            u.walk(tc, k0, k1);
        } else {
            const int k0 = (c - ofirst) * CHUNK_PIXELS, k1 = min(k0 + CHUNK_PIXELS, u.pixels());
            u.walk(tc, k0, k1);
        }
    }
}

/* One batch: every lane holds (at most) one triangle. */
template <int S>
__device__ __forceinline__ void raster_batch(const FrameParams& fp, const TileCtx& tc, bool have, const RasterTri& t)
{
    int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
    const bool hit = have && tri_box_in_tile(fp, tc, t, x0, y0, x1, y1);
    const int area = hit ? (x1 - x0 + 1) * (y1 - y0 + 1) : 0;
    const bool small = tri_is_small(t);
    if (RND_PROBE(fp, 128u)) { if (hit && t.z[0] == 123.0f) tc.keys[0] = 0; return; }
    if (__ballot(hit && small && area > DIRECT_PIXELS)) raster_chunks<S>(fp, tc, hit && small, t, x0, y0, x1, y1);
    else if (hit && small) raster_lane<S>(fp, tc, t, x0, y0, x1, y1);
    /* too large for 32-bit edge products: one at a time by the whole wave (ballot, broadcast, 8x8 blocks) */
    uint64_t big = __ballot(hit && !small);
    while (big) {
        const int src = __ffsll((unsigned long long)big) - 1;
        big &= big - 1;
        RasterTri u;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            u.x[k] = __shfl(t.x[k], src); u.y[k] = __shfl(t.y[k], src); u.z[k] = __shfl(t.z[k], src);
        }
        u.payload_lo = (uint32_t)__shfl((int)t.payload_lo, src);
        u.slot = (uint32_t)__shfl((int)t.slot, src);
        raster_wave<S>(fp, tc, u, __shfl(x0, src), __shfl(y0, src), __shfl(x1, src), __shfl(y1, src));
    }
}

/* ---- fragment stage ---- */

struct Varyings { float light; float uv[2]; };

/* Vertex-shader outputs other than the position, as the vertex kernel left them */
__device__ __forceinline__ void load_varyings(const FrameParams& fp, uint32_t vtx, Varyings& out)
{
    /* three adjacent words: one 12-byte load.  The address is 64 bits per lane: draws that index one vertex buffer
     * from its start each transform it up to their largest index, and a pass of some hundred such draws holds more
     * than 4 GB of records */
    typedef float Triple __attribute__((ext_vector_type(3), aligned(4)));
    typedef const __attribute__((address_space(1))) Triple* GlobalTriple;
    const Triple a = *(GlobalTriple)&fp.svary[vtx];
    out.light = a.x;
    out.uv[0] = a.y;
    out.uv[1] = a.z; /* the vertex kernel has applied the dummy shader's flip of v */
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

/* i and i + 1 wrapped into [0, n): what wrap_coord returns for each, with one remainder instead of
 * two (a mask when the texture's sizes are powers of two, so every level's are) */
__device__ __forceinline__ void wrap_pair(int i, int n, bool tiled, bool pow2, int& a, int& b)
{
    if (tiled) {
        if (pow2) a = i & (n - 1);
        else { const int m = i % n; a = m < 0 ? m + n : m; }
        b = a + 1 == n ? 0 : a + 1;
    } else {
        a = min(max(i, 0), n - 1);
        b = min(max(i + 1, 0), n - 1);
    }
}

__device__ __forceinline__ int wrap_one(int i, int n, bool tiled, bool pow2)
{
    if (tiled) {
        if (pow2) return i & (n - 1);
        const int m = i % n;
        return m < 0 ? m + n : m;
    }
    return min(max(i, 0), n - 1);
}

/* What the sampler needs of one level of the pass-wide texture, kept in LDS by the tile kernel: a lane whose
 * level differs from its neighbours' reads the level's address, its size as the floats and integers the tap
 * arithmetic uses and the clamp bounds in two 16-byte reads, instead of shifting, clamping and converting the
 * texture's size per lane and level. */
struct __attribute__((aligned(16))) LevelRec {
    unsigned long long base; /* first texel */
    float fw, fh;            /* width and height */
    int32_t w, wm1, hm1, wm2; /* width, width - 1, height - 1, max(width - 2, 0) */
};
/* pointer into LDS, typed as such so a per-lane index compiles to a ds_read and not a flat load */
typedef const __attribute__((address_space(3))) LevelRec* LdsLevels;

/* What a sampler needs of a texture, in registers: scalar for the pass-wide texture, per lane otherwise */
struct TexView {
    uint32_t width, height, levels;
    bool tiled, linear, pow2;
    uint32_t table; /* which half of ShadeTables::value decodes its texels */
};

__device__ __forceinline__ TexView view_of(const TexDesc& t)
{
    TexView tv;
    tv.width = t.width; tv.height = t.height; tv.levels = t.levels;
    tv.tiled = t.tiled != 0u; tv.linear = t.filter_linear != 0u;
    tv.pow2 = ((t.width & (t.width - 1u)) | (t.height & (t.height - 1u))) == 0u;
    tv.table = t.srgb ? 0u : 1u;
    return tv;
}

/* The level differs from lane to lane: its base comes from the copy of the pass texture's table in
 * LDS when there is one, else from the descriptor in memory */
template <bool PASS_TEX>
__device__ __forceinline__ const uint8_t* level_base(const TexDesc& t, LdsLevels lds_levels, uint32_t level)
{
    if constexpr (PASS_TEX) return reinterpret_cast<const uint8_t*>(lds_levels[level].base);
    else return t.level[level];
}

/* A level holds at most 16384 x 16384 texels, so a texel's byte offset fits 32 bits.  Texels live
 * in device memory: say so, or a pointer read from a descriptor costs a flat load. */
__device__ __forceinline__ uint32_t load_texel(const uint8_t* lv, uint32_t byte_offset)
{
    typedef const __attribute__((address_space(1))) uint32_t* GlobalTexel;
    return *(GlobalTexel)(lv + byte_offset);
}

template <typename Tables>
__device__ __forceinline__ void decode_texel(const Tables& tb, uint32_t table, uint32_t p, float* rgb)
{
    const float* value = tb.value[table];
    rgb[0] = value[p & 255u]; rgb[1] = value[(p >> 8) & 255u]; rgb[2] = value[(p >> 16) & 255u];
}

/* The four texels a bilinear sample of one level reads, as byte offsets into the level
 * ((x0,y0) (x1,y0) (x0,y1) (x1,y1)), and its two blend fractions.  Offsets first, loads next,
 * arithmetic last: the loads of a sample (eight for a trilinear one) are then in flight together
 * instead of each waiting for the one before it. */
struct Taps {
    uint32_t off[4];
    float ax, ay;
};

__device__ __forceinline__ void linear_taps(const TexView& t, uint32_t level, float u, float v, Taps& tp)
{
    const int w = (int)level_dim(t.width, level), h = (int)level_dim(t.height, level);
    float fu = fma32(u, (float)w, -0.5f), fv = fma32(v, (float)h, -0.5f);
    fu = fminf(fmaxf(fu, -1.0e9f), 1.0e9f);
    fv = fminf(fmaxf(fv, -1.0e9f), 1.0e9f);
    const float iu = floorf(fu), iv = floorf(fv);
    tp.ax = fu - iu; tp.ay = fv - iv;
    int x0, x1, y0, y1;
    wrap_pair((int)iu, w, t.tiled, t.pow2, x0, x1);
    wrap_pair((int)iv, h, t.tiled, t.pow2, y0, y1);
    const uint32_t row0 = __umul24((uint32_t)y0, (uint32_t)w), row1 = __umul24((uint32_t)y1, (uint32_t)w);
    tp.off[0] = (row0 + (uint32_t)x0) << 2; tp.off[1] = (row0 + (uint32_t)x1) << 2;
    tp.off[2] = (row1 + (uint32_t)x0) << 2; tp.off[3] = (row1 + (uint32_t)x1) << 2;
}

/* The same four texels of a CLAMP_TO_EDGE level as two 8-byte loads, one per row: the two columns a
 * sample reads are x0 = clamp(i) and x1 = clamp(i + 1), which are both among {b, b + 1} for
 * b = min(x0, max(w - 2, 0)) -- neighbours in memory.  Half the vector-memory instructions of four
 * single-texel loads for the same bytes.  A level one texel wide reads four bytes past its row; every
 * level is allocated with that much slack behind it (texture_alloc) and the extra texel is never
 * selected.  A REPEAT sampler keeps the four single loads: its second column wraps to the row's start. */
struct PairTaps {
    uint32_t row[2];   /* byte offset of column b in the sample's two rows */
    bool x0_hi, x1_hi; /* which texel of the pair is column x0 / x1 */
    float ax, ay;
};

__device__ __forceinline__ void pair_taps_of(float fw, float fh, int w, int wm1, int hm1, int wm2, float u, float v, PairTaps& tp)
{
    float fu = fma32(u, fw, -0.5f), fv = fma32(v, fh, -0.5f);
    fu = fminf(fmaxf(fu, -1.0e9f), 1.0e9f);
    fv = fminf(fmaxf(fv, -1.0e9f), 1.0e9f);
    const float iu = floorf(fu), iv = floorf(fv);
    tp.ax = fu - iu; tp.ay = fv - iv;
    const int i = (int)iu, j = (int)iv;
    const int x0 = min(max(i, 0), wm1), x1 = min(max(i + 1, 0), wm1);
    const int y0 = min(max(j, 0), hm1), y1 = min(max(j + 1, 0), hm1);
    const int b = min(x0, wm2);
    tp.x0_hi = x0 != b; tp.x1_hi = x1 != b;
    tp.row[0] = (__umul24((uint32_t)y0, (uint32_t)w) + (uint32_t)b) << 2;
    tp.row[1] = (__umul24((uint32_t)y1, (uint32_t)w) + (uint32_t)b) << 2;
}

__device__ __forceinline__ void pair_taps(const TexView& t, uint32_t level, float u, float v, PairTaps& tp)
{
    const int w = (int)level_dim(t.width, level), h = (int)level_dim(t.height, level);
    pair_taps_of((float)w, (float)h, w, w - 1, h - 1, max(w - 2, 0), u, v, tp);
}

/* the same with the level's figures out of the tile kernel's table */
__device__ __forceinline__ void pair_taps(LdsLevels lds_levels, uint32_t level, float u, float v, PairTaps& tp)
{
    const LdsLevels r = lds_levels + level; /* member by member: a structure in LDS is not copied as a whole */
    pair_taps_of(r->fw, r->fh, r->w, r->wm1, r->hm1, r->wm2, u, v, tp);
}

/* two neighbouring texels in one load; the address is a multiple of 4, not of 8 */
struct TexelPair { uint32_t lo, hi; };

__device__ __forceinline__ TexelPair load_texel_pair(const uint8_t* lv, uint32_t byte_offset)
{
    typedef unsigned long long __attribute__((aligned(4))) Packed;
    typedef const __attribute__((address_space(1))) Packed* GlobalPair;
    const unsigned long long q = *(GlobalPair)(lv + byte_offset);
    TexelPair r;
    r.lo = (uint32_t)q; r.hi = (uint32_t)(q >> 32);
    return r;
}

__device__ __forceinline__ void pair_select(const PairTaps& tp, const TexelPair& r0, const TexelPair& r1, uint32_t* p)
{
    p[0] = tp.x0_hi ? r0.hi : r0.lo; p[1] = tp.x1_hi ? r0.hi : r0.lo;
    p[2] = tp.x0_hi ? r1.hi : r1.lo; p[3] = tp.x1_hi ? r1.hi : r1.lo;
}

__device__ __forceinline__ uint32_t nearest_tap(const TexView& t, uint32_t level, float u, float v)
{
    const int w = (int)level_dim(t.width, level), h = (int)level_dim(t.height, level);
    const float fu = fminf(fmaxf(u * (float)w, -1.0e9f), 1.0e9f);
    const float fv = fminf(fmaxf(v * (float)h, -1.0e9f), 1.0e9f);
    const int x = wrap_one((int)floorf(fu), w, t.tiled, t.pow2), y = wrap_one((int)floorf(fv), h, t.tiled, t.pow2);
    return (__umul24((uint32_t)y, (uint32_t)w) + (uint32_t)x) << 2;
}

template <typename Tables>
__device__ __forceinline__ void bilinear(const Tables& tb, uint32_t table, const uint32_t* p, float ax, float ay, float* rgb)
{
    float c00[3], c10[3], c01[3], c11[3];
    decode_texel(tb, table, p[0], c00);
    decode_texel(tb, table, p[1], c10);
    decode_texel(tb, table, p[2], c01);
    decode_texel(tb, table, p[3], c11);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float top = fma32(ax, c10[k] - c00[k], c00[k]);
        const float bot = fma32(ax, c11[k] - c01[k], c01[k]);
        rgb[k] = fma32(ay, bot - top, top);
    }
}

/* texture() with implicit derivatives: lod = log2(longest footprint axis), clamped to
 * [0, levels-1], trilinear (sampler of tools.cpp:60-82) */
template <bool PASS_TEX, bool CLAMP_LINEAR = false, typename Tables>
__device__ __forceinline__ void sample_texture(const Tables& tb, const TexDesc& desc, LdsLevels lds_levels, float u, float v,
                                               float dudx, float dvdx, float dudy, float dvdy, float* rgb)
{
    TexView t = view_of(desc);
    if constexpr (PASS_TEX) t.table = 0u; /* the tile kernel stages the pass texture's table first */
    if constexpr (CLAMP_LINEAR) { t.linear = true; t.tiled = false; } /* the caller has checked the sampler: LINEAR, CLAMP_TO_EDGE (tools.cpp:60-82) */
    const float w0 = (float)t.width, h0 = (float)t.height;
    const float mxu = dudx * w0, mxv = dvdx * h0, myu = dudy * w0, myv = dvdy * h0;
    const float rx = fma32(mxv, mxv, mxu * mxu), ry = fma32(myv, myv, myu * myu);
    float lambda = 0.5f * lod_log2(fmaxf(rx, ry));
    const float max_lod = (float)(t.levels - 1);
    lambda = fminf(fmaxf(lambda, 0.0f), max_lod);
    if (!finite32(u)) u = 0.0f;
    if (!finite32(v)) v = 0.0f;
    const bool base_only = !(lambda > 0.0f) || t.levels == 1; /* magnified: level 0 alone */
    if (!t.linear) {
        uint32_t level = 0;
        if (!base_only) level = (uint32_t)fminf(fmaxf(ceilf(lambda + 0.5f) - 1.0f, 0.0f), max_lod);
        const uint32_t p = load_texel(level_base<PASS_TEX>(desc, lds_levels, level), nearest_tap(t, level, u, v));
        decode_texel(tb, t.table, p, rgb);
        return;
    }
    if (!__ballot(!base_only)) {
        const uint8_t* lv = level_base<PASS_TEX>(desc, lds_levels, 0);
        uint32_t p[4];
        float ax, ay;
        if (!t.tiled) {
            PairTaps tp;
            if constexpr (PASS_TEX) pair_taps(lds_levels, 0, u, v, tp); else pair_taps(t, 0, u, v, tp);
            const TexelPair r0 = load_texel_pair(lv, tp.row[0]), r1 = load_texel_pair(lv, tp.row[1]);
            pair_select(tp, r0, r1, p);
            ax = tp.ax; ay = tp.ay;
        } else {
            Taps tp;
            linear_taps(t, 0, u, v, tp);
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] = load_texel(lv, tp.off[k]);
            ax = tp.ax; ay = tp.ay;
        }
        bilinear(tb, t.table, p, ax, ay, rgb);
        return;
    }
    /* Lanes of one wave usually disagree about base_only.  Rather than run the one-level and the
     * two-level code one after the other, every lane blends two levels; a base-only lane blends
     * level 0 with itself at fraction 0, which returns the level-0 sample unchanged. */
    const float fl = base_only ? 0.0f : floorf(lambda);
    const uint32_t lo = (uint32_t)fl;
    const uint32_t hi = base_only ? 0u : (lo + 1 < t.levels ? lo + 1 : t.levels - 1);
    const float frac = base_only ? 0.0f : lambda - fl;
    const uint8_t* la = level_base<PASS_TEX>(desc, lds_levels, lo);
    const uint8_t* lc = level_base<PASS_TEX>(desc, lds_levels, hi);
    uint32_t pa[4], pc[4];
    float aax, aay, cax, cay;
    if (!t.tiled) {
        PairTaps ta, tc;
        if constexpr (PASS_TEX) { pair_taps(lds_levels, lo, u, v, ta); pair_taps(lds_levels, hi, u, v, tc); }
        else { pair_taps(t, lo, u, v, ta); pair_taps(t, hi, u, v, tc); }
        const TexelPair a0 = load_texel_pair(la, ta.row[0]), a1 = load_texel_pair(la, ta.row[1]);
        const TexelPair c0 = load_texel_pair(lc, tc.row[0]), c1 = load_texel_pair(lc, tc.row[1]);
        pair_select(ta, a0, a1, pa);
        pair_select(tc, c0, c1, pc);
        aax = ta.ax; aay = ta.ay; cax = tc.ax; cay = tc.ay;
    } else {
        Taps ta, tc;
        linear_taps(t, lo, u, v, ta);
        linear_taps(t, hi, u, v, tc);
#pragma unroll
        for (int k = 0; k < 4; ++k) pa[k] = load_texel(la, ta.off[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) pc[k] = load_texel(lc, tc.off[k]);
        aax = ta.ax; aay = ta.ay; cax = tc.ax; cay = tc.ay;
    }
    float a[3], c[3];
    bilinear(tb, t.table, pa, aax, aay, a);
    bilinear(tb, t.table, pc, cax, cay, c);
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[k] = fma32(frac, c[k] - a[k], a[k]);
}

/* Perspective-correct weights of a fragment over the three corners of its triangle, at the pixel
 * centre and (when derivatives are wanted) at the pixel's two neighbours inside its 2x2 quad.
 *
 * The reference arithmetic orients the triangle (positive area), evaluates its three edge functions
 * e at the point, and forms b = float(e) * (1 / float(area)), p = b * (1/w), q = p * (1 / (p0+p1+p2)).
 * Here the edge functions are taken in the corners' own order: orienting negates all of them and
 * the area, and a product of two negated factors rounds the same way, so b is unchanged; the only
 * trace of the orientation is the order in which the denominator is summed (corners 0,2,1 when the
 * area is negative).  A neighbour's edge values differ from the pixel's by a constant per edge.
 * Triangles whose box spans less than 32 pixels fit 24-bit multiplies; the rest take 64 bits. */
struct Corners {
    int32_t x[3], y[3]; /* 24.8 window coordinates */
    float iw[3];        /* 1/w */
};

struct EdgeFloats {
    float inv_s;        /* 1 / signed doubled area */
    bool swapped;       /* negative area */
    float iw[3];
};

/* the three products of a point and their sum, in the order the reference arithmetic adds them */
__device__ __forceinline__ float weight_products(const EdgeFloats& t, float e0, float e1, float e2, float* p)
{
    p[0] = (e0 * t.inv_s) * t.iw[0];
    p[1] = (e1 * t.inv_s) * t.iw[1];
    p[2] = (e2 * t.inv_s) * t.iw[2];
    return (p[0] + (t.swapped ? p[2] : p[1])) + (t.swapped ? p[1] : p[2]);
}

__device__ __forceinline__ void weights_at(const EdgeFloats& t, float e0, float e1, float e2, float* q)
{
    const float rden = recip32(weight_products(t, e0, e1, e2, q));
    q[0] *= rden; q[1] *= rden; q[2] *= rden;
}

/* The same at three points at once.  The refined estimate is taken for all three sums together and the division only
 * where some lane of the wave has a sum outside the estimate's range (recip32), which is asked once for the three:
 * one stretch of straight code instead of three branches one after the other. */
__device__ __forceinline__ void weights_at3(const EdgeFloats& t, const float* e, const float* ex, const float* ey, float* q, float* qx, float* qy)
{
    const float d0 = weight_products(t, e[0], e[1], e[2], q);
    const float d1 = weight_products(t, ex[0], ex[1], ex[2], qx);
    const float d2 = weight_products(t, ey[0], ey[1], ey[2], qy);
    float r0 = recip32_in_range(d0), r1 = recip32_in_range(d1), r2 = recip32_in_range(d2);
    const bool ok0 = recip32_range(d0), ok1 = recip32_range(d1), ok2 = recip32_range(d2);
    if (__ballot(!(ok0 && ok1 && ok2))) {
        if (!ok0) r0 = 1.0f / d0;
        if (!ok1) r1 = 1.0f / d1;
        if (!ok2) r2 = 1.0f / d2;
    }
    q[0] *= r0; q[1] *= r0; q[2] *= r0;
    qx[0] *= r1; qx[1] *= r1; qx[2] *= r1;
    qy[0] *= r2; qy[1] *= r2; qy[2] *= r2;
}

/* q, qx, qy: weights at pixel (x, y), at its quad neighbour in x and at its quad neighbour in y */
__device__ __forceinline__ void corner_weights(const Corners& c, int x, int y, bool want_derivs, float* q, float* qnx, float* qny)
{
    const int32_t dx0 = c.x[2] - c.x[1], dy0 = c.y[2] - c.y[1];
    const int32_t dx1 = c.x[0] - c.x[2], dy1 = c.y[0] - c.y[2];
    const int32_t dx2 = c.x[1] - c.x[0], dy2 = c.y[1] - c.y[0];
    const int32_t px = x * 256 + 128, py = y * 256 + 128;
    const bool left = (x & 1) != 0, up = (y & 1) != 0; /* the neighbour is at x-1 / y-1 */
    const int ext = max(max(c.x[0], max(c.x[1], c.x[2])) - min(c.x[0], min(c.x[1], c.x[2])),
                        max(c.y[0], max(c.y[1], c.y[2])) - min(c.y[0], min(c.y[1], c.y[2])));
    float fe[3], fx[3], fy[3], fs; /* fx, fy: only where derivatives are wanted, and only read there */
    /* a shaded pixel has a sample inside the triangle, so it and its quad neighbours are within two
     * pixels of the box: with a box under 2^13 every factor below stays under 2^14 */
    if (ext < (1 << 13)) {
        const int32_t e0 = __mul24(dx0, py - c.y[1]) - __mul24(dy0, px - c.x[1]);
        const int32_t e1 = __mul24(dx1, py - c.y[2]) - __mul24(dy1, px - c.x[2]);
        const int32_t e2 = __mul24(dx2, py - c.y[0]) - __mul24(dy2, px - c.x[0]);
        fs = (float)(__mul24(dx2, c.y[2] - c.y[0]) - __mul24(dy2, c.x[2] - c.x[0]));
        fe[0] = (float)e0; fe[1] = (float)e1; fe[2] = (float)e2;
        if (want_derivs) {
            /* x by +-256 changes e by -+256*dy; y by +-256 changes e by +-256*dx: one 24-bit multiply-add per value
             * (the differences are below 2^13), the direction in the factor */
            const int32_t sl = left ? 256 : -256, su = up ? -256 : 256;
            fx[0] = (float)(e0 + __mul24(dy0, sl)); fx[1] = (float)(e1 + __mul24(dy1, sl)); fx[2] = (float)(e2 + __mul24(dy2, sl));
            fy[0] = (float)(e0 + __mul24(dx0, su)); fy[1] = (float)(e1 + __mul24(dx1, su)); fy[2] = (float)(e2 + __mul24(dx2, su));
        }
    } else {
        const int64_t e0 = (int64_t)dx0 * (py - c.y[1]) - (int64_t)dy0 * (px - c.x[1]);
        const int64_t e1 = (int64_t)dx1 * (py - c.y[2]) - (int64_t)dy1 * (px - c.x[2]);
        const int64_t e2 = (int64_t)dx2 * (py - c.y[0]) - (int64_t)dy2 * (px - c.x[0]);
        fs = (float)((int64_t)dx2 * (c.y[2] - c.y[0]) - (int64_t)dy2 * (c.x[2] - c.x[0]));
        fe[0] = (float)e0; fe[1] = (float)e1; fe[2] = (float)e2;
        if (want_derivs) {
            const int64_t gx0 = (int64_t)dy0 * 256, gx1 = (int64_t)dy1 * 256, gx2 = (int64_t)dy2 * 256;
            const int64_t gy0 = (int64_t)dx0 * 256, gy1 = (int64_t)dx1 * 256, gy2 = (int64_t)dx2 * 256;
            fx[0] = (float)(left ? e0 + gx0 : e0 - gx0); fx[1] = (float)(left ? e1 + gx1 : e1 - gx1); fx[2] = (float)(left ? e2 + gx2 : e2 - gx2);
            fy[0] = (float)(up ? e0 - gy0 : e0 + gy0); fy[1] = (float)(up ? e1 - gy1 : e1 + gy1); fy[2] = (float)(up ? e2 - gy2 : e2 + gy2);
        }
    }
    EdgeFloats t;
    t.swapped = fs < 0.0f;
    t.inv_s = recip32_of_area(fs); /* a shaded pixel has a sample inside the triangle: the area is not zero */
    t.iw[0] = c.iw[0]; t.iw[1] = c.iw[1]; t.iw[2] = c.iw[2];
    if (want_derivs) weights_at3(t, fe, fx, fy, q, qnx, qny);
    else weights_at(t, fe[0], fe[1], fe[2], q);
}

/* weights over a fan triangle's corners -> weights over its parent's corners */
__device__ __forceinline__ void to_parent(const float* bary, float* q)
{
    float lam[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) lam[j] = fma32(q[2], bary[6 + j], fma32(q[1], bary[3 + j], q[0] * bary[j]));
    q[0] = lam[0]; q[1] = lam[1]; q[2] = lam[2];
}

/* The fragment shader proper: everything after the corners' data are in registers.  c: window positions and
 * 1/w of the triangle the weights are taken over (a fan triangle when clipped); v0..v2: the vertex shader's
 * outputs at the PARENT's corners; bary (clipped only): the fan triangle's corners as weights over the parent's. */
template <bool SPEC, typename Tables>
__device__ __forceinline__ void shade_fragment(const FrameParams& fp, const Tables& tb, LdsLevels pass_levels, uint32_t draw,
                                               int shader, int texture, const Corners& c, const Varyings& v0, const Varyings& v1,
                                               const Varyings& v2, const float* bary, int x, int y, float* rgba)
{
    /* SPEC: every draw of the pass runs the dummy or the test_ktx shader on one texture sampled LINEAR with CLAMP_TO_EDGE
     * (what the reference's samples record): shader and texture are the pass's, nothing about them is asked per lane */
    const bool want_derivs = SPEC || texture >= 0;
    /* weights over the parent's corners at the pixel centre, and at the neighbour in x and the
     * neighbour in y of the 2x2 quad (same primitive, extrapolated) for the derivatives */
    float lam[3], lnx[3], lny[3]; /* lnx, lny: written and read only where derivatives are wanted */
    corner_weights(c, x, y, want_derivs, lam, lnx, lny);
    if (bary) {
        to_parent(bary, lam);
        if (want_derivs) { to_parent(bary, lnx); to_parent(bary, lny); }
    }

    float tex[3] = {1.0f, 1.0f, 1.0f};
    if (SPEC || shader == 0 || shader == 1) {
        if (SPEC || texture >= 0) {
            float u = interp(lam, v0.uv[0], v1.uv[0], v2.uv[0]), v = interp(lam, v0.uv[1], v1.uv[1], v2.uv[1]);
            float unx = interp(lnx, v0.uv[0], v1.uv[0], v2.uv[0]), vnx = interp(lnx, v0.uv[1], v1.uv[1], v2.uv[1]);
            float uny = interp(lny, v0.uv[0], v1.uv[0], v2.uv[0]), vny = interp(lny, v0.uv[1], v1.uv[1], v2.uv[1]);
            if (RND_PROBE(fp, 512u)) { u = 0.5f; v = 0.5f; unx = 0.5001f; vnx = 0.5f; uny = 0.5f; vny = 0.5001f; } /* every lane samples the same texels */
            /* toward the quad neighbour, which lies on either side: the sampler squares these, so the direction
             * does not reach the result (a difference and its negative have the same square) */
            const float dudx = unx - u, dvdx = vnx - v, dudy = uny - u, dvdy = vny - v;
            if constexpr (SPEC) sample_texture<true, true>(tb, fp.pass_tex, pass_levels, u, v, dudx, dvdx, dudy, dvdy, tex);
            else if (fp.pass_texture >= 0) sample_texture<true>(tb, fp.pass_tex, pass_levels, u, v, dudx, dvdx, dudy, dvdy, tex);
            else sample_texture<false>(tb, fp.textures[texture], nullptr, u, v, dudx, dvdx, dudy, dvdy, tex);
        } else {
            tex[0] = tex[1] = tex[2] = 0.0f;
        }
    }
    if (shader == 0) { rgba[0] = tex[0]; rgba[1] = tex[1]; rgba[2] = tex[2]; rgba[3] = 1.0f; return; }
    const float diff = fmaxf(interp(lam, v0.light, v1.light, v2.light), 0.0f);
    if (SPEC || shader == 1) {
        rgba[0] = tex[0] * diff; rgba[1] = tex[1] * diff; rgba[2] = tex[2] * diff;
    } else {
        /* v_color = push colour at every corner, interpolated like any other varying */
        const float* col = fp.draws[draw].push;
        float c0 = interp(lam, col[0], col[0], col[0]), c1 = interp(lam, col[1], col[1], col[1]),
              c2 = interp(lam, col[2], col[2], col[2]);
        rgba[0] = c0 * diff; rgba[1] = c1 * diff; rgba[2] = c2 * diff;
    }
    rgba[3] = 1.0f;
}

/* shader and texture of a draw: from the kernel arguments when every draw of the pass agrees, which
 * saves two dependent loads per pixel; otherwise from the draw's record */
__device__ __forceinline__ void shader_and_texture(const FrameParams& fp, uint32_t draw, int& shader, int& texture)
{
    shader = fp.pass_shader >= 0 ? fp.pass_shader : fp.draws[draw].shader;
    texture = -1;
    if (shader == 0 || shader == 1) texture = fp.pass_texture != -2 ? fp.pass_texture : fp.draws[draw].texture;
}

/* colour of the fragment of (slot, sub) at pixel (x, y), everything gathered from device memory */
template <bool SPEC, typename Tables>
__device__ void shade_pixel(const FrameParams& fp, const Tables& tb, LdsLevels pass_levels,
                            uint32_t slot, uint32_t sub, bool clipped, int x, int y, float* rgba)
{
    PrimRec pr;
    if (fp.direct_index && !clipped) {
        /* a pass of one draw: the slot is the triangle's number, its corners come from the index buffer */
        const uint32_t* ip = fp.direct_ib + (size_t)slot * 3;
        pr.v0 = fp.direct_svtx_base + load_u32(ip); pr.v1 = fp.direct_svtx_base + load_u32(ip + 1); pr.v2 = fp.direct_svtx_base + load_u32(ip + 2);
        pr.draw_flags = 0;
    } else {
        pr = fp.prim_rec[slot];
    }
    uint32_t vtx[3] = {pr.v0, pr.v1, pr.v2};
    if (RND_PROBE(fp, 256u)) { vtx[0] = 0u; vtx[1] = 1u; vtx[2] = 2u; } /* every lane gathers the same three vertices: what the gathers' spread costs */
    uint32_t draw = pr.draw_flags & SLOT_MASK;
    if (clipped) {
        const ClipTri* ct = &fp.clip_tris[pr.v0 + sub];
        draw = ct->draw;
        vtx[0] = ct->v[0]; vtx[1] = ct->v[1]; vtx[2] = ct->v[2];
    }
    int shader, texture;
    if constexpr (SPEC) { shader = fp.pass_shader; texture = fp.pass_texture; }
    else shader_and_texture(fp, draw, shader, texture);
    Varyings v0, v1, v2;
    load_varyings(fp, vtx[0], v0);
    load_varyings(fp, vtx[1], v1);
    load_varyings(fp, vtx[2], v2);
    Corners c;
    const float* bary = nullptr;
    if (!clipped) {
        /* each corner as one 16-byte load, said explicitly: left as member reads, the compiler folds them with the
         * clipped branch's reads into nine single-word loads through selected pointers */
        typedef uint32_t Record __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) Record* GlobalRecord;
        const Record p0 = *(GlobalRecord)&fp.svtx[vtx[0]], p1 = *(GlobalRecord)&fp.svtx[vtx[1]], p2 = *(GlobalRecord)&fp.svtx[vtx[2]];
        c.x[0] = (int32_t)p0.x >> 8; c.y[0] = (int32_t)p0.y; c.iw[0] = __uint_as_float(p0.w);
        c.x[1] = (int32_t)p1.x >> 8; c.y[1] = (int32_t)p1.y; c.iw[1] = __uint_as_float(p1.w);
        c.x[2] = (int32_t)p2.x >> 8; c.y[2] = (int32_t)p2.y; c.iw[2] = __uint_as_float(p2.w);
    } else {
        const ClipTri* ct = &fp.clip_tris[pr.v0 + sub];
#pragma unroll
        for (int k = 0; k < 3; ++k) { c.x[k] = ct->x[k]; c.y[k] = ct->y[k]; c.iw[k] = ct->iw[k]; }
        bary = ct->bary;
    }
    shade_fragment<SPEC>(fp, tb, pass_levels, draw, shader, texture, c, v0, v1, v2, bary, x, y, rgba);
}

template <typename Tables>
__device__ __forceinline__ uint32_t pack_color(const FrameParams& fp, const Tables& tb, const float* rgba)
{
    uint32_t r, g, b;
    if (fp.color_srgb) { r = srgb_encode(tb, rgba[0]); g = srgb_encode(tb, rgba[1]); b = srgb_encode(tb, rgba[2]); }
    else { r = unorm8(rgba[0]); g = unorm8(rgba[1]); b = unorm8(rgba[2]); }
    uint32_t a = unorm8(rgba[3]);
    return fp.color_bgra ? (b | (g << 8) | (r << 16) | (a << 24)) : (r | (g << 8) | (b << 16) | (a << 24));
}

/* Streams n entries through the rasteriser in batches of one entry per lane, as a three-stage
 * software pipeline: while batch b is rasterised, the vertices of batch b+1 and the entries of
 * batch b+2 are already in flight, so a batch does not wait for two dependent memory round trips. */
template <int S, typename Fetch>
__device__ __forceinline__ void raster_stream(const FrameParams& fp, const TileCtx& tc, uint32_t n, Fetch fetch)
{
    if (n == 0) return;
    /* entry of a thread in the batch that starts at `base`: consecutive (coalesced loads) unless
     * that would leave a whole wave idle; such a batch is dealt round-robin to the four waves */
    auto entry_of = [&](uint32_t base) -> uint32_t {
        if (base >= n) return n;
        if (n - base > (uint32_t)(TILE_THREADS - 64)) return base + threadIdx.x < n ? base + threadIdx.x : n;
        const uint32_t j = (threadIdx.x & 63u) * (TILE_THREADS / 64u) + (threadIdx.x >> 6);
        return j < n - base ? base + j : n;
    };
    TileEntry e_in = {0, 0, 0, 0}, e_mid = {0, 0, 0, 0};
    RasterTri t_cur, t_next;
    bool h_cur = false, h_next = false, h_mid = false, h_in = false;
    {
        TileEntry e0;
        const uint32_t i0 = entry_of(0), i1 = entry_of(TILE_THREADS);
        h_cur = i0 < n && fetch(i0, e0);
        if (h_cur) h_cur = load_raster_tri(fp, e0, t_cur);
        h_mid = i1 < n && fetch(i1, e_mid);
    }
    for (uint32_t base = 0; base < n; base += TILE_THREADS) {
        const uint32_t i2 = entry_of(base + 2u * TILE_THREADS);
        h_in = i2 < n && fetch(i2, e_in);
        h_next = h_mid;
        if (h_next) h_next = load_raster_tri(fp, e_mid, t_next);
        raster_batch<S>(fp, tc, h_cur, t_cur);
        t_cur = t_next; h_cur = h_next;
        e_mid = e_in; h_mid = h_in;
    }
}

/* The single-sample kernel's form of the same pipeline, with the data in flight held in LDS instead of registers.
 *
 * A thread's entry of batch b+2 and its three corners of batch b+1 are fetched by loads that write LDS directly
 * (global_load_lds_dwordx4: per-lane source address, destination the wave's 1 KiB piece of the staging array plus
 * 16 bytes per lane), so nothing is held in registers while the batch b is rasterised, nothing of it can be spilled,
 * and the only waits are the two written here.  (With the loads' destinations in registers the compiler waits for
 * the corners right where they are issued, to shift x into place, and carries the prefetched entry over the loop's
 * edge through scratch memory, which makes every batch wait for both round trips.)  The tile's own list and the
 * pass-wide list run through the one loop as entries 0 .. n-1 and n .. n+n_large-1. */
struct StageArrays {
    uint4* entry;        /* [TILE_THREADS] */
    uint4* corner;       /* [3][TILE_THREADS] */
};

/* 16 bytes per lane from each lane's own address into the wave's piece of an LDS array (byte address lds_piece,
 * the same for the whole wave; lane l's bytes land at lds_piece + 16 l).  Written as an assembly statement so that
 * the compiler does not count the copy: knowing of it, it waits for every copy in flight before each LDS atomic of
 * the rasteriser, which it cannot tell from the staging arrays.  Completion is awaited by wait_for_async_copies(). */
__device__ __forceinline__ void async_copy16(const void* src, uint32_t lds_piece)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_piece) : "memory");
}

/* three such copies into three pieces `step` bytes apart, M0 saved and restored once */
__device__ __forceinline__ void async_copy16x3(const void* src0, const void* src1, const void* src2, uint32_t lds_piece, uint32_t step)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_add_u32 m0, m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_add_u32 m0, m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src0), "v"(src1), "v"(src2), "s"(lds_piece), "s"(step) : "memory", "scc");
}

__device__ __forceinline__ uint32_t lds_byte_address(const void* p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

__device__ __forceinline__ void wait_for_async_copies() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_for_lds_reads() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void raster_lists_staged(const FrameParams& fp, const TileCtx& tc, const StageArrays& st, const TileEntry* list,
                                                    uint32_t n_list, uint32_t n_large, int tx, int ty)
{
    const uint32_t n = n_list + n_large;
    if (n == 0) return;
    const uint32_t lane_slot = threadIdx.x, wave_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
    const uint32_t entry_piece = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_byte_address(st.entry + wave_first));
    const uint32_t corner_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_byte_address(st.corner + wave_first));
    /* entry of a thread in the batch that starts at `base`: consecutive (coalesced loads) unless
     * that would leave a whole wave idle; such a batch is dealt round-robin to the four waves */
    auto entry_of = [&](uint32_t base) -> uint32_t {
        if (base >= n) return n;
        if (n - base > (uint32_t)(TILE_THREADS - 64)) return base + threadIdx.x < n ? base + threadIdx.x : n;
        const uint32_t j = (threadIdx.x & 63u) * (TILE_THREADS / 64u) + (threadIdx.x >> 6);
        return j < n - base ? base + j : n;
    };
    auto fetch_entry = [&](uint32_t i) {
        const void* src = i < n_list ? (const void*)(list + i) : (const void*)&fp.large_list[i - n_list].e;
        async_copy16(src, entry_piece);
    };
    /* an entry of the pass-wide list applies to a box of tiles */
    auto applies = [&](uint32_t i) -> bool {
        if (i < n_list) return true;
        const uint32_t box = fp.large_list[i - n_list].box;
        const int bx0 = box & 255u, by0 = (box >> 8) & 255u, bx1 = (box >> 16) & 255u, by1 = box >> 24;
        return tx >= bx0 && tx <= bx1 && ty >= by0 && ty <= by1;
    };
    /* the three 16-byte pieces that hold a triangle's corners: its svtx records, or the head of its ClipTri */
    auto fetch_corners = [&](const uint4& e) {
        const uint8_t *p0, *p1, *p2;
        if (e.w & FLAG_CLIPPED) {
            p0 = reinterpret_cast<const uint8_t*>(&fp.clip_tris[e.x]); p1 = p0 + 16; p2 = p0 + 32;
        } else {
            p0 = reinterpret_cast<const uint8_t*>(&fp.svtx[e.x]); p1 = reinterpret_cast<const uint8_t*>(&fp.svtx[e.y]);
            p2 = reinterpret_cast<const uint8_t*>(&fp.svtx[e.z]);
        }
        async_copy16x3(p0, p1, p2, corner_first, TILE_THREADS * 16u);
    };
    uint32_t flags_cur = 0, i_mid;
    bool h_cur, h_mid;
    {
        const uint32_t i0 = entry_of(0);
        if (i0 < n) fetch_entry(i0);
        wait_for_async_copies();
        const uint4 e = st.entry[lane_slot];
        wait_for_lds_reads();
        h_cur = i0 < n && applies(i0);
        if (h_cur) fetch_corners(e);
        flags_cur = e.w;
        i_mid = entry_of(TILE_THREADS);
        h_mid = i_mid < n;
        if (h_mid) fetch_entry(i_mid);
    }
    for (uint32_t base = 0; base < n; base += TILE_THREADS) {
        wait_for_async_copies(); /* the corners of this batch and the entry of the next have landed */
        const uint4 c0 = st.corner[lane_slot], c1 = st.corner[TILE_THREADS + lane_slot], c2 = st.corner[2 * TILE_THREADS + lane_slot];
        const uint4 e_mid = st.entry[lane_slot];
        wait_for_lds_reads();
        const bool h_next = h_mid && applies(i_mid);
        if (h_next) fetch_corners(e_mid);
        const uint32_t i_in = entry_of(base + 2u * TILE_THREADS);
        const bool h_in = i_in < n;
        if (h_in) fetch_entry(i_in);
        RasterTri t;
        t.slot = flags_cur & SLOT_MASK;
        if (flags_cur & FLAG_CLIPPED) {
            /* ClipTri: x[3], y[3], z[3] in its first nine words */
            t.x[0] = (int32_t)c0.x; t.x[1] = (int32_t)c0.y; t.x[2] = (int32_t)c0.z;
            t.y[0] = (int32_t)c0.w; t.y[1] = (int32_t)c1.x; t.y[2] = (int32_t)c1.y;
            t.z[0] = __uint_as_float(c1.z); t.z[1] = __uint_as_float(c1.w); t.z[2] = __uint_as_float(c2.x);
            t.payload_lo = ((flags_cur >> ENTRY_SUB_SHIFT) & 7u) | 8u;
        } else {
            t.x[0] = (int32_t)c0.x >> 8; t.y[0] = (int32_t)c0.y; t.z[0] = __uint_as_float(c0.z);
            t.x[1] = (int32_t)c1.x >> 8; t.y[1] = (int32_t)c1.y; t.z[1] = __uint_as_float(c1.z);
            t.x[2] = (int32_t)c2.x >> 8; t.y[2] = (int32_t)c2.y; t.z[2] = __uint_as_float(c2.z);
            t.payload_lo = 0;
        }
        raster_batch<1>(fp, tc, h_cur, t);
        h_cur = h_next; flags_cur = e_mid.w; h_mid = h_in; i_mid = i_in;
    }
}

/* sRGB tables global -> LDS, as copies that write LDS directly: a wave moves 1 KiB per instruction and holds no
 * register for it; the copies run under whatever the caller does next and are awaited (wait_for_async_copies, then
 * the workgroup's barrier) before the tables are read */
__device__ __forceinline__ void stage_tables(ShadeTables& tb, const ShadeTables* src, bool swap_values = false)
{
    static_assert(sizeof(ShadeTables) % 16 == 0, "tables are staged in 16-byte pieces");
    constexpr uint32_t PIECES = sizeof(ShadeTables) / 16;
    const uint32_t lds_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_byte_address(&tb));
    const uint32_t wave_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
    const uint8_t* g = reinterpret_cast<const uint8_t*>(src);
    for (uint32_t first = wave_first; first < PIECES; first += blockDim.x) { /* the same for the whole wave */
        const uint32_t i = first + (threadIdx.x & 63u);
        /* swap_values: the two texel tables change places (64 pieces each, the first 128 pieces of the structure) */
        const uint32_t from = (swap_values && i < 128u) ? (i ^ 64u) : i;
        if (i < PIECES) async_copy16(g + (size_t)from * 16u, lds_first + first * 16u);
    }
}

/* the same for the tables of a pass whose texels all decode through table `which`: that table as value[0], then the
 * encoder's thresholds and buckets, which lie one table further on in the source */
__device__ __forceinline__ void stage_tables(PassTables& tb, const ShadeTables* src, bool second_table)
{
    constexpr uint32_t PIECES = sizeof(PassTables) / 16;
    const uint32_t lds_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_byte_address(&tb));
    const uint32_t wave_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
    const uint8_t* g = reinterpret_cast<const uint8_t*>(src);
    for (uint32_t first = wave_first; first < PIECES; first += blockDim.x) {
        const uint32_t i = first + (threadIdx.x & 63u);
        const uint32_t from = i < 64u ? (second_table ? i + 64u : i) : i + 64u;
        if (i < PIECES) async_copy16(g + (size_t)from * 16u, lds_first + first * 16u);
    }
}

/* one word into an image at a 32-bit byte offset: the store takes the image's address as its scalar base and the offset
 * as it is, where an index into a 64-bit pointer costs a 64-bit multiply-add per pixel */
__device__ __forceinline__ void store_word(void* image, uint32_t byte_offset, uint32_t value)
{
    typedef __attribute__((address_space(1))) uint32_t* GlobalWord;
    *(GlobalWord)(static_cast<uint8_t*>(image) + byte_offset) = value;
}

constexpr uint32_t VIS_NONE = 0xFFFFFFFFu; /* owner word of a sample no fragment covers; otherwise slot | fan << 28 | clipped << 31 */

/* Colour of pixel (x, y), whose samples are owned by vis[0..S): shade each distinct owner once,
 * resolve, encode, store. */
template <int S, bool SPEC, typename Tables>
__device__ __forceinline__ void shade_and_store(const FrameParams& fp, const Tables& tb, LdsLevels pass_levels,
                                                int x, int y, uint32_t pi, const uint32_t* vis)
{
    /* no fragment landed on the pixel: the clear colour, encoded once on the host */
    bool untouched = true;
#pragma unroll
    for (int s = 0; s < S; ++s) untouched = untouched && vis[s] == VIS_NONE;
    if (untouched) { store_word(fp.color, pi << 2, fp.clear_packed); return; }
    float smp[S][4];
    if constexpr (S == 1) {
        if (RND_PROBE(fp, 1u)) {
            smp[0][0] = smp[0][1] = smp[0][2] = smp[0][3] = 0.5f;
        } else {
            shade_pixel<SPEC>(fp, tb, pass_levels, vis[0] & SLOT_MASK, (vis[0] >> ENTRY_SUB_SHIFT) & 7u, (vis[0] & FLAG_CLIPPED) != 0u, x, y, smp[0]);
        }
    } else {
        /* Several samples: one shading per distinct owner, as a loop with ONE copy of the shading code.  Each
         * turn takes the first sample not yet coloured, shades its owner and colours every sample that owner
         * holds; a wave runs as many turns as its pixel with the most owners needs (unrolled per sample, the
         * code is four times the size and a wave runs a copy for every sample index at which any of its
         * pixels meets a new owner). */
        uint32_t todo = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (vis[s] == VIS_NONE) {
                smp[s][0] = fp.clear_color[0]; smp[s][1] = fp.clear_color[1]; smp[s][2] = fp.clear_color[2]; smp[s][3] = fp.clear_color[3];
            } else {
                smp[s][0] = smp[s][1] = smp[s][2] = smp[s][3] = 0.0f;
                todo |= 1u << s;
            }
        }
#pragma unroll 1
        while (todo) {
            uint32_t owner = 0;
#pragma unroll
            for (int s = S - 1; s >= 0; --s) if (todo & (1u << s)) owner = vis[s]; /* the first sample still to do */
            float c[4];
            shade_pixel<SPEC>(fp, tb, pass_levels, owner & SLOT_MASK, (owner >> ENTRY_SUB_SHIFT) & 7u, (owner & FLAG_CLIPPED) != 0u, x, y, c);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if ((todo & (1u << s)) && vis[s] == owner) {
                    smp[s][0] = c[0]; smp[s][1] = c[1]; smp[s][2] = c[2]; smp[s][3] = c[3];
                    todo &= ~(1u << s);
                }
            }
        }
    }
    float acc[4];
    if constexpr (S == 1) {
        acc[0] = smp[0][0]; acc[1] = smp[0][1]; acc[2] = smp[0][2]; acc[3] = smp[0][3];
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = ((smp[0][c] + smp[1][c]) + (smp[2][c] + smp[3][c])) * 0.25f;
    }
    store_word(fp.color, pi << 2, pack_color(fp, tb, acc));
}

/* Several samples per pixel: one shading per distinct owner among a pixel's samples, and the wave's shadings shared out
 * among its lanes.  A lane first shades its own pixel's first owner.  What is left -- a pixel on an edge has one more owner,
 * one on a vertex of the mesh up to three more -- is numbered over the wave (a scan of the counts), and lane l of each
 * further turn takes item 64 (turn - 1) + l: it finds the lane that holds the item (as raster_chunks finds a chunk's
 * triangle), fetches that lane's pixel and owner, shades, and the holder fetches the colour back.  A wave then runs
 * 1 + ceil(items / 64) turns of shading instead of as many as its pixel with the most owners has: a mesh of pixel-sized
 * triangles has a pixel with four owners in almost every wave and 1.9 owners per pixel on average.
 *
 * p: the pixel's number in the tile; every lane of the wave calls this (inside: the pixel lies in the render area). */
template <int S, bool SPEC, typename Tables>
__device__ __forceinline__ void shade_shared_and_store(const FrameParams& fp, const Tables& tb, LdsLevels pass_levels, const TileCtx& tc,
                                                       uint32_t p, bool inside, uint32_t pi, const uint32_t* vis)
{
    static_assert(S == 4, "a pixel has at most three owners beyond its first");
    const uint32_t lane = lane_id();
    /* the distinct owners in the order of their first samples */
    uint32_t own[S], n = 0;
#pragma unroll
    for (int k = 0; k < S; ++k) own[k] = VIS_NONE;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        bool fresh = inside && vis[s] != VIS_NONE;
#pragma unroll
        for (int q = 0; q < s; ++q) fresh = fresh && vis[q] != vis[s];
#pragma unroll
        for (int k = 0; k <= s; ++k) if (fresh && n == (uint32_t)k) own[k] = vis[s];
        n += fresh ? 1u : 0u;
    }
    const int extra = n > 1u ? (int)n - 1 : 0;
    const int incl = wave_inclusive_scan(extra, 0, [](int a, int b) { return a + b; });
    const int excl = incl - extra;
    const int total = __builtin_amdgcn_readlane(incl, 63);
    const int turns = 1 + (total + 63) / 64;
    float col[S][4];
#pragma unroll
    for (int k = 0; k < S; ++k) col[k][0] = col[k][1] = col[k][2] = col[k][3] = 0.0f;
#pragma unroll 1
    for (int turn = 0; turn < turns; ++turn) {
        uint32_t at = p, owner = own[0];
        bool valid = n >= 1u;
        const int first_item = (turn - 1) * 64;
        if (turn > 0) {
            const int item = first_item + (int)lane;
            volatile signed char* row = tc.wave_row; /* lanes talk to each other through it: every access is to happen as written */
            row[lane] = -1;
            if (extra > 0 && excl >= first_item && excl < first_item + 64) row[excl - first_item] = (signed char)lane;
            const unsigned long long before = __ballot(extra > 0 && excl <= first_item);
            const int carried = before ? 63 - __builtin_clzll(before) : 0;
            const int o = max(wave_inclusive_scan((int)row[lane], -1, [](int a, int b) { return max(a, b); }), carried);
            const int k = item - __shfl(excl, o); /* which of the holder's further owners */
            const uint32_t o1 = (uint32_t)__shfl((int)own[1], o), o2 = (uint32_t)__shfl((int)own[2], o), o3 = (uint32_t)__shfl((int)own[3], o);
            at = (uint32_t)__shfl((int)p, o);
            owner = k == 0 ? o1 : (k == 1 ? o2 : o3);
            valid = item < total;
        }
        float c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (valid)
            shade_pixel<SPEC>(fp, tb, pass_levels, owner & SLOT_MASK, (owner >> ENTRY_SUB_SHIFT) & 7u, (owner & FLAG_CLIPPED) != 0u,
                              tc.px0 + (int)(at & (TILE_W - 1)), tc.py0 + (int)(at >> TILE_SHIFT), c);
        if (turn == 0) {
            col[0][0] = c[0]; col[0][1] = c[1]; col[0][2] = c[2]; col[0][3] = c[3];
        } else {
#pragma unroll
            for (int kk = 1; kk < S; ++kk) {
                const int j = excl + (kk - 1) - first_item; /* the lane that shaded this lane's owner kk, if it was this turn */
                const bool here = kk <= extra && j >= 0 && j < 64;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float got = __shfl(c[q], j & 63);
                    if (here) col[kk][q] = got;
                }
            }
        }
    }
    if (!inside) return;
    /* no fragment landed on the pixel: the clear colour, encoded once on the host */
    if (n == 0u) { store_word(fp.color, pi << 2, fp.clear_packed); return; }
    float smp[S][4];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        smp[s][0] = fp.clear_color[0]; smp[s][1] = fp.clear_color[1]; smp[s][2] = fp.clear_color[2]; smp[s][3] = fp.clear_color[3];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            if ((uint32_t)k < n && vis[s] == own[k]) { smp[s][0] = col[k][0]; smp[s][1] = col[k][1]; smp[s][2] = col[k][2]; smp[s][3] = col[k][3]; }
        }
    }
    float acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = ((smp[0][c] + smp[1][c]) + (smp[2][c] + smp[3][c])) * 0.25f;
    store_word(fp.color, pi << 2, pack_color(fp, tb, acc));
}

/* One workgroup per 32x32 tile, in two phases.  Rasterise the tile's triangles into one 64-bit
 * word per sample held in LDS (depth, submission order, fan index; the minimum wins).  Then every
 * thread takes the tile's pixels one row of eight at a time: depth word and owner of each sample
 * come out of LDS, the owner is shaded (once per distinct fan triangle), the samples are resolved,
 * encoded and stored; lanes run along x, so depth and colour stores cover whole 128-byte segments.
 * Workgroups in the two phases share a CU: one's arithmetic fills the other's memory waits. */
#ifndef TILE_WAVES
#define TILE_WAVES 5 /* waves per SIMD the single-sample kernel is compiled for (96 VGPRs): the walk wants about 110, and below 5 the spills cost more than the occupancy gains */
#endif
template <int S, int TH, bool SPEC>
__global__ __launch_bounds__(TILE_THREADS, (S == 1 && (TH == TILE_H || SPEC) ? TILE_WAVES : 4)) void k_tile(FrameParams fp, uint32_t magic) /* 40 rows of sample words leave LDS for five workgroups per CU only without the second texel table */
{
    constexpr int TILE_PIXELS = TILE_W * TH; /* TH: rows of pixels per tile (FrameParams::tile_h) */
    __shared__ unsigned long long keys[TILE_PIXELS * S];
    __shared__ typename std::conditional<SPEC, PassTables, ShadeTables>::type tb;
    __shared__ LevelRec pass_levels[MAX_TEXTURE_LEVELS]; /* the levels of the pass-wide texture, indexable per lane */
    __shared__ uint4 stage_entry[S == 1 ? TILE_THREADS : 1], stage_corner[S == 1 ? 3 * TILE_THREADS : 1];
    __shared__ signed char wave_rows[TILE_THREADS];
    uint32_t tile, utx, uty;
    if (fp.shard_count == 1) {
        /* the grid covers the tiles the render area touches */
        rows_of_eight(blockIdx.x, fp.grid_w, magic, utx, uty);
        if (uty >= fp.grid_h) return; /* padding of the grid */
        utx += fp.grid_x0; uty += fp.grid_y0;
        tile = uty * fp.tiles_x + utx;
    } else {
        tile = blockIdx.x * fp.shard_count + fp.shard_index; /* this shard's tiles, in order */
        utx = tile % fp.tiles_x; uty = tile / fp.tiles_x;
    }
    const int tx = (int)utx, ty = (int)uty;
    TileCtx tc;
    tc.px0 = tx * TILE_W; tc.py0 = ty * TH;
    tc.rx0 = max(tc.px0, fp.clamp_x0); tc.ry0 = max(tc.py0, fp.clamp_y0);
    tc.rx1 = min(tc.px0 + TILE_W - 1, fp.clamp_x1); tc.ry1 = min(tc.py0 + TH - 1, fp.clamp_y1);
    tc.keys = keys;
    tc.wave_row = wave_rows + (threadIdx.x & ~63u);
    RND_STAMP_START(stamp);

    /* the length of the tile's list is asked for first: its round trip runs under the staging of
     * the tables and the clearing of the tile's words */
    const bool visible = tc.rx0 <= tc.rx1 && tc.ry0 <= tc.ry1;
    const uint32_t listed = visible ? fp.tile_count[(size_t)tile * TILE_COUNT_STRIDE] : 0u;
    const uint32_t listed_large = visible ? fp.counters[CTR_LARGE] : 0u;
    /* The texel table of the pass-wide texture is staged as value[0] whichever of the two it is: its address in LDS is
     * then a constant of the kernel, which the table reads carry in their offset field -- a channel's lookup costs the
     * shift that scales its byte (taken out of the texel by the instruction's operand selector) and the read. */
    static_assert(offsetof(ShadeTables, value) == 0 && sizeof(float[2][256]) == 128 * 16, "stage_tables swaps the first 2 x 64 pieces");
    stage_tables(tb, fp.tables, fp.pass_texture >= 0 && fp.pass_tex.srgb == 0u);
    if (threadIdx.x < (uint32_t)MAX_TEXTURE_LEVELS) {
        const int lw = (int)level_dim(fp.pass_tex.width, threadIdx.x), lh = (int)level_dim(fp.pass_tex.height, threadIdx.x);
        LevelRec r;
        r.base = reinterpret_cast<unsigned long long>(fp.pass_tex.level[threadIdx.x]);
        r.fw = (float)lw; r.fh = (float)lh; r.w = lw; r.wm1 = lw - 1; r.hm1 = lh - 1; r.wm2 = max(lw - 2, 0);
        pass_levels[threadIdx.x] = r;
    }
    const unsigned long long clear_key = ((unsigned long long)(fp.depth_test ? fp.clear_d24 : 0xFFFFFFu) << 40) |
                                         (fp.depth_less ? 0ull : 0xFFFFFFFFFFull);
    for (uint32_t i = threadIdx.x; i < TILE_PIXELS * S; i += TILE_THREADS) keys[i] = clear_key;
    wait_for_async_copies(); /* the tables */
    __syncthreads();
    RND_STAMP(fp, stamp, 0); /* prologue */

    if (visible && !RND_PROBE(fp, 2u)) {
        /* this tile's list: 16-byte entries, one per lane per batch */
        uint32_t count = min(listed, fp.tile_capacity);
        if (RND_PROBE(fp, 16u)) count = min(count, 256u);
        if (RND_PROBE(fp, 64u)) count = min(count, 512u);
        const TileEntry* list = fp.tile_lists + (size_t)tile * fp.tile_stride;
        /* the pass-wide list: primitives spanning many tiles, and spill of full tile lists */
        const uint32_t n_large = min(listed_large, fp.large_capacity);
        if constexpr (S == 1) {
            StageArrays st;
            st.entry = stage_entry; st.corner = stage_corner;
            raster_lists_staged(fp, tc, st, list, count, n_large, tx, ty);
        } else {
            raster_stream<S>(fp, tc, count, [&](uint32_t i, TileEntry& e) -> bool { e = list[i]; return true; });
            raster_stream<S>(fp, tc, n_large, [&](uint32_t i, TileEntry& e) -> bool {
                const LargeEntry le = fp.large_list[i];
                const int bx0 = le.box & 255u, by0 = (le.box >> 8) & 255u, bx1 = (le.box >> 16) & 255u, by1 = le.box >> 24;
                e = le.e;
                return tx >= bx0 && tx <= bx1 && ty >= by0 && ty <= by1;
            });
        }
    }
    RND_STAMP(fp, stamp, 1); /* rasterisation */
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t ovf = fp.counters[CTR_OVERFLOW];
        /* a list overflowed: the host reports it at the next wait.  The flags go out with their complement beside them: the
         * host takes the pair for a report only when the two agree, so nothing else that might land in that pinned word
         * reads as one.  One aligned 64-bit store: the host, which may look while another frame in flight reports, never
         * sees the flags of one report beside the complement of none */
        if (ovf) *reinterpret_cast<volatile unsigned long long*>(fp.status_host) = ((unsigned long long)~ovf << 32) | ovf;
    }
    __syncthreads();
    /* every thread has read the list length: leave the counter clean for the next frame's binning */
    if (threadIdx.x == 0) fp.tile_count[(size_t)tile * TILE_COUNT_STRIDE] = 0;
    RND_STAMP(fp, stamp, 2); /* waiting for the other waves' rasterisation */

    static_assert(TILE_PIXELS % TILE_THREADS == 0, "every lane of a wave runs the same number of turns of the loop below");
    for (uint32_t p = threadIdx.x; p < TILE_PIXELS; p += TILE_THREADS) {
        const int x = tc.px0 + (int)(p & (TILE_W - 1)), y = tc.py0 + (int)(p >> TILE_SHIFT);
        const bool inside = !(x < fp.area_x0 || x > fp.area_x1 || y < fp.area_y0 || y > fp.area_y1);
        if (S == 1 && !inside) continue; /* several samples: the wave shares its shadings out, every lane goes along */
        const uint32_t pi = (uint32_t)y * fp.width + (uint32_t)x; /* a target is at most 8192 x 8192 pixels of 4 samples: byte offsets fit 32 bits */
        uint32_t vis[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            vis[s] = VIS_NONE;
            if (!inside) continue;
            const unsigned long long k = keys[p * S + s];
            store_word(fp.depth, (pi * S + s) << 2, fp.depth_test ? (uint32_t)(k >> 40) : fp.clear_d24);
            if (k != clear_key) {
                const uint32_t low = (uint32_t)(k >> 8);
                const uint32_t fan_clipped = (uint32_t)(k & 15ull) ^ (fp.depth_less ? 0u : 7u); /* the fan index as the rasteriser complemented it */
                vis[s] = ((fp.depth_less ? low : ~low) & SLOT_MASK) | (fan_clipped << ENTRY_SUB_SHIFT);
            }
        }
        if constexpr (S == 1) shade_and_store<S, SPEC>(fp, tb, (LdsLevels)pass_levels, x, y, pi, vis);
        else shade_shared_and_store<S, SPEC>(fp, tb, (LdsLevels)pass_levels, tc, p, inside, pi, vis);
    }
    RND_STAMP(fp, stamp, 8); /* shading */
    RND_STAMP_END(fp);
}

/* ------------------------------------------------------------------ mip generation (blit LINEAR, tools.cpp:670-707) */

__global__ __launch_bounds__(256) void k_mip(const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh,
                                             int srgb, const ShadeTables* tables)
{
    __shared__ ShadeTables tb;
    stage_tables(tb, tables);
    wait_for_async_copies();
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
            float c00 = tb.value[lin][b00];
            float c10 = tb.value[lin][b10];
            float c01 = tb.value[lin][b01];
            float c11 = tb.value[lin][b11];
            float top = fma32(ax, c10 - c00, c00);
            float bot = fma32(ax, c11 - c01, c01);
            float c = fma32(ay, bot - top, top);
            out |= (lin ? unorm8(c) : srgb_encode(tb, c)) << (8 * k);
        }
        reinterpret_cast<uint32_t*>(dst)[i] = out;
    }
}

/* ------------------------------------------------------------------ the pass that allocates nothing: set-up */

/* One workgroup: the draw's record goes where the frame's kernels expect it, and the words a frame assumes to be zero
 * at its start are zeroed (the array keeps whatever the previous call left). */
__global__ __launch_bounds__(256) void k_noalloc_begin(DrawStatic d, uint8_t* base, uint32_t n_tiles)
{
    if (threadIdx.x == 0) *reinterpret_cast<DrawStatic*>(base + NoAllocLayout::draw) = d;
    uint32_t* derived = reinterpret_cast<uint32_t*>(base + NoAllocLayout::derived);
    if (threadIdx.x < sizeof(DrawDerived) / 4) derived[threadIdx.x] = 0u;
    uint32_t* counters = reinterpret_cast<uint32_t*>(base + NoAllocLayout::counters);
    if (threadIdx.x < CTR_FRAME_COUNT) counters[threadIdx.x] = 0u;
    uint32_t* tile_count = reinterpret_cast<uint32_t*>(base + NoAllocLayout::tile_count);
    for (uint32_t t = threadIdx.x; t < n_tiles; t += blockDim.x) tile_count[(size_t)t * TILE_COUNT_STRIDE] = 0u;
}

void launch_noalloc_begin(const DrawStatic& d, uint8_t* base, uint32_t n_tiles, hipStream_t s)
{
    hipLaunchKernelGGL(k_noalloc_begin, dim3(1), dim3(256), 0, s, d, base, n_tiles);
}

/* ------------------------------------------------------------------ launch wrappers */

void launch_index_range(const IndexRangeJob* jobs, uint32_t* out_max, uint32_t n_jobs, hipStream_t s)
{
    if (!n_jobs) return;
    hipLaunchKernelGGL(k_index_range, dim3(64, n_jobs), dim3(256), 0, s, jobs, out_max);
}

void launch_bounds(const BoundsJob* jobs, uint32_t* out_min_max, uint32_t n_jobs, hipStream_t s)
{
    if (!n_jobs) return;
    hipLaunchKernelGGL(k_bounds, dim3(64, n_jobs), dim3(256), 0, s, jobs, out_min_max);
}

/* RND_LIST_WALK_BLOCKS=n: the grid that walks a list of visible chunks has at most n workgroups (for A/B runs) */
static uint32_t list_walk_blocks()
{
    static const uint32_t v = [] { const char* e = getenv("RND_LIST_WALK_BLOCKS"); return e && atoi(e) > 0 ? (uint32_t)atoi(e) : LIST_WALK_BLOCKS; }();
    return v;
}

void launch_vertex(const FrameParams& fp, const UniformPatch& patch, hipStream_t s)
{
    const uint32_t blocks = (fp.n_vchunks + 3) / 4;
    if (!blocks) return;
    VertexArgs args;
    args.fp = fp; args.up = patch;
    if (fp.compact) {
        hipLaunchKernelGGL(k_draws, dim3((fp.n_draws + DRAWS_PER_BLOCK - 1) / DRAWS_PER_BLOCK), dim3(DRAWS_THREADS), 0, s, args);
        hipLaunchKernelGGL((k_vertex<true, true>), dim3(blocks < list_walk_blocks() ? blocks : list_walk_blocks()), dim3(256), 0, s, args);
    } else if (fp.n_vchunks >= PER_DRAW_KERNEL_MIN_WAVES) {
        hipLaunchKernelGGL(k_draws_by_wave, dim3((fp.n_draws + 3) / 4), dim3(256), 0, s, args);
        hipLaunchKernelGGL((k_vertex<true, false>), dim3(blocks), dim3(256), 0, s, args);
    } else {
        hipLaunchKernelGGL((k_vertex<false, false>), dim3(blocks), dim3(256), 0, s, args);
    }
}

void launch_setup(const FrameParams& fp, hipStream_t s)
{
    if (!fp.n_tchunks) return;
    uint32_t blocks = (fp.n_tchunks + 3) / 4;
    if (fp.direct_index) blocks = ((blocks + 7) / 8) * 8; /* one share per XCD */
    if (fp.compact) hipLaunchKernelGGL(k_setup<true>, dim3(blocks < list_walk_blocks() ? blocks : list_walk_blocks()), dim3(256), 0, s, fp);
    else hipLaunchKernelGGL(k_setup<false>, dim3(blocks), dim3(256), 0, s, fp);
}

/* RND_TILE_GENERIC=1: every pass takes the general instantiation of the tile kernel (for A/B runs and the tests) */
static bool getenv_no_spec()
{
    static const bool v = [] { const char* e = getenv("RND_TILE_GENERIC"); return e && atoi(e) != 0; }();
    return v;
}

void launch_tile(const FrameParams& fp, hipStream_t s, uint32_t lds_pad)
{
    const uint32_t n_tiles = fp.tiles_x * fp.tiles_y;
    const uint32_t mine = n_tiles > fp.shard_index ? (n_tiles - fp.shard_index + fp.shard_count - 1) / fp.shard_count : 0;
    if (!mine) return;
    if (fp.shard_count == 1 && (!fp.grid_w || !fp.grid_h)) return; /* empty render area */
    const uint32_t grid = fp.shard_count == 1 ? rows_of_eight_grid(fp.grid_w, fp.grid_h) : mine;
    const uint32_t magic = rows_of_eight_magic(fp.shard_count == 1 ? fp.grid_w : fp.tiles_x);
    /* the instantiation for passes of the samples' kind: one texture for every draw, sampled LINEAR with CLAMP_TO_EDGE, by the
     * dummy or the test_ktx shader */
    const bool spec = !getenv_no_spec() && fp.pass_texture >= 0 && (fp.pass_shader == 0 || fp.pass_shader == 1) && fp.pass_tex.filter_linear != 0u && fp.pass_tex.tiled == 0u;
    const dim3 g(grid), b(TILE_THREADS);
    if (getenv("RND_PRINT_OCCUPANCY")) {
        /* once per process: how many workgroups of each single-sample instantiation share a CU */
        static bool printed = false;
        if (!printed) {
            printed = true;
            int n32 = 0, n32s = 0, n40 = 0, n40s = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n32, k_tile<1, TILE_H, false>, TILE_THREADS, 0);
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n32s, k_tile<1, TILE_H, true>, TILE_THREADS, 0);
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n40, k_tile<1, TILE_H_TALL, false>, TILE_THREADS, 0);
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n40s, k_tile<1, TILE_H_TALL, true>, TILE_THREADS, 0);
            fprintf(stderr, "tile workgroups per CU: 32 rows %d, 32 rows (pass tables) %d, 40 rows %d, 40 rows (pass tables) %d\n", n32, n32s, n40, n40s);
        }
    }
    if (fp.samples == 4) {
        if (spec) hipLaunchKernelGGL((k_tile<4, TILE_H, true>), g, b, 0, s, fp, magic);
        else hipLaunchKernelGGL((k_tile<4, TILE_H, false>), g, b, 0, s, fp, magic);
    } else if (fp.tile_h == (uint32_t)TILE_H_TALL) {
        if (spec) hipLaunchKernelGGL((k_tile<1, TILE_H_TALL, true>), g, b, 0, s, fp, magic);
        else hipLaunchKernelGGL((k_tile<1, TILE_H_TALL, false>), g, b, 0, s, fp, magic);
    } else {
        if (spec) hipLaunchKernelGGL((k_tile<1, TILE_H, true>), g, b, (size_t)lds_pad, s, fp, magic);
        else hipLaunchKernelGGL((k_tile<1, TILE_H, false>), g, b, (size_t)lds_pad, s, fp, magic);
    }
}

void launch_mip(const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh, int srgb,
                const ShadeTables* tables, hipStream_t s)
{
    size_t total = (size_t)dw * dh;
    uint32_t blocks = (uint32_t)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_mip, dim3(blocks), dim3(256), 0, s, src, sw, sh, dst, dw, dh, srgb, tables);
}

} // namespace rnd
