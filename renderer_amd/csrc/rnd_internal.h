/*
 * rnd_internal.h -- structures shared by the host side of the C ABI and the
 * gfx950 kernels.  Everything here is plain data laid out for device memory.
 */
#ifndef RND_INTERNAL_H
#define RND_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rnd {

constexpr int TILE_W = 32;           /* pixels per tile side: one workgroup owns a tile's depth/owner words in LDS */
/* Rows of pixels per tile: chosen per pass (FrameParams::tile_h; the rule is in rnd_cmd_end), a multiple of 8 -- every
 * thread of a tile's workgroup shades tile_h / 8 pixels.  32 is the rule, 40 the choice for a one-draw mesh on a large target. */
constexpr int TILE_H = 32;
constexpr int TILE_H_TALL = 40;
constexpr int TILE_SHIFT = 5;        /* log2(TILE_W) */
constexpr int TILE_PIXELS_MAX = TILE_W * TILE_H_TALL;
#ifndef TILE_THREADS_N
#define TILE_THREADS_N 256
#endif
constexpr int TILE_THREADS = TILE_THREADS_N; /* a multiple of 64; a build option: 128 gains 6 % at 4K and costs 38 % at 720p, 192 and 320 lose everywhere */
constexpr int CHUNK = 64;            /* primitives (or vertices) handled by one wave */
constexpr int MAX_TILES_PER_PRIM = 16; /* primitives touching more tiles go to the pass-wide list */
/* Words 0..2 are zeroed by each frame's vertex kernel.  Four more words are the lengths of the lists of visible chunks
 * (vertex chunks, primitive chunks) that the per-draw kernel of a pass of many draws builds: two pairs, used by
 * alternate frames, so that a frame's per-draw kernel can zero the pair the next frame will count into -- each in a
 * 128-byte line of its own (visible_length), so that the atomics on the two lists do not queue behind one another. */
enum : uint32_t { CTR_LARGE = 0, CTR_CLIP_TRIS = 1, CTR_OVERFLOW = 2, CTR_FRAME_COUNT = 3, CTR_COUNT = 160 };
/* word of the length of list `which` (0: vertex chunks, 1: primitive chunks) of the frames of parity `parity` */
#ifdef __HIPCC__
__host__ __device__
#endif
constexpr uint32_t visible_length(uint32_t parity, uint32_t which) { return 32u + 32u * (2u * parity + which); }
/* a pass of at least this many draws gets the per-draw kernel, which leaves the chunks of the draws it found visible in
 * two dense lists; the vertex and setup kernels then run over those with a fixed grid, and a culled draw costs them nothing */
constexpr uint32_t COMPACT_MIN_DRAWS = 256;
enum : uint32_t { OVF_LARGE = 1, OVF_CLIP_TRIS = 4 };
constexpr int TILE_COUNT_STRIDE = 32; /* one tile counter per 128-byte line: atomics on neighbouring tiles do not share a line */
constexpr int MAX_TILE_AXIS = 256;   /* tile coordinates are packed in 8 bits */
constexpr int SUBPIXEL_BITS = 8;
constexpr float GUARD_BAND = 2.0f;

/* outcode bits kept in the low byte of svtx.xf */
enum : uint32_t {
    OC_XNEG = 1, OC_XPOS = 2, OC_YNEG = 4, OC_YPOS = 8, OC_ZNEG = 16, OC_ZPOS = 32,
    OC_GUARD = 64, OC_INVALID = 128
};

constexpr uint32_t BBOX_CULLED = 0x000000FFu; /* tx0=255 > tx1=0: nothing to draw */

/* float bits [2^-13, 1) >> 16: 13 octaves x 128 buckets */
constexpr uint32_t SRGB_LUT_SHIFT = 15;                 /* a bucket per exponent and top eight mantissa bits: no bucket holds two thresholds */
constexpr uint32_t SRGB_LUT_FIRST = 0x39000000u >> SRGB_LUT_SHIFT; /* 2^-13 */
constexpr uint32_t SRGB_LUT_LAST = 0x3F800000u >> SRGB_LUT_SHIFT;  /* 1.0 */
constexpr uint32_t SRGB_LUT_SIZE = SRGB_LUT_LAST - SRGB_LUT_FIRST;

/* Texel tables, staged in LDS by the kernels that shade: value[0][k] = linear value of sRGB code k,
 * value[1][k] = float(k) / 255 (a UNORM channel); thr[k] = linear value of sRGB code k-0.5
 * (thr[0] = -inf, thr[256] = +inf); lut[bucket] = number of thresholds not above the bucket's lower edge */
struct __attribute__((aligned(16))) ShadeTables {
    float value[2][256];
    float thr[257];
    uint8_t lut[SRGB_LUT_SIZE];
    uint8_t pad[12]; /* size is a multiple of 16 bytes */
};

/* The same without the second texel table: what the tile kernel stages when every texel of the pass decodes through one table
 * (the pass's texture's, staged as value[0]).  The kilobyte saved is what lets five workgroups of 40-row tiles share a CU. */
struct __attribute__((aligned(16))) PassTables {
    float value[1][256];
    float thr[257];
    uint8_t lut[SRGB_LUT_SIZE];
    uint8_t pad[12];
};
static_assert(sizeof(PassTables) + 1024 == sizeof(ShadeTables) && sizeof(PassTables) % 16 == 0, "PassTables is ShadeTables less value[1]");
constexpr int MAX_TEXTURE_LEVELS = 15; /* a side of at most 16384 texels has at most 15 levels */

/* Bits of TileEntry::slot_flags and PrimRec::draw_flags */
constexpr uint32_t SLOT_MASK = 0x0FFFFFFFu;   /* primitive slot (submission number) */
constexpr uint32_t ENTRY_SUB_SHIFT = 28;      /* bits 28..30: fan index of a clipped primitive's triangle */
constexpr uint32_t FLAG_CLIPPED = 0x80000000u;
constexpr uint32_t MAX_FAN = 7;               /* a triangle clipped by 6 planes has at most 9 corners */

/* One triangle in a tile's list: 16 bytes, read with one dwordx4 per lane.
 * Unclipped: a,b,c index svtx[] and are ordered so the area is positive.
 * Clipped (FLAG_CLIPPED): a indexes clip_tris[]. */
struct TileEntry {
    uint32_t a, b, c;
    uint32_t slot_flags;
};

/* Entry of the pass-wide list (primitives spanning many tiles, and overflow of full tile lists) */
struct LargeEntry {
    TileEntry e;
    uint32_t box;      /* packed tile box the entry applies to */
    uint32_t pad[3];
};

/* Per primitive slot, written by setup for every primitive that survives (in a one-draw pass only for
 * clipped ones: FrameParams::direct_index): what shading needs.
 * Unclipped: v0..v2 index svtx[] in the primitive's own corner order.
 * Clipped (FLAG_CLIPPED in draw_flags): v0 = index of its first ClipTri. */
struct PrimRec {
    uint32_t v0, v1, v2;
    uint32_t draw_flags; /* draw index | FLAG_CLIPPED */
};

/* One fan triangle of a clipped primitive, oriented (positive area), 128 bytes */
struct ClipTri {
    int32_t x[3], y[3];
    float z[3], iw[3];
    float bary[9];      /* bary[k*3+j]: weight of the parent's corner j at this triangle's corner k */
    uint32_t v[3];      /* the parent's corners in svtx[] */
    uint32_t draw;
    uint32_t valid;     /* 0: degenerate or culled, never rasterised */
    uint32_t pad[5];
};

/* Screen-space vertex written by the vertex kernel: 16 bytes, one dwordx4 per corner in setup */
struct SVtx {
    int32_t xf;  /* (x in 24.8) << 8 | outcode */
    int32_t y;   /* 24.8 */
    float z;     /* viewport depth */
    float iw;    /* 1/w */
};

/* What shading needs of a vertex besides its position: 12 bytes, read as one load per corner.
 * `light` is dot(v_normal, normalize(vec3(0, 1, -1))) (test_ktx.frag:14-15, push_constants.frag:12-13) taken at the
 * vertex: the dot product is linear in the normal, so interpolating it is interpolating the normal and taking the
 * dot per fragment, up to float rounding -- and the record holds one float where the normal would take three. */
struct SVary {
    float light;      /* zero for the dummy shader, which does not light */
    float u, v;       /* v_uv: a_uv, with v flipped for the dummy shader (dummy.vert.glsl:25) */
};
static_assert(sizeof(SVary) == 12, "packed");

/* Per-draw record, immutable after recording */
struct DrawStatic {
    const uint8_t* vb;   /* vertex 0 of the draw (binding offset and vertex_offset applied) */
    const uint32_t* ib;  /* index 0 of the draw (binding offset and first_index applied) */
    const float* ubo;    /* 80 floats: projection, view_projection, view, model, mvp */
    uint32_t vb_capacity; /* whole vertices readable from vb */
    uint32_t stride;
    int32_t off_pos, off_nrm, off_uv;
    uint32_t tri_count;
    uint32_t vert_count; /* vertices the vertex kernel transforms: min(max index + 1, vb_capacity) */
    uint32_t svtx_base;
    int32_t shader;
    int32_t texture;
    uint32_t cull_mode, front_face;
    uint32_t prim_base;  /* running primitive number of triangle 0 (for diagnostics) */
    uint32_t has_bounds; /* bounds_min / bounds_max hold the box of the draw's vertices (all finite) */
    float push[20];      /* vec4 colour, mat4 transform */
    float bounds_min[3], bounds_max[3]; /* object space, over vertices [0, vert_count) */
    uint32_t first_tchunk; /* number, within the pass, of the draw's first chunk of 64 primitives: chunk k of the pass owns primitive slots 64k .. 64k+63 */
    uint32_t pad1;
};

/* Per draw, derived from the uniform buffer each time the work executes: object space to clip space (for the
 * clipper, and for the vertex kernel when a kernel of its own prepared it), the normal matrix, and whether
 * the draw is drawn at all this frame. */
struct DrawDerived {
    float clip_m[16];
    float nrm_m[9];   /* written only by the per-draw kernel (passes of many draws) */
    uint32_t visible; /* 0: the draw's box misses the frustum or the rectangle this device renders; setup skips it */
    uint32_t pad[6];
};

/* Record-time job: the box of `count` positions read at vb + i * stride + off_pos.  The result is six
 * words per job, minimum x y z then maximum x y z, as floats mapped to unsigned integers of the same order
 * (float_to_ordered); the caller initialises them to 0xFFFFFFFF x3, 0 x3. */
struct BoundsJob {
    const uint8_t* vb;
    uint32_t stride, off_pos, count, pad;
};

/* A small host write to a uniform buffer that has not reached device memory.  It rides in the
 * arguments of every frame's first kernel, which reads the buffer through it, so the per-frame
 * 320-byte update costs no copy and frames in flight each keep their own bytes. */
constexpr uint32_t PATCH_MAX_BYTES = 512;
struct UniformPatch {
    uint8_t* dst;       /* device address the bytes belong at (4-byte aligned), or null */
    uint32_t bytes;     /* multiple of 4 */
    uint32_t pad;
    uint32_t words[PATCH_MAX_BYTES / 4];
};

struct Chunk {
    uint32_t draw;
    uint32_t first; /* first vertex / triangle of the chunk inside its draw */
};

struct TexDesc {
    const uint8_t* level[16];
    uint32_t width, height, levels;
    uint32_t srgb, tiled, filter_linear;
};

struct IndexRangeJob {
    const uint32_t* ib;
    uint32_t count;
    uint32_t pad;
};

/* Everything a frame's kernels need; passed by value as the kernel argument */
/* Probe masks switch parts of a kernel off to time the rest; the frame is then wrong.  They exist
 * only in a build with -DRND_PROBES (tools/build_variant.sh), which reads the mask from RND_DEBUG. */
#ifdef RND_PROBES
#define RND_PROBE(fp, bit) (((fp).debug_flags & (bit)) != 0u)
/* phase stamps: thread 0 of every workgroup adds the cycles since its previous stamp to slot `i` of the workgroup's own row
 * (no atomics: a row has one writer) */
#define RND_STAMP(fp, last, i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        if (threadIdx.x == 0u && (fp).probe_cycles) { (fp).probe_cycles[(size_t)blockIdx.x * 16 + (i)] += now_ - (last); } (last) = now_; } while (0)
#define RND_STAMP_START(last) unsigned long long last = __builtin_amdgcn_s_memtime(); \
        if (threadIdx.x == 0u && (fp).probe_cycles) { (fp).probe_cycles[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memrealtime(); \
            (fp).probe_cycles[(size_t)blockIdx.x * 16 + 13] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | \
                                                              (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }
#define RND_STAMP_END(fp) do { if (threadIdx.x == 0u && (fp).probe_cycles) (fp).probe_cycles[(size_t)blockIdx.x * 16 + 15] = __builtin_amdgcn_s_memrealtime(); } while (0)
/* when a whole kernel ran (100 MHz ticks): workgroup 0 leaves its start in word 4 * k of the row after the tile kernel's rows,
 * every workgroup its end in a word of its own (no atomics: they would serialise the kernel being timed) from `first` on */
#define RND_KERNEL_BEGIN(fp, k) do { if ((fp).probe_cycles && blockIdx.x == 0u && threadIdx.x == 0u) \
        (fp).probe_cycles[(size_t)(fp).probe_kernel_row * 16 + 4 * (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define RND_KERNEL_END(fp, first) do { if ((fp).probe_cycles && threadIdx.x == 0u) \
        (fp).probe_cycles[(size_t)(fp).probe_kernel_row * 16 + 16 + (first) + blockIdx.x] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RND_KERNEL_BEGIN(fp, k) do { } while (0)
#define RND_KERNEL_END(fp, k) do { } while (0)
#define RND_STAMP_END(fp) do { } while (0)
#define RND_PROBE(fp, bit) false
#define RND_STAMP(fp, last, i) do { } while (0)
#define RND_STAMP_START(last) do { } while (0)
#endif

struct FrameParams {
    const DrawStatic* draws;
    DrawDerived* derived;
    const Chunk* vchunks;
    const Chunk* tchunks;
    const TexDesc* textures;
    SVtx* svtx;
    struct SVary* svary;     /* per vertex: the vertex shader's other outputs */
    PrimRec* prim_rec;       /* per primitive slot */
    uint32_t* tile_count;    /* tile t at [t*TILE_COUNT_STRIDE]: entries appended this frame (may exceed tile_capacity) */
    TileEntry* tile_lists;   /* tile t owns entries [t*tile_stride, t*tile_stride + tile_capacity) */
    LargeEntry* large_list;  /* primitives touching more than MAX_TILES_PER_PRIM tiles + tile-list overflow */
    ClipTri* clip_tris;
    uint32_t* counters;      /* [0] large_list entries, [1] clip_tris used, [2] overflow flags */
    uint32_t* status_host;   /* pinned host word: overflow flags of the last frame that had any */
    uint32_t tile_capacity, large_capacity, clip_tri_capacity;
    uint32_t tile_stride;    /* capacity + 16 entries: lists do not all start on the same power-of-two boundary
                                (equal low address bits would pile every tile onto the same memory channels) */
    uint8_t* color;          /* resolved colour, width*height*4 */
    uint32_t* depth;         /* width*height*samples */
    const struct ShadeTables* tables; /* sRGB decode / encode tables */
    uint32_t n_draws, n_vchunks, n_tchunks, n_textures;
    uint32_t width, height, samples;
    uint32_t color_srgb, color_bgra;
    uint32_t tiles_x, tiles_y;
    uint32_t tile_h;         /* rows of pixels per tile: TILE_H or TILE_H_TALL */
    uint32_t tile_h_magic;   /* ceil(2^32 / tile_h): a pixel row's row of tiles is the high word of row * magic (rows stay below 2^16) */
    uint32_t shard_index, shard_count;
    uint32_t grid_x0, grid_y0, grid_w, grid_h; /* the tiles the render area touches: only these are launched */
    /* Draw culling: the clamp rectangle grown by one pixel, in normalised device coordinates.  A draw whose
     * box of vertices lies wholly beyond one of these four planes, the near plane or the far plane has no
     * fragment inside the rectangle (a device that renders one band of a frame skips the other bands' draws) */
    float cull_lo_x, cull_hi_x, cull_lo_y, cull_hi_y;
    uint32_t cull_draws;     /* 0: the boxes are stale (vertex data was written after recording), nothing is culled */
    int32_t clamp_x0, clamp_y0, clamp_x1, clamp_y1; /* inclusive pixel rectangle fragments may land in */
    int32_t area_x0, area_y0, area_x1, area_y1;     /* inclusive render area (cleared and stored) */
    float half_w, half_h, ox, oy, pz, zmin;
    float clear_color[4];
    uint32_t clear_packed;   /* clear_color as the target stores it (encoded, channel order applied) */
    uint32_t clear_d24;
    uint32_t depth_less;     /* 1: LESS, 0: LESS_OR_EQUAL */
    uint32_t depth_test, depth_write;
    uint32_t direct_index;   /* 1: the pass is one draw, so a primitive's slot is its triangle number and shading reads the
                                index buffer itself; unclipped primitives then get no PrimRec */
    const uint32_t* direct_ib;     /* direct_index: the one draw's index buffer and first svtx[] record, by value (shading reads */
    uint32_t direct_svtx_base;     /* them per pixel: kernel arguments, not a load from the draw's record) */
    int32_t pass_shader;     /* shader of every draw of the pass, or -1 when they differ */
    int32_t pass_texture;    /* texture index of every draw (-1: none bound), or -2 when they differ */
    TexDesc pass_tex;        /* textures[pass_texture] by value */
    unsigned long long* probe_cycles; /* a build with RND_PROBES: wave-cycles summed per phase of the tile kernel; otherwise null */
    uint32_t probe_kernel_row; /* a build with RND_PROBES: the row of probe_cycles that holds whole-kernel start and end times */
    uint32_t debug_flags;    /* probe masks of a build with RND_PROBES (timing experiments; results are wrong); otherwise unused */
    /* a pass of many draws (last, so that the arguments every pass reads stay where the kernels' wide scalar loads want them) */
    Chunk* visible_vchunks;  /* compact != 0: the chunks of the draws this frame found visible, dense, in no particular order; */
    Chunk* visible_tchunks;  /* their lengths are counters[visible_length(list_parity, 0 / 1)] */
    uint32_t compact;        /* 1: the pass has COMPACT_MIN_DRAWS draws or more */
    uint32_t list_parity;    /* which pair of list lengths this frame uses: alternates between successive frames of the command buffer */
};


/* rnd_draw_indexed_noalloc renders one draw with every intermediate buffer inside one static device array instead of
 * allocations made at record time.  Its limits, and where each buffer lies in the array (offsets in bytes, all multiples
 * of 256): */
constexpr uint32_t NA_MAX_VERTICES = 65536;      /* vertices the draw may reference */
constexpr uint32_t NA_MAX_TRIANGLES = 131072;
constexpr uint32_t NA_MAX_TILES = 1024;          /* 32x32 tiles of the target: 1280x720 has 920 */
constexpr uint32_t NA_TILE_CAPACITY = 1024;      /* entries per tile list before spilling to the pass-wide list */
constexpr uint32_t NA_LARGE_CAPACITY = 32768;
constexpr uint32_t NA_CLIP_TRI_CAPACITY = 16384;
struct NoAllocLayout {
    static constexpr size_t draw = 0;                                                     /* DrawStatic[1] */
    static constexpr size_t derived = draw + 256;                                         /* DrawDerived[1] */
    static constexpr size_t counters = derived + 256;                                     /* CTR_FRAME_COUNT words (a pass of one draw keeps no lists of visible chunks) */
    static constexpr size_t ubo = counters + 256;                                         /* where the uniform block would live (320 bytes) */
    static constexpr size_t tile_count = ubo + 512;
    static constexpr size_t svtx = tile_count + (size_t)NA_MAX_TILES * TILE_COUNT_STRIDE * 4;
    static constexpr size_t svary = svtx + (size_t)NA_MAX_VERTICES * sizeof(SVtx);
    static constexpr size_t prim_rec = svary + (size_t)NA_MAX_VERTICES * sizeof(SVary);
    static constexpr size_t tile_lists = prim_rec + (size_t)NA_MAX_TRIANGLES * sizeof(PrimRec);
    static constexpr size_t large_list = tile_lists + (size_t)NA_MAX_TILES * (NA_TILE_CAPACITY + 16) * sizeof(TileEntry);
    static constexpr size_t clip_tris = large_list + (size_t)NA_LARGE_CAPACITY * sizeof(LargeEntry);
    static constexpr size_t total = clip_tris + (size_t)NA_CLIP_TRI_CAPACITY * sizeof(ClipTri);
};
static_assert(sizeof(DrawStatic) <= 256 && sizeof(DrawDerived) <= 256, "a draw's two records each fit the 256-byte slot NoAllocLayout gives them");
static_assert(NoAllocLayout::total % 256 == 0 && NoAllocLayout::total < (size_t(32) << 20), "the static scratch array stays under 32 MiB");

/* this device's address of the static scratch array (hipGetSymbolAddress; the current device's copy) */
hipError_t noalloc_scratch_address(void** out);
/* puts the draw's record into the array and zeroes what a frame expects to find at zero */
void launch_noalloc_begin(const DrawStatic& d, uint8_t* scratch, uint32_t n_tiles, hipStream_t s);

/* launch wrappers (rnd_kernels.hip) */
void launch_index_range(const IndexRangeJob* jobs, uint32_t* out_max, uint32_t n_jobs, hipStream_t s);
void launch_bounds(const BoundsJob* jobs, uint32_t* out_min_max, uint32_t n_jobs, hipStream_t s);
void launch_vertex(const FrameParams& fp, const UniformPatch& patch, hipStream_t s); /* with the per-draw kernel first when the pass has many draws */
void launch_setup(const FrameParams& fp, hipStream_t s);
/* lds_pad: bytes of LDS a tile workgroup claims beyond what it uses (0, or TILE_LDS_PAD_FOUR so that four workgroups
 * fit a CU instead of five; the single-sample kernel only) */
constexpr uint32_t TILE_LDS_PAD_FOUR = 2048;
void launch_tile(const FrameParams& fp, hipStream_t s, uint32_t lds_pad = 0);
void launch_mip(const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh, int srgb,
                const struct ShadeTables* tables, hipStream_t s);

} // namespace rnd
#endif
