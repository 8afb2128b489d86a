/*
 * raster_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement of the indexed-triangle
 * forward-rendering path that the reference hands to its Vulkan driver
 * (reference: src/samples/test_ktx_app/test_ktx_app.cpp:627-746 records the
 * draw list, src/libs/app/app/vk_app.cpp:463-467 submits it).  Nothing in the
 * product library links or calls it; only tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py do.
 *
 * PARITY UNPINNED: the reference holds no tests, golden images or fixtures for
 * this path and its arithmetic lives in an unpinned third-party Vulkan ICD, so
 * the rounding choices below (8 sub-pixel bits, top-left rule, D24 round-half-up,
 * 2x2-quad LOD, linear-space resolve) are this repository's own pinned reading
 * of the Vulkan 1.0 rules.
 *
 * Enum values are the Vulkan 1.0 numeric values so that state taken from the
 * reference's VkPipeline*CreateInfo structures passes straight through.
 */
#ifndef RASTER_ORACLE_H
#define RASTER_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORC_SHADER_DUMMY = 0,          /* test_app/shaders/dummy.{vert,frag}.glsl */
    ORC_SHADER_TEST_KTX = 1,       /* test_ktx_app/shaders/test_ktx.{vert,frag} */
    ORC_SHADER_PUSH_CONSTANTS = 2  /* test_push_constants_app/shaders/push_constants.{vert,frag} */
};

typedef struct orc_texture {
    uint32_t width, height, levels;
    uint32_t srgb;          /* 1: texels are sRGB-encoded (alpha linear) */
    uint32_t tiled;         /* 1: REPEAT, 0: CLAMP_TO_EDGE (tools.cpp:69-71) */
    uint32_t filter_linear; /* 1: LINEAR min/mag/mip, 0: NEAREST (tools.cpp:66-68) */
    const uint8_t* level_data[16]; /* RGBA8, tightly packed, level i is max(w>>i,1) x max(h>>i,1) */
} orc_texture;

typedef struct orc_draw {
    int32_t shader;
    uint32_t stride;
    int32_t off_pos, off_nrm, off_uv; /* byte offsets inside a vertex; <0: attribute absent (reads as 0) */
    int32_t cull_mode;   /* VkCullModeFlagBits: 0 none, 1 front, 2 back */
    int32_t front_face;  /* VkFrontFace: 0 counter-clockwise, 1 clockwise */
    int32_t depth_test, depth_write;
    int32_t depth_op;    /* VkCompareOp: 1 LESS, 3 LESS_OR_EQUAL */
    const uint8_t* vb;   /* vertex buffer at the bound offset */
    uint64_t vb_bytes;
    const uint32_t* ib;  /* index buffer at the bound offset */
    uint32_t index_count, first_index;
    int32_t vertex_offset;
    const float* ubo;    /* 80 floats: projection, view_projection, view, model, mvp */
    float push[20];      /* vec4 colour, mat4 transform (column-major) */
    int32_t texture;     /* index into the texture array, <0: none */
} orc_draw;

typedef struct orc_pass {
    uint32_t width, height;   /* target extent */
    uint32_t samples;         /* 1 or 4 */
    uint32_t color_srgb;      /* 1: sRGB-encode on store */
    uint32_t color_bgra;      /* 1: byte order B,G,R,A */
    float clear_color[4];
    float clear_depth;
    float viewport[6];        /* x, y, w, h, min_depth, max_depth */
    int32_t scissor[4];       /* x, y, w, h */
    int32_t area[4];          /* render area x, y, w, h */
} orc_pass;

/* Renders one pass.  out_color: width*height*4 bytes (resolved when samples>1),
 * out_depth: width*height*samples words, D24 in bits 0..23 (sample-major inside a pixel),
 * out_prim (optional): width*height*samples, running primitive number of the
 * fragment that owns the sample (0xFFFFFFFF = cleared).
 * literal != 0: shade every passing fragment in submission order (the literal
 * pipeline); 0: resolve depth first and shade the surviving fragment only.
 * Both give the same image; threads > 1 splits the target in bands. */
int orc_render(const orc_pass* pass, const orc_draw* draws, int n_draws,
               const orc_texture* textures, int n_textures,
               uint8_t* out_color, uint32_t* out_depth, uint32_t* out_prim,
               int literal, int threads);

/* Mip chain as the reference builds it: levels = (uint32)log2(max(w,h)), each
 * level blitted LINEAR from the previous one (tools.cpp:514, 651-708).
 * out_levels[i] receives level i (i >= 1); level 0 is the input. */
uint32_t orc_mip_level_count(uint32_t w, uint32_t h, int gen_mips);
int orc_gen_mip(const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh, int srgb);

/* Known-answer hooks for the pinned scalar rules. */
uint32_t orc_srgb_encode(float linear);     /* linear -> 0..255 */
float orc_srgb_decode(uint32_t code);       /* 0..255 -> linear */
float orc_log2(float x);                    /* LOD log2 approximation */
int32_t orc_snap(float window_coord);       /* window coordinate -> 24.8 fixed point */
uint32_t orc_depth24(float z);              /* z in [0,1] -> D24 */
void orc_sample_texture(const orc_texture* t, float u, float v, float dudx, float dvdx, float dudy, float dvdy, float out_rgba[4]);

/* glm-equivalent host matrices (column-major, float32), reference call sites:
 * camera.cpp:13-23, base_obj_viewer_app.cpp:122-132, test_ktx_app.cpp:714-715. */
void orc_perspective_fov_rh_no(float fov, float w, float h, float n, float f, float out[16]);
void orc_perspective_fov_rh_zo(float fov, float w, float h, float n, float f, float out[16]);
void orc_look_at_rh(const float eye[3], const float center[3], const float up[3], float out[16]);
void orc_translate(const float m[16], const float v[3], float out[16]);
void orc_scale(const float m[16], const float v[3], float out[16]);
void orc_rotate(const float m[16], float angle, const float axis[3], float out[16]);
void orc_yaw_pitch_roll(float yaw, float pitch, float roll, float out[16]);
void orc_mat4_mul(const float a[16], const float b[16], float out[16]);

#ifdef __cplusplus
}
#endif
#endif
