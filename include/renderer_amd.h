/*
 * renderer_amd.h -- C ABI of the MI355X (gfx950) draw path.
 *
 * The reference (VladislavKhudziakov/renderer) hands indexed-triangle forward
 * rendering to its Vulkan driver.  This library replaces the driver side of that
 * hand-off with HIP kernels.  Each entry point is the image of one Vulkan call
 * (or reference helper) the samples use on this path; the citation names the
 * reference call site it replaces.  Enum values are the Vulkan 1.0 numeric
 * values, so state held in the reference's Vk*CreateInfo structures passes
 * straight through.
 *
 * Conventions
 *   - every function returns 0 on success and a negative rnd_result on failure,
 *     never throws; rnd_last_error() returns the message of the calling
 *     thread's last failure (the host turns it into the reference's
 *     RAISE_ERROR_* macros, error_handler.hpp:9-48);
 *   - the caller owns every handle; *_destroy(NULL) is a no-op
 *     (handlers.hpp:108-113); children are destroyed before their device
 *     (context.cpp:567-574);
 *   - data passed to a *_create or *_write call is copied before the call
 *     returns (staging semantics of tools.cpp:820-830);
 *   - one device is driven from one host thread at a time (the reference is
 *     single-threaded: vk_app.cpp:60-75).
 */
#ifndef RENDERER_AMD_H
#define RENDERER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rnd_device_t* rnd_device;
typedef struct rnd_buffer_t* rnd_buffer;
typedef struct rnd_texture_t* rnd_texture;
typedef struct rnd_target_t* rnd_target;
typedef struct rnd_pipeline_t* rnd_pipeline;
typedef struct rnd_cmd_t* rnd_cmd;
typedef struct rnd_fence_t* rnd_fence;

typedef enum rnd_result {
    RND_SUCCESS = 0,
    RND_ERROR_INVALID_ARGUMENT = -1,
    RND_ERROR_OUT_OF_DEVICE_MEMORY = -2,  /* VK_ERROR_OUT_OF_DEVICE_MEMORY */
    RND_ERROR_INITIALIZATION_FAILED = -3, /* VK_ERROR_INITIALIZATION_FAILED */
    RND_ERROR_DEVICE_LOST = -4,           /* VK_ERROR_DEVICE_LOST: a HIP call failed */
    RND_ERROR_FEATURE_NOT_PRESENT = -8,   /* VK_ERROR_FEATURE_NOT_PRESENT: state outside the supported subset */
    RND_ERROR_FORMAT_NOT_SUPPORTED = -11, /* VK_ERROR_FORMAT_NOT_SUPPORTED */
    RND_ERROR_RECORDING = -12             /* command recorded in the wrong state */
} rnd_result;

/* VkFormat values used by the reference on this path */
enum {
    RND_FORMAT_R8_UNORM = 9, RND_FORMAT_R8_UINT = 13, RND_FORMAT_R8_SRGB = 15,
    RND_FORMAT_R8G8_UNORM = 16, RND_FORMAT_R8G8_UINT = 20, RND_FORMAT_R8G8_SRGB = 22,
    RND_FORMAT_R8G8B8_UNORM = 23, RND_FORMAT_R8G8B8_UINT = 27, RND_FORMAT_R8G8B8_SRGB = 29,
    RND_FORMAT_R8G8B8A8_UNORM = 37, RND_FORMAT_R8G8B8A8_UINT = 41, RND_FORMAT_R8G8B8A8_SRGB = 43,
    RND_FORMAT_B8G8R8A8_UNORM = 44, RND_FORMAT_B8G8R8A8_SRGB = 50,
    RND_FORMAT_R32G32_SFLOAT = 103, RND_FORMAT_R32G32B32_SFLOAT = 106,
    RND_FORMAT_D24_UNORM_S8_UINT = 129
};

/* VkBufferUsageFlagBits */
enum { RND_BUFFER_USAGE_UNIFORM = 0x10, RND_BUFFER_USAGE_INDEX = 0x40, RND_BUFFER_USAGE_VERTEX = 0x80 };
/* VkCullModeFlagBits, VkFrontFace, VkCompareOp, VkFilter, VkIndexType, VkPrimitiveTopology */
enum { RND_CULL_NONE = 0, RND_CULL_FRONT = 1, RND_CULL_BACK = 2 };
enum { RND_FRONT_FACE_COUNTER_CLOCKWISE = 0, RND_FRONT_FACE_CLOCKWISE = 1 };
enum { RND_COMPARE_LESS = 1, RND_COMPARE_LESS_OR_EQUAL = 3 };
enum { RND_FILTER_NEAREST = 0, RND_FILTER_LINEAR = 1 };
enum { RND_INDEX_TYPE_UINT32 = 1 };
enum { RND_TOPOLOGY_TRIANGLE_LIST = 3 };

/* The three shader pairs the reference owns (SPIR-V ingestion is not provided):
 * test_app/shaders/dummy.*, test_ktx_app/shaders/test_ktx.*,
 * test_push_constants_app/shaders/push_constants.* */
enum { RND_SHADER_DUMMY = 0, RND_SHADER_TEST_KTX = 1, RND_SHADER_PUSH_CONSTANTS = 2 };

/* vk_utils::sampler_info (tools.hpp:9-14) */
typedef struct rnd_sampler_info {
    int32_t filtering;     /* RND_FILTER_* */
    int32_t tiled;         /* REPEAT when non-zero, else CLAMP_TO_EDGE (tools.cpp:69-71) */
    float max_anisotropy;  /* accepted, sampling is isotropic */
} rnd_sampler_info;

/* One vertex attribute as the samples declare it (test_ktx_app.cpp:541-605) */
typedef struct rnd_vertex_attribute {
    int32_t format;  /* RND_FORMAT_R32G32B32_SFLOAT / R32G32_SFLOAT, 0 = absent */
    uint32_t offset; /* bytes from the start of a vertex */
} rnd_vertex_attribute;

/* The fixed-function state of VkGraphicsPipelineCreateInfo that the samples set
 * (test_ktx_app.cpp:412-625, dummy_obj_viewer_app.cpp:193-393). */
typedef struct rnd_pipeline_info {
    int32_t shader;                    /* RND_SHADER_* */
    rnd_vertex_attribute position;     /* location 0 */
    rnd_vertex_attribute normal;       /* location 1 */
    rnd_vertex_attribute uv;           /* location 2 */
    uint32_t stride;
    int32_t topology;                  /* RND_TOPOLOGY_TRIANGLE_LIST */
    int32_t cull_mode, front_face;
    int32_t depth_test, depth_write, depth_compare;
    uint32_t samples;                  /* 1 or 4 */
    uint32_t color_write_mask;         /* must be 0xF (RGBA) */
    int32_t blend_enable;              /* must be 0 */
} rnd_pipeline_info;

/* Stage times of the most recent profiled submit, in milliseconds. */
enum { RND_STAGE_VERTEX = 0, RND_STAGE_SETUP, RND_STAGE_TILE, RND_STAGE_COUNT }; /* TILE: rasterisation and shading, one kernel */

const char* rnd_last_error(void);

/* vk_utils::context::init (context.cpp:238-258): one device object per GPU.
 * stream: a hipStream_t to launch on, or NULL for a stream owned by the device. */
int rnd_device_create(int hip_ordinal, void* stream, rnd_device* out);
/* The device goes last, as the reference destroys its context last (context.cpp:567-574): every buffer, texture, target,
 * pipeline, command buffer and fence points back at its device, and destroying it while any of them is alive fails with
 * RND_ERROR_INVALID_ARGUMENT and leaves the device as it is. */
int rnd_device_destroy(rnd_device dev);
int rnd_device_wait_idle(rnd_device dev);                 /* vkDeviceWaitIdle (vk_app.cpp:412-418) */
void* rnd_device_stream(rnd_device dev);                  /* the hipStream_t uploads and copies are queued on */
int rnd_device_set_profiling(rnd_device dev, int enable); /* bracket every stage with HIP events */
int rnd_device_stage_times(rnd_device dev, float* ms, int count);

/* vk_utils::create_buffer (tools.cpp:788-835) */
int rnd_buffer_create(rnd_device dev, uint64_t bytes, uint32_t usage, const void* init, rnd_buffer* out);
/* vmaMapMemory / memcpy / vmaFlushAllocation / vmaUnmapMemory (base_obj_viewer_app.cpp:134-138).
 * Frames submitted before the call keep the old bytes, frames submitted after it see the new
 * ones.  Up to 512 bytes into a buffer whose only usage is UNIFORM cost no copy at all.
 * Vertex data may be rewritten between submits (whole-draw culling is then off for command
 * buffers recorded earlier).  Index data may not: how many vertices a draw transforms is scanned
 * from its indices at rnd_cmd_end, so rnd_submit of a command buffer recorded before a write into
 * an index buffer fails with RND_ERROR_RECORDING until it is recorded again. */
int rnd_buffer_write(rnd_buffer buf, uint64_t offset, const void* src, uint64_t bytes);
/* same, source already in device memory (device-to-device on the device's stream) */
int rnd_buffer_write_device(rnd_buffer buf, uint64_t offset, const void* device_src, uint64_t bytes);
int rnd_buffer_read(rnd_buffer buf, uint64_t offset, void* dst, uint64_t bytes);
int rnd_buffer_destroy(rnd_buffer buf);

/* vk_utils::create_texture_2D (tools.cpp:448-460): upload level 0, optionally
 * build (uint32)log2(max(w,h)) levels by successive linear blits (tools.cpp:514, 651-708). */
int rnd_texture_create_2d(rnd_device dev, uint32_t width, uint32_t height, int32_t format, int32_t gen_mips,
                          const void* texels, const rnd_sampler_info* sampler, rnd_texture* out);
/* vk_utils::create_ktx_texture (tools.cpp:1043-1334): stored levels uploaded as they are */
int rnd_texture_create_levels(rnd_device dev, uint32_t width, uint32_t height, int32_t format, uint32_t level_count,
                              const void* const* level_texels, const rnd_sampler_info* sampler, rnd_texture* out);
int rnd_texture_level_count(rnd_texture tex, uint32_t* out);
int rnd_texture_read_level(rnd_texture tex, uint32_t level, void* rgba8_out); /* no reference analogue: parity checks */
int rnd_texture_destroy(rnd_texture tex);

/* colour + D24S8 attachments of init_framebuffers (test_ktx_app.cpp:164-258) and, with
 * samples = 4, the multisampled pair + resolve image of dummy_obj_viewer_app.cpp:546-639 */
int rnd_target_create(rnd_device dev, uint32_t width, uint32_t height, int32_t color_format, int32_t depth_format,
                      uint32_t samples, rnd_target* out);
/* no reference analogue (the reference only presents): resolved colour, 4 bytes per pixel in
 * the target's byte order; depth words (D24 in bits 0..23), samples per pixel consecutive.
 * Either pointer may be NULL.  Waits for the device. */
int rnd_target_read(rnd_target target, void* color_out, uint32_t* depth_out);
void* rnd_target_color_device_ptr(rnd_target target); /* resolved colour image in device memory, row-major */
/* no reference analogue: queue a device-to-device copy of the resolved colour image
 * (width*height*4 bytes), e.g. into the send buffer of the multi-GPU frame gather.  The copy
 * runs behind the last frame rendered into the target, on that frame's stream (which is
 * rnd_device_stream() only for a device created on a caller-supplied stream), and a later frame
 * into the target follows it.  Does not wait: before another stream or the host reads
 * device_dst, wait for a fence of a later submit, rnd_device_wait_idle or rnd_target_read. */
int rnd_target_copy_color(rnd_target target, void* device_dst);
/* the same for rows [first_row, first_row + row_count) only (row_count*width*4 bytes): what a device
 * that renders one band of the frame (its render area) contributes to the gather */
int rnd_target_copy_color_rows(rnd_target target, uint32_t first_row, uint32_t row_count, void* device_dst);
int rnd_target_destroy(rnd_target target);

/* vkCreateGraphicsPipelines (test_ktx_app.cpp:607-620) */
int rnd_pipeline_create(rnd_device dev, const rnd_pipeline_info* info, rnd_pipeline* out);
int rnd_pipeline_destroy(rnd_pipeline p);

/* command recording: record_command_buffers (test_ktx_app.cpp:627-746) */
int rnd_cmd_create(rnd_device dev, rnd_cmd* out);      /* vkAllocateCommandBuffers :639 */
int rnd_cmd_begin(rnd_cmd cmd);                        /* vkBeginCommandBuffer :698 */
int rnd_cmd_begin_pass(rnd_cmd cmd, rnd_target target, const float clear_color[4], float clear_depth,
                       int32_t area_x, int32_t area_y, uint32_t area_w, uint32_t area_h); /* vkCmdBeginRenderPass :700 */
int rnd_cmd_set_viewport(rnd_cmd cmd, float x, float y, float w, float h, float min_depth, float max_depth); /* :701 */
int rnd_cmd_set_scissor(rnd_cmd cmd, int32_t x, int32_t y, uint32_t w, uint32_t h);                          /* :702 */
int rnd_cmd_bind_pipeline(rnd_cmd cmd, rnd_pipeline p);                                                      /* :723 */
int rnd_cmd_bind_vertex_buffer(rnd_cmd cmd, rnd_buffer buf, uint64_t byte_offset);                           /* :729 */
int rnd_cmd_bind_index_buffer(rnd_cmd cmd, rnd_buffer buf, uint64_t byte_offset, int32_t index_type);        /* :730 */
/* vkCmdBindDescriptorSets (:724) split by descriptor: the uniform block (set 0 binding 0) ... */
int rnd_cmd_bind_uniform(rnd_cmd cmd, uint32_t set, uint32_t binding, rnd_buffer buf, uint64_t offset, uint64_t range);
/* ... and the combined image sampler (set 0 binding 1 in test_ktx, set 1 binding 0 in test_app) */
int rnd_cmd_bind_texture(rnd_cmd cmd, uint32_t set, uint32_t binding, rnd_texture tex);
/* vkCmdPushConstants (:719): bytes are copied at record time */
int rnd_cmd_push_constants(rnd_cmd cmd, uint32_t offset, uint32_t size, const void* data);
/* vkCmdDrawIndexed (:732) */
int rnd_cmd_draw_indexed(rnd_cmd cmd, uint32_t index_count, uint32_t instance_count, uint32_t first_index,
                         int32_t vertex_offset, uint32_t first_instance);
int rnd_cmd_end_pass(rnd_cmd cmd);                     /* vkCmdEndRenderPass :736 */
int rnd_cmd_end(rnd_cmd cmd);                          /* vkEndCommandBuffer :738 */
int rnd_cmd_destroy(rnd_cmd cmd);

/* Multi-GPU sort-first sharding (no reference analogue): this device renders only the
 * tiles t with t % shard_count == shard_index; other pixels of the target are left
 * untouched.  (0,1) renders everything.  Takes effect at the next submit. */
int rnd_cmd_set_tile_shard(rnd_cmd cmd, uint32_t shard_index, uint32_t shard_count);
/* The other way to shard a frame needs no call of its own: give each device a band of the frame as
 * its render area and scissor (rnd_cmd_begin_pass / rnd_cmd_set_scissor, the reference's renderArea
 * and vkCmdSetScissor).  Only the tiles the render area touches are launched, and a draw whose box
 * of vertices misses the scissored rectangle is skipped whole when the work executes, so a band
 * costs its own share of the geometry work as well as of the pixels.  Pixels inside the band are
 * the same as in the unsharded frame. */
/* no reference analogue (a statistic for tests and tuning): how many draws of the command buffer's
 * last executed frame were kept, i.e. had vertices and were not skipped whole.  Waits for the device. */
int rnd_cmd_visible_draws(rnd_cmd cmd, uint32_t* out_count);

/* vkQueueSubmit (vk_app.cpp:463-467).  The command buffer can be resubmitted; uniform
 * buffers are read when the work executes, so a buffer_write between two submits of the
 * same command buffer changes the second frame (base_obj_viewer_app.cpp:134-138).
 * Frames of one command buffer, and frames into one target, execute in submission order;
 * frames of different command buffers into different targets may overlap on the GPU (the
 * reference's frames in flight, vk_app.cpp:109-130) unless the device was created on a
 * caller-supplied stream, which then carries everything.  fence may be NULL. */
int rnd_submit(rnd_device dev, rnd_cmd cmd, rnd_fence fence);

/* No counterpart in the reference.  For callers that cannot allocate device memory at the moment they draw -- code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc: one render pass holding one indexed draw,
 * queued at once on the device's stream, with the same result as recording that pass (begin_pass, set_viewport,
 * set_scissor, bind_*, push_constants, draw_indexed, end) and submitting it.  rnd_cmd_end allocates the buffers a frame's
 * kernels hand to one another; this call keeps them in one static device array of the library instead (about 25 MiB,
 * held by every GPU the library runs on until the process ends) and calls no hipMalloc, hipFree or any routine that
 * allocates.  The buffers, texture, pipeline and target are created beforehand as usual.
 * Limits (beyond them the call fails with RND_ERROR_OUT_OF_DEVICE_MEMORY and draws nothing): vertex_count <= 65 536,
 * index_count / 3 <= 131 072, a target of at most 1024 tiles of 32x32 pixels (1280x720 has 920).  Indices >= vertex_count
 * drop their primitive.  The render area is the whole target.  The static array also holds at most 16 384 fan triangles of
 * clipped primitives and 32 768 entries of the pass-wide list (primitives over many tiles, spill of full tile lists): a
 * draw beyond that renders an incomplete frame and the fence's wait reports the overflow, as the recorded path does.
 * Every call on a GPU shares that GPU's array: calls run one after the other on the device's stream, and two devices
 * objects on one GPU (or a caller-supplied stream per device) must not have such calls in flight together. */
typedef struct rnd_noalloc_draw {
    rnd_target target;
    rnd_pipeline pipeline;
    rnd_texture texture;            /* may be NULL */
    rnd_buffer vertex_buffer; uint64_t vertex_offset_bytes; uint32_t vertex_count;
    rnd_buffer index_buffer; uint64_t index_offset_bytes; uint32_t index_count;
    const float* uniform_block;     /* host memory, 80 floats: projection, view_projection, view, model, mvp */
    const float* push_constants;    /* host memory, 20 floats (vec4 colour, mat4 transform), or NULL for zeros */
    float clear_color[4]; float clear_depth;
    float viewport[6];              /* x, y, width, height, min_depth, max_depth */
    int32_t scissor[4];             /* x, y, width, height */
} rnd_noalloc_draw;
int rnd_draw_indexed_noalloc(rnd_device dev, const rnd_noalloc_draw* draw, rnd_fence fence);

int rnd_fence_create(rnd_device dev, rnd_fence* out);  /* vk_app.cpp:109-130 */
int rnd_fence_wait(rnd_fence fence);                   /* vkWaitForFences vk_app.cpp:435-442 */
int rnd_fence_destroy(rnd_fence fence);

#ifdef __cplusplus
}
#endif
#endif
