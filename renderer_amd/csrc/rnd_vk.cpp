/*
 * rnd_vk.cpp -- librnd_vk.so: the Vulkan-named entry points of include/rnd_vk.h, each a thin call into the HIP
 * library (include/renderer_amd.h).  Host code only; no kernels, no arithmetic of the draw path lives here.
 *
 * How Vulkan's objects map onto the library's:
 *   VkBuffer + VmaAllocation   one object: an rnd_buffer (GPU_ONLY / CPU_TO_GPU) and/or a host block (what vmaMapMemory
 *                              returns; CPU_ONLY staging buffers are only that)
 *   VkRenderPass               the attachment formats and sample counts, kept to build framebuffers and check pipelines
 *   VkFramebuffer              an rnd_target (the library keeps colour, depth and the resolve image together)
 *   VkDescriptorSet            a table binding -> uniform range or texture, replayed by vkCmdBindDescriptorSets
 *   VkPipeline                 an rnd_pipeline built from the fixed-function state (test_ktx_app.cpp:412-625)
 *   VkCommandBuffer            an rnd_cmd for the render pass, plus the buffer copies recorded outside one, which run
 *                              when the command buffer is submitted (tools.cpp:788-835 records and submits them once)
 *   VkFence                    an rnd_fence
 * Semaphores of VkSubmitInfo are accepted and ignored: there is one queue and it executes in order.
 */
#include "../../include/rnd_vk.h"
#include "../../include/renderer_amd.h"

#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

/* the sizes these structures have under the Vulkan 1.0 ABI on an LP64 target: a member left out or mistyped in the header shows here */
static_assert(sizeof(VkBufferCreateInfo) == 56 && sizeof(VkAttachmentDescription) == 36 && sizeof(VkSubpassDescription) == 72 &&
              sizeof(VkSubpassDependency) == 28 && sizeof(VkRenderPassCreateInfo) == 64 && sizeof(VkFramebufferCreateInfo) == 64, "Vulkan ABI");
static_assert(sizeof(VkDescriptorSetLayoutBinding) == 24 && sizeof(VkWriteDescriptorSet) == 64 && sizeof(VkDescriptorImageInfo) == 24 &&
              sizeof(VkDescriptorBufferInfo) == 24 && sizeof(VkPipelineLayoutCreateInfo) == 48 && sizeof(VkDescriptorPoolCreateInfo) == 40, "Vulkan ABI");
static_assert(sizeof(VkPipelineShaderStageCreateInfo) == 48 && sizeof(VkPipelineVertexInputStateCreateInfo) == 48 &&
              sizeof(VkPipelineInputAssemblyStateCreateInfo) == 32 && sizeof(VkPipelineViewportStateCreateInfo) == 48 &&
              sizeof(VkPipelineRasterizationStateCreateInfo) == 64 && sizeof(VkPipelineMultisampleStateCreateInfo) == 48 &&
              sizeof(VkPipelineDepthStencilStateCreateInfo) == 104 && sizeof(VkPipelineColorBlendAttachmentState) == 32 &&
              sizeof(VkPipelineColorBlendStateCreateInfo) == 56 && sizeof(VkPipelineDynamicStateCreateInfo) == 32 &&
              sizeof(VkGraphicsPipelineCreateInfo) == 144, "Vulkan ABI");
static_assert(sizeof(VkCommandBufferAllocateInfo) == 32 && sizeof(VkCommandBufferBeginInfo) == 32 && sizeof(VkRenderPassBeginInfo) == 64 &&
              sizeof(VkSubmitInfo) == 72 && sizeof(VkViewport) == 24 && sizeof(VkRect2D) == 16 && sizeof(VkClearValue) == 16 && sizeof(VkBufferCopy) == 24, "Vulkan ABI");

namespace {

thread_local std::string g_error;

VkResult code_of(int rc)
{
    switch (rc) {
    case RND_SUCCESS: return VK_SUCCESS;
    case RND_ERROR_OUT_OF_DEVICE_MEMORY: return VK_ERROR_OUT_OF_DEVICE_MEMORY;
    case RND_ERROR_INITIALIZATION_FAILED: return VK_ERROR_INITIALIZATION_FAILED;
    case RND_ERROR_DEVICE_LOST: return VK_ERROR_DEVICE_LOST;
    case RND_ERROR_FEATURE_NOT_PRESENT: return VK_ERROR_FEATURE_NOT_PRESENT;
    case RND_ERROR_FORMAT_NOT_SUPPORTED: return VK_ERROR_FORMAT_NOT_SUPPORTED;
    default: return VK_ERROR_UNKNOWN;
    }
}

/* result of a library call -> VkResult, keeping the library's message */
VkResult to_vk(int rc)
{
    if (rc != RND_SUCCESS) g_error = rnd_last_error();
    return code_of(rc);
}

VkResult fail(VkResult r, const std::string& what)
{
    g_error = what;
    return r;
}

} // namespace

/* the handle types of the header, defined */
struct VkDevice_T { rnd_device dev; };
struct VkQueue_T { VkDevice_T* device; };
struct VmaAllocator_T { VkDevice_T* device; };
struct VkBuffer_T {
    VkDevice_T* device;
    rnd_buffer buf;            /* null for CPU_ONLY staging memory */
    std::vector<uint8_t> host; /* what a mapping returns */
    VkDeviceSize size;
    VkBufferUsageFlags usage;
    bool host_visible;
};
struct VmaAllocation_T; /* a VmaAllocation is its VkBuffer_T */
struct VkImageView_T {
    VkFormat format; uint32_t width, height; VkSampleCountFlagBits samples;
    rnd_texture texture; /* sampled views only */
};
struct VkSampler_T { int unused; };
struct VkShaderModule_T { int shader; VkShaderStageFlagBits stage; };
struct VkRenderPass_T {
    std::vector<VkAttachmentDescription> attachments;
    int color = -1, depth = -1, resolve = -1; /* attachment numbers of subpass 0 */
};
struct VkFramebuffer_T { VkDevice_T* device; rnd_target target; uint32_t width, height; };
struct VkDescriptorSetLayout_T { std::vector<VkDescriptorSetLayoutBinding> bindings; };
struct VkPipelineLayout_T { std::vector<VkPushConstantRange> ranges; };
struct VkDescriptorSet_T {
    struct Entry { uint32_t binding; VkDescriptorType type; VkBuffer_T* buffer; VkDeviceSize offset, range; VkImageView_T* view; };
    std::vector<Entry> entries;
};
struct VkDescriptorPool_T { std::vector<VkDescriptorSet_T*> sets; };
struct VkPipeline_T { rnd_pipeline pipeline; };
struct VkCommandPool_T { VkDevice_T* device; };
struct VkCommandBuffer_T {
    VkDevice_T* device;
    rnd_cmd cmd = nullptr;
    bool has_pass = false;
    struct Copy { VkBuffer_T* src; VkBuffer_T* dst; VkBufferCopy region; };
    std::vector<Copy> copies;
    int first_error = 0; /* vkCmd* return void: the first failure surfaces at vkEndCommandBuffer, as validation would put it */
};
struct VkFence_T { rnd_fence fence; bool signaled; bool submitted; };

namespace {
void note(VkCommandBuffer_T* c, int rc) { if (rc != RND_SUCCESS && !c->first_error) { c->first_error = rc; g_error = rnd_last_error(); } }
} // namespace

extern "C" {

const char* rndVkLastError(void) { return g_error.c_str(); }

/* ------------------------------------------------------------------ device */

VkResult rndVkCreateDevice(int hip_ordinal, VkDevice* pDevice, VkQueue* pGraphicsQueue, VmaAllocator* pAllocator)
{
    if (!pDevice) return fail(VK_ERROR_INITIALIZATION_FAILED, "rndVkCreateDevice: null output");
    rnd_device dev = nullptr;
    if (int rc = rnd_device_create(hip_ordinal, nullptr, &dev)) return to_vk(rc);
    VkDevice_T* d = new (std::nothrow) VkDevice_T{dev};
    if (!d) { rnd_device_destroy(dev); return VK_ERROR_OUT_OF_HOST_MEMORY; }
    *pDevice = d;
    if (pGraphicsQueue) *pGraphicsQueue = new VkQueue_T{d};
    if (pAllocator) *pAllocator = new VmaAllocator_T{d};
    return VK_SUCCESS;
}

void rndVkDestroyDevice(VkDevice device)
{
    if (!device) return;
    /* the library refuses to destroy a device that objects still point at: then the wrapper stays too (the handle
     * remains valid for destroying those objects) and rndVkLastError says why */
    if (rnd_device_destroy(device->dev) != RND_SUCCESS) { (void)fail(VK_ERROR_INITIALIZATION_FAILED, std::string("rndVkDestroyDevice: ") + rnd_last_error()); return; }
    delete device; /* the queue and allocator objects are a few bytes each and die with the process */
}

VkResult vkDeviceWaitIdle(VkDevice device) { return device ? to_vk(rnd_device_wait_idle(device->dev)) : VK_ERROR_DEVICE_LOST; }
VkResult vkQueueWaitIdle(VkQueue queue) { return queue ? vkDeviceWaitIdle(queue->device) : VK_ERROR_DEVICE_LOST; }

/* ------------------------------------------------------------------ buffers */

static VkResult create_buffer(VkDevice_T* device, const VkBufferCreateInfo* info, VmaMemoryUsage mem, VkBuffer* out)
{
    if (!device || !info || !out || !info->size) return fail(VK_ERROR_INITIALIZATION_FAILED, "buffer: bad arguments");
    VkBuffer_T* b = new (std::nothrow) VkBuffer_T{};
    if (!b) return VK_ERROR_OUT_OF_HOST_MEMORY;
    b->device = device; b->size = info->size; b->usage = info->usage; b->buf = nullptr;
    b->host_visible = mem == VMA_MEMORY_USAGE_CPU_ONLY || mem == VMA_MEMORY_USAGE_CPU_TO_GPU || mem == VMA_MEMORY_USAGE_GPU_TO_CPU;
    const uint32_t usage = info->usage & (VK_BUFFER_USAGE_VERTEX_BUFFER_BIT | VK_BUFFER_USAGE_INDEX_BUFFER_BIT | VK_BUFFER_USAGE_UNIFORM_BUFFER_BIT);
    if (usage) {
        if (int rc = rnd_buffer_create(device->dev, info->size, usage, nullptr, &b->buf)) { delete b; return to_vk(rc); }
    }
    if (b->host_visible) b->host.assign((size_t)info->size, 0);
    *out = b;
    return VK_SUCCESS;
}

VkResult vkCreateBuffer(VkDevice device, const VkBufferCreateInfo* pCreateInfo, const VkAllocationCallbacks*, VkBuffer* pBuffer)
{
    return create_buffer(device, pCreateInfo, VMA_MEMORY_USAGE_GPU_ONLY, pBuffer);
}

void vkDestroyBuffer(VkDevice, VkBuffer buffer, const VkAllocationCallbacks*)
{
    if (!buffer) return;
    rnd_buffer_destroy(buffer->buf);
    delete buffer;
}

VkResult vmaCreateBuffer(VmaAllocator allocator, const VkBufferCreateInfo* pBufferCreateInfo, const VmaAllocationCreateInfo* pAllocationCreateInfo,
                         VkBuffer* pBuffer, VmaAllocation* pAllocation, VmaAllocationInfo* pAllocationInfo)
{
    if (!allocator || !pAllocationCreateInfo) return fail(VK_ERROR_INITIALIZATION_FAILED, "vmaCreateBuffer: bad arguments");
    const VkResult r = create_buffer(allocator->device, pBufferCreateInfo, pAllocationCreateInfo->usage, pBuffer);
    if (r != VK_SUCCESS) return r;
    if (pAllocation) *pAllocation = reinterpret_cast<VmaAllocation>(*pBuffer);
    if (pAllocationInfo) {
        std::memset(pAllocationInfo, 0, sizeof(*pAllocationInfo));
        pAllocationInfo->size = pBufferCreateInfo->size;
    }
    return VK_SUCCESS;
}

void vmaDestroyBuffer(VmaAllocator, VkBuffer buffer, VmaAllocation) { vkDestroyBuffer(nullptr, buffer, nullptr); }

VkResult vmaMapMemory(VmaAllocator, VmaAllocation allocation, void** ppData)
{
    VkBuffer_T* b = reinterpret_cast<VkBuffer_T*>(allocation);
    if (!b || !ppData || !b->host_visible) return fail(VK_ERROR_MEMORY_MAP_FAILED, "vmaMapMemory: the allocation is not host visible");
    *ppData = b->host.data();
    return VK_SUCCESS;
}

void vmaUnmapMemory(VmaAllocator, VmaAllocation) {}

/* the mapped bytes reach the device here: frames submitted afterwards see them (rnd_buffer_write) */
VkResult vmaFlushAllocation(VmaAllocator, VmaAllocation allocation, VkDeviceSize offset, VkDeviceSize size)
{
    VkBuffer_T* b = reinterpret_cast<VkBuffer_T*>(allocation);
    if (!b || !b->host_visible) return fail(VK_ERROR_MEMORY_MAP_FAILED, "vmaFlushAllocation: the allocation is not host visible");
    if (!b->buf) return VK_SUCCESS; /* staging memory: read by vkCmdCopyBuffer */
    if (offset > b->size) return fail(VK_ERROR_MEMORY_MAP_FAILED, "vmaFlushAllocation: offset past the allocation");
    if (size == VK_WHOLE_SIZE || size > b->size - offset) size = b->size - offset;
    return to_vk(rnd_buffer_write(b->buf, offset, b->host.data() + offset, size));
}

/* ------------------------------------------------------------------ render pass, framebuffer */

VkResult vkCreateRenderPass(VkDevice, const VkRenderPassCreateInfo* info, const VkAllocationCallbacks*, VkRenderPass* pRenderPass)
{
    if (!info || !pRenderPass || info->subpassCount != 1 || !info->pSubpasses)
        return fail(VK_ERROR_FEATURE_NOT_PRESENT, "vkCreateRenderPass: exactly one subpass is supported (what the samples record)");
    const VkSubpassDescription& sp = info->pSubpasses[0];
    if (sp.colorAttachmentCount != 1 || !sp.pColorAttachments || !sp.pDepthStencilAttachment)
        return fail(VK_ERROR_FEATURE_NOT_PRESENT, "vkCreateRenderPass: one colour attachment and a depth attachment are required");
    VkRenderPass_T* rp = new (std::nothrow) VkRenderPass_T{};
    if (!rp) return VK_ERROR_OUT_OF_HOST_MEMORY;
    rp->attachments.assign(info->pAttachments, info->pAttachments + info->attachmentCount);
    rp->color = (int)sp.pColorAttachments[0].attachment;
    rp->depth = (int)sp.pDepthStencilAttachment->attachment;
    rp->resolve = sp.pResolveAttachments ? (int)sp.pResolveAttachments[0].attachment : -1;
    for (int a : {rp->color, rp->depth, rp->resolve})
        if (a >= (int)rp->attachments.size()) { delete rp; return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCreateRenderPass: attachment reference out of range"); }
    /* the samples clear and store colour and depth (test_ktx_app.cpp:78-101); nothing else is implemented */
    if (rp->attachments[rp->color].loadOp != VK_ATTACHMENT_LOAD_OP_CLEAR || rp->attachments[rp->depth].loadOp != VK_ATTACHMENT_LOAD_OP_CLEAR) {
        delete rp;
        return fail(VK_ERROR_FEATURE_NOT_PRESENT, "vkCreateRenderPass: attachments must be cleared on load");
    }
    *pRenderPass = rp;
    return VK_SUCCESS;
}

void vkDestroyRenderPass(VkDevice, VkRenderPass renderPass, const VkAllocationCallbacks*) { delete renderPass; }

VkResult rndVkCreateAttachmentView(VkDevice, VkFormat format, uint32_t width, uint32_t height, VkSampleCountFlagBits samples, VkImageView* pView)
{
    if (!pView) return fail(VK_ERROR_INITIALIZATION_FAILED, "rndVkCreateAttachmentView: null output");
    *pView = new VkImageView_T{format, width, height, samples, nullptr};
    return VK_SUCCESS;
}

void vkDestroyImageView(VkDevice, VkImageView imageView, const VkAllocationCallbacks*)
{
    if (!imageView) return;
    rnd_texture_destroy(imageView->texture);
    delete imageView;
}

void vkDestroySampler(VkDevice, VkSampler sampler, const VkAllocationCallbacks*) { delete sampler; }

VkResult vkCreateFramebuffer(VkDevice device, const VkFramebufferCreateInfo* info, const VkAllocationCallbacks*, VkFramebuffer* pFramebuffer)
{
    if (!device || !info || !pFramebuffer || !info->renderPass) return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCreateFramebuffer: bad arguments");
    const VkRenderPass_T* rp = info->renderPass;
    if (info->attachmentCount != rp->attachments.size()) return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCreateFramebuffer: attachment count differs from the render pass");
    /* with a resolve attachment the colour that is kept is the resolve image's (dummy_obj_viewer_app.cpp:468-520) */
    const int stored = rp->resolve >= 0 ? rp->resolve : rp->color;
    const VkImageView_T* cv = info->pAttachments[stored];
    const VkImageView_T* dv = info->pAttachments[rp->depth];
    if (!cv || !dv) return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCreateFramebuffer: null attachment");
    if (cv->format != rp->attachments[stored].format || dv->format != rp->attachments[rp->depth].format)
        return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCreateFramebuffer: a view's format differs from the render pass attachment's");
    const uint32_t samples = (uint32_t)rp->attachments[rp->color].samples;
    VkFramebuffer_T* fb = new (std::nothrow) VkFramebuffer_T{device, nullptr, info->width, info->height};
    if (!fb) return VK_ERROR_OUT_OF_HOST_MEMORY;
    if (int rc = rnd_target_create(device->dev, info->width, info->height, (int32_t)cv->format, (int32_t)dv->format, samples, &fb->target)) { delete fb; return to_vk(rc); }
    *pFramebuffer = fb;
    return VK_SUCCESS;
}

void vkDestroyFramebuffer(VkDevice, VkFramebuffer framebuffer, const VkAllocationCallbacks*)
{
    if (!framebuffer) return;
    rnd_target_destroy(framebuffer->target);
    delete framebuffer;
}

VkResult rndVkReadFramebuffer(VkFramebuffer framebuffer, void* color_out, uint32_t* depth_out)
{
    if (!framebuffer) return fail(VK_ERROR_INITIALIZATION_FAILED, "rndVkReadFramebuffer: null framebuffer");
    return to_vk(rnd_target_read(framebuffer->target, color_out, depth_out));
}

/* ------------------------------------------------------------------ textures */

VkResult rndVkCreateTexture2D(VkDevice device, uint32_t width, uint32_t height, VkFormat format, VkBool32 gen_mips, const void* texels,
                              VkBool32 linear, VkBool32 tiled, VkImageView* pView, VkSampler* pSampler)
{
    if (!device || !pView || !pSampler) return fail(VK_ERROR_INITIALIZATION_FAILED, "rndVkCreateTexture2D: bad arguments");
    const rnd_sampler_info si{linear ? RND_FILTER_LINEAR : RND_FILTER_NEAREST, tiled ? 1 : 0, 0.0f};
    rnd_texture t = nullptr;
    if (int rc = rnd_texture_create_2d(device->dev, width, height, (int32_t)format, gen_mips ? 1 : 0, texels, &si, &t)) return to_vk(rc);
    *pView = new VkImageView_T{format, width, height, VK_SAMPLE_COUNT_1_BIT, t};
    *pSampler = new VkSampler_T{0};
    return VK_SUCCESS;
}

/* ------------------------------------------------------------------ descriptor sets */

VkResult vkCreateDescriptorSetLayout(VkDevice, const VkDescriptorSetLayoutCreateInfo* info, const VkAllocationCallbacks*, VkDescriptorSetLayout* pSetLayout)
{
    if (!info || !pSetLayout) return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCreateDescriptorSetLayout: bad arguments");
    VkDescriptorSetLayout_T* l = new (std::nothrow) VkDescriptorSetLayout_T{};
    if (!l) return VK_ERROR_OUT_OF_HOST_MEMORY;
    if (info->bindingCount) l->bindings.assign(info->pBindings, info->pBindings + info->bindingCount);
    *pSetLayout = l;
    return VK_SUCCESS;
}
void vkDestroyDescriptorSetLayout(VkDevice, VkDescriptorSetLayout l, const VkAllocationCallbacks*) { delete l; }

VkResult vkCreatePipelineLayout(VkDevice, const VkPipelineLayoutCreateInfo* info, const VkAllocationCallbacks*, VkPipelineLayout* pPipelineLayout)
{
    if (!info || !pPipelineLayout) return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCreatePipelineLayout: bad arguments");
    VkPipelineLayout_T* l = new (std::nothrow) VkPipelineLayout_T{};
    if (!l) return VK_ERROR_OUT_OF_HOST_MEMORY;
    if (info->pushConstantRangeCount) l->ranges.assign(info->pPushConstantRanges, info->pPushConstantRanges + info->pushConstantRangeCount);
    for (const VkPushConstantRange& r : l->ranges)
        if (r.offset + r.size > 128) { delete l; return fail(VK_ERROR_FEATURE_NOT_PRESENT, "vkCreatePipelineLayout: push constants are limited to 128 bytes"); }
    *pPipelineLayout = l;
    return VK_SUCCESS;
}
void vkDestroyPipelineLayout(VkDevice, VkPipelineLayout l, const VkAllocationCallbacks*) { delete l; }

VkResult vkCreateDescriptorPool(VkDevice, const VkDescriptorPoolCreateInfo* info, const VkAllocationCallbacks*, VkDescriptorPool* pDescriptorPool)
{
    if (!info || !pDescriptorPool) return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCreateDescriptorPool: bad arguments");
    *pDescriptorPool = new VkDescriptorPool_T{};
    return VK_SUCCESS;
}
void vkDestroyDescriptorPool(VkDevice, VkDescriptorPool pool, const VkAllocationCallbacks*)
{
    if (!pool) return;
    for (VkDescriptorSet_T* s : pool->sets) delete s; /* sets are freed with their pool, as in Vulkan */
    delete pool;
}

VkResult vkAllocateDescriptorSets(VkDevice, const VkDescriptorSetAllocateInfo* info, VkDescriptorSet* pDescriptorSets)
{
    if (!info || !info->descriptorPool || !pDescriptorSets) return fail(VK_ERROR_INITIALIZATION_FAILED, "vkAllocateDescriptorSets: bad arguments");
    for (uint32_t i = 0; i < info->descriptorSetCount; ++i) {
        VkDescriptorSet_T* s = new (std::nothrow) VkDescriptorSet_T{};
        if (!s) return VK_ERROR_OUT_OF_HOST_MEMORY;
        info->descriptorPool->sets.push_back(s);
        pDescriptorSets[i] = s;
    }
    return VK_SUCCESS;
}

void vkUpdateDescriptorSets(VkDevice, uint32_t descriptorWriteCount, const VkWriteDescriptorSet* pDescriptorWrites, uint32_t, const VkCopyDescriptorSet*)
{
    for (uint32_t i = 0; i < descriptorWriteCount; ++i) {
        const VkWriteDescriptorSet& w = pDescriptorWrites[i];
        if (!w.dstSet || w.descriptorCount < 1) continue;
        VkDescriptorSet_T::Entry e{w.dstBinding, w.descriptorType, nullptr, 0, 0, nullptr};
        if (w.descriptorType == VK_DESCRIPTOR_TYPE_UNIFORM_BUFFER && w.pBufferInfo) {
            e.buffer = w.pBufferInfo->buffer; e.offset = w.pBufferInfo->offset; e.range = w.pBufferInfo->range;
        } else if (w.descriptorType == VK_DESCRIPTOR_TYPE_COMBINED_IMAGE_SAMPLER && w.pImageInfo) {
            e.view = w.pImageInfo->imageView;
        } else {
            continue; /* a descriptor type the shaders of this path do not read */
        }
        bool replaced = false;
        for (VkDescriptorSet_T::Entry& old : w.dstSet->entries) if (old.binding == e.binding) { old = e; replaced = true; }
        if (!replaced) w.dstSet->entries.push_back(e);
    }
}

/* ------------------------------------------------------------------ pipelines */

VkResult rndVkCreateShaderModuleFromFile(VkDevice, const char* path, VkShaderModule* pShaderModule)
{
    if (!path || !pShaderModule) return fail(VK_ERROR_INITIALIZATION_FAILED, "rndVkCreateShaderModuleFromFile: bad arguments");
    std::string name = path;
    const size_t slash = name.find_last_of("/\\");
    if (slash != std::string::npos) name = name.substr(slash + 1);
    for (const char* suffix : {".spv", ".glsl"}) {
        const size_t n = std::strlen(suffix);
        if (name.size() > n && name.compare(name.size() - n, n, suffix) == 0) name.resize(name.size() - n);
    }
    static const struct { const char* file; int shader; VkShaderStageFlagBits stage; } known[] = {
        {"test_ktx.vert", RND_SHADER_TEST_KTX, VK_SHADER_STAGE_VERTEX_BIT}, {"test_ktx.frag", RND_SHADER_TEST_KTX, VK_SHADER_STAGE_FRAGMENT_BIT},
        {"push_constants.vert", RND_SHADER_PUSH_CONSTANTS, VK_SHADER_STAGE_VERTEX_BIT}, {"push_constants.frag", RND_SHADER_PUSH_CONSTANTS, VK_SHADER_STAGE_FRAGMENT_BIT},
        {"dummy.vert", RND_SHADER_DUMMY, VK_SHADER_STAGE_VERTEX_BIT}, {"dummy.frag", RND_SHADER_DUMMY, VK_SHADER_STAGE_FRAGMENT_BIT}};
    for (const auto& k : known)
        if (name == k.file) { *pShaderModule = new VkShaderModule_T{k.shader, k.stage}; return VK_SUCCESS; }
    return fail(VK_ERROR_FEATURE_NOT_PRESENT, "no built-in shader is named '" + name + "' (SPIR-V is not ingested)");
}
void vkDestroyShaderModule(VkDevice, VkShaderModule m, const VkAllocationCallbacks*) { delete m; }

static VkResult one_pipeline(VkDevice_T* device, const VkGraphicsPipelineCreateInfo& ci, VkPipeline* out)
{
    if (!ci.pVertexInputState || !ci.pInputAssemblyState || !ci.pRasterizationState || !ci.pMultisampleState || !ci.pDepthStencilState || !ci.pColorBlendState || !ci.renderPass)
        return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCreateGraphicsPipelines: a state block is missing");
    rnd_pipeline_info info{};
    const VkShaderModule_T* vs = nullptr; const VkShaderModule_T* fs = nullptr;
    for (uint32_t i = 0; i < ci.stageCount; ++i) {
        const VkPipelineShaderStageCreateInfo& st = ci.pStages[i];
        if (!st.module || st.module->stage != st.stage) return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCreateGraphicsPipelines: a module is used for a stage it was not written for");
        if (st.stage == VK_SHADER_STAGE_VERTEX_BIT) vs = st.module;
        if (st.stage == VK_SHADER_STAGE_FRAGMENT_BIT) fs = st.module;
    }
    if (!vs || !fs || vs->shader != fs->shader) return fail(VK_ERROR_FEATURE_NOT_PRESENT, "vkCreateGraphicsPipelines: vertex and fragment shader must be one of the reference's pairs");
    info.shader = vs->shader;
    const VkPipelineVertexInputStateCreateInfo& vi = *ci.pVertexInputState;
    if (vi.vertexBindingDescriptionCount != 1 || vi.pVertexBindingDescriptions[0].inputRate != VK_VERTEX_INPUT_RATE_VERTEX)
        return fail(VK_ERROR_FEATURE_NOT_PRESENT, "vkCreateGraphicsPipelines: one per-vertex binding is supported (test_ktx_app.cpp:541-605)");
    info.stride = vi.pVertexBindingDescriptions[0].stride;
    /* The samples always declare three attributes and leave the ones a sub-geometry lacks at offset 0 with format
     * UNDEFINED-or-stale; locations are pos 0, normal 1, uv 2 (test_ktx.vert:3-5).  An attribute whose format is not
     * the one its location reads is absent. */
    for (uint32_t i = 0; i < vi.vertexAttributeDescriptionCount; ++i) {
        const VkVertexInputAttributeDescription& a = vi.pVertexAttributeDescriptions[i];
        if (a.location == 0 && a.format == VK_FORMAT_R32G32B32_SFLOAT) info.position = {(int32_t)a.format, a.offset};
        else if (a.location == 1 && a.format == VK_FORMAT_R32G32B32_SFLOAT) info.normal = {(int32_t)a.format, a.offset};
        else if (a.location == 2 && a.format == VK_FORMAT_R32G32_SFLOAT) info.uv = {(int32_t)a.format, a.offset};
    }
    if (ci.pInputAssemblyState->primitiveRestartEnable) return fail(VK_ERROR_FEATURE_NOT_PRESENT, "vkCreateGraphicsPipelines: primitive restart is not supported");
    info.topology = (int32_t)ci.pInputAssemblyState->topology;
    const VkPipelineRasterizationStateCreateInfo& rs = *ci.pRasterizationState;
    if (rs.polygonMode != VK_POLYGON_MODE_FILL || rs.depthClampEnable || rs.depthBiasEnable || rs.rasterizerDiscardEnable)
        return fail(VK_ERROR_FEATURE_NOT_PRESENT, "vkCreateGraphicsPipelines: FILL without depth clamp, bias or discard is supported (test_ktx_app.cpp:453-464)");
    info.cull_mode = (int32_t)rs.cullMode; info.front_face = (int32_t)rs.frontFace;
    const VkPipelineDepthStencilStateCreateInfo& ds = *ci.pDepthStencilState;
    if (ds.stencilTestEnable || ds.depthBoundsTestEnable) return fail(VK_ERROR_FEATURE_NOT_PRESENT, "vkCreateGraphicsPipelines: stencil and depth-bounds tests are not supported");
    info.depth_test = ds.depthTestEnable ? 1 : 0; info.depth_write = ds.depthWriteEnable ? 1 : 0; info.depth_compare = (int32_t)ds.depthCompareOp;
    info.samples = (uint32_t)ci.pMultisampleState->rasterizationSamples;
    if (ci.pMultisampleState->sampleShadingEnable) return fail(VK_ERROR_FEATURE_NOT_PRESENT, "vkCreateGraphicsPipelines: sample shading is not supported");
    if ((uint32_t)ci.renderPass->attachments[ci.renderPass->color].samples != info.samples)
        return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCreateGraphicsPipelines: rasterizationSamples differs from the render pass attachment's");
    const VkPipelineColorBlendStateCreateInfo& cb = *ci.pColorBlendState;
    if (cb.attachmentCount != 1 || cb.logicOpEnable) return fail(VK_ERROR_FEATURE_NOT_PRESENT, "vkCreateGraphicsPipelines: one colour attachment without logic op is supported");
    info.blend_enable = cb.pAttachments[0].blendEnable ? 1 : 0;
    info.color_write_mask = cb.pAttachments[0].colorWriteMask;
    rnd_pipeline p = nullptr;
    if (int rc = rnd_pipeline_create(device->dev, &info, &p)) return to_vk(rc);
    *out = new VkPipeline_T{p};
    return VK_SUCCESS;
}

VkResult vkCreateGraphicsPipelines(VkDevice device, VkPipelineCache, uint32_t createInfoCount, const VkGraphicsPipelineCreateInfo* pCreateInfos, const VkAllocationCallbacks*, VkPipeline* pPipelines)
{
    if (!device || !pCreateInfos || !pPipelines) return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCreateGraphicsPipelines: bad arguments");
    for (uint32_t i = 0; i < createInfoCount; ++i) pPipelines[i] = VK_NULL_HANDLE;
    for (uint32_t i = 0; i < createInfoCount; ++i) {
        const VkResult r = one_pipeline(device, pCreateInfos[i], &pPipelines[i]);
        if (r != VK_SUCCESS) return r;
    }
    return VK_SUCCESS;
}

void vkDestroyPipeline(VkDevice, VkPipeline pipeline, const VkAllocationCallbacks*)
{
    if (!pipeline) return;
    rnd_pipeline_destroy(pipeline->pipeline);
    delete pipeline;
}

/* ------------------------------------------------------------------ command buffers */

VkResult vkCreateCommandPool(VkDevice device, const VkCommandPoolCreateInfo*, const VkAllocationCallbacks*, VkCommandPool* pCommandPool)
{
    if (!device || !pCommandPool) return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCreateCommandPool: bad arguments");
    *pCommandPool = new VkCommandPool_T{device};
    return VK_SUCCESS;
}
void vkDestroyCommandPool(VkDevice, VkCommandPool pool, const VkAllocationCallbacks*) { delete pool; }

VkResult vkAllocateCommandBuffers(VkDevice device, const VkCommandBufferAllocateInfo* info, VkCommandBuffer* pCommandBuffers)
{
    if (!device || !info || !pCommandBuffers) return fail(VK_ERROR_INITIALIZATION_FAILED, "vkAllocateCommandBuffers: bad arguments");
    for (uint32_t i = 0; i < info->commandBufferCount; ++i) {
        VkCommandBuffer_T* c = new (std::nothrow) VkCommandBuffer_T{};
        if (!c) return VK_ERROR_OUT_OF_HOST_MEMORY;
        c->device = device;
        if (int rc = rnd_cmd_create(device->dev, &c->cmd)) { delete c; return to_vk(rc); }
        pCommandBuffers[i] = c;
    }
    return VK_SUCCESS;
}

void vkFreeCommandBuffers(VkDevice, VkCommandPool, uint32_t commandBufferCount, const VkCommandBuffer* pCommandBuffers)
{
    for (uint32_t i = 0; i < commandBufferCount; ++i) {
        if (!pCommandBuffers[i]) continue;
        rnd_cmd_destroy(pCommandBuffers[i]->cmd);
        delete pCommandBuffers[i];
    }
}

VkResult vkBeginCommandBuffer(VkCommandBuffer c, const VkCommandBufferBeginInfo*)
{
    if (!c) return fail(VK_ERROR_INITIALIZATION_FAILED, "vkBeginCommandBuffer: null command buffer");
    c->copies.clear(); c->has_pass = false; c->first_error = 0;
    return VK_SUCCESS; /* the library's command buffer begins with its render pass: one that only copies buffers has none */
}

VkResult vkEndCommandBuffer(VkCommandBuffer c)
{
    if (!c) return fail(VK_ERROR_INITIALIZATION_FAILED, "vkEndCommandBuffer: null command buffer");
    if (c->first_error) return code_of(c->first_error); /* the message of the command that failed is kept */
    return c->has_pass ? to_vk(rnd_cmd_end(c->cmd)) : VK_SUCCESS;
}

void vkCmdBeginRenderPass(VkCommandBuffer c, const VkRenderPassBeginInfo* b, VkSubpassContents)
{
    if (!c || !b || !b->framebuffer || !b->renderPass) return;
    const VkRenderPass_T* rp = b->renderPass;
    float color[4] = {0, 0, 0, 0}, depth = 1.0f;
    if ((uint32_t)rp->color < b->clearValueCount) std::memcpy(color, b->pClearValues[rp->color].color.float32, sizeof(color));
    if ((uint32_t)rp->depth < b->clearValueCount) depth = b->pClearValues[rp->depth].depthStencil.depth;
    c->has_pass = true;
    note(c, rnd_cmd_begin(c->cmd));
    note(c, rnd_cmd_begin_pass(c->cmd, b->framebuffer->target, color, depth, b->renderArea.offset.x, b->renderArea.offset.y, b->renderArea.extent.width, b->renderArea.extent.height));
}

void vkCmdEndRenderPass(VkCommandBuffer c) { if (c) note(c, rnd_cmd_end_pass(c->cmd)); }

void vkCmdSetViewport(VkCommandBuffer c, uint32_t firstViewport, uint32_t viewportCount, const VkViewport* v)
{
    if (!c || firstViewport != 0 || viewportCount < 1) return;
    note(c, rnd_cmd_set_viewport(c->cmd, v->x, v->y, v->width, v->height, v->minDepth, v->maxDepth));
}

void vkCmdSetScissor(VkCommandBuffer c, uint32_t firstScissor, uint32_t scissorCount, const VkRect2D* s)
{
    if (!c || firstScissor != 0 || scissorCount < 1) return;
    note(c, rnd_cmd_set_scissor(c->cmd, s->offset.x, s->offset.y, s->extent.width, s->extent.height));
}

void vkCmdBindPipeline(VkCommandBuffer c, VkPipelineBindPoint, VkPipeline pipeline) { if (c && pipeline) note(c, rnd_cmd_bind_pipeline(c->cmd, pipeline->pipeline)); }

void vkCmdBindDescriptorSets(VkCommandBuffer c, VkPipelineBindPoint, VkPipelineLayout, uint32_t firstSet, uint32_t descriptorSetCount,
                             const VkDescriptorSet* pDescriptorSets, uint32_t, const uint32_t*)
{
    if (!c) return;
    for (uint32_t s = 0; s < descriptorSetCount; ++s) {
        if (!pDescriptorSets[s]) continue;
        for (const VkDescriptorSet_T::Entry& e : pDescriptorSets[s]->entries) {
            if (e.type == VK_DESCRIPTOR_TYPE_UNIFORM_BUFFER && e.buffer && e.buffer->buf)
                note(c, rnd_cmd_bind_uniform(c->cmd, firstSet + s, e.binding, e.buffer->buf, e.offset, e.range == VK_WHOLE_SIZE ? e.buffer->size - e.offset : e.range));
            else if (e.type == VK_DESCRIPTOR_TYPE_COMBINED_IMAGE_SAMPLER && e.view && e.view->texture)
                note(c, rnd_cmd_bind_texture(c->cmd, firstSet + s, e.binding, e.view->texture));
        }
    }
}

void vkCmdPushConstants(VkCommandBuffer c, VkPipelineLayout, VkShaderStageFlags, uint32_t offset, uint32_t size, const void* pValues)
{
    if (c) note(c, rnd_cmd_push_constants(c->cmd, offset, size, pValues));
}

void vkCmdBindVertexBuffers(VkCommandBuffer c, uint32_t firstBinding, uint32_t bindingCount, const VkBuffer* pBuffers, const VkDeviceSize* pOffsets)
{
    if (!c || firstBinding != 0 || bindingCount < 1 || !pBuffers[0]) return;
    note(c, rnd_cmd_bind_vertex_buffer(c->cmd, pBuffers[0]->buf, pOffsets[0]));
}

void vkCmdBindIndexBuffer(VkCommandBuffer c, VkBuffer buffer, VkDeviceSize offset, VkIndexType indexType)
{
    if (c && buffer) note(c, rnd_cmd_bind_index_buffer(c->cmd, buffer->buf, offset, (int32_t)indexType));
}

void vkCmdDrawIndexed(VkCommandBuffer c, uint32_t indexCount, uint32_t instanceCount, uint32_t firstIndex, int32_t vertexOffset, uint32_t firstInstance)
{
    if (c) note(c, rnd_cmd_draw_indexed(c->cmd, indexCount, instanceCount, firstIndex, vertexOffset, firstInstance));
}

void vkCmdCopyBuffer(VkCommandBuffer c, VkBuffer srcBuffer, VkBuffer dstBuffer, uint32_t regionCount, const VkBufferCopy* pRegions)
{
    if (!c || !srcBuffer || !dstBuffer) return;
    for (uint32_t i = 0; i < regionCount; ++i) c->copies.push_back({srcBuffer, dstBuffer, pRegions[i]});
}

/* ------------------------------------------------------------------ fences, submit */

VkResult vkCreateFence(VkDevice device, const VkFenceCreateInfo* info, const VkAllocationCallbacks*, VkFence* pFence)
{
    if (!device || !pFence) return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCreateFence: bad arguments");
    rnd_fence f = nullptr;
    if (int rc = rnd_fence_create(device->dev, &f)) return to_vk(rc);
    *pFence = new VkFence_T{f, info && (info->flags & VK_FENCE_CREATE_SIGNALED_BIT), false};
    return VK_SUCCESS;
}

void vkDestroyFence(VkDevice, VkFence fence, const VkAllocationCallbacks*)
{
    if (!fence) return;
    rnd_fence_destroy(fence->fence);
    delete fence;
}

VkResult vkWaitForFences(VkDevice, uint32_t fenceCount, const VkFence* pFences, VkBool32, uint64_t)
{
    for (uint32_t i = 0; i < fenceCount; ++i) {
        VkFence_T* f = pFences[i];
        if (!f || f->signaled) continue;
        if (!f->submitted) return fail(VK_TIMEOUT, "vkWaitForFences: an unsignaled fence that no submit will signal");
        if (int rc = rnd_fence_wait(f->fence)) return to_vk(rc);
        f->signaled = true;
    }
    return VK_SUCCESS;
}

VkResult vkResetFences(VkDevice, uint32_t fenceCount, const VkFence* pFences)
{
    for (uint32_t i = 0; i < fenceCount; ++i) if (pFences[i]) { pFences[i]->signaled = false; pFences[i]->submitted = false; }
    return VK_SUCCESS;
}

VkResult vkQueueSubmit(VkQueue queue, uint32_t submitCount, const VkSubmitInfo* pSubmits, VkFence fence)
{
    if (!queue) return fail(VK_ERROR_DEVICE_LOST, "vkQueueSubmit: null queue");
    rnd_device dev = queue->device->dev;
    VkCommandBuffer_T* last_pass = nullptr;
    for (uint32_t s = 0; s < submitCount; ++s)
        for (uint32_t k = 0; k < pSubmits[s].commandBufferCount; ++k)
            if (pSubmits[s].pCommandBuffers[k] && pSubmits[s].pCommandBuffers[k]->has_pass) last_pass = pSubmits[s].pCommandBuffers[k];
    for (uint32_t s = 0; s < submitCount; ++s) {
        for (uint32_t k = 0; k < pSubmits[s].commandBufferCount; ++k) {
            VkCommandBuffer_T* c = pSubmits[s].pCommandBuffers[k];
            if (!c) continue;
            /* buffer copies recorded outside a render pass: staging memory into device buffers, in order */
            for (const VkCommandBuffer_T::Copy& cp : c->copies) {
                if (!cp.dst->buf || cp.src->host.empty()) return fail(VK_ERROR_FEATURE_NOT_PRESENT, "vkCmdCopyBuffer: supported from host-visible staging memory into a device buffer");
                if (cp.region.srcOffset + cp.region.size > cp.src->size || cp.region.dstOffset + cp.region.size > cp.dst->size)
                    return fail(VK_ERROR_INITIALIZATION_FAILED, "vkCmdCopyBuffer: region outside a buffer");
                if (int rc = rnd_buffer_write(cp.dst->buf, cp.region.dstOffset, cp.src->host.data() + cp.region.srcOffset, cp.region.size)) return to_vk(rc);
            }
            if (c->has_pass) {
                /* the fence is signalled when everything submitted has executed: it rides on the last pass */
                if (int rc = rnd_submit(dev, c->cmd, (fence && c == last_pass) ? fence->fence : nullptr)) return to_vk(rc);
            }
        }
    }
    if (fence) {
        if (!last_pass) {
            /* nothing but copies: they are queued on the device's stream; the fence is complete once that drains */
            if (int rc = rnd_device_wait_idle(dev)) return to_vk(rc);
            fence->signaled = true;
        }
        fence->submitted = true;
    }
    return VK_SUCCESS;
}

} // extern "C"
