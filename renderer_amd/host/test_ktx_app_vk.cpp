// test_ktx_app_vk.cpp -- the reference's test_ktx sample (src/samples/test_ktx_app/test_ktx_app.cpp) written against
// the Vulkan names it uses itself: every step of the sample that lies on the draw path calls the same vk* / vma*
// entry point with the same structures as the reference does, and those entry points are librnd_vk.so's
// (include/rnd_vk.h) over the HIP library.  Steps and their reference lines:
//   init_render_passes      :75-162     vkCreateRenderPass
//   init_framebuffers       :164-258    the depth image's view, vkCreateFramebuffer
//   init_descriptor_set     :261-410    vkCreateDescriptorSetLayout / PipelineLayout / DescriptorPool, vkAllocateDescriptorSets, vkUpdateDescriptorSets
//   init_pipelines          :412-625    vkCreateGraphicsPipelines, one pipeline per sub-geometry
//   record_command_buffers  :627-746    the static draw list
//   draw_frame              base_obj_viewer_app.cpp:90-141 (camera, map / memcpy / flush / unmap) and vk_app.cpp:433-467 (fence, vkQueueSubmit)
//   create_buffer           tools.cpp:788-835 (staging buffer, vkCmdCopyBuffer, one-time submit)
// What the reference gets from its window system is replaced, headless: the swapchain image is an attachment view of
// the preferred format (vk_app.cpp:197-221), presenting is rndVkReadFramebuffer.  The mesh and texture are the same
// procedural stand-ins as in test_ktx_app.cpp of this directory, so both programs print the same checksums.
//
//   test_ktx_app_vk_headless [--width W] [--height H] [--frames N] [--out frame.ppm] [--samples 1|4] [--shader ktx|push_constants]
// --samples 4 builds the render pass and framebuffer of the obj viewer sample (dummy_obj_viewer_app.cpp:461-639: multisampled colour
// and depth, resolve into the swapchain image); --shader push_constants is test_push_constants_app.cpp's shader pair and descriptor set.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/rnd_vk.h"
#include "base_obj_viewer_app.hpp" // camera, global_ubo, the model container's host half
#include "procedural.hpp"

using namespace rnd_host;

namespace {

#define VK_TRY(call, what) do { const VkResult r_ = (call); if (r_ != VK_SUCCESS) RAISE_ERROR_FATAL((int)r_, std::string(what) + ": " + rndVkLastError()); } while (0)

struct push_constant_data { // test_ktx_app.hpp:16-19
    glm_lite::vec4 color;
    glm_lite::mat4 transform;
};

struct vk_model { // obj_loader.hpp:61-89 with Vulkan buffers
    VkBuffer vertex_buffer{VK_NULL_HANDLE};
    VmaAllocation vertex_allocation{nullptr};
    VkBuffer index_buffer{VK_NULL_HANDLE};
    VmaAllocation index_allocation{nullptr};
    std::vector<obj_sub_geometry> sub_geometries{};
    glm_lite::mat4 model_transform{1.0f};
};

class test_ktx_app {
public:
    test_ktx_app(uint32_t w, uint32_t h, uint32_t samples, bool push_constants_sample)
        : m_extent{w, h}, m_samples(samples == 4 ? VK_SAMPLE_COUNT_4_BIT : VK_SAMPLE_COUNT_1_BIT), m_push_constants_sample(push_constants_sample)
    {
        m_camera.eye_position = {0, 0, -2}; // base_obj_viewer_app.cpp:12-14
        m_camera.target_position = {0, 0, 0};
        m_camera.fov = 90.0f;
    }

    ~test_ktx_app()
    {
        if (!m_device) return;
        vkDeviceWaitIdle(m_device);
        vkDestroyFence(m_device, m_fence, nullptr);
        if (!m_command_buffers.empty()) vkFreeCommandBuffers(m_device, m_command_pool, (uint32_t)m_command_buffers.size(), m_command_buffers.data());
        vkDestroyCommandPool(m_device, m_command_pool, nullptr);
        for (VkPipeline p : m_graphics_pipelines) vkDestroyPipeline(m_device, p, nullptr);
        vkDestroyShaderModule(m_device, m_vert_shader, nullptr);
        vkDestroyShaderModule(m_device, m_frag_shader, nullptr);
        vkDestroyDescriptorPool(m_device, m_descriptor_pool, nullptr);
        vkDestroyPipelineLayout(m_device, m_pipeline_layout, nullptr);
        vkDestroyDescriptorSetLayout(m_device, m_descriptor_set_layout, nullptr);
        for (VkFramebuffer f : m_main_pass_framebuffers) vkDestroyFramebuffer(m_device, f, nullptr);
        vkDestroyImageView(m_device, m_main_depth_image_view, nullptr);
        vkDestroyImageView(m_device, m_msaa_color_image_view, nullptr);
        for (VkImageView v : m_swapchain_image_views) vkDestroyImageView(m_device, v, nullptr);
        vkDestroyRenderPass(m_device, m_main_render_pass, nullptr);
        vkDestroySampler(m_device, m_sampler, nullptr);
        vkDestroyImageView(m_device, m_texture_view, nullptr);
        vmaDestroyBuffer(m_allocator, m_ubo, m_ubo_allocation);
        vmaDestroyBuffer(m_allocator, m_model.vertex_buffer, m_model.vertex_allocation);
        vmaDestroyBuffer(m_allocator, m_model.index_buffer, m_model.index_allocation);
        rndVkDestroyDevice(m_device);
    }

    ERROR_TYPE on_vulkan_initialized(int hip_ordinal)
    {
        VK_TRY(rndVkCreateDevice(hip_ordinal, &m_device, &m_graphics_queue, &m_allocator), "context::init");
        // the "swapchain": one image of the format the reference prefers (vk_app.cpp:197-221)
        m_swapchain_image_views.resize(1);
        VK_TRY(rndVkCreateAttachmentView(m_device, m_swapchain_format, m_extent.width, m_extent.height, VK_SAMPLE_COUNT_1_BIT, &m_swapchain_image_views[0]), "swapchain image view");

        VkCommandPoolCreateInfo pool_info{
            .sType = VK_STRUCTURE_TYPE_COMMAND_POOL_CREATE_INFO,
            .pNext = nullptr,
            .flags = VK_COMMAND_POOL_CREATE_RESET_COMMAND_BUFFER_BIT,
            .queueFamilyIndex = 0};
        VK_TRY(vkCreateCommandPool(m_device, &pool_info, nullptr, &m_command_pool), "command pool");

        // base_obj_viewer_app.cpp:29-40: the uniform buffer, host visible
        VkBufferCreateInfo ubo_info{
            .sType = VK_STRUCTURE_TYPE_BUFFER_CREATE_INFO,
            .pNext = nullptr,
            .flags = 0,
            .size = sizeof(global_ubo),
            .usage = VK_BUFFER_USAGE_UNIFORM_BUFFER_BIT,
            .sharingMode = VK_SHARING_MODE_EXCLUSIVE,
            .queueFamilyIndexCount = 0,
            .pQueueFamilyIndices = nullptr};
        VmaAllocationCreateInfo ubo_alloc_info{};
        ubo_alloc_info.usage = VMA_MEMORY_USAGE_CPU_TO_GPU;
        VK_TRY(vmaCreateBuffer(m_allocator, &ubo_info, &ubo_alloc_info, &m_ubo, &m_ubo_allocation, nullptr), "uniform buffer");

        PASS_ERROR(init_model());
        const std::vector<uint8_t> texels = make_checker(1024);
        VK_TRY(rndVkCreateTexture2D(m_device, 1024, 1024, VK_FORMAT_R8G8B8A8_SRGB, VK_TRUE, texels.data(), VK_TRUE, VK_FALSE, &m_texture_view, &m_sampler), "create_texture_2D");

        PASS_ERROR(init_render_passes());
        PASS_ERROR(init_framebuffers());
        PASS_ERROR(init_descriptor_set());
        PASS_ERROR(init_pipelines());
        PASS_ERROR(record_command_buffers());

        VkFenceCreateInfo fence_info{.sType = VK_STRUCTURE_TYPE_FENCE_CREATE_INFO, .pNext = nullptr, .flags = VK_FENCE_CREATE_SIGNALED_BIT};
        VK_TRY(vkCreateFence(m_device, &fence_info, nullptr, &m_fence), "fence");
        RAISE_ERROR_OK();
    }

    // vk_utils::create_buffer (tools.cpp:788-835): fill a host-visible staging buffer, copy it into a device-local one
    ERROR_TYPE create_buffer(VkBuffer& buffer, VmaAllocation& allocation, VkBufferUsageFlags usage, const void* data, size_t size)
    {
        VkBufferCreateInfo staging_info{
            .sType = VK_STRUCTURE_TYPE_BUFFER_CREATE_INFO, .pNext = nullptr, .flags = 0, .size = size,
            .usage = VK_BUFFER_USAGE_TRANSFER_SRC_BIT, .sharingMode = VK_SHARING_MODE_EXCLUSIVE, .queueFamilyIndexCount = 0, .pQueueFamilyIndices = nullptr};
        VmaAllocationCreateInfo staging_alloc{};
        staging_alloc.usage = VMA_MEMORY_USAGE_CPU_ONLY;
        VkBuffer staging{VK_NULL_HANDLE};
        VmaAllocation staging_allocation{nullptr};
        VK_TRY(vmaCreateBuffer(m_allocator, &staging_info, &staging_alloc, &staging, &staging_allocation, nullptr), "staging buffer");
        void* mapped = nullptr;
        VK_TRY(vmaMapMemory(m_allocator, staging_allocation, &mapped), "map staging buffer");
        std::memcpy(mapped, data, size);
        VK_TRY(vmaFlushAllocation(m_allocator, staging_allocation, 0, VK_WHOLE_SIZE), "flush staging buffer");
        vmaUnmapMemory(m_allocator, staging_allocation);

        VkBufferCreateInfo buffer_info = staging_info;
        buffer_info.usage = usage | VK_BUFFER_USAGE_TRANSFER_DST_BIT;
        VmaAllocationCreateInfo alloc{};
        alloc.usage = VMA_MEMORY_USAGE_GPU_ONLY;
        VK_TRY(vmaCreateBuffer(m_allocator, &buffer_info, &alloc, &buffer, &allocation, nullptr), "device buffer");

        VkCommandBufferAllocateInfo cmd_info{
            .sType = VK_STRUCTURE_TYPE_COMMAND_BUFFER_ALLOCATE_INFO, .pNext = nullptr, .commandPool = m_command_pool,
            .level = VK_COMMAND_BUFFER_LEVEL_PRIMARY, .commandBufferCount = 1};
        VkCommandBuffer copy_cmd{VK_NULL_HANDLE};
        VK_TRY(vkAllocateCommandBuffers(m_device, &cmd_info, &copy_cmd), "copy command buffer");
        VkCommandBufferBeginInfo begin{.sType = VK_STRUCTURE_TYPE_COMMAND_BUFFER_BEGIN_INFO, .pNext = nullptr,
                                       .flags = VK_COMMAND_BUFFER_USAGE_ONE_TIME_SUBMIT_BIT, .pInheritanceInfo = nullptr};
        vkBeginCommandBuffer(copy_cmd, &begin);
        VkBufferCopy region{.srcOffset = 0, .dstOffset = 0, .size = size};
        vkCmdCopyBuffer(copy_cmd, staging, buffer, 1, &region);
        vkEndCommandBuffer(copy_cmd);
        VkSubmitInfo submit{.sType = VK_STRUCTURE_TYPE_SUBMIT_INFO, .pNext = nullptr, .waitSemaphoreCount = 0, .pWaitSemaphores = nullptr,
                            .pWaitDstStageMask = nullptr, .commandBufferCount = 1, .pCommandBuffers = &copy_cmd, .signalSemaphoreCount = 0, .pSignalSemaphores = nullptr};
        VK_TRY(vkQueueSubmit(m_graphics_queue, 1, &submit, VK_NULL_HANDLE), "submit copy");
        VK_TRY(vkQueueWaitIdle(m_graphics_queue), "wait for copy");
        vkFreeCommandBuffers(m_device, m_command_pool, 1, &copy_cmd);
        vmaDestroyBuffer(m_allocator, staging, staging_allocation);
        RAISE_ERROR_OK();
    }

    // init_obj_geometry's tail (obj_loader.cpp:144-310) with a procedural sphere for the Git-LFS stub
    ERROR_TYPE init_model()
    {
        model_buffers b;
        PASS_ERROR(build_model_buffers({make_uv_sphere(64, 32, 4.0f)}, b));
        PASS_ERROR(create_buffer(m_model.vertex_buffer, m_model.vertex_allocation, VK_BUFFER_USAGE_VERTEX_BUFFER_BIT, b.vertices.data(), b.vertices.size() * sizeof(float)));
        PASS_ERROR(create_buffer(m_model.index_buffer, m_model.index_allocation, VK_BUFFER_USAGE_INDEX_BUFFER_BIT, b.indices.data(), b.indices.size() * sizeof(uint32_t)));
        m_model.sub_geometries = b.sub_geometries;
        m_model.model_transform = b.model_transform;
        RAISE_ERROR_OK();
    }

    ERROR_TYPE init_render_passes() // :75-162
    {
        VkAttachmentDescription pass_attachments[]{
            {.flags = 0,
             .format = m_swapchain_format,
             .samples = VK_SAMPLE_COUNT_1_BIT,
             .loadOp = VK_ATTACHMENT_LOAD_OP_CLEAR,
             .storeOp = VK_ATTACHMENT_STORE_OP_STORE,
             .stencilLoadOp = VK_ATTACHMENT_LOAD_OP_DONT_CARE,
             .stencilStoreOp = VK_ATTACHMENT_STORE_OP_DONT_CARE,
             .initialLayout = VK_IMAGE_LAYOUT_UNDEFINED,
             .finalLayout = VK_IMAGE_LAYOUT_PRESENT_SRC_KHR},
            {.flags = 0,
             .format = VK_FORMAT_D24_UNORM_S8_UINT,
             .samples = VK_SAMPLE_COUNT_1_BIT,
             .loadOp = VK_ATTACHMENT_LOAD_OP_CLEAR,
             .storeOp = VK_ATTACHMENT_STORE_OP_STORE,
             .stencilLoadOp = VK_ATTACHMENT_LOAD_OP_DONT_CARE,
             .stencilStoreOp = VK_ATTACHMENT_STORE_OP_DONT_CARE,
             .initialLayout = VK_IMAGE_LAYOUT_UNDEFINED,
             .finalLayout = VK_IMAGE_LAYOUT_DEPTH_STENCIL_ATTACHMENT_OPTIMAL},
        };
        VkAttachmentReference pass_attachment_refs[]{
            {.attachment = 0, .layout = VK_IMAGE_LAYOUT_COLOR_ATTACHMENT_OPTIMAL},
            {.attachment = 1, .layout = VK_IMAGE_LAYOUT_DEPTH_STENCIL_ATTACHMENT_OPTIMAL}};
        VkSubpassDescription pass_subpasses[]{
            {.flags = 0,
             .pipelineBindPoint = VK_PIPELINE_BIND_POINT_GRAPHICS,
             .inputAttachmentCount = 0,
             .pInputAttachments = nullptr,
             .colorAttachmentCount = 1,
             .pColorAttachments = pass_attachment_refs,
             .pResolveAttachments = nullptr,
             .pDepthStencilAttachment = pass_attachment_refs + 1,
             .preserveAttachmentCount = 0,
             .pPreserveAttachments = nullptr}};
        VkSubpassDependency pass_deps[]{
            {.srcSubpass = VK_SUBPASS_EXTERNAL,
             .dstSubpass = 0,
             .srcStageMask = VK_PIPELINE_STAGE_COLOR_ATTACHMENT_OUTPUT_BIT,
             .dstStageMask = VK_PIPELINE_STAGE_COLOR_ATTACHMENT_OUTPUT_BIT,
             .srcAccessMask = VK_ACCESS_COLOR_ATTACHMENT_READ_BIT,
             .dstAccessMask = VK_ACCESS_COLOR_ATTACHMENT_WRITE_BIT,
             .dependencyFlags = 0},
            {.srcSubpass = 0,
             .dstSubpass = VK_SUBPASS_EXTERNAL,
             .srcStageMask = VK_PIPELINE_STAGE_COLOR_ATTACHMENT_OUTPUT_BIT,
             .dstStageMask = VK_PIPELINE_STAGE_COLOR_ATTACHMENT_OUTPUT_BIT,
             .srcAccessMask = VK_ACCESS_COLOR_ATTACHMENT_WRITE_BIT,
             .dstAccessMask = VK_ACCESS_COLOR_ATTACHMENT_READ_BIT,
             .dependencyFlags = 0}};
        VkRenderPassCreateInfo pass_info{
            .sType = VK_STRUCTURE_TYPE_RENDER_PASS_CREATE_INFO,
            .pNext = nullptr,
            .flags = 0,
            .attachmentCount = std::size(pass_attachments),
            .pAttachments = pass_attachments,
            .subpassCount = std::size(pass_subpasses),
            .pSubpasses = pass_subpasses,
            .dependencyCount = std::size(pass_deps),
            .pDependencies = pass_deps};
        // The multisampled pass of the obj viewer sample (dummy_obj_viewer_app.cpp:468-520): attachments {resolve image 1x,
        // depth 4x, colour 4x}; the subpass draws into attachment 2 and resolves into attachment 0.
        VkAttachmentDescription msaa_attachments[]{pass_attachments[0], pass_attachments[1], pass_attachments[0]};
        msaa_attachments[0].loadOp = VK_ATTACHMENT_LOAD_OP_DONT_CARE;
        msaa_attachments[1].samples = m_samples;
        msaa_attachments[2].samples = m_samples;
        msaa_attachments[2].finalLayout = VK_IMAGE_LAYOUT_COLOR_ATTACHMENT_OPTIMAL;
        VkAttachmentReference msaa_refs[]{
            {.attachment = 2, .layout = VK_IMAGE_LAYOUT_COLOR_ATTACHMENT_OPTIMAL},
            {.attachment = 1, .layout = VK_IMAGE_LAYOUT_DEPTH_STENCIL_ATTACHMENT_OPTIMAL},
            {.attachment = 0, .layout = VK_IMAGE_LAYOUT_COLOR_ATTACHMENT_OPTIMAL}};
        VkSubpassDescription msaa_subpass = pass_subpasses[0];
        msaa_subpass.pColorAttachments = msaa_refs;
        msaa_subpass.pDepthStencilAttachment = msaa_refs + 1;
        msaa_subpass.pResolveAttachments = msaa_refs + 2;
        if (m_samples != VK_SAMPLE_COUNT_1_BIT) {
            pass_info.attachmentCount = std::size(msaa_attachments);
            pass_info.pAttachments = msaa_attachments;
            pass_info.pSubpasses = &msaa_subpass;
        }
        if (vkCreateRenderPass(m_device, &pass_info, nullptr, &m_main_render_pass) != VK_SUCCESS) {
            RAISE_ERROR_FATAL(-1, std::string("cannot init main pass: ") + rndVkLastError());
        }
        RAISE_ERROR_OK();
    }

    ERROR_TYPE init_framebuffers() // :164-258
    {
        VK_TRY(rndVkCreateAttachmentView(m_device, VK_FORMAT_D24_UNORM_S8_UINT, m_extent.width, m_extent.height, m_samples, &m_main_depth_image_view), "depth image view");
        if (m_samples != VK_SAMPLE_COUNT_1_BIT) // the multisampled colour image (dummy_obj_viewer_app.cpp:546-600)
            VK_TRY(rndVkCreateAttachmentView(m_device, m_swapchain_format, m_extent.width, m_extent.height, m_samples, &m_msaa_color_image_view), "multisampled colour image view");
        m_main_pass_framebuffers.reserve(m_swapchain_image_views.size());
        for (size_t i = 0; i < m_swapchain_image_views.size(); ++i) {
            VkImageView fb_attachments[]{m_swapchain_image_views[i], m_main_depth_image_view, m_msaa_color_image_view};
            VkFramebufferCreateInfo fb_info{
                .sType = VK_STRUCTURE_TYPE_FRAMEBUFFER_CREATE_INFO,
                .pNext = nullptr,
                .flags = 0,
                .renderPass = m_main_render_pass,
                .attachmentCount = m_samples != VK_SAMPLE_COUNT_1_BIT ? 3u : 2u,
                .pAttachments = fb_attachments,
                .width = m_extent.width,
                .height = m_extent.height,
                .layers = 1};
            VkFramebuffer fb{VK_NULL_HANDLE};
            if (vkCreateFramebuffer(m_device, &fb_info, nullptr, &fb) != VK_SUCCESS) {
                RAISE_ERROR_FATAL(-1, std::string("cannot init framebuffer: ") + rndVkLastError());
            }
            m_main_pass_framebuffers.push_back(fb);
        }
        RAISE_ERROR_OK();
    }

    ERROR_TYPE init_descriptor_set() // :261-410
    {
        VkDescriptorSetLayoutBinding bindings[]{
            {.binding = 0, .descriptorType = VK_DESCRIPTOR_TYPE_UNIFORM_BUFFER, .descriptorCount = 1, .stageFlags = VK_SHADER_STAGE_VERTEX_BIT, .pImmutableSamplers = nullptr},
            {.binding = 1, .descriptorType = VK_DESCRIPTOR_TYPE_COMBINED_IMAGE_SAMPLER, .descriptorCount = 1, .stageFlags = VK_SHADER_STAGE_FRAGMENT_BIT, .pImmutableSamplers = nullptr}};
        // test_push_constants_app.cpp:196-345: the same set without the sampler
        const uint32_t n_bindings = m_push_constants_sample ? 1u : (uint32_t)std::size(bindings);
        VkDescriptorSetLayoutCreateInfo desc_set_layout_info{
            .sType = VK_STRUCTURE_TYPE_DESCRIPTOR_SET_LAYOUT_CREATE_INFO, .pNext = nullptr, .flags = 0, .bindingCount = n_bindings, .pBindings = bindings};
        VK_TRY(vkCreateDescriptorSetLayout(m_device, &desc_set_layout_info, nullptr, &m_descriptor_set_layout), "descriptor set layout");

        VkPushConstantRange push_constant_range{.stageFlags = VK_SHADER_STAGE_VERTEX_BIT, .offset = 0, .size = sizeof(push_constant_data)};
        VkPipelineLayoutCreateInfo pipeline_layout_info{
            .sType = VK_STRUCTURE_TYPE_PIPELINE_LAYOUT_CREATE_INFO, .pNext = nullptr, .flags = 0, .setLayoutCount = 1, .pSetLayouts = &m_descriptor_set_layout,
            .pushConstantRangeCount = 1, .pPushConstantRanges = &push_constant_range};
        VK_TRY(vkCreatePipelineLayout(m_device, &pipeline_layout_info, nullptr, &m_pipeline_layout), "pipeline layout");

        VkDescriptorPoolSize pool_sizes[]{{.type = VK_DESCRIPTOR_TYPE_UNIFORM_BUFFER, .descriptorCount = 1}, {.type = VK_DESCRIPTOR_TYPE_COMBINED_IMAGE_SAMPLER, .descriptorCount = 1}};
        VkDescriptorPoolCreateInfo desc_pool_info{
            .sType = VK_STRUCTURE_TYPE_DESCRIPTOR_POOL_CREATE_INFO, .pNext = nullptr, .flags = 0, .maxSets = 1, .poolSizeCount = std::size(pool_sizes), .pPoolSizes = pool_sizes};
        VK_TRY(vkCreateDescriptorPool(m_device, &desc_pool_info, nullptr, &m_descriptor_pool), "descriptor pool");

        VkDescriptorSetAllocateInfo descriptor_sets_alloc_info{
            .sType = VK_STRUCTURE_TYPE_DESCRIPTOR_SET_ALLOCATE_INFO, .pNext = nullptr, .descriptorPool = m_descriptor_pool, .descriptorSetCount = 1, .pSetLayouts = &m_descriptor_set_layout};
        VK_TRY(vkAllocateDescriptorSets(m_device, &descriptor_sets_alloc_info, &m_descriptor_set), "descriptor set");

        VkDescriptorBufferInfo buffer_info{.buffer = m_ubo, .offset = 0, .range = sizeof(global_ubo)};
        VkDescriptorImageInfo image_info{.sampler = m_sampler, .imageView = m_texture_view, .imageLayout = VK_IMAGE_LAYOUT_SHADER_READ_ONLY_OPTIMAL};
        VkWriteDescriptorSet write_ops[]{
            {.sType = VK_STRUCTURE_TYPE_WRITE_DESCRIPTOR_SET, .pNext = nullptr, .dstSet = m_descriptor_set, .dstBinding = 0, .dstArrayElement = 0, .descriptorCount = 1,
             .descriptorType = VK_DESCRIPTOR_TYPE_UNIFORM_BUFFER, .pImageInfo = nullptr, .pBufferInfo = &buffer_info, .pTexelBufferView = nullptr},
            {.sType = VK_STRUCTURE_TYPE_WRITE_DESCRIPTOR_SET, .pNext = nullptr, .dstSet = m_descriptor_set, .dstBinding = 1, .dstArrayElement = 0, .descriptorCount = 1,
             .descriptorType = VK_DESCRIPTOR_TYPE_COMBINED_IMAGE_SAMPLER, .pImageInfo = &image_info, .pBufferInfo = nullptr, .pTexelBufferView = nullptr}};
        vkUpdateDescriptorSets(m_device, n_bindings, write_ops, 0, nullptr);
        RAISE_ERROR_OK();
    }

    ERROR_TYPE init_pipelines() // :412-625
    {
        VK_TRY(rndVkCreateShaderModuleFromFile(m_device, m_push_constants_sample ? "./shaders/push_constants.vert.spv" : "./shaders/test_ktx.vert.spv", &m_vert_shader), "load_shader");
        VK_TRY(rndVkCreateShaderModuleFromFile(m_device, m_push_constants_sample ? "./shaders/push_constants.frag.spv" : "./shaders/test_ktx.frag.spv", &m_frag_shader), "load_shader");
        VkPipelineShaderStageCreateInfo shader_stages[]{
            {.sType = VK_STRUCTURE_TYPE_PIPELINE_SHADER_STAGE_CREATE_INFO, .pNext = nullptr, .flags = 0, .stage = VK_SHADER_STAGE_VERTEX_BIT, .module = m_vert_shader, .pName = "main", .pSpecializationInfo = nullptr},
            {.sType = VK_STRUCTURE_TYPE_PIPELINE_SHADER_STAGE_CREATE_INFO, .pNext = nullptr, .flags = 0, .stage = VK_SHADER_STAGE_FRAGMENT_BIT, .module = m_frag_shader, .pName = "main", .pSpecializationInfo = nullptr}};
        VkPipelineInputAssemblyStateCreateInfo input_assembly_info{
            .sType = VK_STRUCTURE_TYPE_PIPELINE_INPUT_ASSEMBLY_STATE_CREATE_INFO, .pNext = nullptr, .flags = 0,
            .topology = VK_PRIMITIVE_TOPOLOGY_TRIANGLE_LIST, .primitiveRestartEnable = VK_FALSE};
        VkPipelineViewportStateCreateInfo viewport_info{
            .sType = VK_STRUCTURE_TYPE_PIPELINE_VIEWPORT_STATE_CREATE_INFO, .pNext = nullptr, .flags = 0,
            .viewportCount = 1, .pViewports = nullptr, .scissorCount = 1, .pScissors = nullptr};
        VkPipelineRasterizationStateCreateInfo rasterization_info{
            .sType = VK_STRUCTURE_TYPE_PIPELINE_RASTERIZATION_STATE_CREATE_INFO, .pNext = nullptr, .flags = 0,
            .depthClampEnable = VK_FALSE, .rasterizerDiscardEnable = VK_FALSE, .polygonMode = VK_POLYGON_MODE_FILL,
            .cullMode = VK_CULL_MODE_BACK_BIT, .frontFace = VK_FRONT_FACE_COUNTER_CLOCKWISE, .depthBiasEnable = VK_FALSE,
            .depthBiasConstantFactor = 0, .depthBiasClamp = 0, .depthBiasSlopeFactor = 0, .lineWidth = 1};
        VkPipelineMultisampleStateCreateInfo multisample_info{
            .sType = VK_STRUCTURE_TYPE_PIPELINE_MULTISAMPLE_STATE_CREATE_INFO, .pNext = nullptr, .flags = 0,
            .rasterizationSamples = m_samples, .sampleShadingEnable = VK_FALSE, .minSampleShading = 0,
            .pSampleMask = nullptr, .alphaToCoverageEnable = VK_FALSE, .alphaToOneEnable = VK_FALSE};
        VkPipelineDepthStencilStateCreateInfo depth_stencil_info{
            .sType = VK_STRUCTURE_TYPE_PIPELINE_DEPTH_STENCIL_STATE_CREATE_INFO, .pNext = nullptr, .flags = 0,
            .depthTestEnable = VK_TRUE, .depthWriteEnable = VK_TRUE, .depthCompareOp = VK_COMPARE_OP_LESS_OR_EQUAL,
            .depthBoundsTestEnable = VK_FALSE, .stencilTestEnable = VK_FALSE, .front = {}, .back = {}, .minDepthBounds = 0, .maxDepthBounds = 1};
        VkPipelineColorBlendAttachmentState color_blend_attachment{
            .blendEnable = VK_FALSE, .srcColorBlendFactor = VK_BLEND_FACTOR_ONE, .dstColorBlendFactor = VK_BLEND_FACTOR_ZERO, .colorBlendOp = VK_BLEND_OP_ADD,
            .srcAlphaBlendFactor = VK_BLEND_FACTOR_ONE, .dstAlphaBlendFactor = VK_BLEND_FACTOR_ZERO, .alphaBlendOp = VK_BLEND_OP_ADD,
            .colorWriteMask = VK_COLOR_COMPONENT_R_BIT | VK_COLOR_COMPONENT_G_BIT | VK_COLOR_COMPONENT_B_BIT | VK_COLOR_COMPONENT_A_BIT};
        VkPipelineColorBlendStateCreateInfo color_blend_info{
            .sType = VK_STRUCTURE_TYPE_PIPELINE_COLOR_BLEND_STATE_CREATE_INFO, .pNext = nullptr, .flags = 0, .logicOpEnable = VK_FALSE, .logicOp = VK_LOGIC_OP_COPY,
            .attachmentCount = 1, .pAttachments = &color_blend_attachment, .blendConstants = {0, 0, 0, 0}};
        VkDynamicState dyn_state_flags[]{VK_DYNAMIC_STATE_VIEWPORT, VK_DYNAMIC_STATE_SCISSOR};
        VkPipelineDynamicStateCreateInfo dyn_state_info{
            .sType = VK_STRUCTURE_TYPE_PIPELINE_DYNAMIC_STATE_CREATE_INFO, .pNext = nullptr, .flags = 0,
            .dynamicStateCount = std::size(dyn_state_flags), .pDynamicStates = dyn_state_flags};

        m_graphics_pipelines.reserve(m_model.sub_geometries.size());
        for (const obj_sub_geometry& subgeom : m_model.sub_geometries) {
            // :541-605: one binding, three attributes whose offsets accumulate over the sub-geometry's format
            uint32_t stride = 0;
            VkVertexInputAttributeDescription vert_attrs[3]{};
            for (size_t a = 0; a < subgeom.format.size() && a < 3; ++a) {
                vert_attrs[a] = {.location = (uint32_t)a, .binding = 0, .format = (VkFormat)subgeom.format[a], .offset = stride};
                stride += subgeom.format[a] == VK_FORMAT_R32G32_SFLOAT ? 8u : 12u;
            }
            // a sub-geometry without normals keeps its uv at location 2 (test_ktx.vert:3-5)
            if (subgeom.format.size() == 2 && subgeom.format[1] == VK_FORMAT_R32G32_SFLOAT) vert_attrs[1].location = 2;
            VkVertexInputBindingDescription vert_binding{.binding = 0, .stride = stride, .inputRate = VK_VERTEX_INPUT_RATE_VERTEX};
            VkPipelineVertexInputStateCreateInfo vert_input_info{
                .sType = VK_STRUCTURE_TYPE_PIPELINE_VERTEX_INPUT_STATE_CREATE_INFO, .pNext = nullptr, .flags = 0,
                .vertexBindingDescriptionCount = 1, .pVertexBindingDescriptions = &vert_binding,
                .vertexAttributeDescriptionCount = (uint32_t)subgeom.format.size(), .pVertexAttributeDescriptions = vert_attrs};
            VkGraphicsPipelineCreateInfo pipeline_create_info{
                .sType = VK_STRUCTURE_TYPE_GRAPHICS_PIPELINE_CREATE_INFO,
                .pNext = nullptr,
                .flags = 0,
                .stageCount = std::size(shader_stages),
                .pStages = shader_stages,
                .pVertexInputState = &vert_input_info,
                .pInputAssemblyState = &input_assembly_info,
                .pTessellationState = nullptr,
                .pViewportState = &viewport_info,
                .pRasterizationState = &rasterization_info,
                .pMultisampleState = &multisample_info,
                .pDepthStencilState = &depth_stencil_info,
                .pColorBlendState = &color_blend_info,
                .pDynamicState = &dyn_state_info,
                .layout = m_pipeline_layout,
                .renderPass = m_main_render_pass,
                .subpass = 0,
                .basePipelineHandle = VK_NULL_HANDLE,
                .basePipelineIndex = -1};
            VkPipeline pipeline{VK_NULL_HANDLE};
            if (vkCreateGraphicsPipelines(m_device, VK_NULL_HANDLE, 1, &pipeline_create_info, nullptr, &pipeline) != VK_SUCCESS) {
                RAISE_ERROR_FATAL(-1, std::string("cannot init pipeline: ") + rndVkLastError());
            }
            m_graphics_pipelines.push_back(pipeline);
        }
        RAISE_ERROR_OK();
    }

    ERROR_TYPE record_command_buffers() // :627-746
    {
        VkCommandBufferAllocateInfo buffers_alloc_info{
            .sType = VK_STRUCTURE_TYPE_COMMAND_BUFFER_ALLOCATE_INFO,
            .pNext = nullptr,
            .commandPool = m_command_pool,
            .level = VK_COMMAND_BUFFER_LEVEL_PRIMARY,
            .commandBufferCount = static_cast<uint32_t>(m_main_pass_framebuffers.size())};
        std::vector<VkCommandBuffer> cmd_buffers(m_main_pass_framebuffers.size());
        if (vkAllocateCommandBuffers(m_device, &buffers_alloc_info, cmd_buffers.data()) != VK_SUCCESS) {
            RAISE_ERROR_FATAL(-1, "cannot init command buffers");
        }

        VkCommandBufferBeginInfo cmd_buffer_begin_info{
            .sType = VK_STRUCTURE_TYPE_COMMAND_BUFFER_BEGIN_INFO,
            .pNext = nullptr,
            .flags = 0,
            .pInheritanceInfo = nullptr};

        VkClearValue clear_values[]{
            {.color{.float32 = {0.2f, 0.3f, 0.7f, 1.f}}},
            {.depthStencil = {.depth = 1.0f, .stencil = 0}},
            {.color{.float32 = {0.2f, 0.3f, 0.7f, 1.f}}}}; // the multisampled pass clears its third attachment, the colour image

        VkRenderPassBeginInfo pass_begin_info{
            .sType = VK_STRUCTURE_TYPE_RENDER_PASS_BEGIN_INFO,
            .pNext = nullptr,
            .renderPass = m_main_render_pass,
            .framebuffer = VK_NULL_HANDLE,
            .renderArea = {
                .offset = {0, 0},
                .extent = m_extent},
            .clearValueCount = m_samples != VK_SAMPLE_COUNT_1_BIT ? 3u : 2u,
            .pClearValues = clear_values,
        };

        VkViewport viewport{
            .x = 0,
            .y = 0,
            .width = static_cast<float>(m_extent.width),
            .height = static_cast<float>(m_extent.height),
            .minDepth = 0,
            .maxDepth = 1,
        };

        VkRect2D scissor{
            .offset = {0, 0},
            .extent = m_extent};

        static glm_lite::vec4 colors[]{
            {0., 0., 0., 1.},
            {1., 0., 0., 1.},
            {0., 1., 0., 1.},
            {0., 0., 1., 1.},
            {1., 1., 0., 1.},
            {0., 1., 1., 1.},
            {1., 0., 1., 1.},
            {1., 1., 1., 1.},
            {1., 0.5, 0., 1.},
            {0., 0.5, 1., 1.},
        };

        float rot_angle = 360.0f / std::size(colors);

        for (size_t i = 0; i < m_main_pass_framebuffers.size(); ++i) {
            pass_begin_info.framebuffer = m_main_pass_framebuffers[i];

            vkBeginCommandBuffer(cmd_buffers[i], &cmd_buffer_begin_info);

            vkCmdBeginRenderPass(cmd_buffers[i], &pass_begin_info, VK_SUBPASS_CONTENTS_INLINE);
            vkCmdSetViewport(cmd_buffers[i], 0, 1, &viewport);
            vkCmdSetScissor(cmd_buffers[i], 0, 1, &scissor);

            VkPipeline last_pipeline = VK_NULL_HANDLE;

            float curr_angle = 0;

            for (size_t j = 0; j < std::size(colors); j++) {
                push_constant_data data{
                    .color = colors[j],
                    .transform = glm_lite::mat4{1}};

                data.transform = glm_lite::rotate(data.transform, glm_lite::radians(curr_angle), {0.0f, 0.0f, 1.0f});
                data.transform = glm_lite::translate(data.transform, {0, 10, 0});

                curr_angle += rot_angle;

                vkCmdPushConstants(cmd_buffers[i], m_pipeline_layout, VK_SHADER_STAGE_VERTEX_BIT, 0, sizeof(push_constant_data), &data);

                for (size_t g = 0; g < m_model.sub_geometries.size(); g++) {
                    if (m_graphics_pipelines[g] != last_pipeline) {
                        vkCmdBindPipeline(cmd_buffers[i], VK_PIPELINE_BIND_POINT_GRAPHICS, m_graphics_pipelines[g]);
                        vkCmdBindDescriptorSets(cmd_buffers[i], VK_PIPELINE_BIND_POINT_GRAPHICS, m_pipeline_layout, 0, 1, &m_descriptor_set, 0, nullptr);
                        last_pipeline = m_graphics_pipelines[g];
                    }

                    VkDeviceSize vertex_buffer_offset = m_model.sub_geometries[g].vertices_offset;
                    vkCmdBindVertexBuffers(cmd_buffers[i], 0, 1, &m_model.vertex_buffer, &vertex_buffer_offset);
                    vkCmdBindIndexBuffer(cmd_buffers[i], m_model.index_buffer, m_model.sub_geometries[g].indices_offset * sizeof(uint32_t), VK_INDEX_TYPE_UINT32);
                    vkCmdDrawIndexed(cmd_buffers[i], m_model.sub_geometries[g].indices_size, 1, 0, 0, 0);
                }
            }

            vkCmdEndRenderPass(cmd_buffers[i]);

            if (vkEndCommandBuffer(cmd_buffers[i]) != VK_SUCCESS) {
                RAISE_ERROR_FATAL(-1, std::string("cannot record command buffer: ") + rndVkLastError());
            }
        }

        m_command_buffers = std::move(cmd_buffers);

        RAISE_ERROR_OK();
    }

    void orbit(float x_angle_deg, float y_angle_deg) // base_obj_viewer_app.cpp:121-125
    {
        const glm_lite::mat4 r = glm_lite::yawPitchRoll(glm_lite::radians(x_angle_deg), glm_lite::radians(y_angle_deg), 0.0f);
        m_camera.eye_position = glm_lite::mat3_mul(r, m_camera.eye_position);
    }

    void set_model_scale(float s) { m_model.model_transform = glm_lite::scale(glm_lite::mat4(1.0f), {s, s, s}); }

    ERROR_TYPE draw_frame()
    {
        // vk_app::begin_frame (vk_app.cpp:433-442): the image's frame must have left the queue
        VK_TRY(vkWaitForFences(m_device, 1, &m_fence, VK_TRUE, UINT64_MAX), "wait for fence");
        VK_TRY(vkResetFences(m_device, 1, &m_fence), "reset fence");
        const uint32_t image_index = 0;

        // base_obj_viewer_app::draw_frame (base_obj_viewer_app.cpp:126-138)
        PASS_ERROR(m_camera.update((float)m_extent.width, (float)m_extent.height));
        global_ubo ubo_data;
        ubo_data.model = glm_lite::mat4(1.0f) * m_model.model_transform;
        ubo_data.view_projection = m_camera.view_proj_matrix;
        ubo_data.projection = m_camera.proj_matrix;
        ubo_data.view = m_camera.view_matrix;
        ubo_data.mvp = ubo_data.projection * ubo_data.view * ubo_data.model;
        void* mapped_data = nullptr;
        VK_TRY(vmaMapMemory(m_allocator, m_ubo_allocation, &mapped_data), "map uniform buffer");
        std::memcpy(mapped_data, &ubo_data, sizeof(ubo_data));
        VK_TRY(vmaFlushAllocation(m_allocator, m_ubo_allocation, 0, VK_WHOLE_SIZE), "flush uniform buffer");
        vmaUnmapMemory(m_allocator, m_ubo_allocation);

        // vk_app::finish_frame (vk_app.cpp:444-467)
        VkPipelineStageFlags wait_stage = VK_PIPELINE_STAGE_COLOR_ATTACHMENT_OUTPUT_BIT;
        VkSubmitInfo submit_info{
            .sType = VK_STRUCTURE_TYPE_SUBMIT_INFO,
            .pNext = nullptr,
            .waitSemaphoreCount = 0,
            .pWaitSemaphores = nullptr,
            .pWaitDstStageMask = &wait_stage,
            .commandBufferCount = 1,
            .pCommandBuffers = &m_command_buffers[image_index],
            .signalSemaphoreCount = 0,
            .pSignalSemaphores = nullptr};
        VK_TRY(vkQueueSubmit(m_graphics_queue, 1, &submit_info, m_fence), "queue submit");
        RAISE_ERROR_OK();
    }

    ERROR_TYPE read_back(std::vector<uint8_t>& color, std::vector<uint32_t>& depth)
    {
        color.resize((size_t)m_extent.width * m_extent.height * 4);
        depth.resize((size_t)m_extent.width * m_extent.height * (size_t)m_samples);
        VK_TRY(vkWaitForFences(m_device, 1, &m_fence, VK_TRUE, UINT64_MAX), "wait for fence");
        VK_TRY(rndVkReadFramebuffer(m_main_pass_framebuffers[0], color.data(), depth.data()), "read framebuffer");
        RAISE_ERROR_OK();
    }

private:
    VkExtent2D m_extent;
    VkFormat m_swapchain_format{VK_FORMAT_R8G8B8A8_SRGB};
    VkSampleCountFlagBits m_samples;
    bool m_push_constants_sample;
    camera m_camera{};
    VkDevice m_device{VK_NULL_HANDLE};
    VkQueue m_graphics_queue{VK_NULL_HANDLE};
    VmaAllocator m_allocator{nullptr};
    VkCommandPool m_command_pool{VK_NULL_HANDLE};
    VkBuffer m_ubo{VK_NULL_HANDLE};
    VmaAllocation m_ubo_allocation{nullptr};
    vk_model m_model{};
    VkImageView m_texture_view{VK_NULL_HANDLE};
    VkSampler m_sampler{VK_NULL_HANDLE};
    std::vector<VkImageView> m_swapchain_image_views{};
    VkImageView m_main_depth_image_view{VK_NULL_HANDLE};
    VkImageView m_msaa_color_image_view{VK_NULL_HANDLE};
    VkRenderPass m_main_render_pass{VK_NULL_HANDLE};
    std::vector<VkFramebuffer> m_main_pass_framebuffers{};
    VkDescriptorSetLayout m_descriptor_set_layout{VK_NULL_HANDLE};
    VkPipelineLayout m_pipeline_layout{VK_NULL_HANDLE};
    VkDescriptorPool m_descriptor_pool{VK_NULL_HANDLE};
    VkDescriptorSet m_descriptor_set{VK_NULL_HANDLE};
    VkShaderModule m_vert_shader{VK_NULL_HANDLE}, m_frag_shader{VK_NULL_HANDLE};
    std::vector<VkPipeline> m_graphics_pipelines{};
    std::vector<VkCommandBuffer> m_command_buffers{};
    VkFence m_fence{VK_NULL_HANDLE};
};

uint64_t fnv1a(const void* data, size_t n, uint64_t h = 1469598103934665603ull)
{
    const uint8_t* p = static_cast<const uint8_t*>(data);
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

ERROR_TYPE run(int argc, const char** argv)
{
    uint32_t w = 1280, h = 720;
    int frames = 3;
    uint32_t samples = 1;
    bool push_constants_sample = false;
    std::string out;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--width" && i + 1 < argc) w = (uint32_t)std::atoi(argv[++i]);
        else if (a == "--height" && i + 1 < argc) h = (uint32_t)std::atoi(argv[++i]);
        else if (a == "--frames" && i + 1 < argc) frames = std::atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--samples" && i + 1 < argc) samples = (uint32_t)std::atoi(argv[++i]);
        else if (a == "--shader" && i + 1 < argc) push_constants_sample = std::string(argv[++i]) == "push_constants";
        else RAISE_ERROR_FATAL(-1, "invalid argument " + a);
    }
    test_ktx_app app(w, h, samples, push_constants_sample);
    PASS_ERROR(app.on_vulkan_initialized(0));
    app.set_model_scale(0.12f);
    for (int f = 0; f < frames; ++f) {
        if (f) app.orbit(0.5f, 0.0f);
        PASS_ERROR(app.draw_frame());
    }
    std::vector<uint8_t> color;
    std::vector<uint32_t> depth;
    PASS_ERROR(app.read_back(color, depth));
    size_t covered = 0;
    for (uint32_t d : depth) covered += d != 0xFFFFFFu;
    std::printf("frames=%d extent=%ux%u covered=%zu color_fnv=%016llx depth_fnv=%016llx\n", frames, w, h, covered,
                (unsigned long long)fnv1a(color.data(), color.size()), (unsigned long long)fnv1a(depth.data(), depth.size() * 4));
    if (!out.empty()) {
        FILE* fp = std::fopen(out.c_str(), "wb");
        if (!fp) RAISE_ERROR_WARN(-1, "cannot open " + out);
        std::fprintf(fp, "P6\n%u %u\n255\n", w, h);
        for (size_t i = 0; i < (size_t)w * h; ++i) std::fwrite(&color[i * 4], 1, 3, fp);
        std::fclose(fp);
    }
    RAISE_ERROR_OK();
}

} // namespace

int main(int argc, const char** argv)
{
    try {
        run(argc, argv);
    } catch (const errors::error& e) {
        std::fprintf(stderr, "[ERROR] %s\n", e.what());
        return 1;
    }
    return 0;
}
