// handles.hpp -- move-only owners for the library's handles, shaped like the reference's
// vulkan_base_handler (src/libs/vk_utils/vk_utils/handlers.hpp:34-118): non-copyable, move by
// swap, destroy() idempotent and null-safe, implicit conversion to the raw handle.
#pragma once
#include <utility>

#include "../../include/renderer_amd.h"

namespace rnd_host {

template <typename Handle, int (*Destroy)(Handle)>
class handler {
public:
    handler() = default;
    explicit handler(Handle h) : m_handle(h) {}
    handler(const handler&) = delete;
    handler& operator=(const handler&) = delete;
    handler(handler&& o) noexcept { std::swap(m_handle, o.m_handle); }
    handler& operator=(handler&& o) noexcept { if (this != &o) { destroy(); std::swap(m_handle, o.m_handle); } return *this; }
    ~handler() { destroy(); }
    void destroy() { if (m_handle) { Destroy(m_handle); m_handle = nullptr; } }
    Handle* put() { destroy(); return &m_handle; } // out-parameter of a *_create call
    operator Handle() const { return m_handle; }
    Handle handle() const { return m_handle; }
    explicit operator bool() const { return m_handle != nullptr; }
private:
    Handle m_handle{nullptr};
};

using device_handler = handler<rnd_device, rnd_device_destroy>;
using buffer_handler = handler<rnd_buffer, rnd_buffer_destroy>;     // vma_buffer_handler
using texture_handler = handler<rnd_texture, rnd_texture_destroy>;  // vma_image_handler + image_view_handler + sampler_handler
using target_handler = handler<rnd_target, rnd_target_destroy>;     // framebuffer + attachments
using pipeline_handler = handler<rnd_pipeline, rnd_pipeline_destroy>;
using cmd_buffer_handler = handler<rnd_cmd, rnd_cmd_destroy>;
using fence_handler = handler<rnd_fence, rnd_fence_destroy>;

} // namespace rnd_host
