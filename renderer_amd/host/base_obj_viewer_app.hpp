// base_obj_viewer_app.hpp -- the per-frame host work of the reference's sample base class
// (src/samples/base_obj_viewer_app/base_obj_viewer_app.{hpp,cpp}) against the HIP library:
// camera (vk_utils::camera, camera.hpp:11-39 / camera.cpp:9-25), the 320-byte global_ubo
// (base_obj_viewer_app.hpp:36-43) and draw_frame's update + upload (base_obj_viewer_app.cpp:90-141).
// Windowing, swapchain and input callbacks stay with the host application; orbit() takes the
// place of the mouse handlers (base_obj_viewer_app.cpp:98-125).
#pragma once
#include <cstring>

#include "errors.hpp"
#include "glm_lite.hpp"
#include "handles.hpp"
#include "obj_model.hpp"

namespace rnd_host {

class camera {
public:
    ERROR_TYPE update(float extent_width, float extent_height)
    {
        // near and far are hard-coded in the reference (camera.cpp:13); z_near / z_far are not used
        proj_matrix = glm_lite::perspectiveFov(glm_lite::radians(fov), extent_width, extent_height, 0.01f, 100.0f, depth_zero_to_one);
        view_matrix = glm_lite::lookAt(eye_position, target_position, up);
        view_proj_matrix = proj_matrix * view_matrix;
        RAISE_ERROR_OK();
    }
    float fov{90};
    float z_near{0.01f};
    float z_far{100.0f};
    bool depth_zero_to_one{false}; // what the reference build effectively uses, see glm_lite::perspectiveFov
    glm_lite::vec3 eye_position{0, 0, -1};
    glm_lite::vec3 target_position{0, 0, 0};
    glm_lite::vec3 up{0, 1, 0};
    glm_lite::mat4 view_matrix{1.0f};
    glm_lite::mat4 proj_matrix{1.0f};
    glm_lite::mat4 view_proj_matrix{1.0f};
};

struct global_ubo {
    glm_lite::mat4 projection{1.0f};
    glm_lite::mat4 view_projection{1.0f};
    glm_lite::mat4 view{1.0f};
    glm_lite::mat4 model{1.0f};
    glm_lite::mat4 mvp{1.0f};
};
static_assert(sizeof(global_ubo) == 320, "five column-major mat4");

class base_obj_viewer_app {
public:
    base_obj_viewer_app(uint32_t width, uint32_t height) : m_width(width), m_height(height)
    {
        m_camera.eye_position = {0, 0, -2}; // base_obj_viewer_app.cpp:12-14
        m_camera.target_position = {0, 0, 0};
        m_camera.fov = 90.0f;
    }
    virtual ~base_obj_viewer_app() = default;

    // context::init + the uniform buffer (base_obj_viewer_app.cpp:25-44)
    virtual ERROR_TYPE on_vulkan_initialized(int hip_ordinal)
    {
        RND_CHECK(rnd_device_create(hip_ordinal, nullptr, m_device.put()));
        RND_CHECK(rnd_buffer_create(m_device, sizeof(global_ubo), RND_BUFFER_USAGE_UNIFORM, nullptr, m_ubo.put()));
        RAISE_ERROR_OK();
    }

    // one step of the mouse orbit: eye = mat3(yawPitchRoll(x, y, 0)) * eye (base_obj_viewer_app.cpp:121-125)
    void orbit(float x_angle_deg, float y_angle_deg)
    {
        const glm_lite::mat4 r = glm_lite::yawPitchRoll(glm_lite::radians(x_angle_deg), glm_lite::radians(y_angle_deg), 0.0f);
        m_camera.eye_position = glm_lite::mat3_mul(r, m_camera.eye_position);
    }

    // camera + uniform block, written every frame (base_obj_viewer_app.cpp:127-138)
    virtual ERROR_TYPE draw_frame()
    {
        PASS_ERROR(m_camera.update((float)m_width, (float)m_height));
        global_ubo ubo_data;
        ubo_data.model = glm_lite::mat4(1.0f) * m_model.model_transform;
        ubo_data.view_projection = m_camera.view_proj_matrix;
        ubo_data.projection = m_camera.proj_matrix;
        ubo_data.view = m_camera.view_matrix;
        ubo_data.mvp = ubo_data.projection * ubo_data.view * ubo_data.model;
        m_last_ubo = ubo_data;
        RND_CHECK(rnd_buffer_write(m_ubo, 0, &ubo_data, sizeof(ubo_data)));
        RAISE_ERROR_OK();
    }

    const global_ubo& last_ubo() const { return m_last_ubo; }

protected:
    uint32_t m_width, m_height;
    device_handler m_device{};
    obj_model m_model{};
    buffer_handler m_ubo{};
    camera m_camera{};
    global_ubo m_last_ubo{};
};

} // namespace rnd_host
