// test_ktx_app.cpp -- the reference's test_ktx sample (src/samples/test_ktx_app/test_ktx_app.cpp)
// driven through the HIP library instead of Vulkan, headless.  The sample's steps keep their
// names: on_vulkan_initialized (:46-56), init_pipelines (:412-625, state only), record_command_buffers
// (:627-746), draw_frame (:66-73).  A procedural UV sphere and checker texture stand in for the
// reference's OBJ / KTX assets (Git-LFS stubs).  Output: a PPM of the last frame and an FNV-1a
// checksum of colour and depth on stdout.
//
//   test_ktx_app_headless [--width W] [--height H] [--frames N] [--out frame.ppm]
//                         [--shader ktx|push_constants] [--dump-ubo]
//
// Built with -DSAMPLE_PUSH_CONSTANTS the same source is the reference's test_push_constants sample
// (src/samples/test_push_constants_app/test_push_constants_app.cpp:562-681): the same ring of ten
// draws with push_constants.{vert,frag}, whose descriptor set holds the uniform block and no texture.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "base_obj_viewer_app.hpp"
#include "procedural.hpp"

using namespace rnd_host;

namespace {

#ifdef SAMPLE_PUSH_CONSTANTS
constexpr int DEFAULT_SHADER = RND_SHADER_PUSH_CONSTANTS;
#else
constexpr int DEFAULT_SHADER = RND_SHADER_TEST_KTX;
#endif

struct push_constant_data { // test_ktx_app.hpp:16-19
    glm_lite::vec4 color;
    glm_lite::mat4 transform;
};
static_assert(sizeof(push_constant_data) == 80, "vec4 + mat4");

class test_ktx_app : public base_obj_viewer_app {
public:
    test_ktx_app(uint32_t w, uint32_t h, int shader) : base_obj_viewer_app(w, h), m_shader(shader) {}

    ERROR_TYPE on_vulkan_initialized(int hip_ordinal) override
    {
        PASS_ERROR(base_obj_viewer_app::on_vulkan_initialized(hip_ordinal));
        PASS_ERROR(upload_model(m_device, {make_uv_sphere(64, 32, 4.0f)}, m_model));
        const std::vector<uint8_t> texels = make_checker(1024);
        const rnd_sampler_info sampler{RND_FILTER_LINEAR, 0, 0.0f};
        m_model.textures.emplace_back();
        RND_CHECK(rnd_texture_create_2d(m_device, 1024, 1024, RND_FORMAT_R8G8B8A8_SRGB, 1, texels.data(), &sampler, m_model.textures.back().put()));
        // swapchain image + D24S8 (init_framebuffers, test_ktx_app.cpp:164-258)
        RND_CHECK(rnd_target_create(m_device, m_width, m_height, RND_FORMAT_R8G8B8A8_SRGB, RND_FORMAT_D24_UNORM_S8_UINT, 1, m_target.put()));
        PASS_ERROR(init_pipelines());
        PASS_ERROR(record_command_buffers());
        RND_CHECK(rnd_fence_create(m_device, m_fence.put()));
        RAISE_ERROR_OK();
    }

    // one pipeline per sub-geometry, the fixed-function state of test_ktx_app.cpp:435-499
    ERROR_TYPE init_pipelines()
    {
        m_graphics_pipelines.clear();
        for (const obj_sub_geometry& g : m_model.sub_geometries) {
            rnd_pipeline_info info{};
            info.shader = m_shader;
            uint32_t off = 0;
            info.position = {g.format[0], off}; off += 12;
            if (g.format.size() > 1 && g.format[1] == RND_FORMAT_R32G32B32_SFLOAT) { info.normal = {g.format[1], off}; off += 12; }
            if (g.format.back() == RND_FORMAT_R32G32_SFLOAT) { info.uv = {g.format.back(), off}; off += 8; }
            info.stride = vertex_stride(g);
            info.topology = RND_TOPOLOGY_TRIANGLE_LIST;
            info.cull_mode = RND_CULL_BACK;
            info.front_face = RND_FRONT_FACE_COUNTER_CLOCKWISE;
            info.depth_test = 1; info.depth_write = 1; info.depth_compare = RND_COMPARE_LESS_OR_EQUAL;
            info.samples = 1;
            info.color_write_mask = 0xF;
            info.blend_enable = 0;
            m_graphics_pipelines.emplace_back();
            RND_CHECK(rnd_pipeline_create(m_device, &info, m_graphics_pipelines.back().put()));
        }
        RAISE_ERROR_OK();
    }

    // the static draw list: clear, viewport, scissor, ten ring instances (test_ktx_app.cpp:650-741)
    ERROR_TYPE record_command_buffers()
    {
        RND_CHECK(rnd_cmd_create(m_device, m_command_buffer.put()));
        const float clear_color[4] = {0.2f, 0.3f, 0.7f, 1.0f};
        static const glm_lite::vec4 colors[] = {{0, 0, 0, 1}, {1, 0, 0, 1}, {0, 1, 0, 1}, {0, 0, 1, 1}, {1, 1, 0, 1},
                                                {0, 1, 1, 1}, {1, 0, 1, 1}, {1, 1, 1, 1}, {1, 0.5f, 0, 1}, {0, 0.5f, 1, 1}};
        const size_t n_colors = sizeof(colors) / sizeof(colors[0]);
        const float rot_angle = 360.0f / n_colors;
        rnd_cmd cmd = m_command_buffer;
        RND_CHECK(rnd_cmd_begin(cmd));
        RND_CHECK(rnd_cmd_begin_pass(cmd, m_target, clear_color, 1.0f, 0, 0, m_width, m_height));
        RND_CHECK(rnd_cmd_set_viewport(cmd, 0, 0, (float)m_width, (float)m_height, 0, 1));
        RND_CHECK(rnd_cmd_set_scissor(cmd, 0, 0, m_width, m_height));
        rnd_pipeline last_pipeline = nullptr;
        float curr_angle = 0;
        for (size_t j = 0; j < n_colors; j++) {
            push_constant_data data{colors[j], glm_lite::mat4(1.0f)};
            data.transform = glm_lite::rotate(data.transform, glm_lite::radians(curr_angle), {0.0f, 0.0f, 1.0f});
            data.transform = glm_lite::translate(data.transform, {0, 10, 0});
            curr_angle += rot_angle;
            RND_CHECK(rnd_cmd_push_constants(cmd, 0, sizeof(push_constant_data), &data));
            for (size_t g = 0; g < m_model.sub_geometries.size(); g++) {
                if (m_graphics_pipelines[g].handle() != last_pipeline) {
                    RND_CHECK(rnd_cmd_bind_pipeline(cmd, m_graphics_pipelines[g]));
                    RND_CHECK(rnd_cmd_bind_uniform(cmd, 0, 0, m_ubo, 0, sizeof(global_ubo)));
                    if (m_shader == RND_SHADER_TEST_KTX) RND_CHECK(rnd_cmd_bind_texture(cmd, 0, 1, m_model.textures[0]));
                    last_pipeline = m_graphics_pipelines[g];
                }
                const obj_sub_geometry& sg = m_model.sub_geometries[g];
                RND_CHECK(rnd_cmd_bind_vertex_buffer(cmd, m_model.vertex_buffer, sg.vertices_offset));
                RND_CHECK(rnd_cmd_bind_index_buffer(cmd, m_model.index_buffer, sg.indices_offset * sizeof(uint32_t), RND_INDEX_TYPE_UINT32));
                RND_CHECK(rnd_cmd_draw_indexed(cmd, sg.indices_size, 1, 0, 0, 0));
            }
        }
        RND_CHECK(rnd_cmd_end_pass(cmd));
        RND_CHECK(rnd_cmd_end(cmd));
        RAISE_ERROR_OK();
    }

    // update the uniform block, submit the pre-recorded buffer (test_ktx_app.cpp:66-73, vk_app.cpp:433-467)
    ERROR_TYPE draw_frame() override
    {
        RND_CHECK(rnd_fence_wait(m_fence)); // frames in flight = 1
        PASS_ERROR(base_obj_viewer_app::draw_frame());
        RND_CHECK(rnd_submit(m_device, m_command_buffer, m_fence));
        RAISE_ERROR_OK();
    }

    ERROR_TYPE read_back(std::vector<uint8_t>& color, std::vector<uint32_t>& depth)
    {
        color.resize((size_t)m_width * m_height * 4);
        depth.resize((size_t)m_width * m_height);
        RND_CHECK(rnd_fence_wait(m_fence));
        RND_CHECK(rnd_target_read(m_target, color.data(), depth.data()));
        RAISE_ERROR_OK();
    }

    void set_model_scale(float s) { m_model.model_transform = glm_lite::scale(glm_lite::mat4(1.0f), {s, s, s}); }

private:
    int m_shader;
    target_handler m_target{};
    std::vector<pipeline_handler> m_graphics_pipelines{};
    cmd_buffer_handler m_command_buffer{};
    fence_handler m_fence{};
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
    int frames = 3, shader = DEFAULT_SHADER;
    bool dump_ubo = false;
    std::string out;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--width" && i + 1 < argc) w = (uint32_t)std::atoi(argv[++i]);
        else if (a == "--height" && i + 1 < argc) h = (uint32_t)std::atoi(argv[++i]);
        else if (a == "--frames" && i + 1 < argc) frames = std::atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--shader" && i + 1 < argc) shader = std::string(argv[++i]) == "push_constants" ? RND_SHADER_PUSH_CONSTANTS : RND_SHADER_TEST_KTX;
        else if (a == "--dump-ubo") dump_ubo = true;
        else RAISE_ERROR_FATAL(-1, "invalid argument " + a);
    }
    if (dump_ubo) {
        // host maths only, no device: the uniform block of the first frame as float bit patterns
        camera cam;
        cam.eye_position = {0, 0, -2};
        PASS_ERROR(cam.update((float)w, (float)h));
        const glm_lite::mat4 model = glm_lite::scale(glm_lite::mat4(1.0f), {0.12f, 0.12f, 0.12f});
        const glm_lite::mat4 mats[5] = {cam.proj_matrix, cam.view_proj_matrix, cam.view_matrix, model, cam.proj_matrix * cam.view_matrix * model};
        for (const auto& m : mats) for (int k = 0; k < 16; ++k) { uint32_t b; std::memcpy(&b, m.data() + k, 4); std::printf("%08x\n", b); }
        float ang = 0;
        for (int j = 0; j < 10; ++j) {
            glm_lite::mat4 t = glm_lite::rotate(glm_lite::mat4(1.0f), glm_lite::radians(ang), {0, 0, 1});
            t = glm_lite::translate(t, {0, 10, 0});
            ang += 36.0f;
            for (int k = 0; k < 16; ++k) { uint32_t b; std::memcpy(&b, t.data() + k, 4); std::printf("%08x\n", b); }
        }
        RAISE_ERROR_OK();
    }
    test_ktx_app app(w, h, shader);
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
#ifndef NO_EXCEPTIONS
    try {
        run(argc, argv);
    } catch (const errors::error& e) {
        std::fprintf(stderr, "[ERROR] %s\n", e.what());
        return 1;
    }
    return 0;
#else
    const errors::error e = run(argc, argv);
    errors::handle_error(e);
    return e.err_state == errors::OK ? 0 : 1;
#endif
}
