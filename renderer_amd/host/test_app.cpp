// test_app.cpp -- the reference's obj viewer sample (src/samples/test_app: main.cpp,
// dummy_obj_viewer_app.{hpp,cpp}) driven through the HIP library instead of Vulkan, headless:
// load an OBJ model and its diffuse texture, one pipeline per sub-geometry with the sample's state
// (dummy shaders, back-face culling, LESS_OR_EQUAL, 4 samples: dummy_obj_viewer_app.cpp:193-393), a
// multisampled target resolved at the end of the pass (:461-639), one indexed draw per sub-geometry
// (record_obj_model_dummy_draw_commands, :396-458), clear colour (0,0,0,1).
//
//   test_app_headless --obj model.obj [--ktx diffuse.ktx2] [--width W] [--height H] [--samples 1|4]
//                     [--dump PREFIX] [--no-gpu]
//
// --dump writes PREFIX.vb / .ib (the buffers as uploaded), PREFIX.json (sub-geometries, model
// transform, texture levels, uniform block) and, with a GPU, PREFIX.color / .depth (the frame).
// --no-gpu stops after the loaders: it needs no device.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "base_obj_viewer_app.hpp"
#include "ktx2.hpp"
#include "obj_loader.hpp"

using namespace rnd_host;

namespace {

void write_file(const std::string& path, const void* data, size_t bytes)
{
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) return;
    std::fwrite(data, 1, bytes, fp);
    std::fclose(fp);
}

std::string floats_json(const float* f, size_t n)
{
    std::string s = "[";
    char buf[40];
    for (size_t i = 0; i < n; ++i) {
        uint32_t bits;
        std::memcpy(&bits, f + i, 4);
        std::snprintf(buf, sizeof(buf), "%s%u", i ? "," : "", bits); // bit patterns: exact
        s += buf;
    }
    return s + "]";
}

class dummy_obj_viewer_app : public base_obj_viewer_app {
public:
    dummy_obj_viewer_app(uint32_t w, uint32_t h, uint32_t samples) : base_obj_viewer_app(w, h), m_samples(samples) {}

    ERROR_TYPE load(const std::string& obj_path, const std::string& ktx_path)
    {
        PASS_ERROR(load_obj_file(obj_path, m_loaded));
        PASS_ERROR(build_model_buffers(m_loaded.shapes, m_buffers));
        if (!ktx_path.empty()) {
            PASS_ERROR(read_file(ktx_path, m_ktx_bytes));
            PASS_ERROR(parse_ktx2(m_ktx_bytes.data(), m_ktx_bytes.size(), m_ktx));
        }
        RAISE_ERROR_OK();
    }

    ERROR_TYPE on_vulkan_initialized(int hip_ordinal) override
    {
        PASS_ERROR(base_obj_viewer_app::on_vulkan_initialized(hip_ordinal));
        PASS_ERROR(upload_model(m_device, m_loaded.shapes, m_model));
        const rnd_sampler_info sampler{RND_FILTER_LINEAR, 0, 0.0f}; // sampler_info defaults, tools.hpp:9-14
        if (!m_ktx_bytes.empty()) {
            m_model.textures.emplace_back();
            PASS_ERROR(create_ktx_texture(m_device, m_ktx_bytes.data(), m_ktx_bytes.size(), sampler, m_model.textures.back()));
        }
        RND_CHECK(rnd_target_create(m_device, m_width, m_height, RND_FORMAT_R8G8B8A8_SRGB, RND_FORMAT_D24_UNORM_S8_UINT, m_samples, m_target.put()));
        PASS_ERROR(record_obj_model_dummy_draw_commands());
        RND_CHECK(rnd_fence_create(m_device, m_fence.put()));
        RAISE_ERROR_OK();
    }

    // init_geom_pipelines (dummy_obj_viewer_app.cpp:193-393) + the draw list (:396-458)
    ERROR_TYPE record_obj_model_dummy_draw_commands()
    {
        m_pipelines.clear();
        for (const obj_sub_geometry& g : m_model.sub_geometries) {
            rnd_pipeline_info info{};
            info.shader = RND_SHADER_DUMMY;
            uint32_t off = 0;
            info.position = {g.format[0], off}; off += 12;
            if (g.format.size() > 1 && g.format[1] == RND_FORMAT_R32G32B32_SFLOAT) { info.normal = {g.format[1], off}; off += 12; }
            if (g.format.back() == RND_FORMAT_R32G32_SFLOAT) { info.uv = {g.format.back(), off}; off += 8; }
            info.stride = vertex_stride(g);
            info.topology = RND_TOPOLOGY_TRIANGLE_LIST;
            info.cull_mode = RND_CULL_BACK;
            info.front_face = RND_FRONT_FACE_COUNTER_CLOCKWISE;
            info.depth_test = 1; info.depth_write = 1; info.depth_compare = RND_COMPARE_LESS_OR_EQUAL;
            info.samples = m_samples;
            info.color_write_mask = 0xF;
            m_pipelines.emplace_back();
            RND_CHECK(rnd_pipeline_create(m_device, &info, m_pipelines.back().put()));
        }
        RND_CHECK(rnd_cmd_create(m_device, m_command_buffer.put()));
        const float clear_color[4] = {0.0f, 0.0f, 0.0f, 1.0f};
        rnd_cmd cmd = m_command_buffer;
        RND_CHECK(rnd_cmd_begin(cmd));
        RND_CHECK(rnd_cmd_begin_pass(cmd, m_target, clear_color, 1.0f, 0, 0, m_width, m_height));
        RND_CHECK(rnd_cmd_set_viewport(cmd, 0, 0, (float)m_width, (float)m_height, 0, 1));
        RND_CHECK(rnd_cmd_set_scissor(cmd, 0, 0, m_width, m_height));
        for (size_t j = 0; j < m_model.sub_geometries.size(); j++) {
            const obj_sub_geometry& sg = m_model.sub_geometries[j];
            RND_CHECK(rnd_cmd_bind_pipeline(cmd, m_pipelines[j]));
            RND_CHECK(rnd_cmd_bind_vertex_buffer(cmd, m_model.vertex_buffer, sg.vertices_offset));
            RND_CHECK(rnd_cmd_bind_index_buffer(cmd, m_model.index_buffer, sg.indices_offset * sizeof(uint32_t), RND_INDEX_TYPE_UINT32));
            RND_CHECK(rnd_cmd_bind_uniform(cmd, 0, 0, m_ubo, 0, sizeof(global_ubo)));
            RND_CHECK(rnd_cmd_bind_texture(cmd, 1, 0, m_model.textures.empty() ? nullptr : m_model.textures[0].handle()));
            RND_CHECK(rnd_cmd_draw_indexed(cmd, sg.indices_size, 1, 0, 0, 0));
        }
        RND_CHECK(rnd_cmd_end_pass(cmd));
        RND_CHECK(rnd_cmd_end(cmd));
        RAISE_ERROR_OK();
    }

    ERROR_TYPE draw_frame() override
    {
        RND_CHECK(rnd_fence_wait(m_fence));
        PASS_ERROR(base_obj_viewer_app::draw_frame());
        RND_CHECK(rnd_submit(m_device, m_command_buffer, m_fence));
        RAISE_ERROR_OK();
    }

    ERROR_TYPE read_back(std::vector<uint8_t>& color, std::vector<uint32_t>& depth)
    {
        color.resize((size_t)m_width * m_height * 4);
        depth.resize((size_t)m_width * m_height * m_samples);
        RND_CHECK(rnd_fence_wait(m_fence));
        RND_CHECK(rnd_target_read(m_target, color.data(), depth.data()));
        RAISE_ERROR_OK();
    }

    // what the loaders produced, for the tests
    void dump(const std::string& prefix, bool with_ubo) const
    {
        write_file(prefix + ".vb", m_buffers.vertices.data(), m_buffers.vertices.size() * sizeof(float));
        write_file(prefix + ".ib", m_buffers.indices.data(), m_buffers.indices.size() * sizeof(uint32_t));
        std::string j = "{\n \"sub_geometries\": [";
        for (size_t i = 0; i < m_buffers.sub_geometries.size(); ++i) {
            const obj_sub_geometry& g = m_buffers.sub_geometries[i];
            j += std::string(i ? "," : "") + "\n  {\"name\": \"" + m_loaded.shape_names[i] + "\", \"format\": [";
            for (size_t k = 0; k < g.format.size(); ++k) j += (k ? "," : "") + std::to_string(g.format[k]);
            j += "], \"indices_offset\": " + std::to_string(g.indices_offset) + ", \"indices_size\": " + std::to_string(g.indices_size) +
                 ", \"indices_bias\": " + std::to_string(g.indices_bias) + ", \"vertices_offset\": " + std::to_string(g.vertices_offset) +
                 ", \"diffuse\": \"" + m_loaded.diffuse_paths[i] + "\"}";
        }
        j += "\n ],\n \"model_transform_bits\": " + floats_json(m_buffers.model_transform.data(), 16);
        if (with_ubo) j += ",\n \"ubo_bits\": " + floats_json(reinterpret_cast<const float*>(&last_ubo()), 80);
        if (!m_ktx_bytes.empty()) {
            j += ",\n \"ktx\": {\"vk_format\": " + std::to_string(m_ktx.header.vk_format) + ", \"width\": " + std::to_string(m_ktx.header.pixel_width) +
                 ", \"height\": " + std::to_string(m_ktx.header.pixel_height) + ", \"stored_levels\": " + std::to_string(m_ktx.levels.size()) +
                 ", \"texture_levels\": " + std::to_string(m_ktx.texture_levels) + ", \"gen_mips\": " + (m_ktx.gen_mips ? "true" : "false") + ", \"level_offsets\": [";
            for (size_t l = 0; l < m_ktx.levels.size(); ++l) j += (l ? "," : "") + std::to_string(m_ktx.levels[l].byte_offset);
            j += "]}";
        }
        j += "\n}\n";
        write_file(prefix + ".json", j.data(), j.size());
    }

private:
    uint32_t m_samples;
    loaded_obj m_loaded{};
    model_buffers m_buffers{};
    std::vector<uint8_t> m_ktx_bytes{};
    ktx2_info m_ktx{};
    target_handler m_target{};
    std::vector<pipeline_handler> m_pipelines{};
    cmd_buffer_handler m_command_buffer{};
    fence_handler m_fence{};
};

ERROR_TYPE run(int argc, const char** argv)
{
    uint32_t w = 800, h = 600, samples = 4;
    std::string obj, ktx, prefix;
    bool no_gpu = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--obj" && i + 1 < argc) obj = argv[++i];
        else if (a == "--ktx" && i + 1 < argc) ktx = argv[++i];
        else if (a == "--width" && i + 1 < argc) w = (uint32_t)std::atoi(argv[++i]);
        else if (a == "--height" && i + 1 < argc) h = (uint32_t)std::atoi(argv[++i]);
        else if (a == "--samples" && i + 1 < argc) samples = (uint32_t)std::atoi(argv[++i]);
        else if (a == "--dump" && i + 1 < argc) prefix = argv[++i];
        else if (a == "--no-gpu") no_gpu = true;
        else RAISE_ERROR_FATAL(-1, "invalid argument " + a);
    }
    if (obj.empty()) RAISE_ERROR_FATAL(-1, "--obj is required");
    dummy_obj_viewer_app app(w, h, samples);
    PASS_ERROR(app.load(obj, ktx));
    if (no_gpu) {
        if (!prefix.empty()) app.dump(prefix, false);
        RAISE_ERROR_OK();
    }
    PASS_ERROR(app.on_vulkan_initialized(0));
    PASS_ERROR(app.draw_frame());
    std::vector<uint8_t> color;
    std::vector<uint32_t> depth;
    PASS_ERROR(app.read_back(color, depth));
    size_t covered = 0;
    for (uint32_t d : depth) covered += d != 0xFFFFFFu;
    std::printf("extent=%ux%u samples=%u covered_samples=%zu\n", w, h, samples, covered);
    if (!prefix.empty()) {
        app.dump(prefix, true);
        write_file(prefix + ".color", color.data(), color.size());
        write_file(prefix + ".depth", depth.data(), depth.size() * 4);
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
