// render_framework_app.cpp -- the test_ktx scene (ten textured, lit spheres on a ring,
// src/samples/test_ktx_app/test_ktx_app.cpp:650-741) drawn through the reference's render_framework
// seam with the HIP backend: the mesh from a vertex_format and two data blobs, the uniform block
// and the push-constant block as parameters lists whose parameters are loaded one by one, the
// texture from a texture builder, the shader pair and the bindings from a material.
//
//   render_framework_headless [--width W] [--height H] [--layout]
//
// --layout prints what the builders derive (attribute offsets, stride, parameter slots) without a
// device.  Otherwise one frame is rendered and its checksums printed in the format of
// test_ktx_app_headless, whose frame it must reproduce bit for bit.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "base_obj_viewer_app.hpp"
#include "procedural.hpp"
#include "render_framework.hpp"

using namespace rnd_host;
namespace rf = render_framework;

namespace {

uint64_t fnv1a(const void* data, size_t n, uint64_t h = 1469598103934665603ull)
{
    const uint8_t* p = static_cast<const uint8_t*>(data);
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

rf::vertex_format obj_vertex_format()
{
    rf::vertex_format f;
    f.add_attribute({rf::vertex_format::attribute_type::float32, 3}); // position
    f.add_attribute({rf::vertex_format::attribute_type::float32, 3}); // normal
    f.add_attribute({rf::vertex_format::attribute_type::float32, 2}); // uv
    return f;
}

utils::data name_blob(const char* name) { return utils::data::copy_of(name, std::strlen(name)); }

ERROR_TYPE make_parameters(rf::parameter::value_type type, size_t n, std::vector<rf::parameter>& out)
{
    out.resize(n);
    for (rf::parameter& p : out) PASS_ERROR(rf::parameter_builder().set_value_type(type).create(p));
    RAISE_ERROR_OK();
}

ERROR_TYPE print_layout()
{
    // builders that need no device for what is printed
    rf::hip_mesh_builder mb(nullptr);
    mb.set_format(obj_vertex_format());
    rf::detail::hip_vertex_layout layout;
    PASS_ERROR(mb.create_vertex_inputs(layout));
    std::printf("stride=%u position=%u normal=%u uv=%u\n", layout.stride, layout.position.offset, layout.normal.offset, layout.uv.offset);
    // parameter slots: int32, float32, vec4, mat4 in one list
    std::vector<rf::parameter> ps(4);
    const rf::parameter::value_type types[4] = {rf::parameter::value_type::int32, rf::parameter::value_type::float32,
                                                rf::parameter::value_type::vec4, rf::parameter::value_type::mat4};
    for (int i = 0; i < 4; ++i) PASS_ERROR(rf::parameter_builder().set_value_type(types[i]).create(ps[(size_t)i]));
    rf::hip_push_constant_parameters_list_builder lb;
    for (rf::parameter& p : ps) lb.add_parameter(p);
    rf::parameters_list list;
    PASS_ERROR(lb.create(list));
    const auto* impl = static_cast<const rf::detail::hip_parameters_list_impl*>(list.get_impl());
    std::printf("list_size=%zu slots=", impl->get_buffer_size());
    for (const auto& d : impl->get_parameters_data()) std::printf("%zu:%zu ", d.data_offset, d.data_size);
    std::printf("\n");
    // loading a value lands in its slot
    const float v4[4] = {1.0f, 2.0f, 3.0f, 4.0f};
    ps[2].load_value(reinterpret_cast<const uint8_t*>(v4));
    float back[4];
    std::memcpy(back, impl->get_data() + impl->get_parameters_data()[2].data_offset, 16);
    std::printf("vec4_slot=%g,%g,%g,%g\n", back[0], back[1], back[2], back[3]);
    // error paths of the builders
    rf::parameter bad;
    struct probe { static ERROR_TYPE run(rf::parameter& p) { PASS_ERROR(rf::parameter_builder().set_elements_count(0).create(p)); RAISE_ERROR_OK(); } };
#ifndef NO_EXCEPTIONS
    try { probe::run(bad); } catch (const errors::error& e) { std::printf("error=%s\n", e.what()); }
#else
    const errors::error e = probe::run(bad);
    std::printf("error=%s\n", e.message.c_str());
#endif
    RAISE_ERROR_OK();
}

ERROR_TYPE run(int argc, const char** argv)
{
    uint32_t w = 1280, h = 720;
    bool layout_only = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--width" && i + 1 < argc) w = (uint32_t)std::atoi(argv[++i]);
        else if (a == "--height" && i + 1 < argc) h = (uint32_t)std::atoi(argv[++i]);
        else if (a == "--layout") layout_only = true;
        else RAISE_ERROR_FATAL(-1, "invalid argument " + a);
    }
    if (layout_only) { PASS_ERROR(print_layout()); RAISE_ERROR_OK(); }

    device_handler device;
    RND_CHECK(rnd_device_create(0, nullptr, device.put()));

    // mesh: the sphere of the test_ktx sample as two blobs
    const shape_data sphere = make_uv_sphere(64, 32, 4.0f);
    std::vector<float> interleaved;
    for (size_t v = 0; v < sphere.positions.size() / 3; ++v) {
        interleaved.insert(interleaved.end(), &sphere.positions[v * 3], &sphere.positions[v * 3] + 3);
        interleaved.insert(interleaved.end(), &sphere.normals[v * 3], &sphere.normals[v * 3] + 3);
        interleaved.insert(interleaved.end(), &sphere.uvs[v * 2], &sphere.uvs[v * 2] + 2);
    }
    rf::mesh mesh;
    rf::hip_mesh_builder mesh_builder(device);
    mesh_builder.set_format(obj_vertex_format())
        .set_index_format(rf::mesh_builder::index_type::int32)
        .set_vertex_data(utils::data::copy_of(interleaved.data(), interleaved.size() * sizeof(float)))
        .set_index_data(utils::data::copy_of(sphere.indices.data(), sphere.indices.size() * sizeof(uint32_t)));
    PASS_ERROR(mesh_builder.create(mesh));

    // texture
    const std::vector<uint8_t> texels = make_checker(1024);
    rf::texture texture;
    rf::hip_raw_texture_builder texture_builder(device);
    texture_builder.set_texels(1024, 1024, RND_FORMAT_R8G8B8A8_SRGB, true, utils::data::copy_of(texels.data(), texels.size()));
    texture_builder.set_filtering(rf::texture_builder::texture_filtering::linear).set_address_mode(rf::texture_builder::texture_address_mode::clamp_to_egde);
    PASS_ERROR(texture_builder.create(texture));

    // the uniform block: projection, view_projection, view, model, mvp (base_obj_viewer_app.hpp:36-43)
    std::vector<rf::parameter> ubo_params;
    PASS_ERROR(make_parameters(rf::parameter::value_type::mat4, 5, ubo_params));
    rf::parameters_list ubo_list;
    rf::hip_uniform_parameters_list_builder ubo_builder(device);
    for (rf::parameter& p : ubo_params) ubo_builder.add_parameter(p);
    PASS_ERROR(ubo_builder.create(ubo_list));

    // the push-constant block: vec4 colour, mat4 transform (test_ktx_app.hpp:16-19)
    std::vector<rf::parameter> push_params(2);
    PASS_ERROR(rf::parameter_builder().set_value_type(rf::parameter::value_type::vec4).create(push_params[0]));
    PASS_ERROR(rf::parameter_builder().set_value_type(rf::parameter::value_type::mat4).create(push_params[1]));
    rf::parameters_list push_list;
    rf::hip_push_constant_parameters_list_builder push_builder;
    push_builder.add_parameter(push_params[0]).add_parameter(push_params[1]);
    PASS_ERROR(push_builder.create(push_list));

    rf::material material;
    rf::hip_material_builder material_builder;
    material_builder.add_stage(rf::material_builder::shading_stage_type::vertex, name_blob("test_ktx"))
        .add_stage(rf::material_builder::shading_stage_type::fragment, name_blob("test_ktx"))
        .add_texture(texture)
        .add_parameters_list(ubo_list)
        .add_parameters_list(push_list);
    PASS_ERROR(material_builder.create(material));
    auto* material_impl = static_cast<rf::detail::hip_material_impl*>(material.get_impl());

    pipeline_handler pipeline;
    PASS_ERROR(rf::create_pipeline(device, mesh, material, 1, pipeline));
    target_handler target;
    RND_CHECK(rnd_target_create(device, w, h, RND_FORMAT_R8G8B8A8_SRGB, RND_FORMAT_D24_UNORM_S8_UINT, 1, target.put()));

    // per-frame values, loaded parameter by parameter
    camera cam;
    cam.eye_position = {0, 0, -2};
    PASS_ERROR(cam.update((float)w, (float)h));
    const glm_lite::mat4 model = glm_lite::scale(glm_lite::mat4(1.0f), {0.12f, 0.12f, 0.12f});
    const glm_lite::mat4 mats[5] = {cam.proj_matrix, cam.view_proj_matrix, cam.view_matrix, model, cam.proj_matrix * cam.view_matrix * model};
    for (int i = 0; i < 5; ++i) ubo_params[(size_t)i].load_value(reinterpret_cast<const uint8_t*>(mats[i].data()));

    // the draw list of record_command_buffers
    cmd_buffer_handler cmd;
    RND_CHECK(rnd_cmd_create(device, cmd.put()));
    const float clear_color[4] = {0.2f, 0.3f, 0.7f, 1.0f};
    static const glm_lite::vec4 colors[] = {{0, 0, 0, 1}, {1, 0, 0, 1}, {0, 1, 0, 1}, {0, 0, 1, 1}, {1, 1, 0, 1},
                                            {0, 1, 1, 1}, {1, 0, 1, 1}, {1, 1, 1, 1}, {1, 0.5f, 0, 1}, {0, 0.5f, 1, 1}};
    RND_CHECK(rnd_cmd_begin(cmd));
    RND_CHECK(rnd_cmd_begin_pass(cmd, target, clear_color, 1.0f, 0, 0, w, h));
    RND_CHECK(rnd_cmd_set_viewport(cmd, 0, 0, (float)w, (float)h, 0, 1));
    RND_CHECK(rnd_cmd_set_scissor(cmd, 0, 0, w, h));
    RND_CHECK(rnd_cmd_bind_pipeline(cmd, pipeline));
    float angle = 0;
    for (size_t j = 0; j < 10; ++j) {
        glm_lite::mat4 transform = glm_lite::rotate(glm_lite::mat4(1.0f), glm_lite::radians(angle), {0.0f, 0.0f, 1.0f});
        transform = glm_lite::translate(transform, {0, 10, 0});
        angle += 36.0f;
        push_params[0].load_value(reinterpret_cast<const uint8_t*>(&colors[j]));
        push_params[1].load_value(reinterpret_cast<const uint8_t*>(transform.data()));
        RND_CHECK(material_impl->apply(cmd));
        PASS_ERROR(rf::draw_mesh(cmd, mesh));
    }
    RND_CHECK(rnd_cmd_end_pass(cmd));
    RND_CHECK(rnd_cmd_end(cmd));

    fence_handler fence;
    RND_CHECK(rnd_fence_create(device, fence.put()));
    RND_CHECK(rnd_submit(device, cmd, fence));
    RND_CHECK(rnd_fence_wait(fence));
    std::vector<uint8_t> color((size_t)w * h * 4);
    std::vector<uint32_t> depth((size_t)w * h);
    RND_CHECK(rnd_target_read(target, color.data(), depth.data()));
    size_t covered = 0;
    for (uint32_t d : depth) covered += d != 0xFFFFFFu;
    std::printf("frames=1 extent=%ux%u covered=%zu color_fnv=%016llx depth_fnv=%016llx\n", w, h, covered,
                (unsigned long long)fnv1a(color.data(), color.size()), (unsigned long long)fnv1a(depth.data(), depth.size() * 4));
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
