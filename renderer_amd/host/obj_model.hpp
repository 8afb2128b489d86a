// obj_model.hpp -- the reference's mesh container (vk_utils::obj_loader::obj_model and
// obj_sub_geometry, src/libs/vk_utils/vk_utils/obj_loader.hpp:61-89) over the HIP library's
// buffers, and the upload step of init_obj_geometry (obj_loader.cpp:93-310): one interleaved
// vertex buffer for all shapes (pos3 [normal3] [uv2]), one uint32 index buffer with indices
// local to each sub-geometry, and the normalising model transform.  Shapes arrive as arrays, from
// obj_loader.hpp or from a procedural mesh (the reference's own assets are Git-LFS stubs).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "errors.hpp"
#include "glm_lite.hpp"
#include "handles.hpp"

namespace rnd_host {

struct obj_sub_geometry {
    std::vector<int32_t> format{};   // RND_FORMAT_* per attribute, in buffer order
    uint32_t indices_offset{0};      // in indices
    uint32_t indices_size{0};
    uint32_t indices_bias{0};
    uint32_t vertices_offset{0};     // in BYTES (obj_loader.cpp:173-205)
    int32_t diffuse_texture{-1};     // index into obj_model::textures
};

struct obj_model {
    buffer_handler vertex_buffer{};
    buffer_handler index_buffer{};
    std::vector<texture_handler> textures{};
    std::vector<obj_sub_geometry> sub_geometries{};
    glm_lite::mat4 model_transform{1.0f};
};

// one shape as a loader hands it over: de-duplicated vertices, local indices
struct shape_data {
    std::vector<float> positions; // 3 per vertex
    std::vector<float> normals;   // 3 per vertex or empty
    std::vector<float> uvs;       // 2 per vertex or empty
    std::vector<uint32_t> indices;
};

// The host half of init_obj_geometry's tail (obj_loader.cpp:144-205, 294-307): the interleaved
// vertex data, the index data, the sub-geometry table, and model_transform = translate(offset) *
// scale(1 / largest half extent), with fabs where the reference's unqualified abs() may truncate
// and a bounding box that starts empty (the reference's starts at numeric_limits<float>::min()).
struct model_buffers {
    std::vector<float> vertices;
    std::vector<uint32_t> indices;
    std::vector<obj_sub_geometry> sub_geometries;
    glm_lite::mat4 model_transform{1.0f};
};

inline ERROR_TYPE build_model_buffers(const std::vector<shape_data>& shapes, model_buffers& out)
{
    std::vector<float>& vb = out.vertices;
    std::vector<uint32_t>& ib = out.indices;
    vb.clear(); ib.clear(); out.sub_geometries.clear();
    glm_lite::vec3 min_pos{1e30f, 1e30f, 1e30f}, max_pos{-1e30f, -1e30f, -1e30f};
    uint32_t start_vertex = 0;
    for (const shape_data& s : shapes) {
        obj_sub_geometry g;
        g.format.push_back(RND_FORMAT_R32G32B32_SFLOAT);
        if (!s.normals.empty()) g.format.push_back(RND_FORMAT_R32G32B32_SFLOAT);
        if (!s.uvs.empty()) g.format.push_back(RND_FORMAT_R32G32_SFLOAT);
        g.vertices_offset = (uint32_t)(vb.size() * sizeof(float));
        g.indices_offset = (uint32_t)ib.size();
        g.indices_size = (uint32_t)s.indices.size();
        g.indices_bias = start_vertex;
        const size_t n = s.positions.size() / 3;
        for (size_t v = 0; v < n; ++v) {
            const float* p = &s.positions[v * 3];
            vb.insert(vb.end(), p, p + 3);
            min_pos = {std::min(min_pos.x, p[0]), std::min(min_pos.y, p[1]), std::min(min_pos.z, p[2])};
            max_pos = {std::max(max_pos.x, p[0]), std::max(max_pos.y, p[1]), std::max(max_pos.z, p[2])};
            if (!s.normals.empty()) vb.insert(vb.end(), &s.normals[v * 3], &s.normals[v * 3] + 3);
            if (!s.uvs.empty()) vb.insert(vb.end(), &s.uvs[v * 2], &s.uvs[v * 2] + 2);
        }
        ib.insert(ib.end(), s.indices.begin(), s.indices.end());
        start_vertex += (uint32_t)n;
        out.sub_geometries.push_back(g);
    }
    if (vb.empty() || ib.empty()) RAISE_ERROR_FATAL(-1, "model holds no geometry");
    glm_lite::vec3 offset = (max_pos - min_pos) * 0.5f;
    const float sx = std::max(std::fabs(min_pos.x - offset.x), std::fabs(max_pos.x - offset.x));
    const float sy = std::max(std::fabs(min_pos.y - offset.y), std::fabs(max_pos.y - offset.y));
    const float sz = std::max(std::fabs(min_pos.z - offset.z), std::fabs(max_pos.z - offset.z));
    float final_scale = sx > sy ? sx : sy;
    final_scale = sz > final_scale ? 1.0f / sz : 1.0f / final_scale;
    const glm_lite::vec3 scale{final_scale, final_scale, final_scale};
    offset = (min_pos + (max_pos - min_pos) * 0.5f) * final_scale;
    out.model_transform = glm_lite::translate(glm_lite::mat4(1.0f), offset);
    out.model_transform = glm_lite::scale(out.model_transform, scale);
    RAISE_ERROR_OK();
}

// ... and the device half: create_buffer for both (obj_loader.cpp:207-292)
inline ERROR_TYPE upload_model(rnd_device dev, const std::vector<shape_data>& shapes, obj_model& model)
{
    model_buffers b;
    PASS_ERROR(build_model_buffers(shapes, b));
    RND_CHECK(rnd_buffer_create(dev, b.vertices.size() * sizeof(float), RND_BUFFER_USAGE_VERTEX, b.vertices.data(), model.vertex_buffer.put()));
    RND_CHECK(rnd_buffer_create(dev, b.indices.size() * sizeof(uint32_t), RND_BUFFER_USAGE_INDEX, b.indices.data(), model.index_buffer.put()));
    model.sub_geometries = b.sub_geometries;
    model.model_transform = b.model_transform;
    RAISE_ERROR_OK();
}

// stride of a sub-geometry's vertex (12 / 20 / 24 / 32 bytes)
inline uint32_t vertex_stride(const obj_sub_geometry& g)
{
    uint32_t s = 0;
    for (int32_t f : g.format) s += f == RND_FORMAT_R32G32_SFLOAT ? 8u : 12u;
    return s;
}

} // namespace rnd_host
