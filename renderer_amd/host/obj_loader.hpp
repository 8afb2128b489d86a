// obj_loader.hpp -- the step before the draw path: a Wavefront OBJ file to the interleaved vertex
// buffer, the index buffer and the sub-geometry table the draw list binds
// (vk_utils::obj_loader::load_model / init_obj_geometry, src/libs/vk_utils/vk_utils/obj_loader.cpp:14-310).
//
// The reference parses with tinyobjloader (an unfetched submodule); parse_obj() restates the subset
// of the format its models use: v / vn / vt / f (v, v/vt, v//vn, v/vt/vn; negative = relative
// indices; polygons fan-triangulated), o / g starting a new shape, usemtl, mtllib, and map_Kd in the
// material library.  build_shapes() is init_obj_geometry's vertex merge (obj_loader.cpp:93-143):
// per shape, every face corner becomes a vertex unless an equal one (position, normal, texture
// coordinate) exists already, and indices are local to the shape.
//
// Where the reference's code and its evident intent part, the intent is followed and named here:
//  * obj_loader.cpp:101,106,126,130 test `static_cast<size_t>(index) >= 0`, which is always true, so
//    a corner without normal or texture coordinate reads attrib arrays out of bounds; here a shape
//    has normals / texture coordinates if its first corner has them (the reference's apparent rule,
//    :96) and a corner that lacks one the shape has is an error;
//  * the merge is a linear search per corner (:134); here a hash map finds the same first match;
//  * the bounding box starts from numeric_limits<float>::min(), a small positive number (:90), and the
//    scale uses unqualified abs() (:296-298); upload_model() starts from the first vertex and uses fabs.
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "errors.hpp"
#include "obj_model.hpp"

namespace rnd_host {

struct obj_attrib {
    std::vector<float> vertices;  // 3 per position
    std::vector<float> normals;   // 3 per normal
    std::vector<float> texcoords; // 2 per coordinate
};

struct obj_index {
    int vertex_index{-1}, normal_index{-1}, texcoord_index{-1};
};

struct obj_shape {
    std::string name;
    std::vector<obj_index> indices;   // three per triangle
    std::vector<int> material_ids;    // one per triangle, -1: none
};

struct obj_material {
    std::string name;
    std::string diffuse_texname; // map_Kd
};

namespace detail {

inline bool parse_corner(const std::string& tok, int nv, int nn, int nt, obj_index& out)
{
    // v, v/vt, v//vn, v/vt/vn
    int field = 0;
    size_t pos = 0;
    int* dst[3] = {&out.vertex_index, &out.texcoord_index, &out.normal_index};
    const int counts[3] = {nv, nt, nn};
    while (field < 3) {
        const size_t slash = tok.find('/', pos);
        const std::string part = tok.substr(pos, slash == std::string::npos ? std::string::npos : slash - pos);
        if (!part.empty()) {
            char* end = nullptr;
            const long v = std::strtol(part.c_str(), &end, 10);
            if (end == part.c_str() || *end != '\0' || v == 0) return false;
            const long idx = v > 0 ? v - 1 : counts[field] + v; // 1-based, or relative to the end
            if (idx < 0 || idx >= counts[field]) return false;
            *dst[field] = (int)idx;
        } else if (field == 0) {
            return false;
        }
        if (slash == std::string::npos) break;
        pos = slash + 1;
        ++field;
    }
    return out.vertex_index >= 0;
}

} // namespace detail

// material library: newmtl + map_Kd (the other maps of obj_loader.cpp:399-406 are not sampled by any shader)
inline ERROR_TYPE parse_mtl(std::istream& in, std::vector<obj_material>& materials)
{
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string key;
        if (!(ls >> key) || key[0] == '#') continue;
        if (key == "newmtl") {
            materials.emplace_back();
            ls >> materials.back().name;
        } else if (key == "map_Kd" && !materials.empty()) {
            std::string name, last;
            while (ls >> name) last = name; // options first, file name last
            materials.back().diffuse_texname = last;
        }
    }
    RAISE_ERROR_OK();
}

inline ERROR_TYPE parse_obj(std::istream& in, obj_attrib& attrib, std::vector<obj_shape>& shapes, std::vector<std::string>& material_names,
                            std::string& mtllib)
{
    obj_shape cur;
    int cur_material = -1;
    auto flush = [&]() {
        if (!cur.indices.empty()) shapes.push_back(cur);
        cur.indices.clear();
        cur.material_ids.clear();
    };
    std::string line;
    size_t line_no = 0;
    while (std::getline(in, line)) {
        ++line_no;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string key;
        if (!(ls >> key) || key[0] == '#') continue;
        if (key == "v") {
            float x, y, z;
            if (!(ls >> x >> y >> z)) RAISE_ERROR_WARN(-1, "obj line " + std::to_string(line_no) + ": bad position");
            attrib.vertices.insert(attrib.vertices.end(), {x, y, z});
        } else if (key == "vn") {
            float x, y, z;
            if (!(ls >> x >> y >> z)) RAISE_ERROR_WARN(-1, "obj line " + std::to_string(line_no) + ": bad normal");
            attrib.normals.insert(attrib.normals.end(), {x, y, z});
        } else if (key == "vt") {
            float u, v = 0.0f;
            if (!(ls >> u)) RAISE_ERROR_WARN(-1, "obj line " + std::to_string(line_no) + ": bad texture coordinate");
            ls >> v;
            attrib.texcoords.insert(attrib.texcoords.end(), {u, v});
        } else if (key == "f") {
            std::vector<obj_index> corners;
            std::string tok;
            while (ls >> tok) {
                obj_index c;
                if (!detail::parse_corner(tok, (int)(attrib.vertices.size() / 3), (int)(attrib.normals.size() / 3), (int)(attrib.texcoords.size() / 2), c))
                    RAISE_ERROR_WARN(-1, "obj line " + std::to_string(line_no) + ": bad face corner '" + tok + "'");
                corners.push_back(c);
            }
            if (corners.size() < 3) RAISE_ERROR_WARN(-1, "obj line " + std::to_string(line_no) + ": face with fewer than three corners");
            for (size_t k = 1; k + 1 < corners.size(); ++k) { // fan
                cur.indices.push_back(corners[0]);
                cur.indices.push_back(corners[k]);
                cur.indices.push_back(corners[k + 1]);
                cur.material_ids.push_back(cur_material);
            }
        } else if (key == "o" || key == "g") {
            flush();
            cur.name.clear();
            ls >> cur.name;
        } else if (key == "usemtl") {
            std::string name;
            ls >> name;
            cur_material = -1;
            for (size_t m = 0; m < material_names.size(); ++m) if (material_names[m] == name) cur_material = (int)m;
            if (cur_material < 0) { material_names.push_back(name); cur_material = (int)material_names.size() - 1; }
        } else if (key == "mtllib") {
            ls >> mtllib;
        }
    }
    flush();
    if (shapes.empty()) RAISE_ERROR_WARN(-1, "obj file holds no faces");
    RAISE_ERROR_OK();
}

// init_obj_geometry's merge (obj_loader.cpp:93-143): shape -> unique vertices + local indices
inline ERROR_TYPE build_shapes(const obj_attrib& attrib, const std::vector<obj_shape>& shapes, std::vector<shape_data>& out)
{
    struct key {
        uint32_t w[8];
        bool operator==(const key& o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; }
    };
    struct key_hash {
        size_t operator()(const key& k) const
        {
            uint64_t h = 1469598103934665603ull;
            for (uint32_t v : k.w) { h ^= v; h *= 1099511628211ull; }
            return (size_t)h;
        }
    };
    out.clear();
    for (const obj_shape& shape : shapes) {
        shape_data g;
        const bool has_normal = shape.indices.front().normal_index >= 0;
        const bool has_uv = shape.indices.front().texcoord_index >= 0;
        std::unordered_map<key, uint32_t, key_hash> seen;
        for (const obj_index& i : shape.indices) {
            if ((has_normal && i.normal_index < 0) || (has_uv && i.texcoord_index < 0))
                RAISE_ERROR_WARN(-1, "shape '" + shape.name + "': a face corner lacks an attribute the shape's first corner has");
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            std::memcpy(v, &attrib.vertices[3 * (size_t)i.vertex_index], 12);
            if (has_normal) std::memcpy(v + 3, &attrib.normals[3 * (size_t)i.normal_index], 12);
            if (has_uv) std::memcpy(v + 6, &attrib.texcoords[2 * (size_t)i.texcoord_index], 8);
            key k;
            for (int j = 0; j < 8; ++j) {
                const float f = v[j] + 0.0f; // -0 and +0 compare equal in the reference's != (:77-84)
                std::memcpy(&k.w[j], &f, 4);
            }
            bool is_nan = false;
            for (int j = 0; j < 8; ++j) is_nan = is_nan || v[j] != v[j]; // NaN never equals: always a new vertex
            auto it = is_nan ? seen.end() : seen.find(k);
            if (it != seen.end()) {
                g.indices.push_back(it->second);
            } else {
                const uint32_t id = (uint32_t)(g.positions.size() / 3);
                if (!is_nan) seen.emplace(k, id);
                g.indices.push_back(id);
                g.positions.insert(g.positions.end(), v, v + 3);
                if (has_normal) g.normals.insert(g.normals.end(), v + 3, v + 6);
                if (has_uv) g.uvs.insert(g.uvs.end(), v + 6, v + 8);
            }
        }
        out.push_back(std::move(g));
    }
    RAISE_ERROR_OK();
}

struct loaded_obj {
    std::vector<shape_data> shapes;          // one per sub-geometry
    std::vector<std::string> shape_names;
    std::vector<std::string> diffuse_paths;  // per shape: map_Kd of its first face's material, or empty
};

// load_model up to the upload (obj_loader.cpp:14-46): file -> shapes; the material library is
// looked up next to the OBJ file (:368)
inline ERROR_TYPE load_obj_file(const std::string& path, loaded_obj& out)
{
    std::ifstream in(path);
    if (!in) RAISE_ERROR_WARN(-1, "cannot open " + path);
    obj_attrib attrib;
    std::vector<obj_shape> shapes;
    std::vector<std::string> material_names;
    std::string mtllib;
    PASS_ERROR(parse_obj(in, attrib, shapes, material_names, mtllib));
    std::vector<obj_material> materials;
    if (!mtllib.empty()) {
        const size_t slash = path.find_last_of('/');
        const std::string dir = slash == std::string::npos ? std::string() : path.substr(0, slash + 1);
        std::ifstream min(dir + mtllib);
        if (min) PASS_ERROR(parse_mtl(min, materials));
    }
    PASS_ERROR(build_shapes(attrib, shapes, out.shapes));
    for (const obj_shape& s : shapes) {
        out.shape_names.push_back(s.name);
        std::string tex;
        const int mid = s.material_ids.empty() ? -1 : s.material_ids.front(); // :394-399: the first face decides
        if (mid >= 0 && (size_t)mid < material_names.size())
            for (const obj_material& m : materials) if (m.name == material_names[(size_t)mid]) tex = m.diffuse_texname;
        out.diffuse_paths.push_back(tex);
    }
    RAISE_ERROR_OK();
}

} // namespace rnd_host
