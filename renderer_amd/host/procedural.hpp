// procedural.hpp -- the stand-ins for the reference's OBJ / KTX assets (Git-LFS stubs there): a UV
// sphere with normals and texture coordinates, and a checker texture with colour ramps.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "obj_model.hpp"

namespace rnd_host {

inline shape_data make_uv_sphere(int segments, int rings, float radius)
{
    shape_data s;
    const double pi = 3.14159265358979323846;
    for (int r = 0; r <= rings; ++r) {
        const double theta = pi * r / rings;
        for (int k = 0; k <= segments; ++k) {
            const double phi = 2.0 * pi * k / segments;
            const float n[3] = {(float)(std::sin(theta) * std::cos(phi)), (float)std::cos(theta), (float)(std::sin(theta) * std::sin(phi))};
            for (int j = 0; j < 3; ++j) { s.positions.push_back(n[j] * radius); s.normals.push_back(n[j]); }
            s.uvs.push_back((float)k / segments);
            s.uvs.push_back((float)r / rings);
        }
    }
    for (int r = 0; r < rings; ++r)
        for (int k = 0; k < segments; ++k) {
            const uint32_t a = (uint32_t)(r * (segments + 1) + k), b = a + (uint32_t)segments + 1;
            if (r != 0) { s.indices.push_back(a); s.indices.push_back(b); s.indices.push_back(a + 1); }
            if (r != rings - 1) { s.indices.push_back(a + 1); s.indices.push_back(b); s.indices.push_back(b + 1); }
        }
    return s;
}

inline std::vector<uint8_t> make_checker(uint32_t size)
{
    std::vector<uint8_t> t((size_t)size * size * 4);
    for (uint32_t y = 0; y < size; ++y)
        for (uint32_t x = 0; x < size; ++x) {
            const bool c = ((x / (size / 16)) + (y / (size / 16))) & 1u;
            uint8_t* p = &t[((size_t)y * size + x) * 4];
            p[0] = c ? 230 : 60; p[1] = (uint8_t)(40 + 200 * x / size); p[2] = (uint8_t)(40 + 200 * y / size); p[3] = 255;
        }
    return t;
}

} // namespace rnd_host
