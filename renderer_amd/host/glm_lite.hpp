// glm_lite.hpp -- the handful of glm functions the reference's host code calls on the draw
// path, restated (glm itself is an unfetched submodule of the reference: .gitmodules:7-9).
// Column-major, column vectors, float32, products and sums in glm 0.9.9's source order.
// Call sites: camera.cpp:13-23, base_obj_viewer_app.cpp:122-132, obj_loader.cpp:306-307,
// test_ktx_app.cpp:714-715.
#pragma once
#include <cmath>

namespace glm_lite {

struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };

struct mat4 {
    float c[4][4]; // c[column][row]
    explicit mat4(float d = 1.0f) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) c[i][j] = i == j ? d : 0.0f; }
    const float* data() const { return &c[0][0]; }
};

inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline vec3 normalize(vec3 a) { float inv = 1.0f / std::sqrt(dot(a, a)); return a * inv; }

inline mat4 operator*(const mat4& a, const mat4& b)
{
    mat4 r(0.0f);
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row)
            r.c[col][row] = a.c[0][row] * b.c[col][0] + a.c[1][row] * b.c[col][1] + a.c[2][row] * b.c[col][2] + a.c[3][row] * b.c[col][3];
    return r;
}

inline vec3 mat3_mul(const mat4& m, vec3 v) // mat3(m) * v
{
    return {m.c[0][0] * v.x + m.c[1][0] * v.y + m.c[2][0] * v.z,
            m.c[0][1] * v.x + m.c[1][1] * v.y + m.c[2][1] * v.z,
            m.c[0][2] * v.x + m.c[1][2] * v.y + m.c[2][2] * v.z};
}

// glm::perspectiveFov, right handed.  zero_to_one = false is the [-1,1] depth form that is in
// effect in the reference: GLM_FORCE_DEPTH_ZERO_TO_ONE is defined in camera.cpp:5 only after
// camera.hpp:5-6 has already included glm.
inline mat4 perspectiveFov(float fov, float width, float height, float z_near, float z_far, bool zero_to_one = false)
{
    const float h = std::cos(0.5f * fov) / std::sin(0.5f * fov);
    const float w = h * height / width;
    mat4 r(0.0f);
    r.c[0][0] = w;
    r.c[1][1] = h;
    r.c[2][3] = -1.0f;
    if (zero_to_one) {
        r.c[2][2] = z_far / (z_near - z_far);
        r.c[3][2] = -(z_far * z_near) / (z_far - z_near);
    } else {
        r.c[2][2] = -(z_far + z_near) / (z_far - z_near);
        r.c[3][2] = -(2.0f * z_far * z_near) / (z_far - z_near);
    }
    return r;
}

inline mat4 lookAt(vec3 eye, vec3 center, vec3 up)
{
    const vec3 f = normalize(center - eye);
    const vec3 s = normalize(cross(f, up));
    const vec3 u = cross(s, f);
    mat4 r(1.0f);
    r.c[0][0] = s.x; r.c[1][0] = s.y; r.c[2][0] = s.z;
    r.c[0][1] = u.x; r.c[1][1] = u.y; r.c[2][1] = u.z;
    r.c[0][2] = -f.x; r.c[1][2] = -f.y; r.c[2][2] = -f.z;
    r.c[3][0] = -dot(s, eye); r.c[3][1] = -dot(u, eye); r.c[3][2] = dot(f, eye);
    return r;
}

inline mat4 translate(const mat4& m, vec3 v)
{
    mat4 r = m;
    for (int row = 0; row < 4; ++row) r.c[3][row] = m.c[0][row] * v.x + m.c[1][row] * v.y + m.c[2][row] * v.z + m.c[3][row];
    return r;
}

inline mat4 scale(const mat4& m, vec3 v)
{
    mat4 r = m;
    for (int row = 0; row < 4; ++row) { r.c[0][row] = m.c[0][row] * v.x; r.c[1][row] = m.c[1][row] * v.y; r.c[2][row] = m.c[2][row] * v.z; }
    return r;
}

inline mat4 rotate(const mat4& m, float angle, vec3 axis_in)
{
    const float c = std::cos(angle), s = std::sin(angle);
    const vec3 a = normalize(axis_in);
    const vec3 t = a * (1.0f - c);
    float R[3][3];
    R[0][0] = c + t.x * a.x;       R[0][1] = t.x * a.y + s * a.z; R[0][2] = t.x * a.z - s * a.y;
    R[1][0] = t.y * a.x - s * a.z; R[1][1] = c + t.y * a.y;       R[1][2] = t.y * a.z + s * a.x;
    R[2][0] = t.z * a.x + s * a.y; R[2][1] = t.z * a.y - s * a.x; R[2][2] = c + t.z * a.z;
    mat4 r(0.0f);
    for (int i = 0; i < 3; ++i)
        for (int row = 0; row < 4; ++row) r.c[i][row] = m.c[0][row] * R[i][0] + m.c[1][row] * R[i][1] + m.c[2][row] * R[i][2];
    for (int row = 0; row < 4; ++row) r.c[3][row] = m.c[3][row];
    return r;
}

inline mat4 yawPitchRoll(float yaw, float pitch, float roll)
{
    const float ch = std::cos(yaw), sh = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch), cb = std::cos(roll), sb = std::sin(roll);
    mat4 r(1.0f);
    r.c[0][0] = ch * cb + sh * sp * sb; r.c[0][1] = sb * cp; r.c[0][2] = -sh * cb + ch * sp * sb;
    r.c[1][0] = -ch * sb + sh * sp * cb; r.c[1][1] = cb * cp; r.c[1][2] = sb * sh + ch * sp * cb;
    r.c[2][0] = sh * cp; r.c[2][1] = -sp; r.c[2][2] = ch * cp;
    return r;
}

} // namespace glm_lite
