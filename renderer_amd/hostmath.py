"""Host-side matrix maths of the reference, restated in float32 numpy.

The reference builds its per-frame uniform block with glm (unfetched submodule,
.gitmodules:7-9).  These functions follow glm 0.9.9's source order of products
and sums so the bytes uploaded to the uniform buffer match what the C++ host
mirror (renderer_amd/host/glm_lite.hpp) produces.  Matrices are column-major:
``m[c, r]`` is column c, row r, and ``m.reshape(16)`` is the upload order.

Call sites restated: camera.cpp:9-25 (perspectiveFov, lookAt),
base_obj_viewer_app.cpp:90-141 (yawPitchRoll, mvp), obj_loader.cpp:294-307
(translate*scale), test_ktx_app.cpp:714-715 (rotate, translate).
"""
import math

import numpy as np

F = np.float32


def identity():
    return np.eye(4, dtype=F)


def mul(a, b):
    """a * b for column-major storage: result[c] = sum_k a[k] * b[c][k] (glm operator*)."""
    out = np.zeros((4, 4), dtype=F)
    for c in range(4):
        acc = a[0] * b[c, 0]
        acc = (acc + a[1] * b[c, 1]).astype(F)
        acc = (acc + a[2] * b[c, 2]).astype(F)
        acc = (acc + a[3] * b[c, 3]).astype(F)
        out[c] = acc
    return out


def _cos(x):
    return F(math.cos(float(F(x))))


def _sin(x):
    return F(math.sin(float(F(x))))


def radians(deg):
    return F(F(deg) * F(0.01745329251994329576923690768489))


def perspective_fov(fov, width, height, z_near, z_far, zero_to_one=False):
    """glm::perspectiveFov, right handed.  zero_to_one=False is the [-1,1] depth
    form that is in effect in the reference: GLM_FORCE_DEPTH_ZERO_TO_ONE is
    defined after glm was already included (camera.cpp:5 vs camera.hpp:5-6)."""
    fov, width, height, n, f = F(fov), F(width), F(height), F(z_near), F(z_far)
    h = F(_cos(F(0.5) * fov) / _sin(F(0.5) * fov))
    w = F(F(h * height) / width)
    m = np.zeros((4, 4), dtype=F)
    m[0, 0] = w
    m[1, 1] = h
    m[2, 3] = F(-1)
    if zero_to_one:
        m[2, 2] = F(f / F(n - f))
        m[3, 2] = F(-(F(f * n)) / F(f - n))
    else:
        m[2, 2] = F(-(F(f + n)) / F(f - n))
        m[3, 2] = F(-(F(F(F(2) * f) * n)) / F(f - n))
    return m


def _dot(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def _cross(a, b):
    return np.array([F(F(a[1] * b[2]) - F(b[1] * a[2])),
                     F(F(a[2] * b[0]) - F(b[2] * a[0])),
                     F(F(a[0] * b[1]) - F(b[0] * a[1]))], dtype=F)


def _normalize(a):
    inv = F(F(1) / F(np.sqrt(_dot(a, a))))
    return np.array([F(a[0] * inv), F(a[1] * inv), F(a[2] * inv)], dtype=F)


def look_at(eye, center, up):
    """glm::lookAt (lookAtRH), camera.cpp:22."""
    eye, center, up = (np.asarray(v, dtype=F) for v in (eye, center, up))
    f = _normalize((center - eye).astype(F))
    s = _normalize(_cross(f, up))
    u = _cross(s, f)
    m = identity()
    m[0, 0], m[1, 0], m[2, 0] = s
    m[0, 1], m[1, 1], m[2, 1] = u
    m[0, 2], m[1, 2], m[2, 2] = -f
    m[3, 0] = -_dot(s, eye)
    m[3, 1] = -_dot(u, eye)
    m[3, 2] = _dot(f, eye)
    return m


def translate(m, v):
    v = np.asarray(v, dtype=F)
    out = m.copy()
    out[3] = (((m[0] * v[0]).astype(F) + (m[1] * v[1]).astype(F)).astype(F) + (m[2] * v[2]).astype(F)).astype(F) + m[3]
    return out.astype(F)


def scale(m, v):
    v = np.asarray(v, dtype=F)
    out = m.copy()
    out[0] = m[0] * v[0]
    out[1] = m[1] * v[1]
    out[2] = m[2] * v[2]
    return out.astype(F)


def rotate(m, angle, axis):
    """glm::rotate(m, angle, axis), test_ktx_app.cpp:714."""
    a = F(angle)
    c, s = _cos(a), _sin(a)
    ax = _normalize(np.asarray(axis, dtype=F))
    t = (F(F(1) - c) * ax).astype(F)
    R = np.zeros((3, 3), dtype=F)
    R[0, 0] = F(c + F(t[0] * ax[0]))
    R[0, 1] = F(F(t[0] * ax[1]) + F(s * ax[2]))
    R[0, 2] = F(F(t[0] * ax[2]) - F(s * ax[1]))
    R[1, 0] = F(F(t[1] * ax[0]) - F(s * ax[2]))
    R[1, 1] = F(c + F(t[1] * ax[1]))
    R[1, 2] = F(F(t[1] * ax[2]) + F(s * ax[0]))
    R[2, 0] = F(F(t[2] * ax[0]) + F(s * ax[1]))
    R[2, 1] = F(F(t[2] * ax[1]) - F(s * ax[0]))
    R[2, 2] = F(c + F(t[2] * ax[2]))
    out = np.zeros((4, 4), dtype=F)
    for i in range(3):
        out[i] = (((m[0] * R[i, 0]).astype(F) + (m[1] * R[i, 1]).astype(F)).astype(F) + (m[2] * R[i, 2]).astype(F)).astype(F)
    out[3] = m[3]
    return out


def yaw_pitch_roll(yaw, pitch, roll):
    """glm::yawPitchRoll, base_obj_viewer_app.cpp:122."""
    ch, sh = _cos(yaw), _sin(yaw)
    cp, sp = _cos(pitch), _sin(pitch)
    cb, sb = _cos(roll), _sin(roll)
    m = identity()
    m[0, 0] = F(F(ch * cb) + F(F(sh * sp) * sb))
    m[0, 1] = F(sb * cp)
    m[0, 2] = F(F(-sh * cb) + F(F(ch * sp) * sb))
    m[1, 0] = F(F(-ch * sb) + F(F(sh * sp) * cb))
    m[1, 1] = F(cb * cp)
    m[1, 2] = F(F(sb * sh) + F(F(ch * sp) * cb))
    m[2, 0] = F(sh * cp)
    m[2, 1] = F(-sp)
    m[2, 2] = F(ch * cp)
    return m


class Camera:
    """vk_utils::camera (camera.hpp:11-39) with the defaults the sample base class
    sets (base_obj_viewer_app.cpp:12-14)."""

    def __init__(self):
        self.eye_position = np.array([0, 0, -2], dtype=F)
        self.target_position = np.array([0, 0, 0], dtype=F)
        self.up = np.array([0, 1, 0], dtype=F)
        self.fov = F(90.0)
        self.zero_to_one = False  # effective behaviour of the reference build, see perspective_fov
        self.proj_matrix = identity()
        self.view_matrix = identity()
        self.view_proj_matrix = identity()

    def update(self, width, height):
        # near/far are hard-coded in the reference (camera.cpp:13), z_near/z_far members are ignored
        self.proj_matrix = perspective_fov(radians(self.fov), width, height, 0.01, 100.0, self.zero_to_one)
        self.view_matrix = look_at(self.eye_position, self.target_position, self.up)
        self.view_proj_matrix = mul(self.proj_matrix, self.view_matrix)


def global_ubo(camera, model_transform, width, height):
    """base_obj_viewer_app::global_ubo (base_obj_viewer_app.hpp:36-43) as filled by
    draw_frame (base_obj_viewer_app.cpp:127-132): projection, view_projection,
    view, model, mvp -- 80 floats, 320 bytes."""
    camera.update(width, height)
    model = mul(identity(), np.asarray(model_transform, dtype=F))
    mvp = mul(mul(camera.proj_matrix, camera.view_matrix), model)
    return np.concatenate([camera.proj_matrix.reshape(16), camera.view_proj_matrix.reshape(16),
                           camera.view_matrix.reshape(16), model.reshape(16), mvp.reshape(16)]).astype(F)


def ring_transform(j, count=10):
    """Instance j of the ring the samples draw (test_ktx_app.cpp:693, 706-717)."""
    rot_angle = F(F(360.0) / F(count))
    curr = F(0)
    for _ in range(j):
        curr = F(curr + rot_angle)
    m = rotate(identity(), radians(curr), [0.0, 0.0, 1.0])
    return translate(m, [0, 10, 0])


RING_COLORS = np.array([
    [0., 0., 0., 1.], [1., 0., 0., 1.], [0., 1., 0., 1.], [0., 0., 1., 1.], [1., 1., 0., 1.],
    [0., 1., 1., 1.], [1., 0., 1., 1.], [1., 1., 1., 1.], [1., 0.5, 0., 1.], [0., 0.5, 1., 1.],
], dtype=F)  # test_ktx_app.cpp:680-691
