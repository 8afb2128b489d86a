"""Thin object wrappers over the C ABI, shaped like the reference's move-only
RAII handles (src/libs/vk_utils/vk_utils/handlers.hpp:34-118): one owner,
``destroy()`` idempotent, children before the device.

Errors follow the reference's default build: a non-zero return becomes an
exception carrying the code and message (error_handler.hpp:9-48).
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import PipelineInfo, SamplerInfo, VertexAttribute

# Vulkan numeric values (include/renderer_amd.h)
FORMAT_R8G8B8A8_UNORM = 37
FORMAT_R8G8B8A8_SRGB = 43
FORMAT_B8G8R8A8_UNORM = 44
FORMAT_B8G8R8A8_SRGB = 50
FORMAT_R8G8B8_SRGB = 29
FORMAT_R8_SRGB = 15
FORMAT_R32G32_SFLOAT = 103
FORMAT_R32G32B32_SFLOAT = 106
FORMAT_D24_UNORM_S8_UINT = 129
BUFFER_USAGE_UNIFORM, BUFFER_USAGE_INDEX, BUFFER_USAGE_VERTEX = 0x10, 0x40, 0x80
CULL_NONE, CULL_FRONT, CULL_BACK = 0, 1, 2
FRONT_FACE_COUNTER_CLOCKWISE, FRONT_FACE_CLOCKWISE = 0, 1
COMPARE_LESS, COMPARE_LESS_OR_EQUAL = 1, 3
FILTER_NEAREST, FILTER_LINEAR = 0, 1
INDEX_TYPE_UINT32 = 1
TOPOLOGY_TRIANGLE_LIST = 3
SHADER_DUMMY, SHADER_TEST_KTX, SHADER_PUSH_CONSTANTS = 0, 1, 2
STAGE_NAMES = ["vertex", "setup", "tile"]


class RendererError(RuntimeError):
    """errors::error{state, code, msg} of the reference (error_handler.hpp:50-92)."""

    def __init__(self, code, msg):
        super().__init__("rnd error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


def _check(code):
    if code != 0:
        raise RendererError(code, _lib.load().rnd_last_error().decode("utf-8", "replace"))


def _as_bytes(arr):
    a = np.ascontiguousarray(arr)
    return a, a.ctypes.data_as(C.c_void_p), a.nbytes


class _Handle:
    _destroy_name = None

    def __init__(self):
        self.h = C.c_void_p()

    def _child_of(self, dev):
        """Every object created on a device points back at it in the library and is destroyed before it
        (Device.destroy sees to that for the objects still alive)."""
        self.dev = dev
        dev._children.add(self)

    def destroy(self):
        if self.h:
            getattr(_lib.load(), self._destroy_name)(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Device(_Handle):
    """vk_utils::context (context.hpp:65-77): one per GPU."""
    _destroy_name = "rnd_device_destroy"

    def __init__(self, ordinal=0, stream=None):
        super().__init__()
        self._children = weakref.WeakSet()
        _check(_lib.load().rnd_device_create(ordinal, C.c_void_p(stream) if stream else None, C.byref(self.h)))

    def destroy(self):
        """Destroys what was created on the device and is still alive, then the device: the library refuses to destroy a
        device that objects still point at (the reference destroys its context last, context.cpp:567-574)."""
        if self.h:
            for child in list(self._children):
                child.destroy()
            _check(_lib.load().rnd_device_destroy(self.h))
            self.h = C.c_void_p()

    def wait_idle(self):
        _check(_lib.load().rnd_device_wait_idle(self.h))

    @property
    def stream(self):
        return _lib.load().rnd_device_stream(self.h)

    def set_profiling(self, enable):
        _check(_lib.load().rnd_device_set_profiling(self.h, 1 if enable else 0))

    def stage_times(self):
        out = (C.c_float * len(STAGE_NAMES))()
        _check(_lib.load().rnd_device_stage_times(self.h, out, len(STAGE_NAMES)))
        return {n: float(out[i]) for i, n in enumerate(STAGE_NAMES)}

    def submit(self, cmd, fence=None):
        _check(_lib.load().rnd_submit(self.h, cmd.h, fence.h if fence else None))


class Buffer(_Handle):
    _destroy_name = "rnd_buffer_destroy"

    def __init__(self, dev, usage, data=None, nbytes=None):
        super().__init__()
        self._child_of(dev)
        if data is not None:
            a, p, n = _as_bytes(data)
            self.nbytes = n
            _check(_lib.load().rnd_buffer_create(dev.h, n, usage, p, C.byref(self.h)))
        else:
            self.nbytes = int(nbytes)
            _check(_lib.load().rnd_buffer_create(dev.h, self.nbytes, usage, None, C.byref(self.h)))

    def write(self, data, offset=0):
        a, p, n = _as_bytes(data)
        _check(_lib.load().rnd_buffer_write(self.h, offset, p, n))

    def write_device(self, device_ptr, nbytes, offset=0):
        _check(_lib.load().rnd_buffer_write_device(self.h, offset, C.c_void_p(device_ptr), nbytes))

    def read(self, nbytes=None, offset=0):
        n = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(n, dtype=np.uint8)
        _check(_lib.load().rnd_buffer_read(self.h, offset, out.ctypes.data_as(C.c_void_p), n))
        return out


class Texture(_Handle):
    _destroy_name = "rnd_texture_destroy"

    def __init__(self, dev, texels, fmt=FORMAT_R8G8B8A8_SRGB, gen_mips=True, filtering=FILTER_LINEAR, tiled=False,
                 levels=None):
        super().__init__()
        self._child_of(dev)
        si = SamplerInfo(filtering, 1 if tiled else 0, 0.0)
        if levels is None:
            a = np.ascontiguousarray(texels, dtype=np.uint8)
            self.height, self.width = a.shape[0], a.shape[1]
            _check(_lib.load().rnd_texture_create_2d(dev.h, self.width, self.height, fmt, 1 if gen_mips else 0,
                                                    a.ctypes.data_as(C.c_void_p), C.byref(si), C.byref(self.h)))
        else:
            arrs = [np.ascontiguousarray(l, dtype=np.uint8) for l in levels]
            self.height, self.width = arrs[0].shape[0], arrs[0].shape[1]
            ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data_as(C.c_void_p) for a in arrs])
            _check(_lib.load().rnd_texture_create_levels(dev.h, self.width, self.height, fmt, len(arrs), ptrs,
                                                        C.byref(si), C.byref(self.h)))

    @property
    def level_count(self):
        n = C.c_uint32()
        _check(_lib.load().rnd_texture_level_count(self.h, C.byref(n)))
        return n.value

    def read_level(self, level):
        w, h = max(self.width >> level, 1), max(self.height >> level, 1)
        out = np.empty((h, w, 4), dtype=np.uint8)
        _check(_lib.load().rnd_texture_read_level(self.h, level, out.ctypes.data_as(C.c_void_p)))
        return out


class Target(_Handle):
    _destroy_name = "rnd_target_destroy"

    def __init__(self, dev, width, height, color_format=FORMAT_R8G8B8A8_SRGB, samples=1):
        super().__init__()
        self._child_of(dev)
        self.width, self.height, self.samples, self.color_format = width, height, samples, color_format
        _check(_lib.load().rnd_target_create(dev.h, width, height, color_format, FORMAT_D24_UNORM_S8_UINT, samples,
                                            C.byref(self.h)))

    def read(self):
        color = np.empty((self.height, self.width, 4), dtype=np.uint8)
        depth = np.empty((self.height, self.width, self.samples), dtype=np.uint32)
        _check(_lib.load().rnd_target_read(self.h, color.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p)))
        return color, depth

    @property
    def color_device_ptr(self):
        return _lib.load().rnd_target_color_device_ptr(self.h)

    def copy_color(self, device_ptr):
        """Queue a device-to-device copy of the colour image; does not wait."""
        _check(_lib.load().rnd_target_copy_color(self.h, C.c_void_p(device_ptr)))

    def copy_color_rows(self, first_row, row_count, device_ptr):
        """The same for rows [first_row, first_row + row_count) only."""
        _check(_lib.load().rnd_target_copy_color_rows(self.h, first_row, row_count, C.c_void_p(device_ptr)))


class Pipeline(_Handle):
    _destroy_name = "rnd_pipeline_destroy"

    def __init__(self, dev, shader, stride=32, off_pos=0, off_nrm=12, off_uv=24, cull_mode=CULL_BACK,
                 front_face=FRONT_FACE_COUNTER_CLOCKWISE, depth_test=True, depth_write=True,
                 depth_compare=COMPARE_LESS_OR_EQUAL, samples=1, topology=TOPOLOGY_TRIANGLE_LIST,
                 color_write_mask=0xF, blend_enable=False):
        super().__init__()
        self._child_of(dev)
        self.info = PipelineInfo(
            shader,
            VertexAttribute(FORMAT_R32G32B32_SFLOAT, off_pos),
            VertexAttribute(FORMAT_R32G32B32_SFLOAT if off_nrm is not None else 0, off_nrm or 0),
            VertexAttribute(FORMAT_R32G32_SFLOAT if off_uv is not None else 0, off_uv or 0),
            stride, topology, cull_mode, front_face, 1 if depth_test else 0, 1 if depth_write else 0, depth_compare,
            samples, color_write_mask, 1 if blend_enable else 0)
        _check(_lib.load().rnd_pipeline_create(dev.h, C.byref(self.info), C.byref(self.h)))


class Fence(_Handle):
    _destroy_name = "rnd_fence_destroy"

    def __init__(self, dev):
        super().__init__()
        self._child_of(dev)
        _check(_lib.load().rnd_fence_create(dev.h, C.byref(self.h)))

    def wait(self):
        _check(_lib.load().rnd_fence_wait(self.h))


class CommandBuffer(_Handle):
    """The vkCmd* subset the samples record (test_ktx_app.cpp:695-741)."""
    _destroy_name = "rnd_cmd_destroy"

    def __init__(self, dev):
        super().__init__()
        self._child_of(dev)
        self._keep = []
        _check(_lib.load().rnd_cmd_create(dev.h, C.byref(self.h)))

    def begin(self):
        self._keep = []
        _check(_lib.load().rnd_cmd_begin(self.h))

    def begin_pass(self, target, clear_color, clear_depth=1.0, area=None):
        cc = (C.c_float * 4)(*clear_color)
        x, y, w, h = area if area is not None else (0, 0, target.width, target.height)
        self._keep.append(target)
        _check(_lib.load().rnd_cmd_begin_pass(self.h, target.h, cc, clear_depth, x, y, w, h))

    def set_viewport(self, x, y, w, h, min_depth=0.0, max_depth=1.0):
        _check(_lib.load().rnd_cmd_set_viewport(self.h, x, y, w, h, min_depth, max_depth))

    def set_scissor(self, x, y, w, h):
        _check(_lib.load().rnd_cmd_set_scissor(self.h, x, y, w, h))

    def bind_pipeline(self, p):
        self._keep.append(p)
        _check(_lib.load().rnd_cmd_bind_pipeline(self.h, p.h))

    def bind_vertex_buffer(self, buf, offset=0):
        self._keep.append(buf)
        _check(_lib.load().rnd_cmd_bind_vertex_buffer(self.h, buf.h, offset))

    def bind_index_buffer(self, buf, offset=0):
        self._keep.append(buf)
        _check(_lib.load().rnd_cmd_bind_index_buffer(self.h, buf.h, offset, INDEX_TYPE_UINT32))

    def bind_uniform(self, buf, set=0, binding=0, offset=0, range=320):
        self._keep.append(buf)
        _check(_lib.load().rnd_cmd_bind_uniform(self.h, set, binding, buf.h, offset, range))

    def bind_texture(self, tex, set=0, binding=1):
        self._keep.append(tex)
        _check(_lib.load().rnd_cmd_bind_texture(self.h, set, binding, tex.h if tex is not None else None))

    def push_constants(self, data, offset=0):
        a, p, n = _as_bytes(data)
        _check(_lib.load().rnd_cmd_push_constants(self.h, offset, n, p))

    def draw_indexed(self, index_count, instance_count=1, first_index=0, vertex_offset=0, first_instance=0):
        _check(_lib.load().rnd_cmd_draw_indexed(self.h, index_count, instance_count, first_index, vertex_offset,
                                               first_instance))

    def end_pass(self):
        _check(_lib.load().rnd_cmd_end_pass(self.h))

    def end(self):
        _check(_lib.load().rnd_cmd_end(self.h))

    def set_tile_shard(self, index, count):
        _check(_lib.load().rnd_cmd_set_tile_shard(self.h, index, count))

    def visible_draws(self):
        """Draws of the last executed frame that were not skipped whole (waits for the device)."""
        n = C.c_uint32(0)
        _check(_lib.load().rnd_cmd_visible_draws(self.h, C.byref(n)))
        return n.value
