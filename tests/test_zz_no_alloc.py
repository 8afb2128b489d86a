"""rnd_draw_indexed_noalloc: one pass of one indexed draw with every intermediate buffer in the library's static device
array (no hipMalloc on the call path), against the same pass recorded and submitted through the ordinary entry points.
Two calls in a row, the second with a smaller input and a smaller target: the array keeps the first call's contents, so
the second one checks that what must start at zero is cleared.  Depth and colour must be identical."""
import ctypes as C

import numpy as np
import pytest

import helpers
from renderer_amd import _lib, api, scenes
from renderer_amd.runner import SceneRunner

pytestmark = pytest.mark.gpu


def noalloc_render(device, scene, runner):
    """The scene's one draw through rnd_draw_indexed_noalloc, using the buffers, texture, pipeline and target the
    runner created; returns (colour, depth)."""
    lib = _lib.load()
    dr = scene.draws[0]
    pipe = scene.pipelines[dr["pipeline"]]
    d = _lib.NoAllocDraw()
    d.target = runner.target.h
    d.pipeline = runner.pipelines[dr["pipeline"]].h
    tex = dr.get("texture", -1)
    d.texture = runner.textures[tex].h if tex is not None and tex >= 0 else None
    d.vertex_buffer = runner.vbs[dr["vb"]].h
    d.vertex_offset_bytes = dr.get("vb_offset", 0)
    d.vertex_count = (scene.vbs[dr["vb"]].nbytes - dr.get("vb_offset", 0)) // pipe["stride"]
    d.index_buffer = runner.ibs[dr["ib"]].h
    d.index_offset_bytes = dr.get("ib_offset", 0) + 4 * dr.get("first_index", 0)
    d.index_count = dr["index_count"]
    ubo = np.ascontiguousarray(scene.ubos[dr["ubo"]], dtype=np.float32)
    push = np.ascontiguousarray(dr["push"], dtype=np.float32)
    d.uniform_block = ubo.ctypes.data_as(C.POINTER(C.c_float))
    d.push_constants = push.ctypes.data_as(C.POINTER(C.c_float))
    d.clear_color = (C.c_float * 4)(*scene.clear_color)
    d.clear_depth = scene.clear_depth
    d.viewport = (C.c_float * 6)(*scene.viewport)
    d.scissor = (C.c_int32 * 4)(*scene.scissor)
    fence = api.Fence(device)
    try:
        rc = lib.rnd_draw_indexed_noalloc(device.h, C.byref(d), fence.h)
        assert rc == 0, lib.rnd_last_error()
        fence.wait()
        return runner.target.read()
    finally:
        fence.destroy()


def one_draw_scenes():
    big = scenes.config1(640, 352, quads=150, tex_size=128)          # 45 000 triangles, 22 801 vertices, 220 tiles
    clipped = helpers.perspective_soup(320, 200, 600, 41, near_cross=True, shader=scenes.SHADER_TEST_KTX,
                                       texture=dict(texels=scenes.checker_texture(64, 1), gen_mips=True, filtering=1, tiled=False, fmt=43),
                                       pipeline_kw=dict(cull_mode=0))  # smaller, with clipped primitives
    small = scenes.config1(200, 120, quads=24, tex_size=32)          # 1 152 triangles, 28 tiles
    return [big, clipped, small]


def test_noalloc_draw_equals_recorded_pass(device):
    expected, got = [], []
    runners = []
    try:
        for sc in one_draw_scenes():
            assert len(sc.draws) == 1
            r = SceneRunner(sc, device)
            runners.append(r)
            expected.append(r.render())
        # the calls in a row, nothing else in between: larger input first
        for sc, r in zip(one_draw_scenes(), runners):
            got.append(noalloc_render(device, sc, r))
    finally:
        for r in runners:
            r.destroy()
    for k, ((ec, ed), (gc, gd)) in enumerate(zip(expected, got)):
        assert (ed != ed.max()).any(), "scene %d draws nothing" % k
        assert np.array_equal(gd, ed), "scene %d: depth differs at %d samples" % (k, int((gd != ed).sum()))
        assert np.array_equal(gc, ec), "scene %d: colour differs at %d bytes" % (k, int((gc != ec).sum()))


def _free_device_bytes():
    """hipMemGetInfo of the current device, from the HIP runtime the library itself is linked against."""
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_noalloc_calls_leave_free_memory_unchanged(device):
    """What the entry point is for: after one warm-up call (the runtime loads the kernels' code object on first launch)
    a run of calls takes no device memory -- the free bytes hipMemGetInfo reports are the same before and after."""
    lib = _lib.load()
    sc = scenes.config1(320, 200, quads=60, tex_size=64)
    r = SceneRunner(sc, device)
    try:
        d = _lib.NoAllocDraw()
        ubo = np.ascontiguousarray(sc.ubos[0], dtype=np.float32)
        push = np.ascontiguousarray(sc.draws[0]["push"], dtype=np.float32)
        d.target = r.target.h; d.pipeline = r.pipelines[0].h; d.texture = r.textures[0].h
        d.vertex_buffer = r.vbs[0].h; d.vertex_count = sc.vbs[0].shape[0]
        d.index_buffer = r.ibs[0].h; d.index_count = sc.ibs[0].size
        d.uniform_block = ubo.ctypes.data_as(C.POINTER(C.c_float))
        d.push_constants = push.ctypes.data_as(C.POINTER(C.c_float))
        d.clear_color = (C.c_float * 4)(*sc.clear_color); d.clear_depth = sc.clear_depth
        d.viewport = (C.c_float * 6)(*sc.viewport); d.scissor = (C.c_int32 * 4)(*sc.scissor)
        assert lib.rnd_draw_indexed_noalloc(device.h, C.byref(d), None) == 0, lib.rnd_last_error()
        device.wait_idle()
        before = _free_device_bytes()
        for _ in range(20):
            assert lib.rnd_draw_indexed_noalloc(device.h, C.byref(d), None) == 0, lib.rnd_last_error()
        device.wait_idle()
        after = _free_device_bytes()
        assert after == before, "20 calls changed the device's free memory by %d bytes" % (before - after)
    finally:
        r.destroy()


def test_noalloc_limits_are_refused(device):
    lib = _lib.load()
    sc = scenes.config1(200, 120, quads=24, tex_size=32)
    r = SceneRunner(sc, device)
    big_target = api.Target(device, 2048, 1024)  # 2048 tiles
    try:
        d = _lib.NoAllocDraw()
        ubo = np.ascontiguousarray(sc.ubos[0], dtype=np.float32)
        d.target = r.target.h; d.pipeline = r.pipelines[0].h; d.texture = r.textures[0].h
        d.vertex_buffer = r.vbs[0].h; d.vertex_count = sc.vbs[0].shape[0]
        d.index_buffer = r.ibs[0].h; d.index_count = sc.ibs[0].size
        d.uniform_block = ubo.ctypes.data_as(C.POINTER(C.c_float))
        d.viewport = (C.c_float * 6)(*sc.viewport); d.scissor = (C.c_int32 * 4)(*sc.scissor)
        d.vertex_count = 65537
        assert lib.rnd_draw_indexed_noalloc(device.h, C.byref(d), None) == -2
        assert b"65536" in lib.rnd_last_error()
        d.vertex_count = sc.vbs[0].shape[0]
        d.index_count = sc.ibs[0].size + 3
        assert lib.rnd_draw_indexed_noalloc(device.h, C.byref(d), None) == -1  # past the index buffer
        d.index_count = sc.ibs[0].size
        d.target = big_target.h
        assert lib.rnd_draw_indexed_noalloc(device.h, C.byref(d), None) == -2
        assert b"tiles" in lib.rnd_last_error()
        d.target = r.target.h
        assert lib.rnd_draw_indexed_noalloc(device.h, C.byref(d), None) == 0
        device.wait_idle()
    finally:
        big_target.destroy()
        r.destroy()
