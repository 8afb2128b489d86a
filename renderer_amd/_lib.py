"""ctypes binding of librenderer_amd.so (the C ABI of include/renderer_amd.h).

The library is built in-tree by ``renderer_amd/csrc/Makefile``.  There is no
fallback: if the shared object is missing, loading raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RENDERER_AMD_LIB selects another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("RENDERER_AMD_LIB") or os.path.join(_HERE, "librenderer_amd.so")

# every symbol include/renderer_amd.h declares
SYMBOLS = [
    "rnd_last_error",
    "rnd_device_create", "rnd_device_destroy", "rnd_device_wait_idle", "rnd_device_stream",
    "rnd_device_set_profiling", "rnd_device_stage_times",
    "rnd_buffer_create", "rnd_buffer_write", "rnd_buffer_write_device", "rnd_buffer_read", "rnd_buffer_destroy",
    "rnd_texture_create_2d", "rnd_texture_create_levels", "rnd_texture_level_count", "rnd_texture_read_level",
    "rnd_texture_destroy",
    "rnd_target_create", "rnd_target_read", "rnd_target_color_device_ptr", "rnd_target_copy_color", "rnd_target_copy_color_rows", "rnd_target_destroy",
    "rnd_pipeline_create", "rnd_pipeline_destroy",
    "rnd_cmd_create", "rnd_cmd_begin", "rnd_cmd_begin_pass", "rnd_cmd_set_viewport", "rnd_cmd_set_scissor",
    "rnd_cmd_bind_pipeline", "rnd_cmd_bind_vertex_buffer", "rnd_cmd_bind_index_buffer", "rnd_cmd_bind_uniform",
    "rnd_cmd_bind_texture", "rnd_cmd_push_constants", "rnd_cmd_draw_indexed", "rnd_cmd_end_pass", "rnd_cmd_end",
    "rnd_cmd_destroy", "rnd_cmd_set_tile_shard", "rnd_cmd_visible_draws",
    "rnd_submit", "rnd_draw_indexed_noalloc",
    "rnd_fence_create", "rnd_fence_wait", "rnd_fence_destroy",
]


class SamplerInfo(C.Structure):
    _fields_ = [("filtering", C.c_int32), ("tiled", C.c_int32), ("max_anisotropy", C.c_float)]


class VertexAttribute(C.Structure):
    _fields_ = [("format", C.c_int32), ("offset", C.c_uint32)]


class PipelineInfo(C.Structure):
    _fields_ = [
        ("shader", C.c_int32),
        ("position", VertexAttribute), ("normal", VertexAttribute), ("uv", VertexAttribute),
        ("stride", C.c_uint32), ("topology", C.c_int32),
        ("cull_mode", C.c_int32), ("front_face", C.c_int32),
        ("depth_test", C.c_int32), ("depth_write", C.c_int32), ("depth_compare", C.c_int32),
        ("samples", C.c_uint32), ("color_write_mask", C.c_uint32), ("blend_enable", C.c_int32),
    ]


class NoAllocDraw(C.Structure):
    """rnd_noalloc_draw"""
    _fields_ = [
        ("target", C.c_void_p), ("pipeline", C.c_void_p), ("texture", C.c_void_p),
        ("vertex_buffer", C.c_void_p), ("vertex_offset_bytes", C.c_uint64), ("vertex_count", C.c_uint32),
        ("index_buffer", C.c_void_p), ("index_offset_bytes", C.c_uint64), ("index_count", C.c_uint32),
        ("uniform_block", C.POINTER(C.c_float)), ("push_constants", C.POINTER(C.c_float)),
        ("clear_color", C.c_float * 4), ("clear_depth", C.c_float),
        ("viewport", C.c_float * 6), ("scissor", C.c_int32 * 4),
    ]


_lib = None


def load():
    """Load the shared object once and declare the argument types."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "librenderer_amd.so is missing (%s); build it with `make -C renderer_amd/csrc` "
            "or __graft_entry__.build(). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_float
    sig = {
        "rnd_last_error": (C.c_char_p, []),
        "rnd_device_create": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
        "rnd_device_destroy": (C.c_int, [vp]),
        "rnd_device_wait_idle": (C.c_int, [vp]),
        "rnd_device_stream": (vp, [vp]),
        "rnd_device_set_profiling": (C.c_int, [vp, C.c_int]),
        "rnd_device_stage_times": (C.c_int, [vp, C.POINTER(f32), C.c_int]),
        "rnd_buffer_create": (C.c_int, [vp, u64, u32, vp, C.POINTER(vp)]),
        "rnd_buffer_write": (C.c_int, [vp, u64, vp, u64]),
        "rnd_buffer_write_device": (C.c_int, [vp, u64, vp, u64]),
        "rnd_buffer_read": (C.c_int, [vp, u64, vp, u64]),
        "rnd_buffer_destroy": (C.c_int, [vp]),
        "rnd_texture_create_2d": (C.c_int, [vp, u32, u32, i32, i32, vp, C.POINTER(SamplerInfo), C.POINTER(vp)]),
        "rnd_texture_create_levels": (C.c_int, [vp, u32, u32, i32, u32, C.POINTER(vp), C.POINTER(SamplerInfo), C.POINTER(vp)]),
        "rnd_texture_level_count": (C.c_int, [vp, C.POINTER(u32)]),
        "rnd_texture_read_level": (C.c_int, [vp, u32, vp]),
        "rnd_texture_destroy": (C.c_int, [vp]),
        "rnd_target_create": (C.c_int, [vp, u32, u32, i32, i32, u32, C.POINTER(vp)]),
        "rnd_target_read": (C.c_int, [vp, vp, vp]),
        "rnd_target_color_device_ptr": (vp, [vp]),
        "rnd_target_copy_color": (C.c_int, [vp, vp]),
        "rnd_target_copy_color_rows": (C.c_int, [vp, u32, u32, vp]),
        "rnd_target_destroy": (C.c_int, [vp]),
        "rnd_pipeline_create": (C.c_int, [vp, C.POINTER(PipelineInfo), C.POINTER(vp)]),
        "rnd_pipeline_destroy": (C.c_int, [vp]),
        "rnd_cmd_create": (C.c_int, [vp, C.POINTER(vp)]),
        "rnd_cmd_begin": (C.c_int, [vp]),
        "rnd_cmd_begin_pass": (C.c_int, [vp, vp, C.POINTER(f32), f32, i32, i32, u32, u32]),
        "rnd_cmd_set_viewport": (C.c_int, [vp, f32, f32, f32, f32, f32, f32]),
        "rnd_cmd_set_scissor": (C.c_int, [vp, i32, i32, u32, u32]),
        "rnd_cmd_bind_pipeline": (C.c_int, [vp, vp]),
        "rnd_cmd_bind_vertex_buffer": (C.c_int, [vp, vp, u64]),
        "rnd_cmd_bind_index_buffer": (C.c_int, [vp, vp, u64, i32]),
        "rnd_cmd_bind_uniform": (C.c_int, [vp, u32, u32, vp, u64, u64]),
        "rnd_cmd_bind_texture": (C.c_int, [vp, u32, u32, vp]),
        "rnd_cmd_push_constants": (C.c_int, [vp, u32, u32, vp]),
        "rnd_cmd_draw_indexed": (C.c_int, [vp, u32, u32, u32, i32, u32]),
        "rnd_cmd_end_pass": (C.c_int, [vp]),
        "rnd_cmd_end": (C.c_int, [vp]),
        "rnd_cmd_destroy": (C.c_int, [vp]),
        "rnd_cmd_set_tile_shard": (C.c_int, [vp, u32, u32]),
        "rnd_cmd_visible_draws": (C.c_int, [vp, C.POINTER(u32)]),
        "rnd_submit": (C.c_int, [vp, vp, vp]),
        "rnd_draw_indexed_noalloc": (C.c_int, [vp, C.POINTER(NoAllocDraw), vp]),
        "rnd_fence_create": (C.c_int, [vp, C.POINTER(vp)]),
        "rnd_fence_wait": (C.c_int, [vp]),
        "rnd_fence_destroy": (C.c_int, [vp]),
    }
    for name in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype, fn.argtypes = sig[name]
    _lib = lib
    return lib
