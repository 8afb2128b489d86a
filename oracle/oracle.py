"""ctypes binding of the CPU oracle (oracle/libraster_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (renderer_amd/) never imports
this module.  PARITY UNPINNED: the reference holds no fixtures for this path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libraster_oracle.so")


class OrcTexture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("levels", C.c_uint32), ("srgb", C.c_uint32),
                ("tiled", C.c_uint32), ("filter_linear", C.c_uint32), ("level_data", C.c_void_p * 16)]


class OrcDraw(C.Structure):
    _fields_ = [("shader", C.c_int32), ("stride", C.c_uint32), ("off_pos", C.c_int32), ("off_nrm", C.c_int32),
                ("off_uv", C.c_int32), ("cull_mode", C.c_int32), ("front_face", C.c_int32), ("depth_test", C.c_int32),
                ("depth_write", C.c_int32), ("depth_op", C.c_int32), ("vb", C.c_void_p), ("vb_bytes", C.c_uint64),
                ("ib", C.c_void_p), ("index_count", C.c_uint32), ("first_index", C.c_uint32),
                ("vertex_offset", C.c_int32), ("ubo", C.c_void_p), ("push", C.c_float * 20), ("texture", C.c_int32)]


class OrcPass(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples", C.c_uint32), ("color_srgb", C.c_uint32),
                ("color_bgra", C.c_uint32), ("clear_color", C.c_float * 4), ("clear_depth", C.c_float),
                ("viewport", C.c_float * 6), ("scissor", C.c_int32 * 4), ("area", C.c_int32 * 4)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("oracle library missing: build it with `make -C oracle`")
    lib = C.CDLL(LIB_PATH)
    f32, u32, i32 = C.c_float, C.c_uint32, C.c_int32
    lib.orc_render.restype = C.c_int
    lib.orc_render.argtypes = [C.POINTER(OrcPass), C.POINTER(OrcDraw), C.c_int, C.POINTER(OrcTexture), C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.orc_mip_level_count.restype = u32
    lib.orc_mip_level_count.argtypes = [u32, u32, C.c_int]
    lib.orc_gen_mip.restype = C.c_int
    lib.orc_gen_mip.argtypes = [C.c_void_p, u32, u32, C.c_void_p, u32, u32, C.c_int]
    lib.orc_srgb_encode.restype = u32
    lib.orc_srgb_encode.argtypes = [f32]
    lib.orc_srgb_decode.restype = f32
    lib.orc_srgb_decode.argtypes = [u32]
    lib.orc_log2.restype = f32
    lib.orc_log2.argtypes = [f32]
    lib.orc_snap.restype = i32
    lib.orc_snap.argtypes = [f32]
    lib.orc_depth24.restype = u32
    lib.orc_depth24.argtypes = [f32]
    lib.orc_sample_texture.restype = None
    lib.orc_sample_texture.argtypes = [C.POINTER(OrcTexture), f32, f32, f32, f32, f32, f32, C.POINTER(f32)]
    m16 = C.POINTER(f32)
    for name, args in [("orc_perspective_fov_rh_no", [f32] * 5 + [m16]), ("orc_perspective_fov_rh_zo", [f32] * 5 + [m16]),
                       ("orc_look_at_rh", [m16, m16, m16, m16]), ("orc_translate", [m16, m16, m16]),
                       ("orc_scale", [m16, m16, m16]), ("orc_rotate", [m16, f32, m16, m16]),
                       ("orc_yaw_pitch_roll", [f32, f32, f32, m16]), ("orc_mat4_mul", [m16, m16, m16])]:
        fn = getattr(lib, name)
        fn.restype = None
        fn.argtypes = args
    _lib = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def mat_call(name, *args):
    """Call one of the orc_* matrix functions; array arguments are float32, result is (4,4) column-major."""
    lib = load()
    out = np.zeros(16, dtype=np.float32)
    conv, keep = [], []
    for a in args:
        if isinstance(a, (float, int, np.floating)):
            conv.append(C.c_float(float(a)))
        else:
            arr = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))
            keep.append(arr)
            conv.append(_fp(arr))
    getattr(lib, name)(*conv, _fp(out))
    return out.reshape(4, 4)


def mip_chain(texels, gen_mips=True, srgb=True):
    """Level list of a texture as the reference builds it (tools.cpp:514, 651-708)."""
    lib = load()
    base = np.ascontiguousarray(texels, dtype=np.uint8)
    h, w = base.shape[:2]
    n = lib.orc_mip_level_count(w, h, 1 if gen_mips else 0)
    levels = [base]
    for l in range(1, n):
        sw, sh = max(w >> (l - 1), 1), max(h >> (l - 1), 1)
        dw, dh = max(w >> l, 1), max(h >> l, 1)
        dst = np.empty((dh, dw, 4), dtype=np.uint8)
        lib.orc_gen_mip(levels[-1].ctypes.data_as(C.c_void_p), sw, sh, dst.ctypes.data_as(C.c_void_p), dw, dh, 1 if srgb else 0)
        levels.append(dst)
    return levels


_SRGB_FORMATS = (15, 22, 29, 43, 50)


def make_texture(tex):
    """OrcTexture (+ the arrays it points at) from a Scene texture dict."""
    srgb = tex.get("fmt", 43) in _SRGB_FORMATS
    if tex.get("levels") is not None:
        levels = [np.ascontiguousarray(l, dtype=np.uint8) for l in tex["levels"]]
    else:
        levels = mip_chain(tex["texels"], tex.get("gen_mips", True), srgb)
    h, w = levels[0].shape[:2]
    t = OrcTexture(w, h, len(levels), 1 if srgb else 0, 1 if tex.get("tiled", False) else 0,
                   1 if tex.get("filtering", 1) == 1 else 0)
    for i, l in enumerate(levels):
        t.level_data[i] = l.ctypes.data_as(C.c_void_p).value
    return t, levels


def render_scene(scene, literal=False, threads=1, want_prim=False):
    """Render a renderer_amd.scenes.Scene on the CPU.  Returns (colour HxWx4 u8,
    depth HxWxS u32, prim HxWxS u32 or None)."""
    lib = load()
    keep = []
    textures = (OrcTexture * max(len(scene.textures), 1))()
    for i, t in enumerate(scene.textures):
        ot, levels = make_texture(t)
        textures[i] = ot
        keep.append(levels)
    vbs = [np.ascontiguousarray(v, dtype=np.float32) for v in scene.vbs]
    ibs = [np.ascontiguousarray(i, dtype=np.uint32) for i in scene.ibs]
    ubos = [np.ascontiguousarray(u, dtype=np.float32) for u in scene.ubos]
    draws = (OrcDraw * max(len(scene.draws), 1))()
    for k, dr in enumerate(scene.draws):
        p = scene.pipelines[dr["pipeline"]]
        d = draws[k]
        d.shader = p["shader"]
        d.stride = p["stride"]
        d.off_pos = p["off_pos"]
        d.off_nrm = -1 if p["off_nrm"] is None else p["off_nrm"]
        d.off_uv = -1 if p["off_uv"] is None else p["off_uv"]
        d.cull_mode, d.front_face = p["cull_mode"], p["front_face"]
        d.depth_test, d.depth_write, d.depth_op = int(p["depth_test"]), int(p["depth_write"]), p["depth_compare"]
        vb, ib = vbs[dr["vb"]], ibs[dr["ib"]]
        vo, io = dr.get("vb_offset", 0), dr.get("ib_offset", 0)
        d.vb = vb.ctypes.data + vo
        d.vb_bytes = vb.nbytes - vo
        d.ib = ib.ctypes.data + io
        d.index_count, d.first_index, d.vertex_offset = dr["index_count"], dr.get("first_index", 0), dr.get("vertex_offset", 0)
        d.ubo = ubos[dr["ubo"]].ctypes.data
        push = np.asarray(dr["push"], dtype=np.float32).reshape(-1)
        for j in range(20):
            d.push[j] = float(push[j])
        tex = dr.get("texture", -1)
        d.texture = -1 if tex is None else tex
    ps = OrcPass()
    ps.width, ps.height, ps.samples = scene.width, scene.height, scene.samples
    ps.color_srgb = 1 if scene.color_format in (43, 50) else 0
    ps.color_bgra = 1 if scene.color_format in (44, 50) else 0
    for j in range(4):
        ps.clear_color[j] = scene.clear_color[j]
        ps.scissor[j] = scene.scissor[j]
        ps.area[j] = scene.area[j]
    ps.clear_depth = scene.clear_depth
    for j in range(6):
        ps.viewport[j] = scene.viewport[j]
    color = np.zeros((scene.height, scene.width, 4), dtype=np.uint8)
    depth = np.zeros((scene.height, scene.width, scene.samples), dtype=np.uint32)
    prim = np.zeros((scene.height, scene.width, scene.samples), dtype=np.uint32) if want_prim else None
    rc = lib.orc_render(C.byref(ps), draws, len(scene.draws), textures, len(scene.textures),
                        color.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p),
                        prim.ctypes.data_as(C.c_void_p) if want_prim else None, 1 if literal else 0, threads)
    if rc != 0:
        raise RuntimeError("orc_render failed: %d" % rc)
    return color, depth, prim
