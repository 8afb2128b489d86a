"""Synthetic scenes standing in for the reference's assets (all of which are
Git-LFS pointer stubs, .gitattributes:1-11).  Every scene is plain numpy data in
the reference's layouts -- interleaved pos3/normal3/uv2 float32 vertices and
uint32 indices local to each sub-geometry (obj_loader.cpp:98-109, 134-141,
173-205) -- plus the draw list the samples would record for it
(test_ktx_app.cpp:695-741).  The same Scene object is consumed by the GPU
runner and by the CPU oracle binding.

All randomness comes from an integer hash of the element index, so scenes are
reproducible bit for bit from their seed on any machine.
"""
from dataclasses import dataclass, field

import numpy as np

from . import hostmath as hm

F = np.float32

SHADER_DUMMY, SHADER_TEST_KTX, SHADER_PUSH_CONSTANTS = 0, 1, 2


def hash_u32(idx, seed):
    """lowbias32-style avalanche of (index, seed) -> uint32, vectorised."""
    with np.errstate(over="ignore"):
        x = (np.asarray(idx, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
        x = x.astype(np.uint32)
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def hash_unit(idx, seed):
    """uniform float32 in [0,1) from the hash."""
    return (hash_u32(idx, seed) >> np.uint32(8)).astype(F) * F(1.0 / 16777216.0)


@dataclass
class Scene:
    width: int
    height: int
    samples: int = 1
    color_format: int = 43  # R8G8B8A8_SRGB, the swapchain preference of vk_app.cpp:197-221
    clear_color: tuple = (0.2, 0.3, 0.7, 1.0)  # test_ktx_app.cpp:650-653
    clear_depth: float = 1.0
    viewport: tuple = None
    scissor: tuple = None
    area: tuple = None
    vbs: list = field(default_factory=list)       # float32 arrays, one row per vertex
    ibs: list = field(default_factory=list)       # uint32 arrays
    ubos: list = field(default_factory=list)      # float32[80]
    textures: list = field(default_factory=list)  # dicts: texels (h,w,4) u8, gen_mips, filtering, tiled, fmt, levels
    pipelines: list = field(default_factory=list)  # dicts of rnd_pipeline_info fields
    draws: list = field(default_factory=list)     # dicts: pipeline, vb, vb_offset, ib, ib_offset, ubo, texture, push, index_count, first_index, vertex_offset
    name: str = ""
    seed: int = 0

    def __post_init__(self):
        if self.viewport is None:
            self.viewport = (0.0, 0.0, float(self.width), float(self.height), 0.0, 1.0)
        if self.scissor is None:
            self.scissor = (0, 0, self.width, self.height)
        if self.area is None:
            self.area = (0, 0, self.width, self.height)

    # ---- statistics the benchmark reports -------------------------------------------------
    def triangle_count(self):
        return sum(d["index_count"] // 3 for d in self.draws)

    def algorithmic_bytes_by_stage(self):
        """Bytes a frame must move at the least, split by the kernel that moves them: every
        distinct vertex once (vertex), every index once plus one 80-byte constant record per
        draw (setup), the depth image and the colour image stored once each plus one texel per
        pixel, or the whole texture if smaller (tile: rasterisation and shading are one kernel)."""
        vb = sum(int(v.nbytes) for v in self.vbs)
        ib = sum(int(i.nbytes) for i in self.ibs)
        px = self.width * self.height * 4
        tex = 0
        if self.textures:
            tb = 0
            for t in self.textures:
                h, w = t["texels"].shape[:2]
                tb += int(w * h * 4 * 4 / 3) if t.get("gen_mips", True) else w * h * 4
            tex = min(px, tb)
        return {"vertex": vb, "setup": ib + 80 * len(self.draws), "tile": 2 * px + tex}

    def algorithmic_bytes(self):
        """Sum of algorithmic_bytes_by_stage(): the least a whole frame moves."""
        return sum(self.algorithmic_bytes_by_stage().values())


def default_pipeline(shader, samples=1, **kw):
    """Fixed-function state of the samples (test_ktx_app.cpp:435-499)."""
    p = dict(shader=shader, stride=32, off_pos=0, off_nrm=12, off_uv=24, cull_mode=2, front_face=0,
             depth_test=True, depth_write=True, depth_compare=3, samples=samples)
    p.update(kw)
    return p


def push_block(color, transform):
    """push_constant_data {vec4 color; mat4 transform} = 80 bytes (test_ktx_app.hpp:16-19)."""
    return np.concatenate([np.asarray(color, dtype=F).reshape(4), np.asarray(transform, dtype=F).reshape(16)])


# ------------------------------------------------------------------------------------- meshes

def uv_sphere(segments=64, rings=32, radius=1.0, flip=False):
    """UV sphere, (segments*(rings-1)*2) triangles; 64x32 -> 4032 triangles."""
    verts = []
    for r in range(rings + 1):
        theta = np.pi * r / rings
        for s in range(segments + 1):
            phi = 2.0 * np.pi * s / segments
            n = np.array([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)])
            verts.append(np.concatenate([n * radius, n, [s / segments, r / rings]]))
    v = np.array(verts, dtype=F)
    idx = []
    for r in range(rings):
        for s in range(segments):
            a = r * (segments + 1) + s
            b = a + segments + 1
            if r != 0:
                idx.append((a, a + 1, b) if flip else (a, b, a + 1))
            if r != rings - 1:
                idx.append((a + 1, b + 1, b) if flip else (a + 1, b, b + 1))
    return v, np.array(idx, dtype=np.uint32).reshape(-1)


def icosphere(subdiv=2, radius=1.0, flip=False):
    """Icosphere; subdiv 2 -> 162 vertices, 320 triangles."""
    t = (1.0 + np.sqrt(5.0)) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
             (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.array(v, dtype=np.float64) / np.linalg.norm(v) for v in verts]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
             (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
             (8, 6, 7), (9, 8, 1)]
    for _ in range(subdiv):
        cache = {}
        new_faces = []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            new_faces += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = new_faces
    n = np.array(verts)
    uv = np.stack([np.arctan2(n[:, 2], n[:, 0]) / (2 * np.pi) + 0.5, np.arccos(np.clip(n[:, 1], -1, 1)) / np.pi], axis=1)
    v = np.concatenate([n * radius, n, uv], axis=1).astype(F)
    idx = np.array(faces, dtype=np.uint32)
    if flip:
        idx = idx[:, [0, 2, 1]]
    return v, idx.reshape(-1)


def value_noise(x, y, seed, octaves=4):
    """Sum of bilinear value-noise octaves on hashed lattice values, in [0,1)."""
    out = np.zeros(np.broadcast(x, y).shape, dtype=np.float64)
    amp, freq, norm = 1.0, 1.0, 0.0
    for o in range(octaves):
        fx, fy = x * freq, y * freq
        ix, iy = np.floor(fx).astype(np.int64), np.floor(fy).astype(np.int64)
        tx, ty = fx - ix, fy - iy
        tx, ty = tx * tx * (3 - 2 * tx), ty * ty * (3 - 2 * ty)

        def lat(i, j):
            return hash_unit((i & 0xFFFF) * 65537 + (j & 0xFFFF) + o * 0x1000193, seed).astype(np.float64)

        v = (lat(ix, iy) * (1 - tx) + lat(ix + 1, iy) * tx) * (1 - ty) + (lat(ix, iy + 1) * (1 - tx) + lat(ix + 1, iy + 1) * tx) * ty
        out += v * amp
        norm += amp
        amp *= 0.5
        freq *= 2.0
    return out / norm


def noise_grid(quads, x_half, y_half, z_amp, seed, flip=False, y_center=1.0):
    """(quads+1)^2 vertices on [-x_half,x_half]x[y_center-y_half,y_center+y_half], z displaced
    by value noise, normals from central differences; 2*quads^2 triangles in row-major order.
    y_center defaults to 1: the reference's vertex shaders set clip.y = 1 - clip.y before the
    divide (test_ktx.vert:29), which puts world height y=1 on the middle row of the screen."""
    n = quads + 1
    gx, gy = np.meshgrid(np.arange(n), np.arange(n), indexing="xy")
    u, v = gx / quads, gy / quads
    height = value_noise(u * 12.0, v * 12.0, seed) * z_amp
    px, py = (u * 2 - 1) * x_half, (v * 2 - 1) * y_half + y_center
    dzdx = np.gradient(height, axis=1) / (2 * x_half / quads)
    dzdy = np.gradient(height, axis=0) / (2 * y_half / quads)
    nrm = np.stack([-dzdx, -dzdy, -np.ones_like(height)], axis=-1)  # facing -z, towards the default camera
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    verts = np.concatenate([px[..., None], py[..., None], height[..., None], nrm, u[..., None], v[..., None]], axis=-1)
    verts = verts.reshape(-1, 8).astype(F)
    qx, qy = np.meshgrid(np.arange(quads), np.arange(quads), indexing="xy")
    a = (qy * n + qx).reshape(-1).astype(np.uint32)
    b, c, d = a + 1, a + n, a + n + 1
    if flip:
        tri = np.stack([a, b, c, b, d, c], axis=1)
    else:
        tri = np.stack([a, c, b, b, c, d], axis=1)
    return verts, tri.reshape(-1).astype(np.uint32)


# ------------------------------------------------------------------------------------- textures

def checker_texture(size=1024, seed=0):
    """Checker x gradient with hashed per-texel noise, RGBA8 (sRGB-encoded bytes)."""
    y, x = np.mgrid[0:size, 0:size]
    check = (((x // (size // 16)) + (y // (size // 16))) & 1).astype(np.float64)
    grad_u, grad_v = x / (size - 1), y / (size - 1)
    noise = hash_unit(y * size + x, seed).astype(np.float64)
    r = 0.25 + 0.6 * check * grad_u + 0.15 * noise
    g = 0.25 + 0.6 * (1 - check) * grad_v + 0.15 * noise
    b = 0.35 + 0.4 * (1 - grad_u) * grad_v + 0.15 * noise
    rgba = np.stack([r, g, b, np.ones_like(r)], axis=-1)
    return np.clip(rgba * 255.0 + 0.5, 0, 255).astype(np.uint8)


def _texture(texels, gen_mips=True, filtering=1, tiled=False, fmt=43):
    return dict(texels=texels, gen_mips=gen_mips, filtering=filtering, tiled=tiled, fmt=fmt)


# ------------------------------------------------------------------------------------- scenes

def _ubo_for(width, height, model_transform, camera=None):
    cam = camera or hm.Camera()
    return hm.global_ubo(cam, model_transform, width, height)


def config0(width=1280, height=720, shader=SHADER_TEST_KTX, samples=1, tex_size=1024, segments=64, rings=32):
    """The reference's simplest scene: a lit textured mesh drawn as the ten-instance ring
    of test_ktx_app.cpp:708-734.  A UV sphere of radius 4 stands in for sphere.obj and
    model_transform = scale(0.12) for the loader's normalisation (obj_loader.cpp:294-307)."""
    sc = Scene(width, height, samples=samples, name="config0_ring", seed=0)
    if shader == SHADER_DUMMY:
        sc.clear_color = (0.0, 0.0, 0.0, 1.0)  # dummy_obj_viewer_app.cpp:421-425
    v, i = uv_sphere(segments, rings, radius=4.0)
    sc.vbs.append(v)
    sc.ibs.append(i)
    sc.textures.append(_texture(checker_texture(tex_size, 0)))
    model = hm.scale(hm.identity(), [0.12, 0.12, 0.12])
    sc.ubos.append(_ubo_for(width, height, model))
    sc.pipelines.append(default_pipeline(shader, samples))
    for j in range(10):
        sc.draws.append(dict(pipeline=0, vb=0, vb_offset=0, ib=0, ib_offset=0, ubo=0, texture=0,
                             push=push_block(hm.RING_COLORS[j], hm.ring_transform(j)),
                             index_count=int(i.size), first_index=0, vertex_offset=0))
    return sc


def config1(width=1920, height=1080, quads=708, seed=1, tex_size=1024):
    """1080p forward shading of one ~1M-triangle draw: a 708x708-quad grid displaced by
    value noise (1 002 528 triangles, 502 681 vertices) covering about 70% of the screen."""
    sc = Scene(width, height, name="config1_grid%d" % quads, seed=seed)
    v, i = noise_grid(quads, 3.3, 1.8, 0.35, seed)
    sc.vbs.append(v)
    sc.ibs.append(i)
    sc.textures.append(_texture(checker_texture(tex_size, seed)))
    sc.ubos.append(_ubo_for(width, height, hm.identity()))
    sc.pipelines.append(default_pipeline(SHADER_TEST_KTX))
    sc.draws.append(dict(pipeline=0, vb=0, vb_offset=0, ib=0, ib_offset=0, ubo=0, texture=0,
                         push=push_block([1, 1, 1, 1], hm.identity()),
                         index_count=int(i.size), first_index=0, vertex_offset=0))
    return sc


def instanced_spheres(width, height, grid_x, grid_y, layers, seed, back_to_front=True, shader=SHADER_TEST_KTX,
                      tex_size=1024, name="instanced"):
    """layers x grid_x x grid_y icosphere instances (162 vertices / 320 triangles each),
    one draw with its own 80-byte push-constant block per instance, about `layers`-fold
    depth overdraw."""
    sc = Scene(width, height, name=name, seed=seed)
    v, i = icosphere(2, radius=1.0)
    sc.vbs.append(v)
    sc.ibs.append(i)
    sc.textures.append(_texture(checker_texture(tex_size, seed)))
    sc.ubos.append(_ubo_for(width, height, hm.identity()))
    sc.pipelines.append(default_pipeline(shader))
    aspect = width / height
    order = range(layers - 1, -1, -1) if back_to_front else range(layers)
    k = 0
    for layer in order:
        z = 0.4 * layer  # further from the camera (eye at z=-2 looking at +z) as layer grows
        dist = 2.0 + z
        half_y, half_x = dist * 0.98, dist * aspect * 0.98
        cell_x, cell_y = 2 * half_x / grid_x, 2 * half_y / grid_y
        rad = 0.62 * min(cell_x, cell_y)
        for gy in range(grid_y):
            for gx in range(grid_x):
                jx = (float(hash_unit(k, seed)) - 0.5) * 0.3 * cell_x
                jy = (float(hash_unit(k, seed + 101)) - 0.5) * 0.3 * cell_y
                cx = -half_x + (gx + 0.5) * cell_x + jx
                cy = 1.0 - half_y + (gy + 0.5) * cell_y + jy  # y=1 is the screen's middle row (clip.y = 1 - clip.y)
                m = hm.translate(hm.identity(), [cx, cy, z])
                m = hm.scale(m, [rad, rad, rad])
                col = [float(hash_unit(k, seed + 7)), float(hash_unit(k, seed + 8)), float(hash_unit(k, seed + 9)), 1.0]
                sc.draws.append(dict(pipeline=0, vb=0, vb_offset=0, ib=0, ib_offset=0, ubo=0, texture=0,
                                     push=push_block(col, m), index_count=int(i.size), first_index=0, vertex_offset=0))
                k += 1
    return sc


def config2(width=3840, height=2160, back_to_front=True, seed=2):
    """4K, 10 000 instanced meshes in 4 depth layers of 50x50 (3.2 M triangles, ~4x overdraw)."""
    return instanced_spheres(width, height, 50, 50, 4, seed, back_to_front, name="config2_instances")


def config3(width=7680, height=4320, seed=3):
    """8K, 40 000 instances (4 layers of 100x100); the frame whose tiles are sharded over GPUs."""
    return instanced_spheres(width, height, 100, 100, 4, seed, True, name="config3_8k")


def config4_camera(k, n_views=64):
    """Camera k of the orbit: eye_k = Ry(2 pi k / n_views) * (0, 0, -2), looking at the origin.  A full turn:
    view 0 faces the grid, views near n/4 and 3n/4 look along it from inside its extent (everything
    straddles the near plane or is a sliver), views between them see its back (all culled)."""
    cam = hm.Camera()
    ang = 2.0 * np.pi * k / n_views
    cam.eye_position = np.array([-2.0 * np.sin(ang), 0.0, -2.0 * np.cos(ang)], dtype=F)
    return cam


def config4_view(k, n_views=64, width=1920, height=1080, quads=2237, seed=4, mesh=None):
    """View k of 64 cameras on a y-axis orbit around a 2237x2237-quad grid
    (10 008 338 triangles, 5 008 644 vertices).  Pass `mesh` to reuse the arrays."""
    sc = Scene(width, height, name="config4_view%d" % k, seed=seed)
    v, i = mesh if mesh is not None else noise_grid(quads, 3.3, 1.8, 0.35, seed)
    sc.vbs.append(v)
    sc.ibs.append(i)
    sc.textures.append(_texture(checker_texture(1024, seed)))
    sc.ubos.append(_ubo_for(width, height, hm.identity(), config4_camera(k, n_views)))
    sc.pipelines.append(default_pipeline(SHADER_TEST_KTX))
    sc.draws.append(dict(pipeline=0, vb=0, vb_offset=0, ib=0, ib_offset=0, ubo=0, texture=0,
                         push=push_block([1, 1, 1, 1], hm.identity()),
                         index_count=int(i.size), first_index=0, vertex_offset=0))
    return sc
