"""GPU parity on directed and randomised edge cases: clipping, guard band, large primitives,
ties in depth, cull state, viewport/scissor/area rectangles, odd target sizes, 4x
multisampling, sampler modes, target formats, uniform re-read, tile sharding."""
import os

import numpy as np
import pytest

import helpers
from renderer_amd import api, hostmath as hm, scenes
from renderer_amd.runner import SceneRunner

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu_render(scene, device):
    r = SceneRunner(scene, device)
    try:
        return r.render()
    finally:
        r.destroy()


def check(scene, device, oracle, exact_color=True, threads=8):
    color, depth = gpu_render(scene, device)
    ocolor, odepth, _ = oracle.render_scene(scene, threads=threads)
    assert np.array_equal(depth, odepth), "depth words differ at %d samples" % int((depth != odepth).sum())
    diff = np.abs(color.astype(np.int32) - ocolor.astype(np.int32))
    assert diff.max() <= 1, "colour differs by %d at %d channels" % (int(diff.max()), int((diff > 1).sum()))
    if exact_color:
        assert diff.max() == 0, "%d channels off by one" % int((diff != 0).sum())
    return color, depth


def tex(size=64, seed=1, **kw):
    d = dict(texels=scenes.checker_texture(size, seed), gen_mips=True, filtering=1, tiled=False, fmt=43)
    d.update(kw)
    return d


@pytest.mark.parametrize("seed", range(4))
def test_soup_random_depth(device, oracle, seed):
    check(helpers.soup_scene(257, 131, 1500, seed, n_draws=4), device, oracle)


@pytest.mark.parametrize("seed", range(4))
def test_soup_depth_ties(device, oracle, seed):
    """Flat triangles on a few depth levels: equal-depth fragments decide by submission order."""
    check(helpers.soup_scene(200, 120, 1200, 100 + seed, depth_levels=3, n_draws=5), device, oracle)


def test_soup_depth_ties_less(device, oracle):
    check(helpers.soup_scene(200, 120, 1200, 7, depth_levels=3, n_draws=5, pipeline_kw=dict(depth_compare=1)), device, oracle)


def test_soup_near_far_planes(device, oracle):
    """z from -0.4 to 1.4: primitives cross z=0 and z=w and are clipped."""
    check(helpers.soup_scene(192, 128, 800, 21, z_range=(-0.4, 1.4), max_extent=1.0), device, oracle)


@pytest.mark.parametrize("seed", range(3))
def test_perspective_soup_clipping(device, oracle, seed):
    """Triangles all around the eye: near plane, far plane, w <= 0 and the guard band."""
    sc = helpers.perspective_soup(320, 200, 600, 40 + seed, near_cross=True, shader=scenes.SHADER_TEST_KTX, texture=tex(),
                                  pipeline_kw=dict(cull_mode=0))
    check(sc, device, oracle)


def test_perspective_soup_textured_lod(device, oracle):
    sc = helpers.perspective_soup(400, 300, 500, 50, near_cross=False, shader=scenes.SHADER_TEST_KTX, texture=tex(256),
                                  pipeline_kw=dict(cull_mode=0))
    check(sc, device, oracle)


def test_huge_triangles(device, oracle):
    """Primitives far larger than the screen (through the guard band) and exactly screen sized."""
    W, H = 300, 170
    tris = [[(-4000.0, -3000.0, 0.9), (-4000.0, 5000.0, 0.9), (6000.0, 100.0, 0.8)],
            [(0.0, 0.0, 0.5), (0.0, float(H), 0.5), (float(W), 0.0, 0.5)],
            [(float(W), float(H), 0.4), (float(W), 0.0, 0.4), (0.0, float(H), 0.4)],
            [(10.0, 10.0, 0.3), (10.0, 150.0, 0.2), (250.0, 80.0, 0.1)]]
    check(helpers.window_triangles(W, H, tris, pipeline_kw=dict(cull_mode=0)), device, oracle)


def test_fill_rule_ties_on_gpu(device, oracle):
    """Quarter-pixel corners put many samples exactly on edges."""
    rng = np.random.RandomState(3)
    W = H = 48
    tris = []
    for _ in range(150):
        pts = rng.randint(0, 4 * W + 1, size=(3, 2)) / 4.0
        z = rng.randint(1, 8) / 8.0
        tris.append([(float(x), float(y), z) for x, y in pts])
    check(helpers.window_triangles(W, H, tris, pipeline_kw=dict(cull_mode=0)), device, oracle)


@pytest.mark.parametrize("cull,ff", [(0, 0), (1, 0), (2, 0), (2, 1), (3, 0)])
def test_cull_state(device, oracle, cull, ff):
    check(helpers.soup_scene(128, 96, 400, 5, pipeline_kw=dict(cull_mode=cull, front_face=ff)), device, oracle)


def test_viewport_scissor_area(device, oracle):
    sc = helpers.soup_scene(160, 120, 600, 9)
    sc.viewport = (20.0, 10.0, 100.0, 90.0, 0.1, 0.9)
    sc.scissor = (30, 5, 80, 70)
    sc.area = (8, 8, 140, 100)
    color, depth = check(sc, device, oracle)
    assert (depth[0:8] == 0).all() and (color[0:8] == 0).all()  # outside the render area nothing is stored
    assert (depth[8, 8:148, 0] == oracle.load().orc_depth24(1.0)).all()


@pytest.mark.parametrize("wh", [(1, 1), (33, 31), (101, 67), (640, 3)])
def test_odd_target_sizes(device, oracle, wh):
    check(helpers.soup_scene(wh[0], wh[1], 300, 13), device, oracle)


def test_empty_pass_and_empty_draw(device, oracle):
    sc = scenes.Scene(70, 40)
    sc.clear_color = (0.25, 0.5, 0.75, 1.0)
    color, depth = check(sc, device, oracle)
    assert (depth == 0xFFFFFF).all()
    sc = helpers.soup_scene(70, 40, 10, 1)
    sc.draws[0]["index_count"] = 2  # fewer than one triangle
    check(sc, device, oracle)


def test_index_out_of_range_drops_primitive(device, oracle):
    sc = helpers.soup_scene(64, 64, 40, 2)
    ib = sc.ibs[0].copy()
    ib[4] = 10 ** 6
    ib[30] = 0xFFFFFFFF
    sc.ibs[0] = ib
    check(sc, device, oracle)


def test_sub_geometries_share_buffers(device, oracle):
    """The reference's layout: one vertex buffer bound at a byte offset per sub-geometry,
    indices local to it, index buffer bound at indices_offset*4 (test_ktx_app.cpp:728-732)."""
    a = helpers.soup_scene(128, 96, 100, 31)
    b = helpers.soup_scene(128, 96, 150, 32)
    sc = scenes.Scene(128, 96)
    sc.vbs.append(np.concatenate([a.vbs[0], b.vbs[0]]))
    sc.ibs.append(np.concatenate([a.ibs[0], b.ibs[0]]))
    sc.ubos.append(a.ubos[0])
    sc.pipelines.append(scenes.default_pipeline(scenes.SHADER_PUSH_CONSTANTS))
    sc.draws.append(dict(pipeline=0, vb=0, vb_offset=0, ib=0, ib_offset=0, ubo=0, texture=-1,
                         push=a.draws[0]["push"], index_count=300, first_index=0, vertex_offset=0))
    sc.draws.append(dict(pipeline=0, vb=0, vb_offset=int(a.vbs[0].nbytes), ib=0, ib_offset=300 * 4, ubo=0, texture=-1,
                         push=b.draws[0]["push"], index_count=450, first_index=0, vertex_offset=0))
    sc.draws.append(dict(pipeline=0, vb=0, vb_offset=0, ib=0, ib_offset=0, ubo=0, texture=-1,
                         push=b.draws[0]["push"], index_count=150, first_index=300, vertex_offset=300))
    check(sc, device, oracle)


def test_vertex_formats(device, oracle):
    """Position-only (stride 12) and position+uv (stride 20) layouts (obj_loader.cpp:98-109)."""
    base = helpers.soup_scene(96, 64, 200, 17, shader=scenes.SHADER_DUMMY, texture=tex())
    v = base.vbs[0]
    for cols, stride, on, ou in (([0, 1, 2], 12, None, None), ([0, 1, 2, 6, 7], 20, None, 12)):
        sc = helpers.soup_scene(96, 64, 200, 17, shader=scenes.SHADER_DUMMY, texture=tex())
        sc.vbs[0] = np.ascontiguousarray(v[:, cols])
        sc.pipelines[0].update(stride=stride, off_nrm=on, off_uv=ou)
        check(sc, device, oracle)


@pytest.mark.parametrize("shader", [scenes.SHADER_DUMMY, scenes.SHADER_TEST_KTX, scenes.SHADER_PUSH_CONSTANTS])
def test_msaa4(device, oracle, shader):
    """4x multisampling + resolve, the state of the third sample (dummy_obj_viewer_app.cpp:314-323, 468-520)."""
    sc = scenes.config0(200, 120, shader=shader, samples=4, tex_size=64, segments=24, rings=12)
    check(sc, device, oracle)


def test_msaa4_soup_with_clipping(device, oracle):
    sc = helpers.perspective_soup(160, 100, 300, 61, near_cross=True, samples=4, pipeline_kw=dict(cull_mode=0))
    check(sc, device, oracle)
    check(helpers.soup_scene(120, 80, 500, 62, depth_levels=3, samples=4, n_draws=3), device, oracle)


@pytest.mark.parametrize("kw", [dict(filtering=0), dict(tiled=True), dict(gen_mips=False), dict(fmt=37),
                                dict(filtering=0, tiled=True, gen_mips=True)])
def test_sampler_modes(device, oracle, kw):
    sc = helpers.perspective_soup(200, 150, 200, 70, near_cross=False, shader=scenes.SHADER_DUMMY, texture=tex(128, 2, **kw),
                                  pipeline_kw=dict(cull_mode=0))
    sc.vbs[0][:, 6:8] = sc.vbs[0][:, 6:8] * 3.0 - 1.0  # uv outside [0,1]: clamp vs repeat
    check(sc, device, oracle)


def test_npot_texture_and_mip_chain(device, oracle):
    for shape in [(100, 60), (64, 64), (1, 37), (130, 130)]:
        t = scenes.checker_texture(max(shape), 4)[:shape[1], :shape[0]].copy()
        for fmt in (43, 37):
            gtex = api.Texture(device, t, fmt=fmt, gen_mips=True)
            levels = oracle.mip_chain(t, True, srgb=(fmt == 43))
            assert gtex.level_count == len(levels)
            for l, ref in enumerate(levels):
                assert np.array_equal(gtex.read_level(l), ref), "level %d of %s fmt %d" % (l, shape, fmt)
            gtex.destroy()


@pytest.mark.parametrize("shape", [(16384, 2), (3, 16384)])
def test_texture_of_the_largest_extent(device, oracle, shape):
    """A side of 16 384 texels, the limit.  Generated mips: the reference's count, one short of a full chain (tools.cpp:514),
    so 14 levels, equal to the oracle's level by level.  Given levels: the full chain of 15, the most a texture holds.
    Either way a frame that samples it (minified by thousands along the long side, so the last levels are read) is the
    oracle's."""
    rng = np.random.RandomState(9)
    t = rng.randint(0, 256, size=(shape[1], shape[0], 4)).astype(np.uint8)
    gtex = api.Texture(device, t, fmt=43, gen_mips=True)
    levels = oracle.mip_chain(t, True, srgb=True)
    assert gtex.level_count == len(levels) == 14
    for l, ref in enumerate(levels):
        assert np.array_equal(gtex.read_level(l), ref), "level %d" % l
    gtex.destroy()
    full = list(levels) + [rng.randint(0, 256, size=(1, 1, 4)).astype(np.uint8)]
    for tex_dict in (dict(texels=t, gen_mips=True, filtering=1, tiled=False, fmt=43),
                     dict(texels=t, levels=full, gen_mips=False, filtering=1, tiled=False, fmt=43)):
        sc = helpers.perspective_soup(160, 120, 120, 31, near_cross=False, shader=scenes.SHADER_TEST_KTX, texture=tex_dict,
                                      pipeline_kw=dict(cull_mode=0))
        check(sc, device, oracle)
    with pytest.raises(api.RendererError):
        api.Texture(device, t, fmt=43, levels=full + [full[-1]])  # sixteen levels


def test_texture_channel_expansion(device, oracle):
    """1-, 2- and 3-channel uploads take the reference's view swizzles (tools.cpp:474-509)."""
    rgb = scenes.checker_texture(32, 5)[..., :3].copy()
    g = api.Texture(device, rgb, fmt=api.FORMAT_R8G8B8_SRGB, gen_mips=False)
    out = g.read_level(0)
    assert np.array_equal(out[..., :3], rgb) and (out[..., 3] == 255).all()
    g.destroy()
    r = rgb[..., 0].copy()
    g = api.Texture(device, r, fmt=api.FORMAT_R8_SRGB, gen_mips=False)
    out = g.read_level(0)
    assert np.array_equal(out[..., 0], r) and np.array_equal(out[..., 1], r) and np.array_equal(out[..., 2], r)
    g.destroy()


@pytest.mark.parametrize("fmt", [37, 43, 44, 50])
def test_target_formats(device, oracle, fmt):
    sc = scenes.config0(160, 90, shader=scenes.SHADER_PUSH_CONSTANTS, tex_size=32, segments=16, rings=8)
    sc.color_format = fmt
    check(sc, device, oracle)


def test_uniforms_are_read_at_execution(device, oracle):
    """The command buffer is recorded once; writing the uniform buffer between two submits
    changes the second frame (base_obj_viewer_app.cpp:134-138, vk_app.cpp:463-467)."""
    sc = scenes.config0(256, 144, tex_size=64, segments=24, rings=12)
    r = SceneRunner(sc, device)
    try:
        c1, d1 = r.render()
        cam = hm.Camera()
        cam.eye_position = np.array([0.4, 0.3, -2.2], dtype=np.float32)
        ubo2 = hm.global_ubo(cam, sc.ubos[0][48:64].reshape(4, 4), sc.width, sc.height)
        r.set_ubo(0, ubo2)
        c2, d2 = r.render()
    finally:
        r.destroy()
    assert not np.array_equal(d1, d2)
    sc.ubos[0] = ubo2
    oc, od, _ = oracle.render_scene(sc, threads=4)
    assert np.array_equal(d2, od) and np.array_equal(c2, oc)


@pytest.mark.parametrize("images", [2, 3])
def test_frames_in_flight_keep_their_uniforms(device, oracle, images):
    """Frames rotate through `images` command buffers and targets without the host waiting
    (vk_app.cpp:109-130: one fence per frame in flight).  Each frame must be drawn with the
    uniform block written just before its submit, whatever overlaps on the GPU."""
    sc = scenes.config0(320, 180, tex_size=64, segments=48, rings=24)
    model = np.asarray(sc.ubos[0][48:64], dtype=np.float32).reshape(4, 4)
    n_frames = 3 * images + 1
    ubos = []
    for i in range(n_frames):
        cam = hm.Camera()
        cam.eye_position = np.array([0.25 * np.sin(0.9 * i), 0.1 * i, -2.0 - 0.05 * i], dtype=np.float32)
        ubos.append(hm.global_ubo(cam, model, sc.width, sc.height))
    r = SceneRunner(sc, device, frames_in_flight=images)
    try:
        owner = {}
        for i in range(n_frames):
            r.set_ubo(0, ubos[i])
            owner[r.submit()] = i
        device.wait_idle()
        got = {img: r.targets[img].read() for img in owner}
    finally:
        r.destroy()
    for img, frame in owner.items():
        sc.ubos[0] = ubos[frame]
        oc, od, _ = oracle.render_scene(sc, threads=4)
        assert np.array_equal(got[img][1], od), "image %d (frame %d): depth differs" % (img, frame)
        assert np.array_equal(got[img][0], oc), "image %d (frame %d): colour differs" % (img, frame)


@pytest.mark.parametrize("seed", range(16))
def test_seeded_scene_sweep(device, oracle, seed):
    """A sweep over the knobs together: target size (odd sizes, partial tiles), triangle count and
    extent (pixel-sized to screen-filling), depth range across both planes, exact depth ties, draw
    count, shader, texture, cull state, compare op, sample count."""
    rng = np.random.RandomState(1000 + seed)
    w, h = int(rng.randint(33, 400)), int(rng.randint(17, 300))
    shader = [scenes.SHADER_PUSH_CONSTANTS, scenes.SHADER_TEST_KTX, scenes.SHADER_DUMMY][seed % 3]
    texture = None if shader == scenes.SHADER_PUSH_CONSTANTS else tex(int(rng.choice([16, 40, 128])), seed, tiled=bool(seed & 1),
                                                                       filtering=int(rng.randint(0, 2)))
    kw = dict(cull_mode=int(rng.randint(0, 4)), front_face=int(rng.randint(0, 2)), depth_compare=int(rng.choice([1, 3])))
    sc = helpers.soup_scene(w, h, int(rng.choice([40, 700, 6000])), seed, max_extent=float(rng.choice([0.02, 0.3, 2.5])),
                            depth_levels=int(rng.choice([0, 0, 3])), z_range=[(0.05, 0.95), (-0.3, 1.3)][seed & 1],
                            shader=shader, n_draws=int(rng.randint(1, 9)), texture=texture, samples=int(rng.choice([1, 4])),
                            pipeline_kw=kw)
    check(sc, device, oracle)


@pytest.mark.parametrize("samples", [1, 4])
def test_mixed_state_pass(device, oracle, samples):
    """One pass whose draws differ in shader, texture, sampler, cull state and uniform buffer: the
    per-draw path of shading (nothing is uniform over the pass), with clipping and depth ties."""
    sc = helpers.soup_scene(224, 160, 1800, 31, n_draws=12, depth_levels=4, z_range=(-0.2, 1.2), samples=samples)
    rng = np.random.RandomState(5)
    sc.textures.append(tex(64, 3))
    sc.textures.append(dict(texels=scenes.checker_texture(48, 9), gen_mips=False, filtering=0, tiled=True, fmt=37))
    sc.pipelines = [scenes.default_pipeline(scenes.SHADER_PUSH_CONSTANTS, samples, cull_mode=0),
                    scenes.default_pipeline(scenes.SHADER_TEST_KTX, samples, cull_mode=2),
                    scenes.default_pipeline(scenes.SHADER_DUMMY, samples, cull_mode=1),
                    scenes.default_pipeline(scenes.SHADER_TEST_KTX, samples, cull_mode=0, front_face=1)]
    second = np.array(sc.ubos[0], dtype=np.float32).copy()
    second[64:80] = hm.scale(hm.identity(), [0.8, 1.1, 1.0]).reshape(16)  # another mvp for some draws
    sc.ubos.append(second)
    for k, d in enumerate(sc.draws):
        d["pipeline"] = k % 4
        d["texture"] = [-1, 0, 1, 1][k % 4] if k % 8 < 4 else [-1, 1, 0, 0][k % 4]
        d["ubo"] = k % 2
        m = hm.translate(hm.identity(), [0.05 * rng.randn(), 0.05 * rng.randn(), 0.0])
        d["push"] = scenes.push_block([rng.rand(), rng.rand(), rng.rand(), 1.0], m)
    check(sc, device, oracle)


def test_uniform_write_paths(device, oracle):
    """The small uniform write is held back and travels with the submits; everything else that
    touches the buffer must still see it: a read-back, a write to another range (which stores the
    held-back one first), a write too large to be held back, a device-side copy -- with frames in
    flight all the while."""
    sc = scenes.config0(256, 144, tex_size=64, segments=24, rings=12)
    model = np.asarray(sc.ubos[0][48:64], dtype=np.float32).reshape(4, 4)

    def ubo_for(i):
        cam = hm.Camera()
        cam.eye_position = np.array([0.3 * np.cos(i), 0.2 * np.sin(i), -2.0 - 0.1 * i], dtype=np.float32)
        return hm.global_ubo(cam, model, sc.width, sc.height)

    r = SceneRunner(sc, device, frames_in_flight=2)
    spare = api.Buffer(device, api.BUFFER_USAGE_UNIFORM, nbytes=2048)
    try:
        r.set_ubo(0, ubo_for(1))                      # held back
        assert np.array_equal(r.ubos[0].read().view(np.float32), ubo_for(1))  # read-back sees it
        r.submit()
        r.set_ubo(0, ubo_for(2))                      # replaces the held-back bytes
        spare.write(np.arange(80, dtype=np.float32))  # another range: the first is stored to memory
        r.submit()                                    # no held-back write for the frame's buffer: reads memory
        spare.write(np.zeros(256, dtype=np.float32))  # 1 KiB: too large to hold back
        r.submit()
        big = np.concatenate([ubo_for(3), np.zeros(176, np.float32)])
        r.ubos[0].write(big[:80])                     # held back again
        assert np.array_equal(spare.read(1024).view(np.float32), np.zeros(256, np.float32))
        last = r.submit()
        device.wait_idle()
        color, depth = r.targets[last].read()
        assert np.array_equal(r.ubos[0].read().view(np.float32), ubo_for(3))
    finally:
        spare.destroy()
        r.destroy()
    sc.ubos[0] = ubo_for(3)
    oc, od, _ = oracle.render_scene(sc, threads=4)
    assert np.array_equal(depth, od) and np.array_equal(color, oc)


def test_two_command_buffers_share_a_target(device, oracle):
    """Frames of different command buffers that render into one target run one after the other."""
    a = helpers.soup_scene(160, 96, 900, 5, n_draws=3)
    b = helpers.soup_scene(160, 96, 900, 6, n_draws=3)
    ra, rb = SceneRunner(a, device), SceneRunner(b, device)
    try:
        rb.record(rb.cmd, ra.target)  # both draw into ra's target; b is submitted last
        for _ in range(4):
            ra.submit()
            device.submit(rb.cmd)
        device.wait_idle()
        color, depth = ra.target.read()
    finally:
        rb.destroy()
        ra.destroy()
    oc, od, _ = oracle.render_scene(b, threads=4)
    assert np.array_equal(depth, od) and np.array_equal(color, oc)


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_tile_shards_compose_the_frame(device, oracle, shards):
    """Sort-first sharding: shard k draws only tiles t with t % n == k; together they are the frame."""
    sc = scenes.instanced_spheres(300, 200, 6, 4, 3, 5, tex_size=64)
    full_c, full_d = gpu_render(sc, device)
    acc_c, acc_d = np.zeros_like(full_c), np.zeros_like(full_d)
    tiles_x = (sc.width + 31) // 32
    for k in range(shards):
        r = SceneRunner(sc, device)
        try:
            r.cmd.set_tile_shard(k, shards)
            c, d = r.render()
        finally:
            r.destroy()
        for ty in range((sc.height + 31) // 32):
            for tx in range(tiles_x):
                sl = (slice(ty * 32, ty * 32 + 32), slice(tx * 32, tx * 32 + 32))
                if (ty * tiles_x + tx) % shards == k:
                    acc_c[sl], acc_d[sl] = c[sl], d[sl]
                else:
                    assert (c[sl] == 0).all()  # a shard leaves other shards' tiles untouched
    assert np.array_equal(acc_c, full_c) and np.array_equal(acc_d, full_d)


@pytest.mark.parametrize("case", [
    dict(n=9000, extent=0.012),                                   # one or two pixels each: nearly all of them fragment entries
    dict(n=9000, extent=0.03, depth_levels=3),                    # up to a dozen pixels: second entries and ordinary ones mixed, exact depth ties
    dict(n=6000, extent=0.02, depth_levels=2, compare=1),         # LESS: the first of equal depths wins
    dict(n=6000, extent=0.02, depth_test=False),                  # no depth test: the last submitted wins everywhere
    dict(n=6000, extent=0.02, n_draws=5, z_range=(-0.2, 1.2)),    # several draws (primitive records), some clipped
    dict(n=6000, extent=0.02, scissor=(37, 21, 90, 70)),          # boxes cut by the scissor
    dict(n=4000, extent=0.015, mix=400),                          # a few large triangles among the small ones in every wave
    dict(n=6000, extent=0.02, shader=scenes.SHADER_TEST_KTX),     # textured: shading of fragment-entry pixels
])
def test_meshes_of_tiny_triangles(device, oracle, case):
    """Triangles of a few pixels are rasterised by setup and travel as fragments in the tile lists: coverage,
    depth, ties and shading must come out as if the tile kernel had walked them, across tile borders too."""
    shader = case.get("shader", scenes.SHADER_PUSH_CONSTANTS)
    kw = dict(depth_compare=case.get("compare", 3), cull_mode=0)
    if not case.get("depth_test", True):
        kw.update(depth_test=False, depth_write=False)
    sc = helpers.soup_scene(200, 136, case["n"], 41, max_extent=case["extent"], depth_levels=case.get("depth_levels", 0),
                            z_range=case.get("z_range", (0.05, 0.95)), n_draws=case.get("n_draws", 1), shader=shader,
                            texture=tex(64, 2) if shader == scenes.SHADER_TEST_KTX else None, pipeline_kw=kw)
    if "mix" in case:  # every tenth triangle of the first `mix` blown up
        v = sc.vbs[0].reshape(-1, 3, 8)
        for t in range(0, case["mix"], 10):
            c = v[t, :, :3].mean(axis=0)
            v[t, :, :2] = c[:2] + (v[t, :, :2] - c[:2]) * 25.0
    if "scissor" in case:
        sc.scissor = case["scissor"]
    check(sc, device, oracle)


def test_tiny_triangles_under_tile_shards(device):
    """Fragment entries are binned like any other: sharded by tiles, the shards still compose the frame."""
    sc = helpers.soup_scene(200, 136, 8000, 43, max_extent=0.02, depth_levels=2, pipeline_kw=dict(cull_mode=0))
    full_c, full_d = gpu_render(sc, device)
    acc_c, acc_d = np.zeros_like(full_c), np.zeros_like(full_d)
    tiles_x = (sc.width + 31) // 32
    for k in range(3):
        r = SceneRunner(sc, device)
        try:
            r.cmd.set_tile_shard(k, 3)
            c, d = r.render()
        finally:
            r.destroy()
        for ty in range((sc.height + 31) // 32):
            for tx in range(tiles_x):
                if (ty * tiles_x + tx) % 3 == k:
                    sl = (slice(ty * 32, ty * 32 + 32), slice(tx * 32, tx * 32 + 32))
                    acc_c[sl], acc_d[sl] = c[sl], d[sl]
    assert np.array_equal(acc_c, full_c) and np.array_equal(acc_d, full_d)


@pytest.mark.parametrize("bands", [2, 3, 8])
@pytest.mark.parametrize("samples", [1, 4])
def test_bands_compose_the_frame(device, oracle, bands, samples):
    """Sort-first sharding by render area: band k of n is rows of whole tiles given as the pass's render
    area and scissor.  Inside its band a device's pixels are the unsharded frame's, outside it touches
    nothing, and it skips the draws that lie wholly in other bands."""
    from renderer_amd import sharding
    sc = scenes.instanced_spheres(300, 200, 6, 5, 3, 5, tex_size=64)
    sc.samples = samples
    for p in sc.pipelines:
        p["samples"] = samples
    full_c, full_d = check(sc, device, oracle)
    acc_c, acc_d = np.zeros_like(full_c), np.zeros_like(full_d)
    kept = []
    for k in range(bands):
        y0, y1 = sharding.band_rows(sc.height, k, bands)
        r = SceneRunner(sharding.band_scene(sc, k, bands), device)
        try:
            c, d = r.render()
            kept.append(r.cmd.visible_draws())
        finally:
            r.destroy()
        acc_c[y0:y1], acc_d[y0:y1] = c[y0:y1], d[y0:y1]
        assert (c[:y0] == 0).all() and (c[y1:] == 0).all() and (d[:y0] == 0).all() and (d[y1:] == 0).all()
    assert np.array_equal(acc_c, full_c) and np.array_equal(acc_d, full_d)
    assert sum(kept) < bands * len(sc.draws)  # some draws were skipped somewhere ...
    assert sum(kept) >= len(sc.draws)         # ... and every draw was kept by the band(s) it shows in
    if bands == 8:  # seven rows of tiles: bands of one row (two or three of the five rows of spheres), the last band empty
        assert max(kept) <= len(sc.draws) * 2 // 3 and kept[-1] == 0


def test_draws_outside_the_frustum_are_skipped(device, oracle):
    """Draws pushed wholly beyond each of the six planes are skipped; draws that straddle a plane, sit in a
    corner, or enclose the whole frustum are kept; the frame is the oracle's either way."""
    sc = scenes.instanced_spheres(256, 160, 3, 2, 1, 11, tex_size=32)
    base = sc.draws[0]
    n_inside = len(sc.draws)

    def add(tx, ty, tz, s):
        m = hm.scale(hm.translate(hm.identity(), [tx, ty, tz]), [s, s, s])
        d = dict(base)
        d["push"] = scenes.push_block([0.9, 0.5, 0.2, 1.0], m)
        sc.draws.append(d)

    outside = [(-40.0, 1.0, 0.0, 1.0), (40.0, 1.0, 0.0, 1.0), (0.0, 41.0, 0.0, 1.0), (0.0, -39.0, 0.0, 1.0),
               (0.0, 1.0, -6.0, 1.0), (0.0, 1.0, 300.0, 1.0)]
    kept = [(-3.2, 1.0, 0.0, 1.0), (0.0, 3.0, 0.0, 0.7), (0.0, 1.0, -1.6, 0.5), (3.0, 3.0, 0.5, 0.8), (0.0, 1.0, 0.0, 500.0),
            (0.0, 1.0, 99.0, 3.0)]
    for t in outside + kept:
        add(*t)
    r = SceneRunner(sc, device)
    try:
        color, depth = r.render()
        visible = r.cmd.visible_draws()
    finally:
        r.destroy()
    oc, od, _ = oracle.render_scene(sc, threads=8)
    assert np.array_equal(depth, od) and np.array_equal(color, oc)
    assert visible == n_inside + len(kept)


def test_vertex_data_written_after_recording_is_drawn(device, oracle):
    """The boxes used for culling are computed when the command buffer is recorded; vertex data written
    afterwards (here: the mesh moved from far off-screen into view) must still be drawn."""
    sc = scenes.instanced_spheres(192, 128, 3, 2, 1, 7, tex_size=32)
    inside = sc.vbs[0].copy()
    far = inside.copy().reshape(-1, 8)
    far[:, 0] += 500.0  # every draw's box is now beyond the right plane
    sc.vbs[0] = far.reshape(inside.shape)
    r = SceneRunner(sc, device)
    try:
        r.render()
        assert r.cmd.visible_draws() == 0
        r.vbs[0].write(inside)  # same command buffer, new vertex data
        color, depth = r.render()
        assert r.cmd.visible_draws() == len(sc.draws)
    finally:
        r.destroy()
    sc.vbs[0] = inside
    oc, od, _ = oracle.render_scene(sc, threads=8)
    assert np.array_equal(depth, od) and np.array_equal(color, oc)


@pytest.mark.parametrize("grid", [(8, 6), (16, 10)])
def test_scissored_pass_skips_draws_and_tiles_outside(device, oracle, grid):
    """A scissor inside the render area inside the target, none aligned to tiles: draws outside the scissor
    are skipped, pixels inside the area but outside the scissor are cleared, pixels outside the area are
    left alone.  96 draws (every vertex wave tests its draw) and 320 (a kernel of its own tests each draw once)."""
    sc = scenes.instanced_spheres(320, 224, grid[0], grid[1], 2, 3, tex_size=32)
    sc.area = (37, 41, 230, 150)
    sc.scissor = (70, 60, 100, 75)
    r = SceneRunner(sc, device)
    try:
        color, depth = r.render()
        visible = r.cmd.visible_draws()
    finally:
        r.destroy()
    oc, od, _ = oracle.render_scene(sc, threads=8)
    assert np.array_equal(depth, od) and np.array_equal(color, oc)
    assert 0 < visible < len(sc.draws) // 2


def test_lists_of_visible_chunks(device, oracle):
    """A pass of several hundred draws: the per-draw kernel leaves the chunks of the visible draws in dense lists that
    the vertex and setup kernels walk.  Small spheres (3 chunks of vertices, 5 of primitives each: written by the draw's
    lane) among a few instances of a mesh of 2562 vertices and 5120 triangles (41 and 80 chunks: written by the whole
    wave), one of them off screen, plus a draw of a single triangle; three frames of one command buffer with the camera
    moved between them, so that the visible set and the pair of list lengths in use both change; then bands."""
    from renderer_amd import sharding
    width, height = 320, 224
    sc = scenes.instanced_spheres(width, height, 12, 9, 3, 17, tex_size=32)
    v, i = scenes.icosphere(4, radius=1.0)
    sc.vbs.append(v)
    sc.ibs.append(i)
    base = sc.draws[0]

    def big(tx, ty, tz, s, at):
        m = hm.scale(hm.translate(hm.identity(), [tx, ty, tz]), [s, s, s])
        d = dict(base, vb=1, ib=1, index_count=int(i.size), push=scenes.push_block([0.3, 0.8, 0.4, 1.0], m))
        sc.draws.insert(at, d)

    big(0.3, 1.2, 0.6, 0.9, 40)     # in view, among the small ones in submission order
    big(60.0, 1.0, 0.0, 1.0, 41)    # far to the right: culled
    big(-2.6, 0.2, 0.2, 0.8, 200)   # across the left edge
    one = dict(base, index_count=3, first_index=30)
    sc.draws.insert(100, one)
    assert len(sc.draws) >= 256
    r = SceneRunner(sc, device)
    try:
        seen = []
        for eye in ([0.0, 0.0, -2.0], [1.5, 0.4, -1.2], [-2.5, -0.8, -2.4]):
            cam = hm.Camera()
            cam.eye_position = np.array(eye, dtype=np.float32)
            sc.ubos[0] = hm.global_ubo(cam, hm.identity(), width, height)
            r.set_ubo(0, sc.ubos[0])
            color, depth = r.render()
            seen.append(r.cmd.visible_draws())
            oc, od, _ = oracle.render_scene(sc, threads=8)
            assert np.array_equal(depth, od) and np.array_equal(color, oc), eye
    finally:
        r.destroy()
    assert len(set(seen)) > 1 and all(0 < n < len(sc.draws) for n in seen)
    full_c, full_d = color, depth
    acc_c, acc_d = np.zeros_like(full_c), np.zeros_like(full_d)
    for k in range(4):
        y0, y1 = sharding.band_rows(height, k, 4)
        r = SceneRunner(sharding.band_scene(sc, k, 4), device)
        try:
            c, d = r.render()
            c2, d2 = r.render()  # the other pair of list lengths
        finally:
            r.destroy()
        assert np.array_equal(c, c2) and np.array_equal(d, d2)
        acc_c[y0:y1], acc_d[y0:y1] = c[y0:y1], d[y0:y1]
    assert np.array_equal(acc_c, full_c) and np.array_equal(acc_d, full_d)


@pytest.mark.parametrize("shader", [scenes.SHADER_TEST_KTX, scenes.SHADER_PUSH_CONSTANTS])
def test_four_samples_pixels_with_many_owners(device, oracle, shader):
    """Four samples per pixel where pixels have two, three and four owners: fans of thin triangles whose hubs lie inside
    pixels, some of the hubs in every quarter of a tile, all over a background quad, in a render area that cuts tiles
    (lanes outside the area go along with the wave's exchange of shadings, with nothing to shade).  Each frame against
    the oracle."""
    width, height = 150, 101
    rng = np.random.RandomState(11)
    tris = [[(2.0, 3.0, 0.9), (148.0, 2.0, 0.9), (3.0, 99.0, 0.9)], [(148.0, 2.0, 0.9), (147.0, 99.5, 0.9), (3.0, 99.0, 0.9)]]
    for _ in range(40):
        hx, hy = float(rng.uniform(5, 145)), float(rng.uniform(5, 96))
        spokes = int(rng.randint(5, 14))
        radius = float(rng.uniform(1.5, 9.0))
        z = float(rng.uniform(0.1, 0.8))
        first = float(rng.uniform(0, 6.28))
        ring = [(hx + radius * np.cos(first + 6.2831853 * k / spokes), hy + radius * np.sin(first + 6.2831853 * k / spokes)) for k in range(spokes)]
        for k in range(spokes):
            a, b = ring[k], ring[(k + 1) % spokes]
            tris.append([(hx, hy, z), (float(b[0]), float(b[1]), z + 0.01 * k), (float(a[0]), float(a[1]), z)])
    tex = scenes._texture(scenes.checker_texture(64, 5)) if shader == scenes.SHADER_TEST_KTX else None
    sc = helpers.window_triangles(width, height, tris, shader=shader, samples=4, texture=tex, pipeline_kw=dict(cull_mode=0))
    check(sc, device, oracle)
    sc.area = (7, 5, 130, 83)
    sc.scissor = (20, 9, 100, 70)
    r = SceneRunner(sc, device)
    try:
        color, depth = r.render()
    finally:
        r.destroy()
    oc, od, _ = oracle.render_scene(sc, threads=8)
    ax, ay, aw, ah = sc.area
    sl = (slice(ay, ay + ah), slice(ax, ax + aw))
    assert np.array_equal(depth[sl], od[sl]) and np.array_equal(color[sl], oc[sl])


@pytest.mark.parametrize("size", [(8192, 8192, 1), (8192, 2048, 4)])
def test_largest_target(device, oracle, size):
    """The largest extent a target may have, 8192 pixels (256 columns of tiles, the last value a tile box's byte holds):
    triangles in the far corners, one across the whole target, a soup between them; against the oracle."""
    width, height, samples = size
    rng = np.random.RandomState(5)
    w, h = float(width), float(height)
    tris = [[(1.0, 1.0, 0.8), (w - 1.5, 2.0, 0.8), (2.0, h - 1.25, 0.8)],
            [(w - 40.0, h - 40.0, 0.3), (w - 0.5, h - 30.0, 0.3), (w - 20.0, h - 0.5, 0.3)],
            [(w - 33.0, 0.5, 0.2), (w - 0.25, 0.75, 0.2), (w - 0.5, 31.0, 0.2)],
            [(0.5, h - 31.0, 0.25), (30.0, h - 0.5, 0.25), (0.25, h - 0.25, 0.25)]]
    for _ in range(300):
        cx, cy, e = float(rng.uniform(0, w)), float(rng.uniform(0, h)), float(np.exp(rng.uniform(np.log(2.0), np.log(900.0))))
        z = float(rng.uniform(0.05, 0.95))
        tris.append([(cx + float(rng.uniform(-e, e)), cy + float(rng.uniform(-e, e)), z) for _ in range(3)])
    sc = helpers.window_triangles(width, height, tris, samples=samples, pipeline_kw=dict(cull_mode=0))
    check(sc, device, oracle, threads=16)


@pytest.mark.parametrize("samples", [1, 4])
def test_a_tile_with_more_triangles_than_its_list_holds(device, oracle, samples):
    """40 000 small triangles inside one tile (and a few hundred over the rest of a small target): a tile's list holds at
    most 16 384 entries, the rest spill to the pass-wide list, which every tile scans for entries of its own.  The frame is
    the oracle's all the same, and no overflow is reported."""
    rng = np.random.RandomState(21)
    width, height = 96, 70
    n = 40000
    cx, cy = rng.uniform(33.0, 62.0, size=n), rng.uniform(33.0, 62.0, size=n)  # tile (1, 1)
    tris = []
    for k in range(n):
        e = float(rng.uniform(0.4, 3.0))
        z = float(rng.randint(1, 64)) / 64.0
        tris.append([(float(cx[k] + rng.uniform(-e, e)), float(cy[k] + rng.uniform(-e, e)), z) for _ in range(3)])
    for _ in range(300):
        x, y, e = float(rng.uniform(0, width)), float(rng.uniform(0, height)), float(rng.uniform(1.0, 20.0))
        tris.append([(x + float(rng.uniform(-e, e)), y + float(rng.uniform(-e, e)), float(rng.uniform(0.1, 0.9))) for _ in range(3)])
    sc = helpers.window_triangles(width, height, tris, samples=samples, pipeline_kw=dict(cull_mode=0))
    check(sc, device, oracle)
    device.wait_idle()  # an overflow of the lists would be reported here


def test_resubmission_is_idempotent(device):
    sc = helpers.soup_scene(200, 150, 2000, 77, depth_levels=2, n_draws=7)
    r = SceneRunner(sc, device)
    try:
        c1, d1 = r.render()
        for _ in range(5):
            r.submit()
        c2, d2 = r.render()
    finally:
        r.destroy()
    assert np.array_equal(c1, c2) and np.array_equal(d1, d2)


def test_error_codes(device):
    with pytest.raises(api.RendererError) as e:
        api.Target(device, 0, 10)
    assert e.value.code == -1
    with pytest.raises(api.RendererError) as e:
        api.Pipeline(device, api.SHADER_TEST_KTX, depth_compare=7)
    assert e.value.code == -8 and "LESS" in e.value.msg
    with pytest.raises(api.RendererError) as e:
        api.Texture(device, np.zeros((4, 4, 4), np.uint8), fmt=129)
    assert e.value.code == -11 and "unsupported pixels format" in e.value.msg
    cmd = api.CommandBuffer(device)
    with pytest.raises(api.RendererError) as e:
        cmd.draw_indexed(3)
    assert e.value.code == -12
    cmd.destroy()
    cmd.destroy()  # destroy is idempotent (handlers.hpp:108-113)


def test_upload_precedes_every_later_command_buffer(device, oracle):
    """A small stream-ordered vertex write (a copy queued on the device's stream), then frames of TWO command
    buffers into two targets: both must see the new vertices, not only the first one submitted."""
    sc = helpers.soup_scene(160, 96, 40, 5, n_draws=1)
    moved = sc.vbs[0].copy()
    moved[:, 0] = -moved[:, 0]  # mirrored mesh: 40 triangles x 3 vertices x 32 B = 3840 B, one staged copy
    assert moved.nbytes <= 4096
    r = SceneRunner(sc, device, frames_in_flight=2)
    try:
        r.render()
        r.render()  # both command buffers have run once with the old vertices
        r.vbs[0].write(moved)
        a = r.submit()
        b = r.submit()
        assert a != b
        device.wait_idle()
        frames = [r.targets[a].read(), r.targets[b].read()]
    finally:
        r.destroy()
    sc.vbs[0] = moved
    oc, od, _ = oracle.render_scene(sc, threads=4)
    for color, depth in frames:
        assert np.array_equal(depth, od) and np.array_equal(color, oc)


def test_index_write_after_recording_needs_a_new_recording(device, oracle):
    """The number of vertices a draw transforms is scanned from its indices at record time: a command buffer
    recorded before a write into an index buffer refuses to run until it is recorded again."""
    sc = helpers.soup_scene(128, 80, 30, 9, n_draws=1)
    r = SceneRunner(sc, device)
    try:
        r.render()
        new_idx = sc.ibs[0][::-1].copy()
        r.ibs[0].write(new_idx)
        with pytest.raises(api.RendererError) as e:
            r.render()
        assert e.value.code == -12 and "record" in e.value.msg
        r.record()
        color, depth = r.render()
    finally:
        r.destroy()
    sc.ibs[0] = new_idx
    oc, od, _ = oracle.render_scene(sc, threads=4)
    assert np.array_equal(depth, od) and np.array_equal(color, oc)


def test_small_write_into_a_shared_uniform_vertex_buffer_reaches_memory(device, oracle):
    """A buffer created with UNIFORM | VERTEX usage holds vertices: a small write into it must be an ordinary copy
    (the held-back uniform path would leave the vertex kernel reading the old bytes)."""
    sc = helpers.soup_scene(128, 80, 4, 3, n_draws=1)  # 12 vertices = 384 B, below the held-back limit
    moved = sc.vbs[0].copy()
    moved[:, 0] = moved[:, 0] * np.float32(0.5)
    r = SceneRunner(sc, device)
    try:
        r.vbs[0].destroy()
        r.vbs[0] = api.Buffer(device, api.BUFFER_USAGE_VERTEX | api.BUFFER_USAGE_UNIFORM, sc.vbs[0])
        r.record()
        r.render()
        r.vbs[0].write(moved)
        color, depth = r.render()
    finally:
        r.destroy()
    sc.vbs[0] = moved
    oc, od, _ = oracle.render_scene(sc, threads=4)
    assert np.array_equal(depth, od) and np.array_equal(color, oc)


@pytest.mark.parametrize("setting", ["RND_TILE_WORKGROUPS=4", "RND_TILE_WORKGROUPS=5", "RND_STREAM_PRIORITY=high",
                                     "RND_STREAM_PRIORITY=normal", "RND_STREAM_PRIORITY=low"])
def test_scheduling_settings_do_not_change_the_frame(oracle, setting, monkeypatch):
    """How many tile workgroups share a CU (RND_TILE_WORKGROUPS=4, otherwise five) and which pool of hardware queues
    the command buffers' streams come from (RND_STREAM_PRIORITY) change when a frame is finished, never what is in
    it: 60 frames rotate through three command buffers, each on a stream of its own, and the last three are compared
    with the oracle."""
    from renderer_amd import api
    monkeypatch.setenv(*setting.split("="))
    dev = api.Device(0)  # the settings are read when a device is created
    try:
        sc = scenes.config0(320, 180, tex_size=64, segments=48, rings=24)
        model = np.asarray(sc.ubos[0][48:64], dtype=np.float32).reshape(4, 4)
        n_frames = 60
        ubos = []
        for i in range(4):
            cam = hm.Camera()
            cam.eye_position = np.array([0.2 * np.sin(0.7 * i), 0.05 * i, -2.0 - 0.04 * i], dtype=np.float32)
            ubos.append(hm.global_ubo(cam, model, sc.width, sc.height))
        r = SceneRunner(sc, dev, frames_in_flight=3)
        try:
            owner = {}
            for i in range(n_frames):
                r.set_ubo(0, ubos[i % 4])
                owner[r.submit()] = i
            dev.wait_idle()
            got = {img: r.targets[img].read() for img in owner}
        finally:
            r.destroy()
        assert len(owner) == 3
        for img, frame in owner.items():
            sc.ubos[0] = ubos[frame % 4]
            oc, od, _ = oracle.render_scene(sc, threads=4)
            assert np.array_equal(got[img][1], od), "image %d (frame %d): depth differs" % (img, frame)
            assert np.array_equal(got[img][0], oc), "image %d (frame %d): colour differs" % (img, frame)
    finally:
        dev.destroy()


def test_vertex_records_beyond_4gb(oracle):
    """300 draws index one vertex buffer from its start (vertexOffset 0, firstIndex advancing), so each transforms the
    buffer up to its own largest index: 316 million transformed records, 5 GB of positions and 3.8 GB of the vertex
    shader's other outputs.  A record's address does not fit a 32-bit byte offset from the array's base."""
    sc = helpers.soup_scene(320, 180, 700000, 11, max_extent=0.01, depth_levels=0, z_range=(0.05, 0.95),
                            shader=scenes.SHADER_TEST_KTX, n_draws=300,
                            texture=dict(texels=scenes.checker_texture(64, 11), gen_mips=True, filtering=1, tiled=True, fmt=43))
    slots = sum(int(sc.ibs[d["ib"]][d["first_index"]:d["first_index"] + d["index_count"]].max()) + 1 for d in sc.draws)
    assert slots * 16 > (1 << 32) and slots * 12 > (3 << 30)
    r = SceneRunner(sc)
    try:
        color, depth = r.render()
    finally:
        r.destroy()
    oc, od, _ = oracle.render_scene(sc, threads=8)
    assert np.array_equal(depth, od) and np.array_equal(color, oc)


def test_a_device_outlives_what_was_created_on_it():
    """Every object points back at its device and its destroy call follows that pointer (the reference destroys its
    context last, context.cpp:567-574): the library refuses to destroy a device that objects still point at, and the
    Python wrapper destroys the live ones first."""
    from renderer_amd import _lib
    lib = _lib.load()
    dev = api.Device(0)
    buf = api.Buffer(dev, api.BUFFER_USAGE_VERTEX, nbytes=64)
    fence = api.Fence(dev)
    assert lib.rnd_device_destroy(dev.h) != 0
    assert "still alive" in lib.rnd_last_error().decode()
    buf.write(np.arange(16, dtype=np.float32))  # the device is as it was
    assert np.array_equal(buf.read().view(np.float32), np.arange(16, dtype=np.float32))
    buf.destroy()
    assert lib.rnd_device_destroy(dev.h) != 0  # the fence is left
    dev.destroy()  # destroys the fence, then the device
    assert not fence.h and not dev.h
    fence.destroy()  # a second destroy of either is a no-op
    dev.destroy()


def test_a_pass_without_geometry_reads_zeroed_counters(device, oracle):
    """A pass whose scissor is empty launches no geometry kernel, so nothing but the recording has zeroed the list
    counters the tile kernel reads: they must be zero by the time the first frame runs, whatever the allocation held
    before (here the words of a freed target) and whatever else the device was still doing when the pass was recorded
    (here the mip chain of a large texture)."""
    big = scenes.checker_texture(2048, 3)
    for i in range(25):
        warm = SceneRunner(helpers.soup_scene(256, 256, 300, 100 + i, max_extent=1.5), device)
        try:
            warm.render()
        finally:
            warm.destroy()
        texture = api.Texture(device, big, gen_mips=True)
        sc = helpers.soup_scene(64, 48, 50, i)
        sc.scissor = (5, 5, 0, 10)
        r = SceneRunner(sc, device)
        try:
            color, depth = r.render()
        finally:
            r.destroy()
            texture.destroy()
        oc, od, _ = oracle.render_scene(sc, threads=4)
        assert np.array_equal(depth, od) and np.array_equal(color, oc)


@pytest.mark.parametrize("height", ["40", "32"])
def test_tile_height_does_not_change_the_frame(device, oracle, height, monkeypatch):
    """The rows of pixels per tile (32, or 40 for a one-draw mesh on a large target; RND_TILE_HEIGHT, read when a command
    buffer is recorded, forces one) change how the frame is cut into workgroups, never what is in it: soups over every knob,
    recorded state cases and a target whose height is a multiple of neither, all against the oracle."""
    import fuzz_soup
    import fuzz_state
    monkeypatch.setenv("RND_TILE_HEIGHT", height)
    cases = [fuzz_soup.case(s) for s in range(3100, 3112)] + [fuzz_state.case(s) for s in range(200, 208)]
    cases.append(scenes.config1(333, 217, quads=120, tex_size=64))   # one draw, 217 rows: 5 full rows of 40-row tiles and 17 rows more
    cases.append(scenes.config0(160, 90, samples=4, tex_size=32, segments=16, rings=8))  # four samples keep 32 rows whatever is asked
    for k, sc in enumerate(cases):
        color, depth = gpu_render(sc, device)
        oc, od, _ = oracle.render_scene(sc, threads=8)
        ax, ay, aw, ah = sc.area
        sl = (slice(ay, ay + ah), slice(ax, ax + aw))
        assert np.array_equal(depth[sl], od[sl]), "case %d: depth differs at %d samples" % (k, int((depth[sl] != od[sl]).sum()))
        assert np.array_equal(color[sl], oc[sl]), "case %d: colour differs" % k


def test_tile_shard_set_after_recording_with_tall_tiles(device, monkeypatch):
    """Interleaved tile shards count tiles of 32 rows.  A command buffer recorded with 40-row tiles and then given a shard is
    compiled again with 32 rows: the shards still compose the frame."""
    monkeypatch.setenv("RND_TILE_HEIGHT", "40")
    sc = scenes.config1(300, 200, quads=100, tex_size=64)
    full_c, full_d = gpu_render(sc, device)
    acc_c, acc_d = np.zeros_like(full_c), np.zeros_like(full_d)
    tiles_x = (sc.width + 31) // 32
    shards = 3
    for k in range(shards):
        r = SceneRunner(sc, device)          # recorded with 40 rows here ...
        try:
            r.cmd.set_tile_shard(k, shards)  # ... and compiled again with 32 by this call
            c, d = r.render()
        finally:
            r.destroy()
        for ty in range((sc.height + 31) // 32):
            for tx in range(tiles_x):
                sl = (slice(ty * 32, ty * 32 + 32), slice(tx * 32, tx * 32 + 32))
                if (ty * tiles_x + tx) % shards == k:
                    acc_c[sl], acc_d[sl] = c[sl], d[sl]
                else:
                    assert (c[sl] == 0).all()
    assert np.array_equal(acc_c, full_c) and np.array_equal(acc_d, full_d)


def test_more_command_buffers_than_streams(oracle, monkeypatch):
    """The device keeps a pool of streams for its command buffers (three unless the environment says otherwise); command buffers
    beyond that share a stream with an earlier one.  Seven command buffers on a pool of two, 40 frames rotating through them
    with the uniform block rewritten before every submit: each image holds the frame of its last submit."""
    from renderer_amd import api
    monkeypatch.setenv("RND_CMD_STREAMS", "2")
    monkeypatch.setenv("RND_DEBUG_SYNC", "1")
    dev = api.Device(0)
    try:
        sc = scenes.config0(200, 120, tex_size=32, segments=24, rings=12)
        model = np.asarray(sc.ubos[0][48:64], dtype=np.float32).reshape(4, 4)
        ubos = []
        for i in range(5):
            cam = hm.Camera()
            cam.eye_position = np.array([0.3 * np.sin(0.9 * i), 0.04 * i, -2.0 - 0.05 * i], dtype=np.float32)
            ubos.append(hm.global_ubo(cam, model, sc.width, sc.height))
        r = SceneRunner(sc, dev, frames_in_flight=7)
        try:
            owner = {}
            for i in range(40):
                r.set_ubo(0, ubos[i % 5])
                owner[r.submit()] = i
            dev.wait_idle()
            got = {img: r.targets[img].read() for img in owner}
        finally:
            r.destroy()
        assert len(owner) == 7
        for img, frame in owner.items():
            sc.ubos[0] = ubos[frame % 5]
            oc, od, _ = oracle.render_scene(sc, threads=4)
            assert np.array_equal(got[img][1], od) and np.array_equal(got[img][0], oc), "image %d (frame %d) differs" % (img, frame)
    finally:
        dev.destroy()


def test_general_tile_kernel_renders_the_same_bits(oracle, monkeypatch):
    """RND_TILE_GENERIC=1 (read once per process by the library, so this test runs the frames in a child process): a pass of
    the samples' kind rendered by the general instantiation of the tile kernel equals the oracle like the specialised one."""
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from renderer_amd import scenes; from renderer_amd.runner import SceneRunner; from oracle import oracle\n"
            "for sc in (scenes.config0(320, 180, tex_size=64, segments=32, rings=16), scenes.config1(240, 136, quads=80, tex_size=64)):\n"
            "    r = SceneRunner(sc, ordinal=0); c, d = r.render(); r.destroy()\n"
            "    oc, od, _ = oracle.render_scene(sc, threads=4)\n"
            "    assert np.array_equal(d, od) and np.array_equal(c, oc), sc.name\n"
            "print('same bits')\n") % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, RND_TILE_GENERIC="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=150)
    assert out.returncode == 0 and "same bits" in out.stdout, out.stderr[-2000:]
