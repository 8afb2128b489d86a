"""Known-answer tests that pin the oracle's scalar rules and its fill rule (CPU only).
The reference holds no fixtures for this path (parity unpinned), so these are the repo's
own known answers for the Vulkan 1.0 rules the oracle restates."""
import math

import numpy as np
import pytest

import helpers
from renderer_amd import hostmath as hm, scenes


def test_srgb_round_trip_and_monotone(oracle):
    lib = oracle.load()
    dec = [lib.orc_srgb_decode(k) for k in range(256)]
    assert dec[0] == 0.0 and dec[255] == 1.0
    assert all(b > a for a, b in zip(dec, dec[1:]))
    for k in range(256):
        assert lib.orc_srgb_encode(dec[k]) == k
        exact = (k / 255.0) / 12.92 if k / 255.0 <= 0.04045 else ((k / 255.0 + 0.055) / 1.055) ** 2.4
        assert abs(dec[k] - exact) <= 1e-7
    assert lib.orc_srgb_encode(-1.0) == 0 and lib.orc_srgb_encode(float("nan")) == 0
    assert lib.orc_srgb_encode(2.0) == 255 and lib.orc_srgb_encode(float("inf")) == 255
    # encode is round-to-nearest of 255 * srgb(c)
    rng = np.random.RandomState(0)
    for c in rng.uniform(0, 1, 2000).astype(np.float32):
        s = 12.92 * c if c <= 0.0031308 else 1.055 * c ** (1 / 2.4) - 0.055
        k = lib.orc_srgb_encode(float(c))
        assert abs(k - 255.0 * s) <= 0.5 + 1e-3


def test_depth24_known_answers(oracle):
    lib = oracle.load()
    assert lib.orc_depth24(0.0) == 0
    assert lib.orc_depth24(1.0) == 0xFFFFFF
    assert lib.orc_depth24(0.5) == 8388608  # 8388607.5: the tie goes to the even neighbour
    assert lib.orc_depth24(0.71875) == 12058623  # 12058623.28: down (adding 0.5 in float and truncating gives ...624)
    assert lib.orc_depth24(0.75) == 12582911  # 12582911.25
    assert lib.orc_depth24(-3.0) == 0 and lib.orc_depth24(7.0) == 0xFFFFFF
    assert lib.orc_depth24(1.0 / 16777215.0) == 1


def test_snap_known_answers(oracle):
    lib = oracle.load()
    assert lib.orc_snap(0.0) == 0
    assert lib.orc_snap(1.0) == 256
    assert lib.orc_snap(0.5) == 128
    assert lib.orc_snap(1.0 / 512.0) == 1      # exactly half a sub-pixel rounds up
    assert lib.orc_snap(-1.0 / 512.0) == 0     # floor(x + 0.5): -0.5 + 0.5 = 0
    assert lib.orc_snap(-0.75) == -192
    assert lib.orc_snap(1919.99609375) == 491519


def test_lod_log2_accuracy(oracle):
    lib = oracle.load()
    for x in np.exp(np.random.RandomState(1).uniform(-20, 20, 2000)):
        assert abs(lib.orc_log2(float(np.float32(x))) - math.log2(float(np.float32(x)))) < 2e-5
    assert lib.orc_log2(1.0) == 0.0 and lib.orc_log2(4.0) == 2.0
    assert lib.orc_log2(0.0) == -126.0


def test_matrices_against_hand_values(oracle):
    # perspectiveFov(90 deg, 800, 600): h = cot(45 deg) = 1, w = h * 600 / 800
    p = oracle.mat_call("orc_perspective_fov_rh_no", hm.radians(90.0), 800.0, 600.0, 0.01, 100.0)
    assert abs(p[1, 1] - 1.0) < 1e-6 and abs(p[0, 0] - 0.75) < 1e-6
    assert p[2, 3] == -1.0
    assert abs(p[2, 2] - (-(100.01) / 99.99)) < 1e-6 and abs(p[3, 2] - (-2.0 / 99.99)) < 1e-7
    z = oracle.mat_call("orc_perspective_fov_rh_zo", hm.radians(90.0), 800.0, 600.0, 0.01, 100.0)
    assert abs(z[2, 2] - (100.0 / (0.01 - 100.0))) < 1e-6 and abs(z[3, 2] - (-1.0 / 99.99)) < 1e-7
    # lookAt with the sample's defaults: f=(0,0,1), s=(-1,0,0), u=(0,1,0)
    v = oracle.mat_call("orc_look_at_rh", [0, 0, -2], [0, 0, 0], [0, 1, 0])
    expect = np.array([[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, 0], [0, 0, -2, 1]], dtype=np.float32)
    assert np.array_equal(v, expect)
    t = oracle.mat_call("orc_translate", np.eye(4), [0, 10, 0])
    assert t[3, 1] == 10.0 and t[3, 3] == 1.0


@pytest.mark.parametrize("wh", [(800, 600), (1280, 720)])
def test_matrices_python_vs_oracle(oracle, wh):
    """The numpy host maths and the C restatement agree (trig may differ in the last place)."""
    w, h = wh
    a = hm.perspective_fov(hm.radians(90.0), w, h, 0.01, 100.0)
    b = oracle.mat_call("orc_perspective_fov_rh_no", hm.radians(90.0), float(w), float(h), 0.01, 100.0)
    assert np.allclose(a, b, rtol=3e-7, atol=0)
    a = hm.perspective_fov(hm.radians(90.0), w, h, 0.01, 100.0, zero_to_one=True)
    b = oracle.mat_call("orc_perspective_fov_rh_zo", hm.radians(90.0), float(w), float(h), 0.01, 100.0)
    assert np.allclose(a, b, rtol=3e-7, atol=0)
    eye, ctr, up = [0.3, 0.2, -2], [0, 0.1, 0], [0, 1, 0]
    assert np.allclose(hm.look_at(eye, ctr, up), oracle.mat_call("orc_look_at_rh", eye, ctr, up), rtol=1e-6, atol=1e-7)
    for j in range(10):
        r = oracle.mat_call("orc_rotate", np.eye(4), hm.radians(36.0 * j), [0, 0, 1])
        r = oracle.mat_call("orc_translate", r, [0, 10, 0])
        assert np.allclose(hm.ring_transform(j), r, rtol=1e-6, atol=1e-6)
    y = oracle.mat_call("orc_yaw_pitch_roll", hm.radians(0.5), hm.radians(-0.5), 0.0)
    assert np.allclose(hm.yaw_pitch_roll(hm.radians(0.5), hm.radians(-0.5), 0.0), y, rtol=1e-6, atol=1e-7)
    m1, m2 = hm.ring_transform(3), hm.look_at(eye, ctr, up)
    assert np.allclose(hm.mul(m1, m2), oracle.mat_call("orc_mat4_mul", m1, m2), rtol=1e-6, atol=1e-6)
    assert np.allclose(hm.scale(m1, [2, 3, 4]), oracle.mat_call("orc_scale", m1, [2, 3, 4]))


def test_global_ubo_layout():
    """320 bytes, order projection, view_projection, view, model, mvp (base_obj_viewer_app.hpp:36-43)."""
    cam = hm.Camera()
    model = hm.scale(hm.identity(), [0.5, 0.5, 0.5])
    u = hm.global_ubo(cam, model, 800, 600)
    assert u.dtype == np.float32 and u.nbytes == 320
    assert np.array_equal(u[0:16].reshape(4, 4), cam.proj_matrix)
    assert np.array_equal(u[16:32].reshape(4, 4), cam.view_proj_matrix)
    assert np.array_equal(u[32:48].reshape(4, 4), cam.view_matrix)
    assert np.array_equal(u[48:64].reshape(4, 4), model)
    assert np.array_equal(u[64:80].reshape(4, 4), hm.mul(hm.mul(cam.proj_matrix, cam.view_matrix), model))


# ------------------------------------------------------------------ fill rule

def _coverage(oracle, sc):
    _, _, prim = oracle.render_scene(sc, want_prim=True)
    return prim[..., 0] != 0xFFFFFFFF


def test_top_left_rule_on_pixel_centres(oracle):
    """Corners exactly on pixel centres: the top and left edges own their samples, the
    hypotenuse does not."""
    W = H = 16
    tri = [(2.5, 2.5, 0.5), (2.5, 10.5, 0.5), (10.5, 2.5, 0.5)]  # counter-clockwise on screen
    cov = _coverage(oracle, helpers.window_triangles(W, H, [tri]))
    assert cov.any()
    fixed = [(int(x * 256), int(y * 256)) for x, y, _ in tri]
    assert np.array_equal(cov, helpers.exact_coverage(fixed, W, H))
    assert cov[2, 2] and cov[2, 9] and cov[9, 2]      # top edge y=2.5 and left edge x=2.5 are owned
    assert not cov[2, 10] and not cov[10, 2]           # the far corners lie on the hypotenuse
    assert not cov[6, 6]                               # (6.5,6.5) is on the hypotenuse: not owned


def test_fill_rule_random_triangles(oracle):
    """Random triangles with corners on 1/4-pixel positions (so samples often land exactly on
    edges): the oracle's coverage equals the rule evaluated independently in integers."""
    rng = np.random.RandomState(7)
    W = H = 16
    n_edge_hits = 0
    for _ in range(60):
        pts = rng.randint(0, 4 * W + 1, size=(3, 2)) / 4.0
        tri = [(float(x), float(y), 0.5) for x, y in pts]
        sc = helpers.window_triangles(W, H, [tri], pipeline_kw=dict(cull_mode=0))
        fixed = [(int(x * 256), int(y * 256)) for x, y, _ in tri]
        expect = helpers.exact_coverage(fixed, W, H)
        assert np.array_equal(_coverage(oracle, sc), expect)
        n_edge_hits += int(expect.sum())
    assert n_edge_hits > 0


def test_shared_edge_no_gap_no_double_hit(oracle):
    """A fan of triangles around a common corner: every pixel centre inside the union is hit
    by exactly one triangle."""
    rng = np.random.RandomState(11)
    W = H = 32
    for trial in range(10):
        c = (16.0 + rng.randint(-8, 9) / 4.0, 16.0 + rng.randint(-8, 9) / 4.0)
        ang = 2 * np.pi * (np.arange(7) + rng.uniform(0.1, 0.9, 7)) / 7.0  # every gap is below 180 degrees
        ring = [(c[0] + round(12 * np.cos(a) * 4) / 4.0, c[1] + round(12 * np.sin(a) * 4) / 4.0) for a in ang]
        tris = []
        for k in range(len(ring)):
            a, b = ring[k], ring[(k + 1) % len(ring)]
            tris.append([(c[0], c[1], 0.5), (a[0], a[1], 0.5), (b[0], b[1], 0.5)])
        total = np.zeros((H, W), dtype=np.int32)
        for t in tris:
            total += _coverage(oracle, helpers.window_triangles(W, H, [t], pipeline_kw=dict(cull_mode=0)))
        union = _coverage(oracle, helpers.window_triangles(W, H, tris, pipeline_kw=dict(cull_mode=0)))
        assert total.max() == 1
        assert np.array_equal(union, total == 1)


def test_degenerate_backface_and_sliver(oracle):
    W = H = 16
    zero_area = [(1.0, 1.0, 0.5), (5.0, 5.0, 0.5), (9.0, 9.0, 0.5)]
    assert not _coverage(oracle, helpers.window_triangles(W, H, [zero_area], pipeline_kw=dict(cull_mode=0))).any()
    ccw = [(2.0, 2.0, 0.5), (2.0, 12.0, 0.5), (12.0, 2.0, 0.5)]
    cw = [ccw[0], ccw[2], ccw[1]]
    assert _coverage(oracle, helpers.window_triangles(W, H, [ccw])).any()          # front (CCW), cull BACK
    assert not _coverage(oracle, helpers.window_triangles(W, H, [cw])).any()       # back face culled
    assert _coverage(oracle, helpers.window_triangles(W, H, [cw], pipeline_kw=dict(front_face=1))).any()
    assert not _coverage(oracle, helpers.window_triangles(W, H, [ccw], pipeline_kw=dict(cull_mode=1))).any()
    sliver = [(3.25, 3.2, 0.5), (3.3, 9.7, 0.5), (3.4, 3.2, 0.5)]  # between two pixel-centre columns
    assert not _coverage(oracle, helpers.window_triangles(W, H, [sliver], pipeline_kw=dict(cull_mode=0))).any()


def test_equal_depth_submission_order(oracle):
    """LESS_OR_EQUAL: of two coplanar fragments the later primitive wins; LESS: the earlier."""
    W = H = 16
    a = [(1.0, 1.0, 0.5), (1.0, 14.0, 0.5), (14.0, 1.0, 0.5)]
    b = [(2.0, 2.0, 0.5), (2.0, 15.0, 0.5), (15.0, 2.0, 0.5)]
    sc = helpers.window_triangles(W, H, [a, b])
    _, _, prim = oracle.render_scene(sc, want_prim=True)
    both = prim[4, 4, 0]
    assert both == 1
    sc = helpers.window_triangles(W, H, [a, b], pipeline_kw=dict(depth_compare=1))
    _, _, prim = oracle.render_scene(sc, want_prim=True)
    assert prim[4, 4, 0] == 0
    # a fragment at exactly the clear depth passes LESS_OR_EQUAL and fails LESS
    far = [(1.0, 1.0, 1.0), (1.0, 14.0, 1.0), (14.0, 1.0, 1.0)]
    assert _coverage(oracle, helpers.window_triangles(W, H, [far])).any()
    assert not _coverage(oracle, helpers.window_triangles(W, H, [far], pipeline_kw=dict(depth_compare=1))).any()


def test_near_plane_clip_matches_geometry(oracle):
    """A triangle that crosses z = 0: only the part in front of the near plane is drawn, and
    the literal (shade every fragment) and deferred oracle modes agree."""
    W = H = 64
    # z goes from -0.5 (behind) at the left corner to 0.5 at the right side; the z=0 line is x = 32
    tri = [(0.0, 32.0, -0.5), (64.0, 60.0, 0.5), (64.0, 4.0, 0.5)]
    sc = helpers.window_triangles(W, H, [tri], pipeline_kw=dict(cull_mode=0))
    cov = _coverage(oracle, sc)
    assert cov.any()
    assert not cov[:, :31].any()       # nothing left of the z=0 crossing survives
    assert cov[32, 40] and cov[32, 63]
    c1, d1, _ = oracle.render_scene(sc)
    c2, d2, _ = oracle.render_scene(sc, literal=True)
    assert np.array_equal(c1, c2) and np.array_equal(d1, d2)
    assert d1[cov].min() == 131072  # first surviving column has its centre at x = 32.5: z = 0.5/64 -> 2^17


def test_literal_equals_deferred_on_soups(oracle):
    for seed in range(4):
        sc = helpers.soup_scene(96, 64, 300, seed, depth_levels=3 if seed % 2 else 0, n_draws=3)
        a = oracle.render_scene(sc, want_prim=True)
        b = oracle.render_scene(sc, literal=True, want_prim=True, threads=3)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    sc = helpers.perspective_soup(96, 64, 200, 5, shader=scenes.SHADER_TEST_KTX,
                                  texture=dict(texels=scenes.checker_texture(64, 1), gen_mips=True, filtering=1, tiled=False, fmt=43))
    a = oracle.render_scene(sc, want_prim=True)
    b = oracle.render_scene(sc, literal=True, want_prim=True, threads=4)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_mip_chain_rule(oracle):
    """levels = (uint32)log2(max(w,h)): one short of a full chain (tools.cpp:514); an exact
    2:1 linear blit is the mean of 2x2 texels taken in linear light."""
    lib = oracle.load()
    assert lib.orc_mip_level_count(1024, 1024, 1) == 10
    assert lib.orc_mip_level_count(2048, 512, 1) == 11
    assert lib.orc_mip_level_count(1000, 1000, 1) == 9
    assert lib.orc_mip_level_count(1024, 1024, 0) == 1
    tex = scenes.checker_texture(16, 3)
    levels = oracle.mip_chain(tex, True, srgb=False)
    assert [l.shape[0] for l in levels] == [16, 8, 4, 2]
    mean = tex.reshape(8, 2, 8, 2, 4).astype(np.float64).mean(axis=(1, 3))
    assert np.abs(levels[1].astype(np.float64) - mean).max() <= 0.5 + 1e-6
    srgb_levels = oracle.mip_chain(tex, True, srgb=True)
    assert np.array_equal(srgb_levels[1][..., 3], levels[1][..., 3])  # alpha is linear either way
    assert (srgb_levels[1][..., :3].astype(int) >= levels[1][..., :3].astype(int)).all()  # linear-light mean is brighter
