"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same scenes.
Coverage and depth must match bit for bit; colour is compared with a tolerance of one
unorm8 step per channel (the bar BASELINE.json states) and the exact-match count is printed."""
import numpy as np
import pytest

from renderer_amd import scenes
from renderer_amd.runner import SceneRunner

pytestmark = pytest.mark.gpu


def compare(scene, device, oracle, threads=8):
    r = SceneRunner(scene, device)
    try:
        color, depth = r.render()
    finally:
        r.destroy()
    ocolor, odepth, _ = oracle.render_scene(scene, threads=threads)
    assert np.array_equal(depth, odepth), "depth words differ at %d samples" % int((depth != odepth).sum())
    cov, ocov = depth != depth.max(), odepth != odepth.max()
    assert np.array_equal(cov, ocov)
    diff = np.abs(color.astype(np.int32) - ocolor.astype(np.int32))
    assert diff.max() <= 1, "colour differs by %d at %d channels" % (int(diff.max()), int((diff > 1).sum()))
    return int((diff != 0).sum())


@pytest.mark.parametrize("shader", [scenes.SHADER_TEST_KTX, scenes.SHADER_PUSH_CONSTANTS, scenes.SHADER_DUMMY])
def test_config0_small(device, oracle, shader):
    sc = scenes.config0(320, 180, shader=shader, tex_size=128, segments=32, rings=16)
    mism = compare(sc, device, oracle)
    assert mism == 0


def test_config0_full(device, oracle):
    mism = compare(scenes.config0(), device, oracle)
    assert mism == 0


def test_grid_small(device, oracle):
    mism = compare(scenes.config1(480, 270, quads=300, tex_size=256), device, oracle)
    assert mism == 0


def test_instances_small(device, oracle):
    for b2f in (True, False):
        mism = compare(scenes.instanced_spheres(512, 288, 12, 8, 4, 2, back_to_front=b2f, tex_size=128), device, oracle)
        assert mism == 0


def test_config1_full(device, oracle):
    """BASELINE configs[1] at full size: 1920x1080, 1 002 528 triangles."""
    mism = compare(scenes.config1(), device, oracle, threads=16)
    assert mism == 0


def test_config2_full(device, oracle):
    """BASELINE configs[2] at full size: 3840x2160, 10 000 instanced meshes, ~4x overdraw,
    submitted back to front (the worst case for early depth rejection)."""
    mism = compare(scenes.config2(), device, oracle, threads=16)
    assert mism == 0


def test_config3_full_and_tile_shards(device, oracle):
    """BASELINE configs[3] at full size: 7680x4320, 40 000 instances (12.8 M triangles).  The whole
    frame against the oracle, and the property the multi-GPU mode rests on: the frames of two tile
    shards, composed by tile owner, are the unsharded frame."""
    sc = scenes.config3()
    r = SceneRunner(sc, device)
    try:
        color, depth = r.render()
        shards = []
        for k in range(2):
            r.cmd.set_tile_shard(k, 2)
            shards.append(r.render())
    finally:
        r.destroy()
    ocolor, odepth, _ = oracle.render_scene(sc, threads=16)
    assert np.array_equal(depth, odepth)
    assert np.abs(color.astype(np.int32) - ocolor.astype(np.int32)).max() <= 1
    tiles_x = (sc.width + 31) // 32
    ty, tx = np.mgrid[0:sc.height, 0:sc.width] // 32
    owner = (ty * tiles_x + tx) % 2
    for k, (c, d) in enumerate(shards):
        mine = owner == k
        assert np.array_equal(c[mine], color[mine]) and np.array_equal(d[mine], depth[mine])


def test_config3_full_bands(device):
    """The other way configs[3] is divided over GPUs, at full size: each of 8 ranks renders a band of whole rows
    of tiles as its render area and skips the draws outside it.  Every band's rows equal the unsharded frame's
    (checked against the oracle by the test above), and no rank keeps more than a quarter of the draws."""
    from renderer_amd import sharding
    sc = scenes.config3()
    r = SceneRunner(sc, device)
    try:
        color, depth = r.render()
    finally:
        r.destroy()
    kept = []
    for k in range(8):
        y0, y1 = sharding.band_rows(sc.height, k, 8)
        r = SceneRunner(sharding.band_scene(sc, k, 8), device)
        try:
            c, d = r.render()
            kept.append(r.cmd.visible_draws())
        finally:
            r.destroy()
        assert np.array_equal(c[y0:y1], color[y0:y1]) and np.array_equal(d[y0:y1], depth[y0:y1]), "band %d differs" % k
    assert sum(kept) >= len(sc.draws) and max(kept) <= len(sc.draws) // 4


_config4_mesh = []


@pytest.mark.parametrize("view", [0, 5, 16, 32, 48])
def test_config4_view_full(device, oracle, view):
    """Views of BASELINE configs[4] at full size: 1920x1080 over a 2237x2237-quad grid (10 008 338 triangles,
    most of them smaller than a pixel) from the 64-camera orbit: facing the grid (0), oblique (5), looking
    along the grid from inside its extent so that it straddles the near plane (16 and 48), and from
    behind, where every triangle is culled (32)."""
    if not _config4_mesh:
        _config4_mesh.append(scenes.noise_grid(2237, 3.3, 1.8, 0.35, 4))
    mism = compare(scenes.config4_view(view, mesh=_config4_mesh[0]), device, oracle, threads=16)
    assert mism == 0
