"""A bounded run of the randomised parity sweeps fuzz_soup.py and fuzz_cull.py (the same generators, fewer cases): triangle
soups over all the knobs together, and instanced meshes under random cameras, transforms, render areas and
scissors (whole-draw culling)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_clip  # noqa: E402
import fuzz_cull  # noqa: E402
import fuzz_shards  # noqa: E402
import fuzz_soup  # noqa: E402
import fuzz_state  # noqa: E402
from renderer_amd.runner import SceneRunner  # noqa: E402

pytestmark = pytest.mark.gpu


def inside_area_equal(sc, device, oracle):
    r = SceneRunner(sc, device)
    try:
        color, depth = r.render()
    finally:
        r.destroy()
    oc, od, _ = oracle.render_scene(sc, threads=8)
    ax, ay, aw, ah = sc.area
    sl = (slice(ay, ay + ah), slice(ax, ax + aw))
    return np.array_equal(depth[sl], od[sl]) and np.array_equal(color[sl], oc[sl])


@pytest.mark.parametrize("block", range(4))
def test_soup_sweep(device, oracle, block):
    bad = [s for s in range(1000 + 25 * block, 1025 + 25 * block) if not inside_area_equal(fuzz_soup.case(s), device, oracle)]
    assert not bad, "seeds that differ from the oracle: %s" % bad


@pytest.mark.parametrize("block", range(4))
def test_culling_sweep(device, oracle, block):
    bad = [s for s in range(5000 + 40 * block, 5040 + 40 * block) if not inside_area_equal(fuzz_cull.case(s), device, oracle)]
    assert not bad, "seeds that differ from the oracle: %s" % bad


@pytest.mark.parametrize("seed", [45003])
def test_soup_seeds_that_once_differed(device, oracle, seed):
    """45003: no depth test, a sliver clipped by the near and the far plane whose fan triangles overlap after snapping --
    the last fan triangle submitted owns the pixel, like the last primitive."""
    assert inside_area_equal(fuzz_soup.case(seed), device, oracle)


def test_shards_compose_the_frame_sweep(device):
    """fuzz_shards.py, bounded: soups as interleaved tile shards and instanced scenes as bands, composed, against the
    same frames rendered whole."""
    bad = []
    for seed in range(2000, 2040):
        n = int(np.random.RandomState(70000 + seed).choice([2, 3, 4, 5, 7, 8]))
        if seed % 2 == 0:
            sc = fuzz_soup.case(seed)
            sc.area = (0, 0, sc.width, sc.height)
            ok = fuzz_shards.tiles_compose(sc, device, n)
        else:
            sc = fuzz_cull.case(seed)
            sc.area = (0, 0, sc.width, sc.height)
            sc.scissor = (0, 0, sc.width, sc.height)
            ok = fuzz_shards.bands_compose(sc, device, n)
        if not ok:
            bad.append(seed)
    assert not bad, "seeds whose shards do not compose the frame: %s" % bad


def test_clipped_slivers_sweep(device, oracle):
    """fuzz_clip.py, bounded: soups across both clip planes, nothing culled, every depth mode, one and four samples.  The
    listed seeds are cases the build before the fan triangles' tie order followed the submission order got wrong."""
    seeds = [34, 247, 408, 421, 857, 1175, 1328, 3991, 4127, 4717] + list(range(20000, 20030))
    bad = [s for s in seeds if not inside_area_equal(fuzz_clip.case(s), device, oracle)]
    assert not bad, "seeds that differ from the oracle: %s" % bad


def test_recorded_state_sweep(device, oracle):
    """fuzz_state.py, bounded: vertex layouts, shared vertices, buffer offsets, firstIndex and vertexOffset, several
    pipelines, uniform blocks and textures in one pass, viewports with a depth range, colour formats, clear values."""
    bad = [s for s in range(60) if not inside_area_equal(fuzz_state.case(s), device, oracle)]
    assert not bad, "seeds that differ from the oracle: %s" % bad



@pytest.mark.parametrize("seed", [0, 2, 5, 7, 11])
def test_sequences_of_interleaved_submits(oracle, seed):
    """fuzz_sequence.py, five fixed sequences: several recorded scenes alive on one device, one to three frames in flight
    each, submits interleaved without waiting, uniform blocks of different buffers rewritten between submits (each such
    write makes the device's stream follow every frame in flight), scenes recorded again.  Sequences 2 and 7 are the ones
    that once never came back from rnd_device_wait_idle.  The device is made with RND_DEBUG_SYNC=1: a stream that does not
    drain is reported by name instead of the call blocking."""
    import fuzz_sequence
    from renderer_amd import api
    kept = {k: os.environ.get(k) for k in ("RND_DEBUG_SYNC", "RND_DEBUG_SYNC_MS")}
    os.environ["RND_DEBUG_SYNC"] = "1"
    os.environ["RND_DEBUG_SYNC_MS"] = "30000"
    try:
        dev = api.Device(0)
    finally:
        for k, v in kept.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    bad = fuzz_sequence.run(seed, dev)  # a stream reported stuck raises here: the device is then left alone, not destroyed (destroying waits for it)
    dev.destroy()
    assert not bad, "images that differ from the oracle: %s" % bad
