"""File-backed vectors: the oracle must reproduce tests/golden/*.npz and known_answers.json
(CPU), and the GPU path must match the same files (gpu)."""
import json
import os

import numpy as np
import pytest

import golden_scenes
from renderer_amd import hostmath as hm

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(golden_scenes.SCENES))
def test_oracle_reproduces_golden(oracle, name):
    ref = np.load(os.path.join(GOLD, name + ".npz"))
    color, depth, prim = oracle.render_scene(golden_scenes.SCENES[name](), want_prim=True, threads=2)
    assert np.array_equal(depth, ref["depth"])
    assert np.array_equal(prim, ref["prim"])
    assert np.array_equal(color, ref["color"])
    assert (ref["prim"] != 0xFFFFFFFF).any()


def test_known_answers(oracle):
    kat = json.load(open(os.path.join(GOLD, "known_answers.json")))
    lib = oracle.load()
    assert [int(np.float32(lib.orc_srgb_decode(k)).view(np.uint32)) for k in range(256)] == kat["srgb_decode_bits"]
    for z, d in kat["depth24"].items():
        assert lib.orc_depth24(float(z)) == d
    for v, s in kat["snap"].items():
        assert lib.orc_snap(float(v)) == s
    for v, b in kat["log2_bits"].items():
        assert int(np.float32(lib.orc_log2(float(v))).view(np.uint32)) == b
    # sRGB encode table: code k starts at exactly the recorded float, and decode / encode round-trip
    for k, bits in enumerate(kat["srgb_encode_first_bits"], start=1):
        assert lib.orc_srgb_encode(float(np.uint32(bits).view(np.float32))) == k
        assert lib.orc_srgb_encode(float(np.uint32(bits - 1).view(np.float32))) == k - 1
        assert lib.orc_srgb_encode(lib.orc_srgb_decode(k)) == k
    # texture() sample points
    import ctypes as C
    from renderer_amd import scenes
    cache = {}
    for pt in kat["sample_points"]:
        w, h, tiled, srgb = pt["texture"]
        if (w, h, srgb) not in cache:
            cache[(w, h, srgb)] = oracle.mip_chain(scenes.checker_texture(64, 3)[:h, :w].copy(), True, bool(srgb))
        levels = cache[(w, h, srgb)]
        assert len(levels) == pt["levels"]
        t = oracle.OrcTexture(w, h, len(levels), srgb, tiled, 1)
        for i, l in enumerate(levels):
            t.level_data[i] = l.ctypes.data_as(C.c_void_p).value
        args = np.array(pt["args_bits"], dtype=np.uint32).view(np.float32)
        out = (C.c_float * 4)()
        lib.orc_sample_texture(C.byref(t), *[C.c_float(float(a)) for a in args], out)
        assert [int(b) for b in np.array(list(out), dtype=np.float32).view(np.uint32)] == pt["rgba_bits"]
    mats = kat["matrices_float_bits"]
    for (w, h) in [(800, 600), (1280, 720)]:
        ubo = hm.global_ubo(hm.Camera(), hm.scale(hm.identity(), [0.12, 0.12, 0.12]), w, h)
        assert [int(x) for x in ubo.view(np.uint32)] == mats["global_ubo_%dx%d" % (w, h)]
    for j in range(10):
        assert [int(x) for x in hm.ring_transform(j).reshape(16).view(np.uint32)] == mats["ring_transform_%d" % j]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(golden_scenes.SCENES))
def test_gpu_matches_golden(device, name):
    from renderer_amd.runner import SceneRunner
    ref = np.load(os.path.join(GOLD, name + ".npz"))
    r = SceneRunner(golden_scenes.SCENES[name](), device)
    try:
        color, depth = r.render()
    finally:
        r.destroy()
    assert np.array_equal(depth, ref["depth"])
    assert np.abs(color.astype(int) - ref["color"].astype(int)).max() <= 1
    assert np.array_equal(color, ref["color"])
