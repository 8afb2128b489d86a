#!/usr/bin/env python3
"""Regenerates the fixtures in this directory from the CPU oracle.

    python tests/golden/make_golden.py

The reference ships no tests, golden images or fixtures for the draw path (parity
unpinned), so these vectors are the repository's own: inputs are rebuilt from seeds by
tests/golden_scenes.py, expected outputs are what oracle/ produced when this script ran.
They pin the oracle against drift (compiler, flags, refactors) and give the GPU tests a
second, file-backed reference."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_scenes  # noqa: E402
from oracle import oracle  # noqa: E402
from renderer_amd import hostmath as hm  # noqa: E402


def main():
    for name, build in golden_scenes.SCENES.items():
        sc = build()
        color, depth, prim = oracle.render_scene(sc, want_prim=True)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), color=color, depth=depth, prim=prim)
        print(name, color.shape, "covered", int((prim != 0xFFFFFFFF).sum()))
    lib = oracle.load()
    kat = {
        "srgb_decode_bits": [int(np.float32(lib.orc_srgb_decode(k)).view(np.uint32)) for k in range(256)],
        "depth24": {repr(z): int(lib.orc_depth24(z)) for z in [0.0, 1e-7, 0.25, 0.5, 0.75, 0.999999, 1.0]},
        "snap": {repr(v): int(lib.orc_snap(v)) for v in [0.0, 0.001953125, 0.5, 1.0, 719.99609375, -0.75, 1919.5]},
        "log2_bits": {repr(v): int(np.float32(lib.orc_log2(v)).view(np.uint32)) for v in [0.3, 1.0, 1.5, 2.0, 7.25, 1000.0]},
    }
    # sRGB encode: for every code k >= 1 the bits of the smallest float that encodes to it (floats of one sign order as
    # their bit patterns, and the encoder is monotone: bisection over the bits)
    first = []
    for k in range(1, 256):
        lo, hi = 0, 0x3F800000  # encode(bits lo) < k <= encode(bits hi)
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if lib.orc_srgb_encode(float(np.uint32(mid).view(np.float32))) >= k:
                hi = mid
            else:
                lo = mid
        first.append(hi)
    kat["srgb_encode_first_bits"] = first
    # texture(): sample points on small textures -- CLAMP_TO_EDGE and REPEAT outside [0, 1], a non-power-of-two size,
    # footprints from magnified to beyond the last level ((uint32)log2(max(w, h)) levels, tools.cpp:514)
    import ctypes as C
    from renderer_amd import scenes
    rng = np.random.RandomState(2024)
    points = []
    for (w, h, tiled, srgb) in [(64, 64, 0, 1), (64, 64, 1, 1), (40, 24, 0, 0), (17, 33, 1, 1)]:
        levels = oracle.mip_chain(scenes.checker_texture(64, 3)[:h, :w].copy(), True, bool(srgb))
        t = oracle.OrcTexture(w, h, len(levels), srgb, tiled, 1)
        for i, l in enumerate(levels):
            t.level_data[i] = l.ctypes.data_as(C.c_void_p).value
        for k in range(8):
            u, v = rng.uniform(-0.6, 1.6, 2)
            foot = 2.0 ** (k * 1.5 - 3.0) / max(w, h)
            args = np.array([u, v, foot, 0.3 * foot, -0.2 * foot, 0.8 * foot], dtype=np.float32)
            out = (C.c_float * 4)()
            lib.orc_sample_texture(C.byref(t), *[C.c_float(float(a)) for a in args], out)
            points.append({"texture": [w, h, tiled, srgb], "levels": len(levels), "args_bits": [int(b) for b in args.view(np.uint32)],
                           "rgba_bits": [int(b) for b in np.array(list(out), dtype=np.float32).view(np.uint32)]})
    kat["sample_points"] = points
    mats = {}
    for (w, h) in [(800, 600), (1280, 720)]:
        cam = hm.Camera()
        ubo = hm.global_ubo(cam, hm.scale(hm.identity(), [0.12, 0.12, 0.12]), w, h)
        mats["global_ubo_%dx%d" % (w, h)] = [int(x) for x in ubo.view(np.uint32)]
        mats["perspective_rh_zo_%dx%d" % (w, h)] = [int(x) for x in hm.perspective_fov(hm.radians(90.0), w, h, 0.01, 100.0, True).reshape(16).view(np.uint32)]
    for j in range(10):
        mats["ring_transform_%d" % j] = [int(x) for x in hm.ring_transform(j).reshape(16).view(np.uint32)]
    kat["matrices_float_bits"] = mats
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)
    print("known_answers.json written")


if __name__ == "__main__":
    main()
