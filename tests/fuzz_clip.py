#!/usr/bin/env python3
"""Randomised parity sweep on a GPU aimed at clipped primitives: triangle soups whose depth range crosses both the near
and the far plane, nothing culled, thin and small triangles (their clipped polygons' fan triangles fold over one another
once snapped), every depth mode, one and four samples, each frame against the oracle.

    python tests/fuzz_clip.py [CASES] [FIRST_SEED]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from renderer_amd import api, scenes  # noqa: E402
from renderer_amd.runner import SceneRunner  # noqa: E402
from oracle import oracle  # noqa: E402


def case(seed):
    rng = np.random.RandomState(130000 + seed)
    w, h = int(rng.randint(8, 300)), int(rng.randint(8, 200))
    shader = [scenes.SHADER_PUSH_CONSTANTS, scenes.SHADER_TEST_KTX, scenes.SHADER_DUMMY][int(rng.randint(0, 3))]
    texture = None
    if shader != scenes.SHADER_PUSH_CONSTANTS:
        texture = dict(texels=scenes.checker_texture(int(rng.choice([16, 40, 128])), seed), gen_mips=True, filtering=1,
                       tiled=bool(rng.randint(0, 2)), fmt=int(rng.choice([37, 43])))
    mode = int(rng.randint(0, 3))
    kw = dict(cull_mode=0, front_face=int(rng.randint(0, 2)), depth_compare=[1, 3, 3][mode])
    if mode == 2:
        kw.update(depth_test=False, depth_write=False)
    return helpers.soup_scene(w, h, int(rng.choice([200, 2000, 12000])), seed, max_extent=float(rng.choice([0.004, 0.011, 0.03, 0.3])),
                              depth_levels=int(rng.choice([0, 1, 3])), z_range=(-0.4, 1.4), shader=shader,
                              n_draws=int(rng.choice([1, 2, 6])), texture=texture, samples=int(rng.choice([1, 4])), pipeline_kw=kw)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = api.Device(0)
    bad = 0
    for seed in range(first, first + cases):
        sc = case(seed)
        r = SceneRunner(sc, dev)
        try:
            color, depth = r.render()
        finally:
            r.destroy()
        oc, od, _ = oracle.render_scene(sc, threads=8)
        if not (np.array_equal(depth, od) and np.array_equal(color, oc)):
            bad += 1
            print("seed %d: MISMATCH depth %d colour %d (%dx%d s%d, %d tris in %d draws, %s)" % (
                seed, int((depth != od).sum()), int((color != oc).sum()), sc.width, sc.height, sc.samples, sc.triangle_count(),
                len(sc.draws), sc.pipelines[0]), flush=True)
        elif seed % 50 == 0:
            print("seed %d ok" % seed, flush=True)
    dev.destroy()
    print("%d cases, %d mismatches" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
