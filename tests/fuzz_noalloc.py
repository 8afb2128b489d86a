#!/usr/bin/env python3
"""Randomised sweep of rnd_draw_indexed_noalloc on a GPU: one-draw triangle soups over the knobs of fuzz_soup.py (within
the entry point's limits), each through the entry point that allocates nothing and through the recorded path, calls in
a row on one device so that every call meets what the call before left in the static array.

    python tests/fuzz_noalloc.py [CASES] [FIRST_SEED]
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
from test_zz_no_alloc import noalloc_render  # noqa: E402


def case(seed):
    rng = np.random.RandomState(90000 + seed)
    w, h = int(rng.randint(1, 420)), int(rng.randint(1, 300))
    shader = [scenes.SHADER_PUSH_CONSTANTS, scenes.SHADER_TEST_KTX, scenes.SHADER_DUMMY][int(rng.randint(0, 3))]
    texture = None
    if shader != scenes.SHADER_PUSH_CONSTANTS:
        texture = dict(texels=scenes.checker_texture(int(rng.choice([16, 24, 40, 128, 200])), seed), gen_mips=bool(rng.randint(0, 2)),
                       filtering=int(rng.randint(0, 2)), tiled=bool(rng.randint(0, 2)), fmt=int(rng.choice([37, 43])))
    kw = dict(cull_mode=int(rng.randint(0, 4)), front_face=int(rng.randint(0, 2)), depth_compare=int(rng.choice([1, 3])))
    if rng.rand() < 0.1:
        kw.update(depth_test=False, depth_write=False)
    sc = helpers.soup_scene(w, h, int(rng.choice([1, 40, 700, 6000, 20000])), seed,
                            max_extent=float(rng.choice([0.011, 0.02, 0.05, 0.3, 2.5])),
                            depth_levels=int(rng.choice([0, 0, 1, 3])), z_range=[(0.05, 0.95), (-0.3, 1.3)][int(rng.randint(0, 2))],
                            shader=shader, n_draws=1, texture=texture, samples=int(rng.choice([1, 1, 4])), pipeline_kw=kw)
    if rng.rand() < 0.3:
        sx, sy = int(rng.randint(0, w)), int(rng.randint(0, h))
        sc.scissor = (sx, sy, int(rng.randint(0, w - sx + 1)), int(rng.randint(0, h - sy + 1)))
    return sc


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = api.Device(0)
    bad = refused = 0
    for seed in range(first, first + cases):
        sc = case(seed)
        r = SceneRunner(sc, dev)
        try:
            ec, ed = r.render()
            try:
                gc, gd = noalloc_render(dev, sc, r)
            except api.RendererError as e:
                # the static array holds 16 384 fan triangles of clipped primitives and 32 768 pass-wide entries: a soup
                # beyond that is reported at the fence, as documented, and the next call starts clean
                if "overflowed" not in str(e):
                    raise
                refused += 1
                continue
        finally:
            r.destroy()
        if not (np.array_equal(gd, ed) and np.array_equal(gc, ec)):
            bad += 1
            print("seed %d: MISMATCH depth %d colour %d (%dx%d s%d, %d tris)" % (
                seed, int((gd != ed).sum()), int((gc != ec).sum()), sc.width, sc.height, sc.samples, sc.triangle_count()), flush=True)
        elif seed % 50 == 0:
            print("seed %d ok" % seed, flush=True)
    dev.destroy()
    print("%d cases, %d mismatches, %d beyond the static array's lists (reported)" % (cases, bad, refused))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
