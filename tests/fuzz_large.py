#!/usr/bin/env python3
"""One-off randomised parity sweep at full target sizes (720p to 4K, up to a million triangles, up to 300 draws,
1 or 4 samples), each frame against the oracle.  Run on a GPU: python tests/fuzz_large.py [SEED...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
from renderer_amd import api, scenes
from renderer_amd.runner import SceneRunner
from oracle import oracle
dev = api.Device(0)
bad = 0
for seed in ([int(a) for a in sys.argv[1:]] or range(40)):
    rng = np.random.RandomState(900 + seed)
    w, h = int(rng.choice([1280, 1920, 2560, 3840])), int(rng.choice([720, 1080, 1440, 2160]))
    shader = [scenes.SHADER_PUSH_CONSTANTS, scenes.SHADER_TEST_KTX, scenes.SHADER_DUMMY][seed % 3]
    texture = None if shader == scenes.SHADER_PUSH_CONSTANTS else dict(texels=scenes.checker_texture(int(rng.choice([256, 1024])), seed), gen_mips=True, filtering=1, tiled=bool(seed & 1), fmt=43)
    sc = helpers.soup_scene(w, h, int(rng.choice([50000, 300000, 1000000])), seed, max_extent=float(rng.choice([0.003, 0.01, 0.05, 0.4])),
                            depth_levels=int(rng.choice([0, 4])), z_range=[(0.05, 0.95), (-0.2, 1.2)][seed & 1], shader=shader,
                            n_draws=int(rng.choice([1, 3, 300])), texture=texture, samples=int(rng.choice([1, 1, 4])),
                            pipeline_kw=dict(cull_mode=int(rng.randint(0, 3))))
    r = SceneRunner(sc, dev)
    try:
        color, depth = r.render()
    finally:
        r.destroy()
    oc, od, _ = oracle.render_scene(sc, threads=16)
    ok = np.array_equal(depth, od) and np.array_equal(color, oc)
    print(seed, w, h, sc.samples, sc.triangle_count(), len(sc.draws), "ok" if ok else "MISMATCH depth %d colour %d" % ((depth != od).sum(), (color != oc).sum()), flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
