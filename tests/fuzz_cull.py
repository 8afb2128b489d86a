#!/usr/bin/env python3
"""Randomised check of whole-draw culling and render-area tile grids against the oracle (run on a GPU):
instanced meshes under random cameras, random per-draw transforms (many partly or wholly outside the frustum,
some crossing the near plane, some enclosing the camera), random render areas and scissors.

    python tests/fuzz_cull.py [CASES] [FIRST_SEED]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from renderer_amd import api, hostmath as hm, scenes  # noqa: E402
from renderer_amd.runner import SceneRunner  # noqa: E402
from oracle import oracle  # noqa: E402


def case(seed):
    rng = np.random.RandomState(seed)
    w, h = int(rng.randint(40, 260)), int(rng.randint(30, 200))
    samples = int(rng.choice([1, 1, 4]))
    shader = [scenes.SHADER_TEST_KTX, scenes.SHADER_PUSH_CONSTANTS, scenes.SHADER_DUMMY][seed % 3]
    sc = scenes.instanced_spheres(w, h, 2, 2, 1, seed, tex_size=32, shader=shader)
    sc.samples = samples
    for p in sc.pipelines:
        p["samples"] = samples
        p["cull_mode"] = int(rng.randint(0, 4))
    cam = hm.Camera()
    cam.eye_position = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), -2.0 + rng.uniform(-1, 1)], dtype=np.float32)
    model = hm.identity()
    sc.ubos[0] = hm.global_ubo(cam, model, w, h)
    base = sc.draws[0]
    sc.draws = []
    n_draws = int(rng.randint(2, 40))
    big = None
    if seed % 5 == 4:
        # a pass of hundreds of draws: the per-draw kernel lists the chunks of the visible ones, and the vertex and
        # setup kernels walk the lists; some draws of a mesh of 41 / 80 chunks, whose chunks the whole wave lists
        n_draws = int(rng.randint(256, 600))
        v, i = scenes.icosphere(4, radius=1.0)
        sc.vbs.append(v)
        sc.ibs.append(i)
        big = dict(base, vb=1, ib=1, index_count=int(i.size))
    for k in range(n_draws):
        kind = rng.randint(0, 6)
        s = float(np.exp(rng.uniform(np.log(0.05), np.log(3.0))))
        t = rng.uniform(-1, 1, size=3) * [4.0, 4.0, 3.0]
        if kind == 0:
            t *= 6.0                       # mostly far outside
        elif kind == 1:
            t = cam.eye_position + rng.uniform(-0.3, 0.3, size=3)  # around the eye: crosses the near plane
        elif kind == 2:
            s *= 40.0                      # encloses the whole frustum
        elif kind == 3:
            t[2] = rng.uniform(60, 140)    # around the far plane
        m = hm.translate(hm.identity(), [float(t[0]), float(1.0 + t[1]), float(t[2])])
        m = hm.scale(m, [s, s * float(rng.uniform(0.3, 1.5)), s])
        d = dict(big if big is not None and k % 97 == 5 else base)
        d["push"] = scenes.push_block([rng.rand(), rng.rand(), rng.rand(), 1.0], m)
        if shader == scenes.SHADER_PUSH_CONSTANTS:
            d["texture"] = -1
        sc.draws.append(d)
    if rng.rand() < 0.6:
        ax, ay = int(rng.randint(0, w // 2)), int(rng.randint(0, h // 2))
        aw, ah = int(rng.randint(1, w - ax + 1)), int(rng.randint(1, h - ay + 1))
        sc.area = (ax, ay, aw, ah)
    if rng.rand() < 0.6:
        sx, sy = int(rng.randint(0, w)), int(rng.randint(0, h))
        sc.scissor = (sx, sy, int(rng.randint(0, w - sx + 1)), int(rng.randint(0, h - sy + 1)))
    return sc


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = api.Device(0)
    bad = 0
    culled_total = draws_total = 0
    for seed in range(first, first + cases):
        sc = case(seed)
        r = SceneRunner(sc, dev)
        try:
            color, depth = r.render()
            kept = r.cmd.visible_draws()
            if len(sc.draws) >= 256:  # the next frame counts into the other pair of list lengths
                color2, depth2 = r.render()
                if not (np.array_equal(color, color2) and np.array_equal(depth, depth2)):
                    print("seed %d: the second frame of the command buffer differs from the first" % seed, flush=True)
                    bad += 1
        finally:
            r.destroy()
        oc, od, _ = oracle.render_scene(sc, threads=8)
        # pixels outside the render area are never written by either side; compare inside it
        ax, ay, aw, ah = sc.area
        sl = (slice(ay, ay + ah), slice(ax, ax + aw))
        ok = np.array_equal(depth[sl], od[sl]) and np.array_equal(color[sl], oc[sl])
        culled_total += len(sc.draws) - kept
        draws_total += len(sc.draws)
        if not ok:
            bad += 1
            print("seed %d: MISMATCH depth %d colour %d (draws %d kept %d, %dx%d s%d area %s scissor %s)" % (
                seed, int((depth[sl] != od[sl]).sum()), int((color[sl] != oc[sl]).sum()), len(sc.draws), kept,
                sc.width, sc.height, sc.samples, sc.area, sc.scissor), flush=True)
        elif seed % 20 == 0:
            print("seed %d ok (draws %d kept %d)" % (seed, len(sc.draws), kept), flush=True)
    dev.destroy()
    print("%d cases, %d mismatches, %d of %d draws culled" % (cases, bad, culled_total, draws_total))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
