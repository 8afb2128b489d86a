#!/usr/bin/env python3
"""Randomised parity sweep on a GPU over the state a frame is recorded with, each frame against the oracle: several
vertex buffers with different layouts (stride, attribute offsets, absent normal or uv), index buffers that share
vertices among triangles, non-zero buffer offsets, firstIndex and vertexOffset, several pipelines in one pass (shader,
cull state, layout), several uniform blocks (identity and the camera's), several textures (format, filter, mips,
tiling), per-draw transforms, a viewport away from the target's rectangle with a depth range inside [0,1], the four
colour formats, clear values, render area and scissor.

    python tests/fuzz_state.py [CASES] [FIRST_SEED]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from renderer_amd import api, hostmath as hm, scenes  # noqa: E402
from renderer_amd.runner import SceneRunner  # noqa: E402
from oracle import oracle  # noqa: E402

F = np.float32
SHADERS = [scenes.SHADER_PUSH_CONSTANTS, scenes.SHADER_TEST_KTX, scenes.SHADER_DUMMY]


def layout(rng, shader):
    """A vertex layout: the three attributes in a random order with random padding; the normal or the uv is left
    out at times, more often when the shader does not read it."""
    want_nrm = rng.rand() < (0.5 if shader == scenes.SHADER_DUMMY else 0.85)
    want_uv = rng.rand() < (0.5 if shader == scenes.SHADER_PUSH_CONSTANTS else 0.85)
    parts = [("pos", 12)] + ([("nrm", 12)] if want_nrm else []) + ([("uv", 8)] if want_uv else [])
    order = rng.permutation(len(parts))
    off, at = 0, {}
    for k in order:
        off += 4 * int(rng.randint(0, 3))
        at[parts[k][0]] = off
        off += parts[k][1]
    stride = off + 4 * int(rng.randint(0, 3))
    return dict(stride=stride, off_pos=at["pos"], off_nrm=at.get("nrm"), off_uv=at.get("uv"))


def fold(t):
    """Triangle wave: folds the reals onto [-1, 1]."""
    return np.abs((t + 1.0) % 4.0 - 2.0) - 1.0


def vertices(rng, n, lay, perspective, walk):
    """n vertices in lay's layout (rows of stride/4 floats, the unused words random).  With walk, positions follow a
    random walk along the buffer, so that triangles over neighbouring vertices are small; else each is anywhere."""
    words = lay["stride"] // 4
    vb = rng.uniform(-1, 1, size=(n, words)).astype(F)
    p = lay["off_pos"] // 4
    if walk:
        unit = fold(np.cumsum(rng.uniform(-1, 1, size=(n, 3)) * float(rng.choice([0.01, 0.04, 0.15])), axis=0) + rng.uniform(-1, 1, size=3))
    else:
        unit = rng.uniform(-1, 1, size=(n, 3))
    if perspective:
        vb[:, p:p + 3] = unit * 2.5 + np.array([0, 1, 0.5])
    else:
        vb[:, p] = unit[:, 0] * 1.3
        vb[:, p + 1] = 1.0 - unit[:, 1] * 1.3
        vb[:, p + 2] = 0.5 + unit[:, 2] * 0.7
    if lay["off_nrm"] is not None:
        q = lay["off_nrm"] // 4
        nrm = rng.normal(size=(n, 3)) + np.array([0, 1.2, -1.2])
        vb[:, q:q + 3] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    if lay["off_uv"] is not None:
        q = lay["off_uv"] // 4
        vb[:, q:q + 2] = rng.uniform(-0.5, 1.5, size=(n, 2))
    return vb


def indices(rng, n_tris, n_verts, local):
    """Triangles over n_verts vertices; with local, each triangle's corners lie near one another in the buffer
    (mesh-like reuse), else anywhere."""
    if local:
        base = rng.randint(0, n_verts, size=(n_tris, 1))
        idx = (base + rng.randint(0, min(n_verts, 12), size=(n_tris, 3))) % n_verts
    else:
        idx = rng.randint(0, n_verts, size=(n_tris, 3))
    return idx.astype(np.uint32).reshape(-1)


def case(seed):
    rng = np.random.RandomState(70000 + seed)
    w, h = int(rng.randint(8, 300)), int(rng.randint(8, 220))
    samples = int(rng.choice([1, 1, 4]))
    sc = scenes.Scene(w, h, samples=samples, color_format=int(rng.choice([37, 43, 44, 50])),
                      clear_color=tuple(float(c) for c in rng.rand(4)), clear_depth=float(rng.choice([1.0, 1.0, 1.0, 0.9, 0.5 + 0.5 * rng.rand()])))
    depth_kw = dict(depth_compare=int(rng.choice([1, 3])))
    if rng.rand() < 0.1:
        depth_kw.update(depth_test=False, depth_write=False)
    cam = hm.Camera()
    cam.eye_position = np.array([rng.uniform(-1, 1), rng.uniform(0, 2), -2.0 + rng.uniform(-1, 1)], dtype=F)
    sc.ubos = [helpers.identity_ubo(), hm.global_ubo(cam, hm.identity(), w, h)]
    for k in range(int(rng.randint(1, 4))):
        sc.textures.append(dict(texels=scenes.checker_texture(int(rng.choice([16, 24, 40, 128])), seed + k), gen_mips=bool(rng.randint(0, 2)),
                                filtering=int(rng.randint(0, 2)), tiled=bool(rng.randint(0, 2)), fmt=int(rng.choice([37, 43]))))
    n_sets = int(rng.randint(1, 4))
    for s in range(n_sets):
        shader = SHADERS[int(rng.randint(0, 3))]
        lay = layout(rng, shader)
        perspective = bool(rng.randint(0, 2))
        n_verts = int(rng.choice([3, 17, 200, 3000]))
        pad_rows = int(rng.randint(0, 5))
        vb = np.concatenate([rng.uniform(-9, 9, size=(pad_rows, lay["stride"] // 4)).astype(F), vertices(rng, n_verts, lay, perspective, bool(rng.randint(0, 2)))])
        sc.vbs.append(vb)
        n_tris = int(rng.choice([1, 30, 500, 5000]))
        pad_idx = int(rng.randint(0, 7))
        ib = np.concatenate([rng.randint(0, n_verts, size=pad_idx).astype(np.uint32), indices(rng, n_tris, n_verts, bool(rng.randint(0, 2)))])
        sc.ibs.append(ib)
        sc.pipelines.append(scenes.default_pipeline(shader, samples, cull_mode=int(rng.choice([0, 0, 1, 1, 2, 2, 3])), front_face=int(rng.randint(0, 2)),
                                                    **lay, **depth_kw))
        # the draws of this set: consecutive index ranges; vertexOffset moves a draw's window of the vertex buffer
        # and the indices are rebased so every draw still reads inside the buffer
        n_draws = int(rng.choice([1, 1, 2, 5]))
        per = (n_tris + n_draws - 1) // n_draws
        for k in range(n_draws):
            first = k * per
            cnt = min(per, n_tris - first)
            if cnt <= 0:
                break
            rows_off = int(rng.randint(0, pad_rows + 1))      # whole rows skipped by the binding's byte offset
            vertex_offset = pad_rows - rows_off                  # and the rest by vertexOffset
            idx_off = int(rng.randint(0, pad_idx + 1))
            m = hm.identity()
            if rng.rand() < 0.5:
                m = hm.translate(m, [float(t) for t in rng.uniform(-0.3, 0.3, size=3)])
                m = hm.scale(m, [float(t) for t in rng.uniform(0.5, 1.5, size=3)])
            sc.draws.append(dict(pipeline=s, vb=s, vb_offset=rows_off * lay["stride"], ib=s, ib_offset=idx_off * 4,
                                 ubo=1 if perspective else 0,
                                 texture=-1 if shader == scenes.SHADER_PUSH_CONSTANTS else int(rng.randint(0, len(sc.textures))),
                                 push=scenes.push_block([0.3 + 0.7 * rng.rand(), 0.3 + 0.7 * rng.rand(), 0.3 + 0.7 * rng.rand(), rng.rand()], m),
                                 index_count=cnt * 3, first_index=pad_idx - idx_off + first * 3, vertex_offset=vertex_offset))
    order = rng.permutation(len(sc.draws))
    sc.draws = [sc.draws[k] for k in order]
    if rng.rand() < 0.5:
        z0, z1 = sorted(float(t) for t in rng.rand(2))
        if rng.rand() < 0.2:
            z0, z1 = z1, z0
        sc.viewport = (float(rng.uniform(-0.3, 0.3) * w), float(rng.uniform(-0.3, 0.3) * h),
                       float(rng.uniform(0.4, 1.5) * w), float(rng.uniform(0.4, 1.5) * h), z0, z1)
    if rng.rand() < 0.3:
        ax, ay = int(rng.randint(0, max(w // 2, 1))), int(rng.randint(0, max(h // 2, 1)))
        sc.area = (ax, ay, int(rng.randint(1, w - ax + 1)), int(rng.randint(1, h - ay + 1)))
    if rng.rand() < 0.3:
        sx, sy = int(rng.randint(0, w)), int(rng.randint(0, h))
        sc.scissor = (sx, sy, int(rng.randint(0, w - sx + 1)), int(rng.randint(0, h - sy + 1)))
    return sc


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
        except api.RendererError as e:
            bad += 1
            print("seed %d: ERROR %s (%d tris in %d draws, area %s scissor %s)" % (seed, e, sc.triangle_count(), len(sc.draws), sc.area, sc.scissor), flush=True)
            continue
        finally:
            r.destroy()
        oc, od, _ = oracle.render_scene(sc, threads=8)
        ax, ay, aw, ah = sc.area
        sl = (slice(ay, ay + ah), slice(ax, ax + aw))
        if not (np.array_equal(depth[sl], od[sl]) and np.array_equal(color[sl], oc[sl])):
            bad += 1
            print("seed %d: MISMATCH depth %d colour %d (%dx%d s%d fmt %d, %d tris in %d draws, viewport %s area %s scissor %s)" % (
                seed, int((depth[sl] != od[sl]).sum()), int((color[sl] != oc[sl]).sum()), sc.width, sc.height, sc.samples,
                sc.color_format, sc.triangle_count(), len(sc.draws), sc.viewport, sc.area, sc.scissor), flush=True)
        elif seed % 25 == 0:
            print("seed %d ok" % seed, flush=True)
    dev.destroy()
    print("%d cases, %d mismatches" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
