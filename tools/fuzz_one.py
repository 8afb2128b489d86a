#!/usr/bin/env python3
"""One case of tests/fuzz_soup.py by seed, with the pixels that differ from the oracle:  python tools/fuzz_one.py SEED"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_soup  # noqa: E402
from renderer_amd import api  # noqa: E402
from renderer_amd.runner import SceneRunner  # noqa: E402
from oracle import oracle  # noqa: E402

seed = int(sys.argv[1])
sc = fuzz_soup.case(seed)
dev = api.Device(0)
r = SceneRunner(sc, dev)
color, depth = r.render()
r.destroy()
oc, od, _ = oracle.render_scene(sc, threads=8)
print("lib", os.environ.get("RENDERER_AMD_LIB", "in-tree"), "seed", seed, sc.width, sc.height, "samples", sc.samples, "tris", sc.triangle_count(),
      "pipelines", [{k: p[k] for k in p if k in ("shader", "samples", "cull_mode", "depth_compare")} for p in sc.pipelines][:2],
      "textures", [(t["texels"].shape, t.get("fmt"), t.get("filtering"), t.get("tiled"), t.get("gen_mips")) for t in sc.textures][:2])
ys, xs = np.nonzero((color != oc).any(axis=2))
for y, x in list(zip(ys, xs))[:10]:
    print("pixel", x, y, "gpu", color[y, x], "oracle", oc[y, x], "depth", depth[y, x], od[y, x])
print("colour channels differing:", int((color != oc).sum()), "depth words differing:", int((depth != od).sum()))
