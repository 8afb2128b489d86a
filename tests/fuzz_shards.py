#!/usr/bin/env python3
"""Randomised sweep of the two ways one frame is divided among devices, on one GPU and without the oracle: the soups of
fuzz_soup.py rendered as n interleaved tile shards (rnd_cmd_set_tile_shard) and the instanced scenes of fuzz_cull.py
rendered as n bands (sharding.band_scene), each composed and compared with the same frame rendered whole.

    python tests/fuzz_shards.py [CASES] [FIRST_SEED]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_cull  # noqa: E402
import fuzz_soup  # noqa: E402
from renderer_amd import api, sharding  # noqa: E402
from renderer_amd.runner import SceneRunner  # noqa: E402


def render(sc, dev, shard=None):
    r = SceneRunner(sc, dev)
    try:
        if shard is not None:
            r.cmd.set_tile_shard(*shard)
        return r.render()
    finally:
        r.destroy()


def tiles_compose(sc, dev, n):
    full_c, full_d = render(sc, dev)
    acc_c, acc_d = np.zeros_like(full_c), np.zeros_like(full_d)
    tiles_x = (sc.width + 31) // 32
    for k in range(n):
        c, d = render(sc, dev, (k, n))
        for ty in range((sc.height + 31) // 32):
            for tx in range(tiles_x):
                if (ty * tiles_x + tx) % n == k:
                    sl = (slice(ty * 32, ty * 32 + 32), slice(tx * 32, tx * 32 + 32))
                    acc_c[sl], acc_d[sl] = c[sl], d[sl]
    return np.array_equal(acc_c, full_c) and np.array_equal(acc_d, full_d)


def bands_compose(sc, dev, n):
    full_c, full_d = render(sc, dev)
    ok = True
    for k in range(n):
        y0, y1 = sharding.band_rows(sc.height, k, n)
        if y1 <= y0:
            continue
        c, d = render(sharding.band_scene(sc, k, n), dev)
        ok = ok and np.array_equal(c[y0:y1], full_c[y0:y1]) and np.array_equal(d[y0:y1], full_d[y0:y1])
    return ok


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = api.Device(0)
    bad = 0
    for seed in range(first, first + cases):
        rng = np.random.RandomState(70000 + seed)
        n = int(rng.choice([2, 3, 4, 5, 7, 8]))
        if seed % 2 == 0:
            sc, what, ok = fuzz_soup.case(seed), "tiles", None
            if sc.area != (0, 0, sc.width, sc.height):
                sc.area = (0, 0, sc.width, sc.height)  # a shard leaves the other shards' tiles alone; the composition below takes whole tiles
            ok = tiles_compose(sc, dev, n)
        else:
            sc, what = fuzz_cull.case(seed), "bands"
            sc.area = (0, 0, sc.width, sc.height)
            sc.scissor = (0, 0, sc.width, sc.height)
            ok = bands_compose(sc, dev, n)
        if not ok:
            bad += 1
            print("seed %d: %s of %d do not compose the frame (%dx%d s%d, %d tris in %d draws)" % (
                seed, what, n, sc.width, sc.height, sc.samples, sc.triangle_count(), len(sc.draws)), flush=True)
        elif seed % 50 == 0:
            print("seed %d ok" % seed, flush=True)
    dev.destroy()
    print("%d cases, %d mismatches" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
