#!/usr/bin/env python3
"""What a change of the oracle's text changed in its output: for each listed commit, the oracle as it was before the
commit and as it was after it are both built (from git history, into /tmp) and render every scene of
tests/golden_scenes.py; the D24 words and colour bytes that differ are counted.

    python tools/oracle_change_counts.py [COMMIT ...]        (default: the two round-2 commits that restated the oracle)

Writes tests/golden/oracle_change_counts.json.  The rule this serves: a kernel reformulation that needs
oracle/raster_oracle.cpp edited is accepted only with tests/test_oracle_independent.py green AND these counts committed."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_scenes  # noqa: E402
from oracle import oracle  # noqa: E402

FLAGS = "-O2 -std=c++17 -fPIC -march=x86-64-v3 -ffp-contract=off -fno-fast-math -pthread".split()


def build(rev, tag):
    d = "/tmp/oracle_%s" % tag
    os.makedirs(d, exist_ok=True)
    for name in ("raster_oracle.cpp", "raster_oracle.h"):
        with open(os.path.join(d, name), "wb") as f:
            f.write(subprocess.check_output(["git", "show", "%s:oracle/%s" % (rev, name)], cwd=ROOT))
    so = os.path.join(d, "libraster_oracle.so")
    subprocess.check_call(["g++"] + FLAGS + ["-shared", "-o", so, os.path.join(d, "raster_oracle.cpp")])
    return so


def render_all(so):
    oracle.LIB_PATH = so
    oracle._lib = None
    out = {}
    for name, make in golden_scenes.SCENES.items():
        color, depth, _ = oracle.render_scene(make(), threads=4)
        out[name] = (color, depth)
    return out


def main(commits):
    result = {}
    for c in commits:
        subject = subprocess.check_output(["git", "log", "-1", "--format=%s", c], cwd=ROOT, text=True).strip()
        before, after = render_all(build(c + "^", c + "_before")), render_all(build(c, c + "_after"))
        per_scene = {}
        for name in before:
            (c0, d0), (c1, d1) = before[name], after[name]
            dc = np.abs(c0.astype(np.int32) - c1.astype(np.int32))
            per_scene[name] = {"samples": int(d0.size), "depth_words_changed": int((d0 != d1).sum()),
                               "largest_depth_change": int(np.abs(d0.astype(np.int64) - d1.astype(np.int64)).max()),
                               "colour_bytes": int(c0.size), "colour_bytes_changed": int((dc != 0).sum()), "largest_colour_change": int(dc.max())}
            print(c, name, per_scene[name], flush=True)
        result[c] = {"subject": subject[:160], "scenes": per_scene}
    with open(os.path.join(ROOT, "tests", "golden", "oracle_change_counts.json"), "w") as f:
        json.dump(result, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:] or ["bfcaf33", "d6b43d9"])
