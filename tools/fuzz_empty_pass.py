#!/usr/bin/env python3
"""Frames of passes that launch no geometry kernel (empty scissor), one scene after another on one device: the tile
kernel then reads list counters that only the recording has zeroed.  Counts the frames that end in an error.

    python tools/fuzz_empty_pass.py [CASES] [FIRST_SEED]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_state  # noqa: E402
from renderer_amd import api  # noqa: E402
from renderer_amd.runner import SceneRunner  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = api.Device(0)
    errors = 0
    for seed in range(first, first + cases):
        sc = fuzz_state.case(seed)
        if seed % 2:
            sc.scissor = (0, 0, 0, 0)
        r = SceneRunner(sc, dev)
        try:
            r.render()
        except api.RendererError as e:
            errors += 1
            print("seed %d: %s" % (seed, e), flush=True)
        finally:
            r.destroy()
        if seed % 500 == 0:
            print("seed %d" % seed, flush=True)
    dev.destroy()
    print("%d frames, %d errors" % (cases, errors))
    sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
