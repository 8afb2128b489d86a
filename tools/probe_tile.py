"""Timing probe: per-frame stage times for config 1 with a fixed and a moving camera."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderer_amd import scenes
from renderer_amd.runner import SceneRunner
import bench

scene = scenes.config1()
run = SceneRunner(scene)
run.dev.set_profiling(True)
for mode in ("fixed", "orbit"):
    ts = []
    for i in range(24):
        if mode == "orbit":
            run.set_ubo(0, bench.orbit_ubo(scene, i))
        run.submit()
        run.dev.wait_idle()
        st = run.dev.stage_times()
        ts.append((st["tile"], st["setup"]))
    print(mode, "tile", [round(t[0] * 1e3, 1) for t in ts])
    print(mode, "setup", [round(t[1] * 1e3, 1) for t in ts])
run.destroy()
