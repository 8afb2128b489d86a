"""Feasibility probe: throughput of F independent devices (streams) on one GPU, frames dealt
round-robin, against one stream.  python tools/probe_inflight.py CONFIG F [F...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from renderer_amd.runner import SceneRunner

cfg = int(sys.argv[1])
scene = bench.build_scene(cfg)
steps, warm = 300, 30
ubos = [bench.orbit_ubo(scene, i) for i in range(steps + warm)]
for F in [int(a) for a in sys.argv[2:]]:
    runs = [SceneRunner(scene) for _ in range(F)]
    def step(i):
        r = runs[i % F]
        r.set_ubo(0, ubos[i]); r.submit()
    for i in range(warm): step(i)
    for r in runs: r.dev.wait_idle()
    t0 = time.perf_counter()
    for i in range(steps): step(warm + i)
    for r in runs: r.dev.wait_idle()
    el = time.perf_counter() - t0
    print("config", cfg, "streams", F, "ms/frame %.5f" % (el / steps * 1e3), flush=True)
    for r in runs: r.destroy()
