import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from renderer_amd import scenes
from renderer_amd.runner import SceneRunner
import bench
for cfg in (0, 1):
    sc = bench.build_scene(cfg)
    for fif in (1, 2, 4):
        run = SceneRunner(sc, ordinal=0, frames_in_flight=fif)
        ubos = [bench.orbit_ubo(sc, i, 0) for i in range(64)]
        for i in range(20):
            run.set_ubo(0, ubos[i]); run.submit()
        run.dev.wait_idle()
        # host cost: a short burst that fits the queue
        t0 = time.perf_counter()
        for i in range(32):
            run.set_ubo(0, ubos[i]); run.submit()
        t1 = time.perf_counter()
        run.dev.wait_idle()
        t2 = time.perf_counter()
        # steady state
        n = 2000
        t3 = time.perf_counter()
        for i in range(n):
            run.set_ubo(0, ubos[i & 63]); run.submit()
        run.dev.wait_idle()
        t4 = time.perf_counter()
        print("config %d, %d in flight: host %.1f us per frame to queue 32 frames, %.1f us per frame until they finished; steady state %.1f us per frame" % (
            cfg, fif, (t1 - t0) / 32 * 1e6, (t2 - t0) / 32 * 1e6, (t4 - t3) / n * 1e6), flush=True)
        run.destroy()
