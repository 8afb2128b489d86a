"""Steady-state overlap of the tile kernels of frames in flight (a build with -DRND_PROBES named by RENDERER_AMD_LIB):
submits FRAMES frames with F command buffers in flight, then lets the library dump the workgroup times of each command
buffer's last frame (RND_PROBE_DUMP) and prints them side by side.  python tools/probe_steady.py CONFIG F [FRAMES]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dump = "/tmp/rnd_probe_dump.txt"
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    if os.path.exists(dump):
        os.remove(dump)
    env = dict(os.environ, RND_PROBE_DUMP=dump)
    subprocess.run([sys.executable, __file__, "--child"] + sys.argv[1:], env=env, check=True)
    import bench
    cols = (bench.build_scene(int(sys.argv[1])).width + 31) // 32  # tile columns: workgroup number -> tile
    subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "probe_overlap.py"), dump, str(cols)], check=True)
    sys.exit(0)
import bench
from renderer_amd.runner import SceneRunner
cfg, F = int(sys.argv[2]), int(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 40
scene = bench.build_scene(cfg)
run = SceneRunner(scene, frames_in_flight=F)
ubos = [bench.orbit_ubo(scene, i) for i in range(frames)]  # not inside the loop: the host must stay ahead of the device
for i in range(frames):
    run.set_ubo(0, ubos[i])
    run.submit()
run.dev.wait_idle()
run.destroy()
