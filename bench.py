#!/usr/bin/env python3
"""Benchmark of the draw path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 0|1|2|3|4] [--mode views|tiles|bands]

One step = one batch of --frames-per-step frames (64), each with a camera of its own: per frame, write the
320-byte uniform block (as the reference's draw_frame does every frame, base_obj_viewer_app.cpp:90-141) and
submit the pre-recorded command buffer (vk_app.cpp:463-467), --frames-in-flight of them in flight (3, the
reference's swapchain image count, vk_app.cpp:109-130, 154).  A region of a few single frames would measure how
the pipeline fills and drains between the two device waits that bracket it: 20 frames read 47 us per frame where
1280 read 42.9.  `ms_per_step` is per batch, `ms_per_frame` per frame; `value` counts every frame's pixels.
Vertex, index, texture and instance data are resident in HBM before the timed region starts.  The timed loop
itself is compiled host code (renderer_amd/host/bench_loop.cpp, handed the library's entry points): no interpreter
work is inside the clock; --python-loop drives the same calls from Python instead.

N = 1 renders BASELINE.json's configs[1]: 1920x1080, one draw of 1 002 528 triangles.

N > 1 runs one rank per GPU.  Launched by torch.distributed.run (WORLD_SIZE set) the process is a
rank; launched plainly with --gpus N it starts the N ranks itself as a child process, before
anything touches a GPU, and relays rank 0's line.  The N-GPU line carries three measurements:
  value / ms_per_step   every rank renders its own view of configs[1] (independent views, no
                        collective on the data path: weak scaling) -- the same workload as the one-GPU
                        line, so that `value` compares across N;
  "strong_scaling"      configs[3]: ONE 7680x4320 frame divided into bands of rows of tiles, one per
   (= "bands_8k")       rank (each rank's render area; the library skips the draws outside it), the
                        finished bands exchanged by one RCCL all_gather overlapped with the next
                        frame's rendering (render and exchange times apart, the gathered frame checked
                        against the unsharded one), beside the same frame rendered whole by rank 0
                        alone in the same run: speedup and efficiency = speedup / N are in the line;
  "multiview_10m"       configs[4]: the 64 cameras of the orbit over the 10 M-triangle scene dealt
                        to the ranks (64 / N views each), no exchange.
--mode tiles | bands with --config C runs that one sharded mode alone, as before.

Prints ONE JSON line on rank 0.
"""
import argparse
import copy
import ctypes as C
import glob
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak
VALU_CYCLES = 4.0      # cycles of a SIMD per wave-wide vector instruction: a SIMD issues one per four-cycle turn (the tile kernel
                       # of configs[2] takes 4.08 per vector instruction it issues); only unmixed streams of add / sub / and / or / xor /
                       # shift-right / mov / fp32 add and mul from several waves pair up at 2.5 (tools/micro/issue_rate.hip)
SIMDS = 1024           # 256 CUs x 4
CLOCK_HZ = 2.4e9       # the device holds 2.39 GHz while the frames run (tools/probe_clock.sh)
FRAME_KERNELS = ("k_draws", "k_vertex", "k_setup", "k_tile")
PRIME_STEPS = 300      # untimed frames before the caller's warm-up: the first launches of a process run at a lower clock
N_VIEWS = 64           # cameras of configs[4]
LEGS_TIME_LIMIT_S = 420 # N > 1: the two sharded legs together; beyond it the line is printed without the leg that hangs


# ----------------------------------------------------------------------------- launching N ranks

def rank_launch_command(n, argv, port):
    """The child command that runs this script as n ranks of one node (argv: this script's own arguments)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """Starts the ranks as a CHILD process (this process has not touched a GPU and never will), relays
    rank 0's JSON line and returns the child's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.run(rank_launch_command(n, argv, free_port()), env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in child.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    else:
        sys.stderr.write(child.stdout[-4000:])
    return child.returncode if line is not None or child.returncode else 1


# ----------------------------------------------------------------------------- scenes and cameras

def build_scene(cfg):
    from renderer_amd import scenes
    if cfg == 0:
        return scenes.config0()
    if cfg == 1:
        return scenes.config1()
    if cfg == 2:
        return scenes.config2()
    if cfg == 3:
        return scenes.config3()
    if cfg == 4:
        return scenes.config4_view(0)  # the scene of the 64 views; the camera of each comes from view_ubo
    raise SystemExit("unknown --config %r" % cfg)


def orbit_ubo(scene, step, rank=0):
    """Per-frame uniform block: the camera swings around the y axis like the sample's
    mouse orbit (base_obj_viewer_app.cpp:98-125)."""
    from renderer_amd import hostmath as hm
    cam = hm.Camera()
    ang = 0.02 * np.sin(0.37 * step + 0.9 * rank) + 0.05 * rank
    cam.eye_position = np.array([-2.0 * np.sin(ang), 0.0, -2.0 * np.cos(ang)], dtype=np.float32)
    model = np.asarray(scene.ubos[0][48:64], dtype=np.float32).reshape(4, 4)
    return hm.global_ubo(cam, model, scene.width, scene.height)


def view_ubo(scene, view):
    """Uniform block of camera `view` of configs[4]'s 64-camera orbit."""
    from renderer_amd import hostmath as hm, scenes
    model = np.asarray(scene.ubos[0][48:64], dtype=np.float32).reshape(4, 4)
    return hm.global_ubo(scenes.config4_camera(view, N_VIEWS), model, scene.width, scene.height)


# ----------------------------------------------------------------------------- side measurements

def cpu_baseline(scene, budget_s=12.0):
    """The CPU oracle on the host cores of this box, same scene, bounded sample."""
    from oracle import oracle
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)), 64))
    oracle.render_scene(scene, threads=cores)  # warm-up (page in, build tables)
    frames, t0 = 0, time.perf_counter()
    while True:
        oracle.render_scene(scene, threads=cores)
        frames += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    mpix = scene.width * scene.height * frames / el / 1e6
    out = {"value": round(mpix, 3), "unit": "Mpix/s", "cores": cores, "kind": "port",
           "mtri_per_s": round(scene.triangle_count() * frames / el / 1e6, 3),
           "sample": "%d frame(s) of %s (%dx%d, %d triangles) in %.2f s" % (
               frames, scene.name, scene.width, scene.height, scene.triangle_count(), el)}
    out.update(lavapipe_probe())
    return out


def lavapipe_probe():
    """The reference has no software path of its own; the CPU path BASELINE.json names next is the same sample on
    Mesa lavapipe.  Looks for the Vulkan loader and an lvp ICD manifest on this box and says what it found.  Nothing
    is loaded or run: without the reference's toolchain (Vulkan headers, GLFW, glm, shaderc) there is no harness to
    time it with, so a present lavapipe is reported, not timed."""
    import ctypes.util
    loader = ctypes.util.find_library("vulkan")
    manifests = []
    for d in ("/usr/share/vulkan/icd.d", "/etc/vulkan/icd.d", "/usr/local/share/vulkan/icd.d",
              os.path.expanduser("~/.local/share/vulkan/icd.d")):
        manifests += glob.glob(os.path.join(d, "*lvp*.json"))
    if loader and manifests:
        return {"lavapipe": "present, not timed (no harness: the reference's samples cannot be built here)",
                "lavapipe_found": {"loader": loader, "icd": manifests}}
    return {"lavapipe": "unavailable", "lavapipe_found": {"loader": loader, "icd": manifests}}


def measured_copy_peak(dev, reps=8):
    """Device-to-device copy bandwidth on this box, GB/s counting bytes read plus bytes written: what a kernel
    that streams its input once and its output once can reach, beside the 8 TB/s on the datasheet.  The copy
    is the library's own (rnd_buffer_write_device: hipMemcpyAsync device-to-device on the device's stream)
    of a 256 MiB image (an 8192 x 8192 colour target) into a buffer."""
    from renderer_amd import api
    side = 8192
    n = side * side * 4
    tgt = api.Target(dev, side, side)
    buf = api.Buffer(dev, api.BUFFER_USAGE_UNIFORM | api.BUFFER_USAGE_VERTEX, nbytes=n)
    try:
        src = tgt.color_device_ptr
        buf.write_device(src, n)
        dev.wait_idle()
        t0 = time.perf_counter()
        for _ in range(reps):
            buf.write_device(src, n)
        dev.wait_idle()
        el = time.perf_counter() - t0
        return 2.0 * n * reps / el / 1e9
    finally:
        buf.destroy()
        tgt.destroy()


KERNEL_SOURCES = ("rnd_kernels.hip", "rnd_device.h", "rnd_internal.h")


def kernel_source_id():
    """First 16 hex digits of the SHA-256 over the kernels' source files as they are in this tree: what a counter pass
    belongs to (the counters change when the kernels do, not when host code does)."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "renderer_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def counter_table(config):
    """The committed counter pass of this workload (profiles/traffic.json, written by tools/summarize_profiles.py from
    rocprofv3 --pmc runs of this same command) and where it came from: ({kernel: row}, {"tag", "kernel_source_id",
    "matches_this_tree"}).  The table is empty when the pass was collected on other kernel sources than this tree's:
    its numbers then describe another build and the line says null instead."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            whole = json.load(f)
    except (OSError, ValueError):
        return {}, None
    table = whole.get("config%d" % config, {})
    origin = whole.get("collected", {}).get("config%d" % config)
    if not table:
        return {}, origin
    if origin is None:
        return {}, {"tag": None, "kernel_source_id": None, "matches_this_tree": False}
    origin = dict(origin, matches_this_tree=origin.get("kernel_source_id") == kernel_source_id())
    return (table if origin["matches_this_tree"] else {}), origin


def measured_counters(kernel, config):
    """HBM bytes per launch (size-resolved read requests + WRITE_SIZE) and vector-ALU instructions per launch
    (SQ_INSTS_VALU) of `kernel` from the committed counter pass; (None, None) when there is none for this build."""
    table, _ = counter_table(config)
    for name, row in table.items():
        if name.split("<")[0] == kernel:
            return int(row["read_bytes"] + row["write_bytes"]), row.get("valu_insts")
    return None, None


def measured_frame_totals(config):
    """Vector-ALU instructions and L2<->HBM bytes of one frame's kernels together, from the same committed counter pass
    ((None, None) without one)."""
    table, _ = counter_table(config)
    rows = [row for name, row in table.items() if name.split("<")[0] in FRAME_KERNELS]
    valu = sum(row.get("valu_insts") or 0 for row in rows)
    traffic = sum((row.get("read_bytes") or 0) + (row.get("write_bytes") or 0) for row in rows)
    return (valu or None), (int(traffic) or None)


# ----------------------------------------------------------------------------- the compiled timed loop

class NativeLoop:
    """renderer_amd/host/bench_loop.cpp: frames i = first .. first+steps-1 as `rnd_buffer_write(ubo, block i)` +
    `rnd_submit(dev, cmds[image])`, then rnd_device_wait_idle, all inside one C++ function."""

    def __init__(self, run, blocks):
        from renderer_amd import _lib
        path = os.path.join(ROOT, "renderer_amd", "libbench_loop.so")
        if not os.path.exists(path):
            raise SystemExit("renderer_amd/libbench_loop.so is missing: make -C renderer_amd/host (or --python-loop)")
        self.so = C.CDLL(path)
        lib = _lib.load()
        self.fn = self.so.rnd_bench_frames
        self.fn.restype = C.c_int
        self.fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int,
                            C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        self.entry = [C.cast(getattr(lib, n), C.c_void_p) for n in ("rnd_buffer_write", "rnd_submit", "rnd_device_wait_idle")]
        self.run = run
        self.blocks = np.ascontiguousarray(np.stack(blocks), dtype=np.float32)
        self.cmds = (C.c_void_p * len(run.cmds))(*[c.h.value for c in run.cmds])
        self.seconds = (C.c_double * 2)()

    def frames(self, first, steps):
        run = self.run
        rc = self.fn(self.entry[0], self.entry[1], self.entry[2], run.dev.h, self.cmds, len(run.cmds), run.next_image,
                     run.ubos[0].h, self.blocks.ctypes.data_as(C.POINTER(C.c_float)), self.blocks.shape[0], self.blocks.shape[1],
                     first, steps, self.seconds)
        if rc != 0:
            from renderer_amd import _lib
            raise SystemExit("timed loop failed: %d %s" % (rc, _lib.load().rnd_last_error().decode("utf-8", "replace")))
        run.next_image = (run.next_image + steps) % len(run.cmds)
        return self.seconds[0], self.seconds[1]


# ----------------------------------------------------------------------------- one measurement

def bench_once(args, ctx, config, mode, steps, warmup, profile=True, cpu=False, copy_peak=False, frames_per_step=1):
    """One workload measured as the contract says: warm-up, a barrier and a device wait on both sides of exactly
    `steps` steps of `frames_per_step` frames each, the maximum over ranks.  Returns the result line as a dict
    (complete on rank 0)."""
    from renderer_amd import api, sharding
    from renderer_amd.runner import SceneRunner
    rank, world, local_rank = ctx["rank"], ctx["world"], ctx["local_rank"]
    dist, torch, backend = ctx["dist"], ctx["torch"], ctx["backend"]

    scene = build_scene(config)
    if args.samples != 1:
        scene.samples = args.samples
        for p in scene.pipelines:
            p["samples"] = args.samples
    bands = mode == "bands"
    tiles_mode = mode in ("tiles", "bands") and dist is not None
    whole_scene = scene
    if bands and (tiles_mode or args.one_shard_of > 1):
        scene = sharding.band_scene(scene, *((rank, world) if tiles_mode else (args.shard_rank, args.one_shard_of)))
    # Bands over RCCL are pipelined: while the bands of frame i are exchanged, frame i + 1 is rendered into the
    # other image.  The tiles mode (and any run over gloo) gathers every frame before the next starts.
    pipelined = tiles_mode and bands and backend == "nccl" and args.frames_in_flight >= 2 and not args.no_overlap
    run = SceneRunner(scene, ordinal=local_rank,
                      frames_in_flight=2 if pipelined else 1 if tiles_mode else args.frames_in_flight)
    if tiles_mode and not bands:
        run.cmd.set_tile_shard(rank, world)
    elif args.one_shard_of > 1 and not bands:
        for c in run.cmds:
            c.set_tile_shard(args.shard_rank, args.one_shard_of)

    def sync():
        run.dev.wait_idle()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # the uniform blocks of every frame are built before the clock starts: the timed loop
    # only writes 320 bytes and submits, like the reference's draw_frame
    multiview = config == 4 and mode == "views"
    if multiview:
        # the 64 cameras dealt to the ranks: rank r renders views r, r + world, ...
        mine = list(range(rank, N_VIEWS, world)) or [rank % N_VIEWS]
        ubos = [view_ubo(scene, v) for v in mine]
        block_of = lambda i: i % len(ubos)  # noqa: E731
    else:
        n_ubos = PRIME_STEPS + (warmup + max(steps, args.profile_steps)) * frames_per_step
        ubos = [orbit_ubo(scene, i, 0 if tiles_mode else rank) for i in range(min(n_ubos, 512))]
        block_of = lambda i: i % len(ubos)  # noqa: E731

    def render_step(i):
        run.set_ubo(0, ubos[block_of(i)])
        run.submit()

    step = render_step
    native = None
    if not tiles_mode and not args.python_loop:
        native = NativeLoop(run, ubos)

    gather_s = [0.0]  # tiles mode: time spent after the rank's own tiles are rendered (copy-out, all_gather, put-back)
    if tiles_mode:
        if bands:
            y0, y1 = sharding.band_rows(scene.height, rank, world)
            mine_img = torch.empty((y1 - y0, scene.width, 4), dtype=torch.uint8, device="cuda")
        else:
            mine_img = torch.empty((scene.height, scene.width, 4), dtype=torch.uint8, device="cuda")

        def step(i):  # noqa: F811 - one frame = render own tiles, then gather the colour image
            render_step(i)
            if bands:
                run.target.copy_color_rows(y0, y1 - y0, mine_img.data_ptr())
            else:
                run.target.copy_color(mine_img.data_ptr())
            run.dev.wait_idle()
            t_g = time.perf_counter()
            part = mine_img if backend == "nccl" else mine_img.cpu()
            frame = sharding.gather_bands(part, dist, world, scene.height) if bands else sharding.gather_frame(part, dist, world, rank)
            torch.cuda.synchronize()
            gather_s[0] += time.perf_counter() - t_g
            return frame

    finish = None
    if pipelined:
        # The exchange reads each image's band where the library rendered it (no copy-out), as equal-sized slabs
        # of rows (sharding.band_send_rows).  Order per image: render (library stream) -> fence -> all_gather
        # (RCCL's stream) -> host wait on it before the image is rendered into again.
        s0, tallest = sharding.band_send_rows(scene.height, rank, world)
        pitch = scene.width * 4
        src = [sharding.device_bytes(t.color_device_ptr, scene.height * pitch)[s0 * pitch:(s0 + tallest) * pitch]
               for t in run.targets]
        slabs = [torch.empty((world, tallest, scene.width, 4), dtype=torch.uint8, device="cuda") for _ in run.targets]
        fences = [api.Fence(run.dev) for _ in run.targets]
        work = [None] * len(run.targets)
        waiting = [None]  # image whose frame is rendered (or rendering) and not yet exchanged
        last_frame = [None]

        def exchange(img):
            fences[img].wait()
            work[img] = dist.all_gather_into_tensor(slabs[img].view(-1), src[img], async_op=True)

        def collect(img):
            """The frame of image img's last exchange, once it is done (None if there was none)."""
            if work[img] is None:
                return None
            work[img].wait()
            work[img] = None
            frame = sharding.compose_bands(slabs[img], scene.height)
            torch.cuda.current_stream().synchronize()
            return frame

        def step(i):  # noqa: F811
            img = run.next_image
            f = collect(img)
            if f is not None:
                last_frame[0] = f
            run.set_ubo(0, ubos[block_of(i)])
            run.next_image = (img + 1) % run.frames_in_flight
            run.dev.submit(run.cmds[img], fences[img])
            if waiting[0] is not None:
                exchange(waiting[0])
            waiting[0] = img

        def finish():
            if waiting[0] is not None:
                exchange(waiting[0])
                waiting[0] = None
            for img in range(len(run.targets)):
                f = collect(img)
                if f is not None:
                    last_frame[0] = f
            return last_frame[0]

    def frames(first, count):
        """`count` frames from frame number `first`, finished on return; seconds spent queueing (native loop) or None"""
        if native is not None:
            return native.frames(first, count)[1]
        for i in range(count):
            step(first + i)
        if finish:
            finish()
        run.dev.wait_idle()
        return None

    fps = 1 if tiles_mode else max(1, int(frames_per_step))
    frames(0, PRIME_STEPS if not tiles_mode else min(PRIME_STEPS, 4))
    frames(PRIME_STEPS, warmup * fps)
    sync()
    gather_s[0] = 0.0
    t0 = time.perf_counter()
    queue_s = frames(PRIME_STEPS + warmup * fps, steps * fps)
    if dist is not None:
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same region again, a few times: the spread of the same measurement on this box (not part of `value`)
    repeats = []
    if not tiles_mode and args.repeats > 0:
        for _ in range(args.repeats):
            sync()
            t1 = time.perf_counter()
            frames(PRIME_STEPS + warmup * fps, steps * fps)
            repeats.append((time.perf_counter() - t1) / max(steps, 1) * 1e3)

    gather_ms = gather_s[0] / steps * 1e3 if tiles_mode and not pipelined else None
    tiles_check = None
    if tiles_mode:
        # the gathered frame against the same frame rendered whole on this rank
        if pipelined:
            step(0)
            frame = finish().cpu().numpy()
        else:
            frame = step(0).cpu().numpy()
        whole = SceneRunner(whole_scene, ordinal=local_rank)
        whole.set_ubo(0, ubos[0])
        color, _ = whole.render()
        whole.destroy()
        tiles_check = "exact" if np.array_equal(frame, color) else "%d bytes differ" % int((frame != color).sum())
        if tiles_check != "exact":
            raise RuntimeError("tile-sharded frame differs from the unsharded frame: " + tiles_check)
    frames_total = steps * fps * (1 if tiles_mode else world)
    if args.one_shard_of > 1 and not tiles_mode:
        frames_total /= args.one_shard_of  # a rehearsal renders one rank's share of every frame
    pixels = scene.width * scene.height
    mpix = pixels * frames_total / elapsed / 1e6
    mtri = scene.triangle_count() * frames_total / elapsed / 1e6

    # per-kernel durations, HIP events on the library's own stream
    roofline = None
    stages = {}
    if rank == 0 and profile and args.profile_steps > 0:
        run.dev.set_profiling(True)
        acc = {}
        for i in range(args.profile_steps):
            render_step(i)  # this rank alone: no exchange here
            run.dev.wait_idle()
            for k, v in run.dev.stage_times().items():
                acc[k] = acc.get(k, 0.0) + v
        run.dev.set_profiling(False)
        stages = {k: v / max(args.profile_steps, 1) for k, v in acc.items()}
        # Roofline of the longest kernel: the part of the frame's algorithmic bytes that kernel
        # moves (scenes.Scene.algorithmic_bytes_by_stage) over its own duration.  The event pair
        # around a kernel also spans the event records themselves, a few microseconds; the
        # rocprofv3 averages under profiles/ are the durations without that.
        dom = max(stages, key=stages.get)
        share = scene.algorithmic_bytes_by_stage()[dom]
        if tiles_mode:
            share = share / world
        elif args.one_shard_of > 1:
            share = share / args.one_shard_of  # a rehearsal renders one rank's share of the frame
        achieved = share / (stages[dom] * 1e-3) / 1e9
        traffic, valu = measured_counters("k_" + dom, config) if args.samples == 1 and not tiles_mode and args.one_shard_of == 1 else (None, None)  # counter passes are whole frames at one sample
        roofline = {"bound": "hbm", "kernel": "k_" + dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic,
                    "algorithmic_bytes_per_launch": share,
                    "kernel_ms": round(stages[dom], 5)}
        if valu:
            # Vector-ALU occupation, for the record beside the HBM fraction: the kernel's vector instructions at
            # VALU_CYCLES cycles of a SIMD each, over the SIMD cycles of its duration
            roofline["valu_busy"] = round(valu * VALU_CYCLES / (SIMDS * stages[dom] * 1e-3 * CLOCK_HZ), 4)
            roofline["valu_cycles_per_instruction"] = VALU_CYCLES
        if copy_peak:
            peak = measured_copy_peak(run.dev)
            roofline["peak_measured"] = round(peak, 1)  # device-to-device copy on this box, read + written bytes
            roofline["frac_of_measured"] = round(achieved / peak, 5)

    # the same for the whole frame in the timed region (frames in flight overlap their kernels): what bounds the pipeline
    whole_frames = args.samples == 1 and not tiles_mode and args.one_shard_of == 1 and (not multiview or world == 1)
    frame_valu, frame_traffic = measured_frame_totals(config) if whole_frames else (None, None)
    frame_valu_busy = frame_traffic_frac = None
    if frame_valu:
        frame_valu_busy = round(frame_valu * VALU_CYCLES * frames_total / world / (SIMDS * elapsed * CLOCK_HZ), 4)
    if frame_traffic:
        # what the frames really moved between L2 and HBM (intermediates and re-read lines included), against the same peak
        frame_traffic_frac = round(frame_traffic * frames_total / world / elapsed / 1e9 / HBM_PEAK_GBS, 4)

    cpu_line = cpu_baseline(scene) if cpu and rank == 0 else None

    out = None
    if rank == 0:
        sharding_text = (("%s of one frame over %d GPUs + RCCL all_gather%s" % (
            mode, world, ", exchange of a frame overlapped with the next frame's rendering" if pipelined else "")) if tiles_mode
            else ("rehearsal: the %s of rank %d of %d only" % (mode if bands else "tiles", args.shard_rank, args.one_shard_of)) if args.one_shard_of > 1
            else ("the %d views of the orbit dealt to %d GPU%s, no exchange" % (N_VIEWS, world, "" if world == 1 else "s")) if multiview
            else ("one view per GPU, %d GPU%s%s" % (world, "" if world == 1 else "s", "" if world == 1 else
                  " (weak scaling: every GPU renders the one-GPU line's workload; one frame divided among the GPUs is under strong_scaling)")))
        out = {
            "metric": "shaded_mpix_per_s", "value": round(mpix, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 5),
            "frames_per_step": fps, "ms_per_frame": round(elapsed / (steps * fps) * 1e3, 5),
            "higher_is_better": True, "scaling": "strong" if tiles_mode else "weak", "vs_baseline": None,
            "dtype": "f32+i32", "data": "synthetic",
            "mtri_per_s": round(mtri, 2),
            "config": {"workload": "%s: %dx%d, %d triangles, %d draws, test_ktx shader, %d sample%s/pixel; a step is %d frame%s, each with its own camera" % (
                scene.name if not multiview else "config4_orbit", scene.width, scene.height, scene.triangle_count(), len(scene.draws), scene.samples,
                "" if scene.samples == 1 else "s", fps, "" if fps == 1 else "s"),
                "sharding": sharding_text, "frames_in_flight": run.frames_in_flight,
                "host": "python loop over ctypes" if native is None else "compiled loop (renderer_amd/host/bench_loop.cpp), no interpreter inside the clock",
                "seed": scene.seed},
            "frame_algorithmic_bytes": scene.algorithmic_bytes(),
            "frame_hbm_frac": round(scene.algorithmic_bytes() * frames_total / elapsed / 1e9 / (HBM_PEAK_GBS * world), 5),
            "frame_valu_busy": frame_valu_busy,
            "frame_traffic_bytes": frame_traffic, "frame_traffic_hbm_frac": frame_traffic_frac,
            "host_queue_ms_per_step": None if queue_s is None else round(queue_s / steps * 1e3, 5),
            "ms_per_step_repeats": [round(x, 5) for x in repeats],
            "ms_per_frame_repeats": [round(x / fps, 5) for x in repeats],
            "stage_ms": {k: round(v, 5) for k, v in stages.items()},
            "tiles_check": tiles_check,
            "gather_ms": None if gather_ms is None else round(gather_ms, 5),
            "render_ms": None if gather_ms is None else round(elapsed / steps * 1e3 - gather_ms, 5),
            "roofline": roofline, "cpu_baseline": cpu_line,
            # where roofline.traffic / valu_busy / frame_valu_busy / frame_traffic_* come from: the tag of the committed
            # rocprofv3 counter pass and the kernel sources it was collected on; they are null when this tree's differ
            "counters_from": counter_table(config)[1] if whole_frames else None,
        }
    run.destroy()
    return out


def brief(line):
    """What the N-GPU line keeps of a secondary measurement."""
    keys = ("value", "unit", "mtri_per_s", "ms_per_step", "steps", "scaling", "tiles_check", "gather_ms", "render_ms", "frame_hbm_frac",
            "one_gpu_ms_per_frame", "speedup_vs_one_gpu", "efficiency")
    out = {k: line[k] for k in keys if line.get(k) is not None}
    out["workload"] = line["config"]["workload"]
    out["sharding"] = line["config"]["sharding"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-step", type=int, default=64,
                    help="frames in one step (one batch of synthetic input: that many camera positions of the orbit, rendered "
                         "with --frames-in-flight of them in flight); 1 in the tile- and band-sharded modes, whose every frame ends in an exchange")
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--samples", type=int, choices=[1, 4], default=1,
                    help="samples per pixel (4: what test_app renders with, dummy_obj_viewer_app.cpp:319; resolved in linear light)")
    ap.add_argument("--mode", choices=["views", "tiles", "bands"], default="views")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="command buffers / targets the frames rotate through (the reference's swapchain image count)")
    ap.add_argument("--one-shard-of", type=int, default=1, metavar="N",
                    help="single-GPU rehearsal of the tiles mode (or, with --mode bands, the bands mode): render only what "
                         "rank --shard-rank of N would own, no exchange")
    ap.add_argument("--shard-rank", type=int, default=0, help="which rank's share --one-shard-of renders")
    ap.add_argument("--no-overlap", action="store_true", help="bands mode: gather every frame before the next is rendered")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="N > 1: only the views measurement, not the 8K bands and the 64-view legs")
    ap.add_argument("--python-loop", action="store_true", help="drive the timed loop from Python (ctypes) instead of the compiled loop")
    ap.add_argument("--repeats", type=int, default=4, help="how many more times the timed region is run for ms_per_step_repeats")
    ap.add_argument("--profile-steps", type=int, default=20)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: run the N ranks as a child, before this process touches a GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    # BENCH_BACKEND=gloo rehearses the multi-rank paths on a box with fewer GPUs than ranks: the ranks
    # share the GPUs that are there and the exchange goes through host memory (numbers mean nothing)
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if world > 1 or ("RANK" in os.environ and args.mode != "views"):  # a one-rank launch of a sharded mode runs the same exchange code
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            # a launcher that shows every rank one device of its own (a visible-devices mask per rank) leaves
            # LOCAL_RANK beyond what the rank can see: its device is then the one it sees
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)
    if args.gpus != world:
        raise SystemExit("--gpus %d but the launcher started %d rank%s" % (args.gpus, world, "" if world == 1 else "s"))
    ctx = {"rank": rank, "world": world, "local_rank": local_rank, "dist": dist, "torch": torch, "backend": backend}

    out = bench_once(args, ctx, args.config, args.mode, args.steps, args.warmup,
                     cpu=(world == 1 and not args.no_cpu), copy_peak=(world == 1), frames_per_step=args.frames_per_step)
    if world > 1 and args.mode == "views" and args.config == 1 and not args.no_extras:
        # what dist saw, and the two sharded workloads BASELINE.json names for several GPUs
        extra = copy.copy(args)
        extra.repeats = 0
        # a leg that fails says so in its place: the line's own measurement above is already taken
        def bands_with_baseline():
            """The 8K frame in bands over all ranks, then the same frame whole on rank 0 alone (the other ranks wait at
            the barrier that follows): what dividing one frame among N GPUs gains over one GPU, from one run."""
            line = bench_once(extra, ctx, 3, "bands", max(10, min(args.steps, 40)), max(2, min(args.warmup, 5)), profile=False)
            solo = None
            if rank == 0:
                alone = dict(ctx, world=1, dist=None)
                one = copy.copy(extra)
                one.frames_in_flight = 2  # as the bands leg: one frame rendered while the previous is exchanged
                solo = bench_once(one, alone, 3, "views", max(10, min(args.steps, 40)), max(2, min(args.warmup, 5)), profile=False, frames_per_step=1)
            dist.barrier()
            if rank == 0:
                line["one_gpu_ms_per_frame"] = solo["ms_per_frame"]
                line["speedup_vs_one_gpu"] = round(solo["ms_per_frame"] / line["ms_per_step"], 3)
                line["efficiency"] = round(solo["ms_per_frame"] / line["ms_per_step"] / world, 3)
            return line

        legs = (("bands_8k", bands_with_baseline),
                ("multiview_10m", lambda: bench_once(extra, ctx, 4, "views", 3 * max(1, N_VIEWS // world), max(1, N_VIEWS // world), profile=False)))
        # The line's own measurement is taken.  A leg in which one rank fails and the others wait for it in a collective would
        # keep every rank waiting until the process group's own time limit ends them all without a line: every rank therefore
        # carries a timer over the legs (they take well under a minute together), and when it fires rank 0 prints the line
        # with the leg marked as timed out and every rank ends with status 3: the line is there, the exit status says
        # that a leg is not.
        state = {"leg": legs[0][0]}

        def give_up():
            if rank == 0:
                out[state["leg"]] = {"error": "timed out after %d s (some rank did not reach a collective)" % LEGS_TIME_LIMIT_S}
                out["n_gpus"] = world
                print(json.dumps(out), flush=True)
            os._exit(3)

        timer = threading.Timer(LEGS_TIME_LIMIT_S, give_up)
        timer.daemon = True
        timer.start()
        for name, leg in legs:
            state["leg"] = name
            try:
                line = leg()
                if rank == 0:
                    out[name] = brief(line)
                    if name == "bands_8k":
                        # the strong-scaling figure under a name that says so, at the top level
                        out["strong_scaling"] = dict(out[name], n_gpus=world)
            except Exception as e:  # noqa: BLE001 - reported, not hidden
                if rank == 0:
                    out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                # the other ranks may still be inside this leg's collectives: entering the next leg's would pair
                # collectives that do not belong together.  Print what there is and end every rank here.
                if rank == 0:
                    out["n_gpus"] = world
                    print(json.dumps(out), flush=True)
                os._exit(3)
        timer.cancel()
    if rank == 0:
        out["n_gpus"] = dist.get_world_size() if dist is not None else 1
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
