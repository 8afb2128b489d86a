#!/usr/bin/env python3
"""Benchmark of the draw path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 0|1|2|3|4] [--mode views|tiles|bands]

One step = one frame: write the 320-byte uniform block (a new camera, as the reference's
draw_frame does every frame, base_obj_viewer_app.cpp:90-141) and submit the pre-recorded
command buffer (vk_app.cpp:463-467).  Vertex, index, texture and instance data are
resident in HBM before the timed region starts.

N = 1 renders BASELINE.json's configs[1]: 1920x1080, one draw of 1 002 528 triangles.
N > 1 (launched by torch.distributed.run, one rank per GPU): every rank renders its own
view of that scene (independent views sharded over GPUs, no collective on the data path,
weak scaling); --mode tiles instead shards the 32x32 tiles of ONE frame over the ranks
and gathers the finished colour image with an RCCL all_gather (strong scaling); --mode bands
gives every rank a band of whole rows of tiles as its render area, so that the library also skips
the draws outside the band, and gathers the bands the same way.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from renderer_amd import api, hostmath as hm, scenes  # noqa: E402
from renderer_amd.runner import SceneRunner  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak
VALU_CYCLES = 2.4      # cycles a SIMD spends per wave-wide vector instruction when it has waves to choose from


def build_scene(cfg):
    if cfg == 0:
        return scenes.config0()
    if cfg == 1:
        return scenes.config1()
    if cfg == 2:
        return scenes.config2()
    if cfg == 3:
        return scenes.config3()
    if cfg == 4:
        return scenes.config4_view(0)  # one of the 64 views; each rank renders its own (orbit_ubo)
    raise SystemExit("unknown --config %r" % cfg)


def orbit_ubo(scene, step, rank=0):
    """Per-frame uniform block: the camera swings around the y axis like the sample's
    mouse orbit (base_obj_viewer_app.cpp:98-125)."""
    cam = hm.Camera()
    ang = 0.02 * np.sin(0.37 * step + 0.9 * rank) + 0.05 * rank
    cam.eye_position = np.array([-2.0 * np.sin(ang), 0.0, -2.0 * np.cos(ang)], dtype=np.float32)
    model = np.asarray(scene.ubos[0][48:64], dtype=np.float32).reshape(4, 4)
    return hm.global_ubo(cam, model, scene.width, scene.height)


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
    return {"value": round(mpix, 3), "unit": "Mpix/s", "cores": cores, "kind": "port",
            "mtri_per_s": round(scene.triangle_count() * frames / el / 1e6, 3),
            "sample": "%d frame(s) of %s (%dx%d, %d triangles) in %.2f s" % (
                frames, scene.name, scene.width, scene.height, scene.triangle_count(), el)}


def measured_counters(kernel, config):
    """The committed counter pass of this workload for `kernel` (profiles/traffic.json, written by
    tools/summarize_profiles.py from rocprofv3 --pmc runs of this same command): HBM bytes per
    launch (FETCH_SIZE / WRITE_SIZE) and vector-ALU instructions per launch (SQ_INSTS_VALU).
    (None, None) when there is no pass for this workload."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            table = json.load(f).get("config%d" % config, {})
    except (OSError, ValueError):
        return None, None
    for name, row in table.items():
        if name.split("<")[0] == kernel:
            return int(row["read_bytes"] + row["write_bytes"]), row.get("valu_insts")
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--samples", type=int, choices=[1, 4], default=1,
                    help="samples per pixel (4: what test_app renders with, dummy_obj_viewer_app.cpp:319; resolved in linear light)")
    ap.add_argument("--mode", choices=["views", "tiles", "bands"], default="views")
    ap.add_argument("--frames-in-flight", type=int, default=2,
                    help="command buffers / targets the frames rotate through (the reference's swapchain image count)")
    ap.add_argument("--one-shard-of", type=int, default=1, metavar="N",
                    help="single-GPU rehearsal of the tiles mode (or, with --mode bands, the bands mode): render only what "
                         "rank --shard-rank of N would own, no exchange")
    ap.add_argument("--shard-rank", type=int, default=0, help="which rank's share --one-shard-of renders")
    ap.add_argument("--no-overlap", action="store_true", help="bands mode: gather every frame before the next is rendered")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--profile-steps", type=int, default=20)
    args = ap.parse_args()

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
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    scene = build_scene(args.config)
    if args.samples != 1:
        scene.samples = args.samples
        for p in scene.pipelines:
            p["samples"] = args.samples
    bands = args.mode == "bands"
    tiles_mode = args.mode in ("tiles", "bands") and dist is not None
    whole_scene = scene
    if bands and (tiles_mode or args.one_shard_of > 1):
        from renderer_amd import sharding
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
    n_ubos = args.warmup + max(args.steps, args.profile_steps)
    ubos = [orbit_ubo(scene, i, 0 if tiles_mode else rank) for i in range(n_ubos)]

    def render_step(i):
        run.set_ubo(0, ubos[i])
        run.submit()

    step = render_step

    gather_s = [0.0]  # tiles mode: time spent after the rank's own tiles are rendered (copy-out, all_gather, put-back)
    if tiles_mode:
        from renderer_amd import sharding
        if bands:
            y0, y1 = sharding.band_rows(scene.height, rank, world)
            mine = torch.empty((y1 - y0, scene.width, 4), dtype=torch.uint8, device="cuda")
        else:
            mine = torch.empty((scene.height, scene.width, 4), dtype=torch.uint8, device="cuda")

        def step(i):  # noqa: F811 - one frame = render own tiles, then gather the colour image
            render_step(i)
            if bands:
                run.target.copy_color_rows(y0, y1 - y0, mine.data_ptr())
            else:
                run.target.copy_color(mine.data_ptr())
            run.dev.wait_idle()
            t_g = time.perf_counter()
            part = mine if backend == "nccl" else mine.cpu()
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
            run.set_ubo(0, ubos[i])
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

    for i in range(args.warmup):
        step(i)
    if finish:
        finish()
    sync()
    gather_s[0] = 0.0
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if finish:
        finish()
    run.dev.wait_idle()
    if dist is not None:
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gather_ms = gather_s[0] / args.steps * 1e3 if tiles_mode and not pipelined else None
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
            raise SystemExit("tile-sharded frame differs from the unsharded frame: " + tiles_check)
    frames_total = args.steps * (1 if tiles_mode else world)
    if args.one_shard_of > 1 and not tiles_mode:
        frames_total /= args.one_shard_of  # a rehearsal renders one rank's share of every frame
    pixels = scene.width * scene.height
    mpix = pixels * frames_total / elapsed / 1e6
    mtri = scene.triangle_count() * frames_total / elapsed / 1e6

    # per-kernel durations, HIP events on the library's own stream
    roofline = None
    stages = {}
    if rank == 0:
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
        achieved = share / (stages[dom] * 1e-3) / 1e9
        traffic, valu = measured_counters("k_" + dom, args.config) if args.samples == 1 and not tiles_mode and args.one_shard_of == 1 else (None, None)  # counter passes are whole frames at one sample
        roofline = {"bound": "hbm", "kernel": "k_" + dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic,
                    "algorithmic_bytes_per_launch": share,
                    "kernel_ms": round(stages[dom], 5)}
        if valu:
            # Vector-ALU occupation, for the record beside the HBM fraction: a SIMD with two or more waves to
            # choose from issues one wave-wide vector instruction per VALU_CYCLES cycles (measured,
            # tools/micro/valu_rate.hip: 2.1 for v_fma_f32, 2.4-2.6 for 32-bit integer adds; a wave on its own
            # gets one per 4.5); 256 CUs x 4 SIMDs at 2.4 GHz
            roofline["valu_busy"] = round(valu * VALU_CYCLES / (1024 * stages[dom] * 1e-3 * 2.4e9), 4)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:  # the CPU leg belongs to the single-GPU line only
        cpu = cpu_baseline(scene)

    if rank == 0:
        out = {
            "metric": "shaded_mpix_per_s", "value": round(mpix, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "strong" if tiles_mode else "weak", "vs_baseline": None,
            "dtype": "f32+i32", "data": "synthetic",
            "mtri_per_s": round(mtri, 2),
            "config": {"workload": "%s: %dx%d, %d triangles, %d draws, test_ktx shader, %d sample%s/pixel" % (
                scene.name, scene.width, scene.height, scene.triangle_count(), len(scene.draws), scene.samples,
                "" if scene.samples == 1 else "s"),
                "sharding": ("%s of one frame over %d GPUs + RCCL all_gather%s" % (
                    args.mode, world, ", exchange of a frame overlapped with the next frame's rendering" if pipelined else "")) if tiles_mode
                else ("rehearsal: the %s of rank %d of %d only" % (args.mode if bands else "tiles", args.shard_rank, args.one_shard_of)) if args.one_shard_of > 1
                else ("one view per GPU, %d GPUs" % world), "frames_in_flight": run.frames_in_flight,
                "seed": scene.seed},
            "frame_algorithmic_bytes": scene.algorithmic_bytes(),
            "frame_hbm_frac": round(scene.algorithmic_bytes() * frames_total / elapsed / 1e9 / (HBM_PEAK_GBS * world), 5),
            "stage_ms": {k: round(v, 5) for k, v in stages.items()},
            "tiles_check": tiles_check,
            "gather_ms": None if gather_ms is None else round(gather_ms, 5),
            "render_ms": None if gather_ms is None else round(elapsed / args.steps * 1e3 - gather_ms, 5),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    run.destroy()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
