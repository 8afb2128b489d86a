#!/usr/bin/env python3
"""How the kernels of frames in flight overlap, from a rocprofv3 kernel trace of the bench command run with its
default three frames in flight (`rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ...`).

    python tools/overlap_from_trace.py DIR OUT.json [LAST_N_FRAMES]

A frame is one run of [k_draws] k_vertex k_setup k_tile on one hardware queue.  For the last N frames (the steady
state; 300 by default) the script reports each kernel's mean duration while other frames' kernels run beside it, the
frame period (mean distance between the ends of successive tile kernels), how much of a tile kernel's duration has
another frame's vertex / setup / tile kernel running inside it, and how many tile kernels run at once on average --
which is why the tile kernel's duration times the frame count exceeds the wall time."""
import csv
import glob
import json
import sys


def main(src, out, last_n=300):
    path = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        kind = "tile" if "k_tile" in n else "setup" if "k_setup" in n else "vertex" if "k_vertex" in n else "draws" if "k_draws" in n else None
        if kind:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id", "0")))
    rows.sort()
    # frames: per queue, kernels in order; a tile kernel closes a frame
    frames, open_frames = [], {}
    for s, e, kind, q in rows:
        f = open_frames.setdefault(q, {})
        f.setdefault(kind, (s, e))
        if kind == "tile":
            if "vertex" in f and "setup" in f:
                f["queue"] = q
                frames.append(f)
            open_frames[q] = {}
    frames.sort(key=lambda f: f["tile"][1])
    frames = frames[-last_n:]
    if len(frames) < 10:
        raise SystemExit("only %d complete frames in the trace" % len(frames))

    def overlap(a, b):
        return max(0, min(a[1], b[1]) - max(a[0], b[0]))

    ends = [f["tile"][1] for f in frames]
    period = (ends[-1] - ends[0]) / (len(ends) - 1)
    dur = {k: [f[k][1] - f[k][0] for f in frames if k in f] for k in ("draws", "vertex", "setup", "tile")}
    inside = {"vertex": 0, "setup": 0, "tile": 0, "draws": 0}
    tile_total = 0
    for i, f in enumerate(frames):
        t = f["tile"]
        tile_total += t[1] - t[0]
        for g in frames[max(0, i - 6):i + 7]:
            if g is f:
                continue
            for k in inside:
                if k in g:
                    inside[k] += overlap(t, g[k])
    wall = frames[-1]["tile"][1] - frames[0]["tile"][0]
    mean = lambda v: sum(v) / len(v) if v else None  # noqa: E731
    res = {
        "trace": path.split("gpurun_out/")[-1],
        "frames": len(frames), "queues": sorted({f["queue"] for f in frames}),
        "frame_period_us": round(period / 1e3, 3),
        "kernel_us_with_frames_in_flight": {k: round(mean(v) / 1e3, 3) for k, v in dur.items() if v},
        "tile_kernel_us_times_frames_over_wall": round(tile_total / wall, 3),
        "share_of_a_tile_kernels_duration_with_another_frames_kernel_inside": {k: round(v / tile_total, 3) for k, v in inside.items()},
        "note": "tile kernels of successive frames run on different hardware queues and overlap: the sum of their durations "
                "is tile_kernel_us_times_frames_over_wall times the wall time of the same frames",
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 300)
