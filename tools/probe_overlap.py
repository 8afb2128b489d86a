"""Reads a RND_PROBE_DUMP file (a build with -DRND_PROBES; one block per command buffer: workgroup, start tick, end tick
of the tile kernel's workgroups in the command buffer's LAST frame, 100 MHz) and prints, per 2-microsecond bin, how many
workgroups of each command buffer were resident: whether tile kernels of frames in flight overlap."""
import sys
blocks, cur, kernels, ids = [], None, [], {}
for line in open(sys.argv[1]):
    if line.startswith("#"):
        cur = []
        blocks.append(cur)
        f = line.split()
        if "vertex" in f:
            kernels.append([int(x) for x in (f[f.index("vertex") + 1], f[f.index("vertex") + 2], f[f.index("setup") + 1], f[f.index("setup") + 2])])
    else:
        f = line.split()
        if int(f[1]) and int(f[2]) >= int(f[1]):
            cur.append((int(f[1]), int(f[2])))
            ids.setdefault(len(blocks) - 1, []).append(int(f[0]))
t0 = min(s for b in blocks for s, _ in b)
t1 = max(e for b in blocks for _, e in b)
for k, b in enumerate(blocks):
    print("command buffer %d: %d workgroups, first start +%.1f us, last start +%.1f us, last end +%.1f us" % (
        k, len(b), (min(s for s, _ in b) - t0) / 100.0, (max(s for s, _ in b) - t0) / 100.0, (max(e for _, e in b) - t0) / 100.0))
for k, ks in enumerate(kernels):
    print("command buffer %d: vertex kernel +%.1f .. +%.1f us, setup kernel +%.1f .. +%.1f us" % tuple([k] + [(x - t0) / 100.0 for x in ks]))
# mean lifetime of the workgroups by when they started: does a workgroup live longer while the device is full?
for k, b in enumerate(blocks):
    bins = {}
    for s_, e_ in b:
        bins.setdefault((s_ - t0) // 500, []).append(e_ - s_)
    print("command buffer %d: mean workgroup lifetime (us) by start time: " % k +
          "  ".join("+%d:%.1f(n=%d)" % (q * 5, sum(v) / len(v) / 100.0, len(v)) for q, v in sorted(bins.items())))
# the workgroups that end last, and how lifetimes are spread
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for k, b in enumerate(blocks):
    life = sorted(e_ - s_ for s_, e_ in b)
    print("command buffer %d: lifetime percentiles (us) 10/50/90/99/max: %s" % (k, " ".join("%.1f" % (life[min(len(life) - 1, int(len(life) * q))] / 100.0) for q in (0.1, 0.5, 0.9, 0.99, 1.0))))
    order = sorted(range(len(b)), key=lambda i: -b[i][1])[:12]
    for i in order:
        wg = ids[k][i]
        where = ""
        if cols:
            where = " tile (%d, %d)" % ((wg % (8 * cols)) >> 3, (wg // (8 * cols)) * 8 + (wg & 7))
        print("   workgroup %5d%s: start +%.1f, end +%.1f, lifetime %.1f us" % (wg, where, (b[i][0] - t0) / 100.0, (b[i][1] - t0) / 100.0, (b[i][1] - b[i][0]) / 100.0))
step = 200
for t in range(t0, min(t1, t0 + 400 * step), step):  # at most 400 lines
    print("+%6.1f us  " % ((t - t0) / 100.0) + "  ".join("%5d" % sum(1 for s, e in b if s <= t < e) for b in blocks))
