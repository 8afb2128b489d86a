"""Reads a RND_PROBE_DUMP file (a build with -DRND_PROBES; one block per command buffer: workgroup, start tick, end tick
of the tile kernel's workgroups in the command buffer's LAST frame, 100 MHz) and prints, per 2-microsecond bin, how many
workgroups of each command buffer were resident: whether tile kernels of frames in flight overlap."""
import sys
blocks, cur, kernels = [], None, []
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
t0 = min(s for b in blocks for s, _ in b)
t1 = max(e for b in blocks for _, e in b)
for k, b in enumerate(blocks):
    print("command buffer %d: %d workgroups, first start +%.1f us, last start +%.1f us, last end +%.1f us" % (
        k, len(b), (min(s for s, _ in b) - t0) / 100.0, (max(s for s, _ in b) - t0) / 100.0, (max(e for _, e in b) - t0) / 100.0))
for k, ks in enumerate(kernels):
    print("command buffer %d: vertex kernel +%.1f .. +%.1f us, setup kernel +%.1f .. +%.1f us" % tuple([k] + [(x - t0) / 100.0 for x in ks]))
step = 200
for t in range(t0, min(t1, t0 + 400 * step), step):  # at most 400 lines
    print("+%6.1f us  " % ((t - t0) / 100.0) + "  ".join("%5d" % sum(1 for s, e in b if s <= t < e) for b in blocks))
