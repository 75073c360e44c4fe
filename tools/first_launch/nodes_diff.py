"""Where do the node rows of two builds differ?  -> per repetition: rows off, and for each the tile / row-in-tile / column pattern.
    python tools/first_launch/nodes_diff.py good.npy bad.npy"""
import sys
import numpy as np
g, b = np.load(sys.argv[1]), np.load(sys.argv[2])
ref = g[0]
print("reference self-consistency over its repetitions: max |diff| %.3e" % max(np.abs(x - ref).max() for x in g))
scale = np.abs(ref).max()
for r, x in enumerate(b):
    d = np.abs(x - ref)
    rows = np.nonzero(d.max(1) > 1e-5 * scale)[0]
    print(f"rep {r}: {len(rows)} of {ref.shape[0]} node rows differ (max |diff| {d.max():.3e}, scale {scale:.3e})")
    for row in rows[:40]:
        cols = np.nonzero(d[row] > 1e-5 * scale)[0]
        # node row u (machine) = tile rows 2u, 2u+1 -> tile (2u)//16, rows-in-tile 2u%16 (+1); workgroup-relative tile = tile % 12
        tile = (2 * row) // 16
        print("   machine row %6d  workgroup %4d  tile-in-workgroup %2d (wave %d, round %d)  tile rows %2d,%2d  lane q=%d  columns off: %3d  first %s  blocks(c) %s  m %s" % (
            row, tile // 12, tile % 12, (tile % 12) % 8, (tile % 12) // 8, (2 * row) % 16, (2 * row) % 16 + 1, ((2 * row) % 16) // 4, len(cols), cols[:8].tolist(),
            sorted(set((cols // 16).tolist())), sorted(set((cols % 16).tolist()))[:16]))

# values of the first few differing rows: expected | got | ratio (a pattern — stale, scaled, zero — names the instruction)
x = b[0]
d = np.abs(x - ref)
rows = np.nonzero(d.max(1) > 1e-5 * scale)[0][:4]
np.set_printoptions(precision=5, linewidth=220, suppress=True)
for row in rows:
    cols = np.nonzero(d[row] > 1e-5 * scale)[0]
    print("row", row, "columns", cols[0], "..", cols[-1])
    print("  expected", ref[row, cols])
    print("  got     ", x[row, cols])
    print("  got/exp ", x[row, cols] / ref[row, cols])
    nb = [c for c in (cols[0] - 16, cols[-1] + 1) if 0 <= c < 128]
    for c0 in nb:
        print("  neighbour block", c0, "expected", ref[row, c0:c0 + 16][:6], "got", x[row, c0:c0 + 16][:6])
    # is the wrong block equal to some other row's expected block (a stale / misplaced accumulator)?
    blk = x[row, cols[0]:cols[0] + 16]
    cand = np.abs(ref[:, cols[0]:cols[0] + 16] - blk).max(1)
    print("  closest expected row for this block:", int(cand.argmin()), "max |diff| %.3e" % cand.min(), "(this row: %d)" % row)
    for cb in range(8):
        c2 = np.abs(ref[row - 8:row + 8, 16 * cb:16 * cb + 16] - blk).max(1)
        if c2.min() < 1e-4:
            print("  == expected block", cb, "of row", row - 8 + int(c2.argmin()))
