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
