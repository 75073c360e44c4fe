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

# which linear combination of the last pass's z0 / z1 did the wrong block receive?  (true: 0.5 al0 z0 + 0.5 (1 + al1) z1)
import os, sys as _sys
inp = sys.argv[2] + ".inputs.npz"
if os.path.exists(inp):
    ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _sys.path.insert(0, ROOT)
    import torch, mtfjsp_amd  # noqa
    from importlib import import_module
    enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
    w = {k: torch.as_tensor(v).double() for k, v in enc_mod.random_init_weights(7)[1].items()}
    f = np.load(inp)
    f1 = torch.as_tensor(f["mfea1"]).double().reshape(-1, 6); f2 = torch.as_tensor(f["mfea2"]).double().reshape(-1, 8)
    n0 = f1 @ w["m_fea_1_fcl.weight"].t(); n1 = f2 @ w["m_fea_2_fcl.weight"].t()
    W = w["gat_layer.W"]; a = w["gat_layer.a"].reshape(-1); a_src, a_dst = a[:128], a[128:]
    for it in range(3):
        z0, z1 = n0 @ W, n1 @ W
        e00 = torch.nn.functional.leaky_relu(z0 @ a_src + z0 @ a_dst, 0.2); e01 = torch.nn.functional.leaky_relu(z0 @ a_src + z1 @ a_dst, 0.2)
        att = torch.softmax(torch.stack([e00, e01], 1), 1)
        if it < 2:
            n0 = torch.nn.functional.elu(att[:, 0:1] * z0 + att[:, 1:2] * z1); n1 = torch.nn.functional.elu(z1)
    z0, z1, att = z0.numpy(), z1.numpy(), att.numpy()
    print("rows whose last-pass attention is (0.5, 0.5) to 1e-6 (node 0 == node 1 after an earlier saturated pass): %d of %d" % (int((np.abs(att[:, 0] - 0.5) < 1e-6).sum()), att.shape[0]))
    print("oracle check: max |0.5 (al0 z0 + (1 + al1) z1) - textual build| = %.3e" % np.abs(0.5 * (att[:, :1] * z0 + (1 + att[:, 1:]) * z1) - ref).max())
    for r, x in enumerate(b[:2]):
        d = np.abs(x - ref)
        for row in np.nonzero(d.max(1) > 1e-5 * scale)[0][:12]:
            cols = np.nonzero(d[row] > 1e-5 * scale)[0]
            A = np.stack([z0[row, cols], z1[row, cols]], 1)
            coef, res, _, _ = np.linalg.lstsq(A, x[row, cols].astype(np.float64), rcond=None)
            print("rep %d row %6d block %d: attention (al0, al1) = (%.6f, %.6f)  true coefficients (%.6f, %.6f)  fitted (%.6f, %.6f)  residual %.2e" % (
                r, row, cols[0] // 16, att[row, 0], att[row, 1], 0.5 * att[row, 0], 0.5 * (1 + att[row, 1]), coef[0], coef[1], float(np.abs(A @ coef - x[row, cols]).max())))
