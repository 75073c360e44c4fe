"""Synthetic MT-FJSP instance generator — same distribution AND same legacy-numpy random stream as the
reference's Instance_Dataset (instance/generate_allsize_mofjsp_dataset.py:133-296, scope = instance/config_ins.json),
so that seeds 0/1/3 reproduce the reference's train/eval/test sets bit for bit
(pinned by tests/test_instances.py against tests/golden/instances_generator.npz).

Layout returned: t, p [S,T,M] f64 (negative = machine infeasible for the task), tt [S,M,M] f64 symmetric with zero
diagonal, edge [S,E,M/E] int (machine ids per shop).  T = n_job * n_machine (ops per job == n_machine).
"""
import numpy as np

DEFAULT_SCOPE = dict(t_low=1, t_high=99, p_low=1, p_high=20, transT_in_low=1, transT_in_high=10,
                     transT_out_low=1, transT_out_high=20, equal_edge=True, weight_low=0.8, weight_high=1.2)


def split_machines(n_machine, n_edge):
    """machines in index order, evenly; the last shop takes the remainder (generate…py:330-345)."""
    avg = n_machine // n_edge
    out, lst = [], list(range(n_machine))
    for i in range(n_edge):
        size = len(lst) if i == n_edge - 1 else avg
        out.append(lst[:size])
        lst = lst[size:]
    return out


def _uniform_rows(rs, low, high, samples, tail, first, count, chunk=256):
    """rs.uniform(low, high, (samples,) + tail)[first:first+count] without materialising the other rows: the legacy stream fills
    an array element by element in C order, so consecutive chunks of samples are the same draws"""
    out = np.empty((count,) + tuple(tail))
    s = 0
    while s < samples:
        c = min(chunk, samples - s)
        x = rs.uniform(low, high, (c,) + tuple(tail))
        a, b = max(s, first), min(s + c, first + count)
        if a < b:
            out[a - first:b - first] = x[a - s:b - s]
        s += c
    return out


def _native():
    try:
        from . import capi
        return capi.lib()
    except Exception:
        return None


def generate_instances(samples, n_job=6, n_machine=6, n_edge=2, seed=None, scope=None, first=0, count=None, native=None):
    """-> rows [first, first+count) (default: all) of the reference's Instance_Dataset(samples, n_job, n_machine, n_edge, seed).
    The random stream is sequential, so every draw of the whole set is taken, but only the requested rows are kept: a rank of a
    sharded run generates ITS instances in O(shard) memory.  native (default: when libmtfjsp.so is present): the two per-draw
    loops of the reference (infeasible machines, transport times) run in csrc/mtfjsp_hostgen.cpp on the same MT19937 state —
    seconds instead of minutes at 16384 x J20M20; native=False is the pure-numpy form, bit-identical (tests/test_instances.py)."""
    sc = dict(DEFAULT_SCOPE)
    if scope:
        sc.update(scope)
    S, J, M, E = int(samples), int(n_job), int(n_machine), int(n_edge)
    T = J * M
    first = int(first)
    n = S - first if count is None else int(count)
    if first < 0 or n < 0 or first + n > S:
        raise ValueError("rows [first, first+count) must lie inside the set")
    rs = np.random.RandomState(seed) if seed is not None else np.random.mtrand._rand
    L = _native() if native is None or native else None
    if native and L is None:
        raise RuntimeError("native generator requested but libmtfjsp.so is not built")
    # draw order of generate…py:161-176
    avg_t = rs.uniform(sc["t_low"], sc["t_high"], (S, T))[first:first + n]
    avg_p = rs.uniform(sc["p_low"], sc["p_high"], (S, T))[first:first + n]
    t_w = _uniform_rows(rs, sc["weight_low"], sc["weight_high"], S, (T, M), first, n)
    p_w = _uniform_rows(rs, sc["weight_low"], sc["weight_high"], S, (T, M), first, n)
    _uniform_rows(rs, 1, 5, S, (1, M), 0, 0)           # idle powers m_p2: drawn, never used downstream (env:371 forces 1)
    t = np.ascontiguousarray(avg_t[:, :, None] * t_w)
    p = avg_p[:, :, None] * p_w
    del t_w, p_w
    shops = split_machines(M, E)
    if len({len(x) for x in shops}) != 1:
        raise ValueError("n_machine must be divisible by n_edge (the reference stacks the shop lists into an array)")
    shop_of = np.zeros(M, np.int64)
    for e, ms in enumerate(shops):
        shop_of[ms] = e
    in_lo, in_hi, out_hi = sc["transT_in_low"], sc["transT_in_high"], sc["transT_out_high"]
    tt = np.zeros((n, M, M))
    if L is not None:
        import ctypes as C
        st = rs.get_state()
        key = np.ascontiguousarray(st[1], np.uint32)
        pos = C.c_int32(int(st[2]))
        if L.mtfjsp_hostgen_infeasible(key.ctypes.data, C.byref(pos), S, T, M, first, n, t.ctypes.data) != 0:
            raise RuntimeError("mtfjsp_hostgen_infeasible failed")
        if L.mtfjsp_hostgen_transport(key.ctypes.data, C.byref(pos), S, M, shop_of.ctypes.data, float(in_lo), float(in_hi), float(out_hi),
                                      first, n, tt.ctypes.data) != 0:
            raise RuntimeError("mtfjsp_hostgen_transport failed")
        rs.set_state((st[0], key, int(pos.value), 0, 0.0))
    else:
        # generate…py:204-216: per task a uniform number k in [0,M) of machines made infeasible
        for s in range(S):
            ts = t[s - first] if first <= s < first + n else None
            for row in range(T):
                k = rs.randint(0, M)
                idx = rs.choice(M, size=k, replace=False)
                if ts is not None:
                    ts[row, idx] *= -1
        for s in range(S):
            a = np.zeros((M, M))
            for i in range(M):
                for j in range(M):
                    if i == j:
                        continue
                    d = abs(int(shop_of[i]) - int(shop_of[j]))
                    if d == 0:
                        a[i, j] = rs.uniform(low=in_lo, high=in_hi, size=1).item()
                    else:
                        a[i, j] = rs.uniform(low=in_hi * d, high=out_hi * d, size=1).item()   # generate…py:262-266
            if first <= s < first + n:
                U = np.triu(a, k=1)
                tt[s - first] = U + U.T - np.diag(np.diag(a))
    neg = t < 0
    p[neg] = -p[neg]
    edge = np.tile(np.array(shops, dtype=np.int64)[None], (n, 1, 1))
    return t, p, tt, edge


def export_pickle(path, t, p, tt, edge):
    """write a generated set in the reference's dataset layout (generate…py:272-275 / 277-290): a pickled list
    [t, p, transT, edge] of numpy arrays, which `Instance_Dataset(dataset_pth=...)` loads unchanged"""
    import pickle
    with open(path, "wb") as f:
        pickle.dump([np.asarray(t), np.asarray(p), np.asarray(tt), np.asarray(edge)], f)


def random_weights(batch, rng=None, kind="01", config_weights=(0.4, 0.4, 0.2)):
    """Reward weights w3 [B,3] exactly as env.generate_random_weights (env:1253-1270): three python
    `random.uniform(0,1)` draws per instance in instance order, normalised by their numpy sum."""
    import random as _random
    r = rng if rng is not None else _random
    out = np.zeros((batch, 3))
    if kind == "01":          # all instances at once: same draws in the same order, same left-to-right sum of three per row
        w = np.array([r.uniform(0, 1) for _ in range(3 * batch)]).reshape(batch, 3)
        return w / np.sum(w, axis=-1, keepdims=True)
    for b in range(batch):
        if kind == "0.1":
            nums = [round(r.uniform(0, 1), 1) for _ in range(3)]
            tot = sum(nums)
            out[b] = np.array([round(x / tot, 1) for x in nums])
        elif kind == "eval":
            out[b] = np.array(config_weights)
        else:
            raise ValueError(kind)
    return out
