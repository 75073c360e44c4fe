"""Exact big-batch BatchNorm over shards (optional; include/mtfjsp.h mtfjsp_encoder_set_stats_reduce): two
processes, each with HALF of the instances of the reference's golden batch, all-reduce every BatchNorm's column sums (gloo
through host copies here; RCCL on a multi-GPU node) and must reproduce the reference's outputs for the WHOLE batch on their
rows — while per-shard statistics (the default) do not."""
import os
from importlib import import_module

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _worker(rank, world, port, exact, q):
    import torch
    import torch.distributed as td
    import mtfjsp_amd  # noqa: F401
    from oracle import encoder_oracle as eo
    enc_mod = import_module("e2e-mappo-for-mt-fjsp_amd.encoder")
    D = import_module("e2e-mappo-for-mt-fjsp_amd.dist")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.load(os.path.join(GOLDEN, "encoder_j6m6e2_rand.npz"))
        J, M, E, B = [int(x) for x in g["meta"]]
        T = J * M
        lo, hi = D.shard_range(B, rank, world)
        n = hi - lo
        ja, ma = eo.split_weights(g)
        enc = enc_mod.Encoder(J, M, n, obs_dtype="f32")
        enc.load_weights(ja, ma, eo.critic_weights(g))
        if exact:
            enc.set_stats_reduce(D.bn_stats_allreduce(), B)
            assert not enc.check()                                  # the single-launch GIN kernel cannot reduce across shards
        t = lambda x, dt=None: (torch.as_tensor(np.ascontiguousarray(x)).cuda().to(dt) if dt is not None else torch.as_tensor(np.ascontiguousarray(x)).cuda())
        worst = {"job_prob": 0.0, "mch_prob": 0.0, "h_o": 0.0}
        for s in g["steps"]:
            p = f"s{int(s)}_"
            col, val = eo.ell_from_dense(g[p + "adj"][lo:hi])
            hm_in = g[p + "h_m_in"]
            prob, h_o, job_v = enc.job_actor_forward(
                t(g[p + "tfea"][lo * T:hi * T], torch.float32), t(col.reshape(n * T, 2).astype(np.int32)), t(val.reshape(n * T, 2).astype(np.float32)),
                t(g[p + "cand"][lo:hi].astype(np.int32)), t(g[p + "mask"][lo:hi].astype(np.uint8)),
                None if hm_in.size == 0 else t(hm_in[lo:hi].astype(np.float32)))
            mprob, h_m, mach_v = enc.machine_actor_forward(t(g[p + "mfea1"][lo:hi], torch.float32), t(g[p + "mfea2"][lo:hi], torch.float32),
                                                           t(g[p + "h_o"][lo:hi].astype(np.float32)), t(g[p + "mmask"][lo:hi].reshape(n, M).astype(np.uint8)))
            torch.cuda.synchronize()
            worst["job_prob"] = max(worst["job_prob"], float(np.abs(prob.cpu().numpy() - g[p + "job_prob"][lo:hi]).max()))
            worst["mch_prob"] = max(worst["mch_prob"], float(np.abs(mprob.cpu().numpy() - g[p + "mch_prob"][lo:hi]).max()))
            worst["h_o"] = max(worst["h_o"], float(np.abs(h_o.cpu().numpy() - g[p + "h_o"][lo:hi]).max() / max(1.0, float(np.abs(g[p + "h_o"]).max()))))
        q.put((rank, worst))
    finally:
        td.destroy_process_group()


def _run(exact):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200) + (50 if exact else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, exact, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    out = [q.get(timeout=300) for _ in procs]
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    return dict(out)


def test_two_shards_with_reduced_statistics_reproduce_the_big_batch():
    res = _run(True)
    for rank, w in res.items():
        assert w["job_prob"] < 1e-4 and w["mch_prob"] < 1e-4 and w["h_o"] < 1e-4, (rank, w)


def test_per_shard_statistics_differ_from_the_big_batch():
    """the default (each shard = a reference run with env_batch = its shard) is a different computation"""
    res = _run(False)
    assert max(w["h_o"] for w in res.values()) > 1e-3
