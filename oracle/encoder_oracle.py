"""fp32 restatement of the rollout forward passes (SURVEY.md §8a rows E1–E4) in
plain tensor ops — TEST INFRASTRUCTURE ONLY (checker for the HIP encoder kernels).

Parity status: PINNED by tests/test_encoder_oracle_golden.py against
tests/golden/encoder_*.npz (outputs of the reference modules themselves, shipped
top1 weights and seeded random weights).

Reference citations: gcn = model/gcn_mlp.py, gat = model/gat.py,
ac = model/actor_critic.py, agent = algorithm/agent_func.py.
Weights are addressed by the reference's state_dict key names.
"""
import numpy as np
import torch

H_EPS = 1e-5  # nn.BatchNorm1d default eps


def _bn(x, gamma, beta):
    """training-mode BatchNorm1d over ALL rows (never .eval() in the reference)."""
    mean = x.mean(0)
    var = x.var(0, unbiased=False)
    return (x - mean) / torch.sqrt(var + H_EPS) * gamma + beta


def _lin(x, w, prefix):
    y = x @ w[prefix + ".weight"].t()
    b = w.get(prefix + ".bias")
    return y if b is None else y + b


def _mlp_tanh(x, w, prefix):
    """MLPActor / MLPCritic (gcn:322-433): L2 tanh(L1 tanh(L0 x))."""
    h = torch.tanh(_lin(x, w, prefix + ".linears.0"))
    h = torch.tanh(_lin(h, w, prefix + ".linears.1"))
    return _lin(h, w, prefix + ".linears.2")


def ell_from_dense(adj):
    """dense adj_wrk [B,T,T] (row = destination, self loop 1) -> ELL (col [B,T,2] int, val [B,T,2])."""
    adj = np.asarray(adj)
    B, T, _ = adj.shape
    col = -np.ones((B, T, 2), np.int32)
    val = np.zeros((B, T, 2), np.float64)
    for b in range(B):
        for v in range(T):
            nz = [u for u in np.flatnonzero(adj[b, v]) if u != v]
            assert len(nz) <= 2
            for s, u in enumerate(nz):
                col[b, v, s] = u
                val[b, v, s] = adj[b, v, u]
    return col, val


def gin_encoder(w, tfea, ell_col, ell_val, B, T, dtype=torch.float32):
    """E1+E2: GraphCNN.forward with neighbor_pooling_type='average', learn_eps=False (gcn:109-197).
    Aggregation in f64 on f32-representable inputs, then cast to f32 (gcn:125,244).
    dtype=torch.float64: the same network in binary64 on the f32-rounded inputs and weights (yardstick, not the reference)."""
    pre = "encoder.feature_extract."
    h = torch.as_tensor(np.asarray(tfea), dtype=torch.float64).float().to(dtype)  # ac:143 .float()
    col = torch.as_tensor(np.asarray(ell_col).reshape(B * T, 2), dtype=torch.long)
    val = torch.as_tensor(np.asarray(ell_val).reshape(B * T, 2), dtype=torch.float64).float().double()
    base = (torch.arange(B * T) // T * T).unsqueeze(1)
    has = col >= 0
    gidx = torch.where(has, col + base, torch.zeros_like(col))
    deg = 1.0 + has.sum(1).double()
    for layer in range(2):
        hd = h.double()
        pooled = hd.clone()                                                      # self loop, weight 1
        for s in range(2):
            pooled = pooled + torch.where(has[:, s:s + 1], val[:, s:s + 1] * hd[gidx[:, s]], torch.zeros_like(hd))
        pooled = (pooled / deg.unsqueeze(1)).to(dtype)
        m = pre + f"mlps.{layer}."
        z = torch.relu(_bn(_lin(pooled, w, m + "linears.0"), w[m + "batch_norms.0.weight"], w[m + "batch_norms.0.bias"]))
        z = torch.relu(_bn(_lin(z, w, m + "linears.1"), w[m + "batch_norms.1.weight"], w[m + "batch_norms.1.bias"]))
        z = _lin(z, w, m + "linears.2")
        h = torch.relu(_bn(z, w[pre + f"batch_norms.{layer}.weight"], w[pre + f"batch_norms.{layer}.bias"]))
    h_pooled = h.reshape(B, T, -1).mean(1)                                       # gcn:192 (1/T graph pool)
    return h, h_pooled


def job_actor_forward(w, tfea, ell_col, ell_val, cand, mask, h_m_prev, B, T, dtype=torch.float32):
    """E3: Operation_Actor_JointAction_selfCritic.forward (ac:104-296), greedy + sampling-free outputs."""
    w = {k: torch.as_tensor(v).to(dtype) for k, v in w.items()}
    h, h_pooled = gin_encoder(w, tfea, ell_col, ell_val, B, T, dtype=dtype)
    J = np.asarray(cand).shape[1]
    Hd = h.shape[1]
    cand_t = torch.as_tensor(np.asarray(cand), dtype=torch.long)
    feat = torch.gather(h.reshape(B, T, Hd), 1, cand_t.unsqueeze(-1).expand(B, J, Hd))
    hp = h_pooled.unsqueeze(1).expand(B, J, Hd)
    if h_m_prev is None or np.asarray(h_m_prev).size == 0:
        hm = w["_input"][None, None, :].expand(B, J, Hd)                          # ac:229-233
    else:
        hm = torch.as_tensor(np.asarray(h_m_prev), dtype=torch.float32).to(dtype).unsqueeze(1).expand(B, J, Hd)
    score = _mlp_tanh(torch.cat([feat, hp, hm], -1), w, "o_policy").squeeze(-1)
    score = score.masked_fill(torch.as_tensor(np.asarray(mask)).bool(), float("-inf"))
    prob = torch.softmax(score, -1)
    job_v = _mlp_tanh(h_pooled, w, "job_critic")
    idx = prob.max(1)[1]                                                          # agent:41-52
    return dict(prob=prob.numpy(), h_nodes=h.numpy(), h_pooled=h_pooled.numpy(), job_v=job_v.numpy(),
                greedy_job=idx.numpy(), greedy_task=cand_t[torch.arange(B), idx].numpy(),
                greedy_logp=torch.log(prob[torch.arange(B), idx]).numpy())


def machine_actor_forward(w, mfea1, mfea2, h_pooled_o, mmask, B, M, dtype=torch.float32):
    """E4: Machine_Actor_JointAction_selfGAT_selfCritic.forward (ac:359-498) with the single shared
    GATLayer (gat:82-159) applied three times on the fixed 2-node graph [[1,1],[0,1]].
    dtype=torch.float64: the same network evaluated in binary64 on the f32-rounded inputs and weights — the yardstick that
    tells the round-off of the reference's own f32 evaluation from a kernel's (tests at full batch sizes)."""
    w = {k: torch.as_tensor(v).to(dtype) for k, v in w.items()}
    f1 = torch.as_tensor(np.asarray(mfea1), dtype=torch.float64).float().reshape(B * M, 6).to(dtype)
    f2 = torch.as_tensor(np.asarray(mfea2), dtype=torch.float64).float().reshape(B * M, 8).to(dtype)
    n0 = f1 @ w["m_fea_1_fcl.weight"].t()
    n1 = f2 @ w["m_fea_2_fcl.weight"].t()
    W = w["gat_layer.W"]
    a = w["gat_layer.a"].reshape(-1)
    Hd = W.shape[1]
    a_src, a_dst = a[:Hd], a[Hd:]
    for it in range(3):
        z0, z1 = n0 @ W, n1 @ W
        e00 = torch.nn.functional.leaky_relu(z0 @ a_src + z0 @ a_dst, 0.2)
        e01 = torch.nn.functional.leaky_relu(z0 @ a_src + z1 @ a_dst, 0.2)
        att = torch.softmax(torch.stack([e00, e01], 1), 1)
        n0 = att[:, 0:1] * z0 + att[:, 1:2] * z1
        n1 = z1                                                                   # node 1 attends only itself
        if it < 2:
            n0 = torch.nn.functional.elu(n0)
            n1 = torch.nn.functional.elu(n1)
    node = _bn((n0 + n1) / 2, w["bn.weight"], w["bn.bias"]).reshape(B, M, Hd)    # ac:420-434
    h_pooled = node.mean(1)
    ho = torch.as_tensor(np.asarray(h_pooled_o), dtype=torch.float32).to(dtype)
    cat = torch.cat([node, h_pooled.unsqueeze(1).expand(B, M, Hd), ho.unsqueeze(1).expand(B, M, Hd)], -1)
    score = _mlp_tanh(cat, w, "m_policy").squeeze(-1) * 10
    score = score.masked_fill(torch.as_tensor(np.asarray(mmask)).reshape(B, M).bool(), float("-inf"))
    prob = torch.softmax(score, -1)
    v = _mlp_tanh(h_pooled, w, "machine_critic")
    return dict(prob=prob.numpy(), h_pooled=h_pooled.numpy(), mach_v=v.numpy(), node=node.numpy())


def split_weights(npz):
    ja = {k[len("w_ja."):]: npz[k] for k in npz.files if k.startswith("w_ja.")}
    ma = {k[len("w_ma."):]: npz[k] for k in npz.files if k.startswith("w_ma.")}
    return ja, ma


def critic_weights(npz):
    return {k[len("w_gc."):]: npz[k] for k in npz.files if k.startswith("w_gc.")}


def _gat_machine_nodes(w, mfea1, mfea2, B, M):
    """shared by the machine actor and the global critic: input projections, 3x the same GATLayer, node mean, BN"""
    f1 = torch.as_tensor(np.asarray(mfea1), dtype=torch.float64).float().reshape(B * M, 6)
    f2 = torch.as_tensor(np.asarray(mfea2), dtype=torch.float64).float().reshape(B * M, 8)
    n0 = f1 @ w["m_fea_1_fcl.weight"].t()
    n1 = f2 @ w["m_fea_2_fcl.weight"].t()
    W = w["gat_layer.W"]
    a = w["gat_layer.a"].reshape(-1)
    Hd = W.shape[1]
    a_src, a_dst = a[:Hd], a[Hd:]
    for it in range(3):
        z0, z1 = n0 @ W, n1 @ W
        e00 = torch.nn.functional.leaky_relu(z0 @ a_src + z0 @ a_dst, 0.2)
        e01 = torch.nn.functional.leaky_relu(z0 @ a_src + z1 @ a_dst, 0.2)
        att = torch.softmax(torch.stack([e00, e01], 1), 1)
        n0 = att[:, 0:1] * z0 + att[:, 1:2] * z1
        n1 = z1
        if it < 2:
            n0 = torch.nn.functional.elu(n0)
            n1 = torch.nn.functional.elu(n1)
    return _bn((n0 + n1) / 2, w["bn.weight"], w["bn.bias"]).reshape(B, M, Hd)


def global_critic_forward(w, tfea, ell_col, ell_val, mfea1, mfea2, B, T, M):
    """Global_Critic_JointAction_GAT.forward (ac:587-750): its own GIN encoder -> graph pool, its own GAT
    machine path -> pool, MLPCritic(256 -> 128 -> 128 -> 4) on [pooled_m, pooled_o]."""
    w = {k: torch.as_tensor(v) for k, v in w.items()}
    _, h_o = gin_encoder(w, tfea, ell_col, ell_val, B, T)
    h_m = _gat_machine_nodes(w, mfea1, mfea2, B, M).mean(1)
    return _mlp_tanh(torch.cat([h_m, h_o], -1), w, "critic").numpy()
