"""mtfjsp_normalize_advantages / mtfjsp_pack_views (the hand-off's normalisation without torch kernels, ppo:485,532,668-671)
against the torch expressions of the reference on the same numbers."""
import os
import sys
from importlib import import_module

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _env(B):
    import mtfjsp_amd  # noqa: F401
    be = import_module("e2e-mappo-for-mt-fjsp_amd.batch_env")
    return be.DeviceBatchEnv(6, 6, 2, B, obs_dtype="f32")


@pytest.mark.parametrize("world,rank", [(1, 0), (2, 1), (3, 0)])
def test_normalisation_kernel_equals_the_reference_expression(world, rank):
    B, S, K, Kt = 40, 72, 8, 16
    env = _env(B)
    g = torch.Generator(device="cuda").manual_seed(3 + world)
    G = torch.randn(world, Kt, S, B, device="cuda", generator=g) * 3.0 + 0.7
    vals3 = torch.randn(S, B, 2, device="cuda", generator=g)                    # strided [S,B] views, like job_v[..., 0]
    values = [vals3[..., k % 2] for k in range(K)]
    norm = torch.empty(K, S, B, device="cuda"); targets = torch.empty_like(norm)
    full = torch.empty(Kt, S, world * B, device="cuda")
    env.normalize_advantages(G, K, world, rank, values, norm, targets, full)
    torch.cuda.synchronize()
    for k in range(Kt):
        ref_full = torch.cat([G[w, k] for w in range(world)], dim=1)               # rank-major column blocks: the single-process tensor
        assert torch.equal(full[k], ref_full)
        if k < K:
            want = (G[rank, k] - ref_full.mean()) / (ref_full.std() + 1e-5)        # ppo:485 (torch's unbiased std)
            np.testing.assert_allclose(norm[k].cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(targets[k].cpu().numpy(), (want + values[k]).cpu().numpy(), rtol=2e-5, atol=4e-6)
    # bit-reproducible (fixed-order partial sums, no atomics)
    norm2 = torch.empty_like(norm)
    env.normalize_advantages(G, K, world, rank, values, norm2, None, None)
    torch.cuda.synchronize()
    assert torch.equal(norm, norm2)


def test_pack_views_copies_strided_views():
    B, S = 24, 36
    env = _env(B)
    src = torch.randn(S, B, 4, device="cuda")
    out = torch.zeros(6, S, B, device="cuda")
    env.pack_views([src[..., i] for i in range(4)], out[2:])
    torch.cuda.synchronize()
    assert float(out[:2].abs().max()) == 0.0
    for i in range(4):
        assert torch.equal(out[2 + i], src[..., i])
