"""Import the *reference* implementation in this build container.

ORACLE HARNESS ONLY — never imported by the product, by `-m gpu` tests,
by smoke() or by bench.py; /root/reference does not exist on the GPU box.
It exists to (a) generate the committed golden fixtures under tests/golden/
and (b) re-measure the reference CPU path.  Shims follow SURVEY.md §8(c):
fake gym / cv2, patched importlib.metadata.version, matplotlib colour-map
accessor, a CPU `trainer.train_device`, fake pynvml, Tensor.cuda no-op.
"""
import importlib.metadata as _md
import os
import sys
import types

REF = os.environ.get("MTFJSP_REFERENCE", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))


def bootstrap(models: bool = True):
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}")
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    for p in (os.path.join(REF, "graph-jsp-env", "src"), REF,
              os.path.join(_HERE, "shims")):
        if p not in sys.path:
            sys.path.insert(0, p)
    if not hasattr(plt.cm, "get_cmap"):
        plt.cm.get_cmap = matplotlib.colormaps.get_cmap
    _v = _md.version
    if not getattr(_md.version, "_mtfjsp_patched", False):
        def version(name):
            return "0.0.0" if name == "graph_jsp_env" else _v(name)
        version._mtfjsp_patched = True
        _md.version = version
    if models:
        import torch
        import trainer  # reference package (namespace)
        td = types.ModuleType("trainer.train_device")
        td.device = torch.device("cpu")
        sys.modules["trainer.train_device"] = td
        trainer.train_device = td
        sys.modules.setdefault("pynvml", types.ModuleType("pynvml"))
        torch.Tensor.cuda = lambda self, *a, **k: self
    return REF


class AttrDict(dict):
    """config dict that also answers attribute access (the env reads
    configs.n_job etc. through a `Variant` wrapper of its own)."""
    __getattr__ = dict.__getitem__


def default_config(n_job=6, n_machine=6, n_edge=2, env_batch=16, **over):
    cfg = dict(
        n_job=n_job, n_machine=n_machine, n_edge=n_edge, env_batch=env_batch,
        weight_mk=0.4, weight_ec=0.4, weight_tt=0.2, m_scaling=1,
        reward_scaling={"scaling_divisor": 1}, GAMMA=0.99, LAMDA=0.98,
        epsilon=0.2, ENTROPY_BETA=0.01, gcn_layer=3, mlp_fea_extract_layer=3,
        gcn_input_dim=12, gcn_hidden_dim=128, learn_eps=False,
        neighbor_pooling_type="average", mlp_actor_layer=3,
        machine_hidden_dim=128, mlp_critic_layer=3, critic_input_dim=128,
        critic_hidden_dim=128, use_orthogonal=False, mask_value=1,
        device="cpu", LR=1e-3, lr_eps=1e-5, K_epochs=5, buffer_size=5,
        use_grad_clip=True, CLIP_GRAD=0.5, use_lr_decay=False,
        decay_step_size=20, decay_ratio=0.96, random_weight_type="01",
    )
    cfg.update(over)
    return cfg
