"""Dataset plumbing for the filter.py / rank.py drop-ins (rank.py:38-56, :83-126).

The reference downloads OGB datasets through the ``ogb`` package at run time; neither ``ogb`` nor a
network exists on the build / GPU boxes.  Two sources are supported instead:

* ``$EPS_DATA_ROOT/<name>.pt`` -- a ``torch.save``d dict {edge_index [2,E] long, edge_weight [E] (optional),
  x [N,F] float (optional), num_nodes, split_edge {train:{edge}, valid:{edge, edge_neg}, test:{edge, edge_neg}}}
  (INTEGRATION.md shows the 10-line export script for a machine that has ogb).
* ``--synthetic``: seeded stand-ins with the node / edge counts and degree skew of ddi / collab / ppa
  (SURVEY 8d S1..S3), scaled by ``EPS_SYNTH_SCALE`` (default 1.0) so tests can run them small.
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch

from .graph import CSRGraph
from . import synth

_SHAPES = {  # name: (num_nodes, undirected train edges, feature dim, weighted, n_valid, n_test, n_neg)
    "ddi": (4267, 1_067_911, 0, False, 133_489, 133_489, 100_000),
    "collab": (235_868, 1_179_052 // 2, 128, True, 60_084, 46_329, 100_000),
    "ppa": (576_289, 21_231_931, 58, False, 6_062_562, 3_031_780, 3_000_000),
}


def _synthetic(name: str, device) -> dict:
    n, m, f, weighted, n_val, n_test, n_neg = _SHAPES[name]
    sc = float(os.environ.get("EPS_SYNTH_SCALE", "1.0"))
    n, m = max(64, int(n * sc)), max(256, int(m * sc))
    n_val, n_test, n_neg = max(16, int(n_val * sc)), max(16, int(n_test * sc)), max(64, int(n_neg * sc))
    scale = max(6, (n - 1).bit_length())
    seed = {"ddi": 1, "collab": 2, "ppa": 3}[name]
    abc = dict(a=0.3, b=0.25, c=0.25) if name == "ddi" else dict(a=0.45, b=0.22, c=0.22)
    ei = synth.rmat_edges(scale, int(m * 1.15) + n_val + n_test, seed, device, **abc)
    gen = torch.Generator(device=device).manual_seed(seed + 1)
    perm = torch.randperm(1 << scale, generator=gen, device=device)
    ei = perm[ei] % n
    ei = ei[:, ei[0] != ei[1]]
    lo, hi = torch.minimum(ei[0], ei[1]), torch.maximum(ei[0], ei[1])
    key = torch.unique(lo * n + hi)
    key = key[torch.randperm(key.numel(), generator=gen, device=device)]
    und = torch.stack([torch.div(key, n, rounding_mode="floor"), key % n])
    n_val, n_test = min(n_val, und.shape[1] // 10), min(n_test, und.shape[1] // 10)
    valid, test, train = und[:, :n_val], und[:, n_val:n_val + n_test], und[:, n_val + n_test:]
    if name == "collab":  # collab's edge_index lists both directions and carries integer weights
        w = torch.randint(1, 6, (train.shape[1],), generator=gen, device=device).float()
        edge_index = torch.cat([train, train.flip(0)], 1)
        edge_weight = torch.cat([w, w])
    else:
        edge_index = torch.cat([train, train.flip(0)], 1)   # OGB ddi/ppa edge_index is bidirectional
        edge_weight = None
    x = None
    if f:
        if name == "ppa":
            x = torch.nn.functional.one_hot(torch.randint(0, f, (n,), generator=gen, device=device), f).float()
        else:
            x = torch.randn(n, f, generator=gen, device=device)
    neg = lambda k: torch.randint(0, n, (k, 2), generator=gen, device=device)  # noqa: E731
    split_edge = {"train": {"edge": train.t().contiguous()},
                  "valid": {"edge": valid.t().contiguous(), "edge_neg": neg(n_neg)},
                  "test": {"edge": test.t().contiguous(), "edge_neg": neg(n_neg)}}
    return {"edge_index": edge_index, "edge_weight": edge_weight, "x": x, "num_nodes": n, "split_edge": split_edge}


def load_raw(name: str, synthetic: bool = False, device="cpu") -> dict:
    if synthetic:
        if name not in _SHAPES:
            raise ValueError(f"no synthetic stand-in for dataset '{name}' (have {sorted(_SHAPES)})")
        return _synthetic(name, device)
    root = os.environ.get("EPS_DATA_ROOT")
    path = os.path.join(root, f"{name}.pt") if root else None
    if not path or not os.path.exists(path):
        raise FileNotFoundError(
            f"dataset '{name}': set EPS_DATA_ROOT to a directory holding {name}.pt (see INTEGRATION.md for the export "
            f"script) or pass --synthetic; the ogb downloader is not available offline")
    return torch.load(path)


def get_data(args):
    """rank.py:83-126: -> (edge_index, edge_weight, split_edge, data) with data.adj_t symmetric and
    ``eval_train`` a random subset of the training edges the size of the validation set (:93-95)."""
    raw = load_raw(args.dataset, synthetic=getattr(args, "synthetic", False))
    edge_index = raw["edge_index"].cpu()
    n = int(raw["num_nodes"])
    edge_weight = torch.ones(edge_index.size(1))
    if raw.get("edge_weight") is not None:
        edge_weight = raw["edge_weight"].view(-1).float().cpu()
    split_edge = {k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in raw["split_edge"].items()}
    idx = torch.randperm(split_edge['train']['edge'].size(0))
    idx = idx[:split_edge['valid']['edge'].size(0)]
    split_edge['eval_train'] = {'edge': split_edge['train']['edge'][idx]}
    # T.ToSparseTensor() + to_symmetric() (rank.py:97-98): edge_weight becomes the value when present
    adj_t = CSRGraph.from_edge_index(edge_index, edge_weight if raw.get("edge_weight") is not None else None,
                                     sparse_sizes=(n, n)).to_symmetric()
    x = raw.get("x")
    data = SimpleNamespace(x=(x.float().cpu() if (x is not None and args.use_feature) else None), adj_t=adj_t,
                           full_adj_t=adj_t, num_nodes=n, edge_index=edge_index, edge_weight=edge_weight)

    def to(device):
        data.adj_t = data.adj_t.to(device)
        data.full_adj_t = data.full_adj_t.to(device)
        if data.x is not None:
            data.x = data.x.to(device)
        return data

    data.to = to
    return edge_index, edge_weight, split_edge, data
