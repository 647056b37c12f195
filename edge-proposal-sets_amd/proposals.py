"""Proposal-set ordering and the inter-stage file format (filter.py:160-165; consumer rank.py:219, :294).

The reference sorts all E scores on the host with an UNSTABLE ``torch.sort(descending=True)`` and saves a
float32 ``[E,3]`` tensor of rows ``(u, v, score)``.  Here the order is declared -- score descending, then
candidate index ascending (== a stable descending sort over the reference's candidate order) -- realised by
sorting packed int64 keys (csrc/topk_keys.hip), which also makes a sharded top-K merge exact.
"""
from __future__ import annotations

import torch

from . import ops

_F32_EXACT_ID = 1 << 24


def sort_order(scores: torch.Tensor, id_base: int = 0) -> torch.Tensor:
    """Permutation realising the declared rule for float32 device scores."""
    keys = ops.pack_keys(scores.contiguous(), id_base=id_base)
    return torch.sort(keys, descending=True).indices


def top_k_keys(scores: torch.Tensor, k: int, id_base: int = 0) -> torch.Tensor:
    """The k best (score, global id) of this shard as sorted int64 keys (descending)."""
    keys = ops.pack_keys(scores.contiguous(), id_base=id_base)
    k = min(k, keys.numel())
    return torch.topk(keys, k, largest=True, sorted=True).values


def merge_top_k(key_lists, k: int) -> torch.Tensor:
    """k-way merge of per-shard sorted key lists: the keys carry the global candidate id, so the result is
    the same for any sharding."""
    allk = torch.cat(list(key_lists))
    k = min(k, allk.numel())
    return torch.topk(allk, k, largest=True, sorted=True).values


def sorted_edges_tensor(edges: torch.Tensor, scores: torch.Tensor, order: torch.Tensor = None) -> torch.Tensor:
    """float32 [E,3] rows (u, v, score) in proposal order -- the legacy layout of filter.py:119,:126,:160-161.
    Node ids are stored as float32 like the reference does, exact only below 2**24."""
    if edges.numel() and int(edges.max()) >= _F32_EXACT_ID:
        raise ValueError("node ids >= 2**24 are not exact in the legacy float32 [E,3] proposal format")
    if order is None:
        order = sort_order(scores)
    e = edges[:, order] if edges.size(0) == 2 else edges[order].t()
    return torch.cat([e.t().to(torch.float32), scores[order].unsqueeze(1)], 1)


def save_sorted_edges(path: str, sorted_edges: torch.Tensor) -> None:
    torch.save(sorted_edges.cpu(), path)      # filter.py:164-165


def load_proposals(path: str, num: int) -> torch.Tensor:
    """rank.py:219 + :294: the first ``num`` proposal rows as a LongTensor [2,num]."""
    t = torch.load(path)
    return t[:int(num), :2].t().long()
