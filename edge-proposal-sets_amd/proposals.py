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


def shard_path(path: str, rank: int, world: int) -> str:
    """File of rank ``rank``'s chunk of a proposal list that a sharded filter run left on the ranks that ordered it (r06)."""
    return f"{path}.shard{rank}of{world}"


def save_sorted_edges_shard(path: str, sorted_edges: torch.Tensor, rank: int, world: int) -> None:
    torch.save(sorted_edges.cpu(), shard_path(path, rank, world))


def load_sorted_edges(path: str) -> torch.Tensor:
    """The [E,3] rows of a proposal file -- or, when a sharded run wrote ``path.shard{r}of{N}`` instead (each rank the chunk of the
    declared order it ordered), the shards concatenated in rank order: the same rows."""
    import glob
    import os
    import re
    if os.path.exists(path):
        return torch.load(path)
    shards = glob.glob(glob.escape(path) + ".shard*of*")
    if not shards:
        raise FileNotFoundError(path)
    worlds = {int(re.search(r"\.shard(\d+)of(\d+)$", f).group(2)) for f in shards}
    if len(worlds) != 1:
        raise ValueError(f"{path}: shards of runs with different numbers of ranks")
    world = worlds.pop()
    parts = [torch.load(shard_path(path, r, world)) for r in range(world)]          # (a missing shard raises: the list would have a hole)
    return torch.cat(parts, 0)


def load_proposals(path: str, num: int) -> torch.Tensor:
    """rank.py:219 + :294: the first ``num`` proposal rows as a LongTensor [2,num]."""
    t = load_sorted_edges(path)
    return t[:int(num), :2].t().long()


class StreamingTopK:
    """Top-K of a candidate stream under the declared rule (score descending, candidate order ascending) when the
    blocks ARRIVE in candidate order: a stable descending sort keeps earlier candidates first among equal scores, so
    no candidate index has to be carried (the full ppa candidate set exceeds 2**32 entries).  Once K entries are held,
    a block is first cut to the scores strictly above the current K-th one -- an equal score from a later block can
    never displace an earlier candidate -- which leaves next to nothing to sort for most blocks.  A block that is
    still much larger than K after that (the first ones: hundreds of millions of candidates) is cut at a threshold
    read off a strided sample and VERIFIED to keep at least K entries, so only ~1.5 K scores are ever sorted; the
    result is the exact top-K either way.  Pairs may arrive as int32 or int64, or as an object with ``select(idx)``
    (candidates.ColumnBlock: the pairs of the surviving positions are formed on demand); ``result()`` returns int64."""

    PRECUT_FACTOR = 4          # pre-cut blocks larger than this many times K
    SAMPLE_STRIDE = 64

    def __init__(self, k: int):
        self.k = int(k)
        self.scores = None     # float32 [<=k], descending
        self.pairs = None      # int [2, <=k]

    def _precut(self, scores: torch.Tensor, need: int):
        """Mask of the entries >= a threshold that provably keeps >= ``need`` of them, or None (sort everything)."""
        n = scores.numel()
        if need <= 0 or n <= self.PRECUT_FACTOR * max(need, 1) or n < 1 << 16:
            return None
        sample = scores[:: self.SAMPLE_STRIDE]
        want = min(sample.numel() - 1, int(1.5 * need / self.SAMPLE_STRIDE) + 64)
        t = torch.sort(sample, descending=True).values[want]
        m = scores >= t
        return m if int(m.sum()) >= need else None

    def bar(self):
        """The score a new candidate has to EXCEED to enter the list (None while fewer than K are held)."""
        if self.scores is None or self.scores.numel() < self.k:
            return None
        return float(self.scores[-1].item())

    def push(self, pairs, scores: torch.Tensor = None) -> None:
        lazy = not isinstance(pairs, torch.Tensor)                       # candidates.ColumnBlock: pairs on demand
        keep = None                                                       # surviving positions of the block, ascending
        held = 0 if self.scores is None else self.scores.numel()
        if lazy and getattr(pairs, "survivors", None) is not None and held >= self.k:
            # the expansion kernel already applied a cut at (or below) the current bar: start from its short list
            keep, scores = pairs.survivors
            sel = scores > self.scores[-1]
            keep, scores = keep[sel], scores[sel]
            if keep.numel() == 0:
                return
        elif scores is None or scores.numel() == 0:
            return
        elif held >= self.k:
            keep = torch.nonzero(scores > self.scores[-1]).squeeze(1)     # ascending: candidate order is kept
            if keep.numel() == 0:
                return
            scores = scores[keep]
        elif lazy and getattr(pairs, "padded", False):
            keep = torch.nonzero(scores > float("-inf")).squeeze(1)       # drop the padding between the columns
            if keep.numel() == 0:
                return
            scores = scores[keep]
        # every entry of the final top-K that comes from this block is among the block's own best K
        m = self._precut(scores, self.k)
        if m is not None:
            idx = torch.nonzero(m).squeeze(1)
            scores = scores[idx]
            keep = idx if keep is None else keep[idx]
        if lazy:
            pairs = pairs.select(keep) if keep is not None else pairs.pairs()
        elif keep is not None:
            pairs = pairs[:, keep]
        if self.scores is not None:
            if pairs.dtype != self.pairs.dtype:
                pairs, self.pairs = pairs.long(), self.pairs.long()
            scores = torch.cat([self.scores, scores])          # held entries come first: they are earlier candidates
            pairs = torch.cat([self.pairs, pairs], 1)
        order = torch.sort(scores, descending=True, stable=True).indices[: self.k]
        self.scores, self.pairs = scores[order], pairs[:, order]

    def result(self):
        if self.scores is None:
            dev = "cpu"
            return torch.zeros((2, 0), dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.float32, device=dev)
        return self.pairs.long(), self.scores


def merge_ranked_lists(pair_lists, score_lists, k: int):
    """Merge per-shard top-K lists whose shards are CONTIGUOUS candidate ranges given in shard order: concatenate
    in that order and stable-sort -- the result equals the single-process top-K for any number of shards."""
    scores = torch.cat(list(score_lists))
    pairs = torch.cat(list(pair_lists), 1)
    order = torch.sort(scores, descending=True, stable=True).indices[:k]
    return pairs[:, order], scores[order]
