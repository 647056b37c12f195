"""Candidate generation of the filter stage (filter.py:96-109): every 2-hop non-edge (u, v) --
(A @ A)[u,v] != 0, u != v, A[u,v] == 0 -- in the reference's order: column-major (v ascending,
then u ascending), both directions present.

This is SURVEY 8(f) row 1 (the step immediately before the hot path).  Round-1 form: a blocked
expansion on device tensor ops (sort/unique per column block) that streams column blocks to the
scoring kernels, so the full candidate set never has to exist at once (the reference materialises
A @ A on the host: the memory wall on ppa).
"""
from __future__ import annotations

from typing import Iterator, Tuple

import torch

from .graph import CSRGraph


def two_hop_block(g: CSRGraph, v_lo: int, v_hi: int) -> torch.Tensor:
    """Candidates whose column v lies in [v_lo, v_hi): int64 [2, E_blk] = (u; v), column-major."""
    dev = g.device
    rowptr, col = g.rowptr, g.col.to(torch.int64)
    n = g.n_rows
    deg = rowptr[1:] - rowptr[:-1]
    vs = torch.arange(v_lo, v_hi, device=dev, dtype=torch.int64)
    cnt1 = deg[vs]
    v1 = torch.repeat_interleave(vs, cnt1)                                   # (v, w): w in N(v)
    off1 = torch.arange(v1.numel(), device=dev) - torch.repeat_interleave(torch.cumsum(cnt1, 0) - cnt1, cnt1)
    w1 = col[rowptr[v1] + off1]
    cnt2 = deg[w1]
    v2 = torch.repeat_interleave(v1, cnt2)                                   # (v, u): u in N(w)
    w2 = torch.repeat_interleave(w1, cnt2)
    off2 = torch.arange(v2.numel(), device=dev) - torch.repeat_interleave(torch.cumsum(cnt2, 0) - cnt2, cnt2)
    u2 = col[rowptr[w2] + off2]
    key = torch.unique(v2 * n + u2)                                          # column-major, sorted, deduplicated
    vv = torch.div(key, n, rounding_mode="floor")
    uu = key - vv * n
    keep = (uu != vv) & ~torch.isin(key, v1 * n + w1)                        # no diagonal, no known edge
    return torch.stack([uu[keep], vv[keep]])


def iter_candidate_blocks(g: CSRGraph, max_paths: int = 1 << 27) -> Iterator[Tuple[int, int, torch.Tensor]]:
    """Yield (v_lo, v_hi, pairs[2,E_blk]) over all columns; blocks are sized so that the number of 2-hop
    paths expanded at once stays below ``max_paths`` (bounds the temporary memory)."""
    deg = (g.rowptr[1:] - g.rowptr[:-1])
    rows = g.row_index()
    paths = torch.zeros(g.n_rows, dtype=torch.int64, device=g.device)
    paths.index_add_(0, rows, deg[g.col.to(torch.int64)])                    # paths(v) = sum_{w in N(v)} deg(w)
    cum = torch.cumsum(paths, 0).cpu()
    n = g.n_rows
    v = 0
    while v < n:
        base = int(cum[v - 1]) if v > 0 else 0
        hi = int(torch.searchsorted(cum, torch.tensor(base + max_paths), right=True))
        hi = max(hi, v + 1)
        hi = min(hi, n)
        yield v, hi, two_hop_block(g, v, hi)
        v = hi


def all_candidates(g: CSRGraph, max_paths: int = 1 << 27) -> torch.Tensor:
    """The whole candidate list [2,E] (small graphs / tests)."""
    blocks = [b for _, _, b in iter_candidate_blocks(g, max_paths)]
    return torch.cat(blocks, 1) if blocks else torch.zeros((2, 0), dtype=torch.int64, device=g.device)
